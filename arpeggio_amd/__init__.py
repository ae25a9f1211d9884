"""arpeggio_amd — MI355X-native implementation of arpeggio's contact-detection hot path.

Drop-in surface: ``arpeggio_amd.core.InteractionComplex`` mirrors
``arpeggio.core.InteractionComplex`` for ``run_arpeggio`` / ``get_contacts``
(reference: arpeggio/core/interactions.py:329-347, 172-212).  All arithmetic runs
in hand-written HIP kernels behind the C ABI of include/arpeggio_hip.h; there is no
CPU fallback.
"""
__version__ = '0.1.0'
