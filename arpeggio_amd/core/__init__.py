from .exceptions import (HydrogenError, OBBioMatchError, AtomSerialError,  # noqa: F401
                         SiftMatchError, SelectionError, NativeLibraryError, IncompleteStructureError)
from .packed import PackedComplex  # noqa: F401
from .interactions import InteractionComplex, pack_from_reference_objects  # noqa: F401
