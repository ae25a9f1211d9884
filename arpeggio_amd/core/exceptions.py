"""Exception classes of the drop-in API.

The reference's error convention (arpeggio/core/exceptions.py:6-38) is part of the boundary: five exception
types, each of which writes one ``logging.error`` line when it is constructed and never calls
``Exception.__init__`` (``e.args`` is what ``BaseException.__new__`` stored: the constructor arguments).  Here that
convention lives in one base class; the subclasses only say what to log.
"""
import logging


class ArpeggioError(Exception):
    """Logs ``describe(*args)`` on construction."""
    message = ''

    def __init__(self, *args):
        logging.error(self.describe(*args))

    def describe(self, *args):
        return self.message


class HydrogenError(ArpeggioError):
    message = 'Please remove all hydrogens from the structure then re-run.'

    def __init__(self):
        ArpeggioError.__init__(self)


class AtomSerialError(ArpeggioError):
    message = 'One or more atom serial numbers are duplicated.'

    def __init__(self):
        ArpeggioError.__init__(self)


class SiftMatchError(ArpeggioError):
    message = 'Seeing is not believing.'

    def __init__(self):
        ArpeggioError.__init__(self)


class OBBioMatchError(ArpeggioError):
    def __init__(self, serial=''):
        ArpeggioError.__init__(self, serial)

    def describe(self, serial):
        if serial:
            return f'OpenBabel OBAtom with serial number {serial} could not be matched to a BioPython counterpart.'
        return 'An OpenBabel atom could not be matched to a BioPython counterpart.'


class SelectionError(ArpeggioError):
    def __init__(self, selection):
        ArpeggioError.__init__(self, selection)

    def describe(self, selection):
        return f'Invalid selector: {selection}'


class IncompleteStructureError(ArpeggioError, NotImplementedError):
    """A structure read from a file alone (``core.protein_reader.read_mmcif``) lacks something the requested run needs and
    only the reference's OpenBabel preparation could supply (I:53-105, 288-327).  Not part of the reference: it prepares
    every structure with OpenBabel.  ``InteractionComplex(path, allow_incomplete=True)`` turns the refusal into a warning."""

    def __init__(self, needs):
        self.needs = tuple(needs)
        ArpeggioError.__init__(self, self.needs)

    def describe(self, needs):
        return ('The structure was read without OpenBabel and the run needs: ' + '; '.join(needs) +
                '.  Pass a PackedComplex prepared with OpenBabel (pack_from_reference_objects) or allow_incomplete=True.')

    def __str__(self):
        return self.describe(self.needs)


class NativeLibraryError(RuntimeError):
    """The HIP library is missing or a HIP call failed.  There is no CPU fallback."""
