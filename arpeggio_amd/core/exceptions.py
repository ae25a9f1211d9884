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


class NativeLibraryError(RuntimeError):
    """The HIP library is missing or a HIP call failed.  There is no CPU fallback."""
