from .exceptions import (HydrogenError, OBBioMatchError, AtomSerialError,  # noqa: F401
                         SiftMatchError, SelectionError, NativeLibraryError)
from .packed import PackedComplex  # noqa: F401
