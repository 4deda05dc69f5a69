from . import _Unavailable

GridSpec = _Unavailable
