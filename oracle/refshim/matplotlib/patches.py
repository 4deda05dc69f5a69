from . import _Unavailable

Patch = Rectangle = _Unavailable
