"""Import-only stand-in for xarray_beam (see ../apache_beam).  TEST
INFRASTRUCTURE ONLY."""


class Key:

  def __init__(self, offsets=None, vars=None):  # pylint: disable=redefined-builtin
    self.offsets, self.vars = dict(offsets or {}), vars


class _Unavailable:

  def __init__(self, *a, **k):
    raise NotImplementedError('wb2shim: xarray_beam is not available')


ChunksToZarr = DatasetToChunks = Mean = _Unavailable
