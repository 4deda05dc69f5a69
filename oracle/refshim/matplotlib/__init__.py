"""Import-only stand-in for matplotlib: lets the reference's visualization.py
be imported for its one numerical helper on the hot path's output,
compute_spread_skill_ratio (visualization.py:136-141).  TEST INFRASTRUCTURE
ONLY -- nothing can be plotted with it."""


class _Unavailable:

  def __init__(self, *a, **k):
    raise NotImplementedError('wb2shim: matplotlib is not available')

  def __getattr__(self, name):
    raise NotImplementedError('wb2shim: matplotlib is not available')


class axes:  # annotations evaluated at import time (visualization.py:103)
  Axes = _Unavailable


class figure:
  Figure = _Unavailable
