"""Import-only stand-in for apache_beam: lets the reference's evaluation.py be
imported so that its plain-Python functions (_metric_and_region_loop,
make_latitude_increasing, ...) can run.  No pipeline can be built with it.
TEST INFRASTRUCTURE ONLY."""


class PTransform:

  def expand(self, pcoll):
    raise NotImplementedError('wb2shim: apache_beam is not available')


class DoFn:
  pass


class PCollection:
  pass


class _Unavailable:

  def __init__(self, *a, **k):
    raise NotImplementedError('wb2shim: apache_beam is not available')


Pipeline = Map = MapTuple = Reshuffle = _Unavailable


class combiners:
  ToList = _Unavailable
