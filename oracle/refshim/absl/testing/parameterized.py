"""absl.testing.parameterized subset: parameters / named_parameters expand a
test method into one method per parameter set at class creation."""
import functools
import unittest

from . import absltest

_PARAMS = '_wb2shim_params'


def _decorate(sets, named):
  def deco(fn):
    setattr(fn, _PARAMS, (list(sets), named))
    return fn
  return deco


def parameters(*sets):
  if len(sets) == 1 and not isinstance(sets[0], (tuple, dict)) and hasattr(
      sets[0], '__iter__') and not isinstance(sets[0], str):
    sets = tuple(sets[0])
  return _decorate(sets, False)


def named_parameters(*sets):
  if len(sets) == 1 and not isinstance(sets[0], (tuple, dict)) and hasattr(
      sets[0], '__iter__'):
    sets = tuple(sets[0])
  return _decorate(sets, True)


def product(**kwargs):
  import itertools
  keys = list(kwargs)
  return _decorate([dict(zip(keys, vals)) for vals in
                    itertools.product(*[kwargs[k] for k in keys])], False)


class _Meta(type):

  def __new__(mcs, name, bases, ns):
    for attr, fn in list(ns.items()):
      spec = getattr(fn, _PARAMS, None)
      if spec is None:
        continue
      sets, named = spec
      del ns[attr]
      for i, p in enumerate(sets):
        if named:
          if isinstance(p, dict):
            p = dict(p)
            suffix = p.pop('testcase_name')
            args, kw = (), p
          else:
            suffix, args, kw = p[0], tuple(p[1:]), {}
        else:
          suffix = str(i)
          if isinstance(p, dict):
            args, kw = (), p
          elif isinstance(p, tuple):
            args, kw = p, {}
          else:
            args, kw = (p,), {}

        def make(fn=fn, args=args, kw=kw):
          @functools.wraps(fn)
          def test(self):
            return fn(self, *args, **kw)
          return test
        t = make()
        t.__name__ = f'{attr}_{suffix}'.replace(' ', '_')
        if hasattr(t, _PARAMS):
          delattr(t, _PARAMS)
        ns[t.__name__] = t
    return super().__new__(mcs, name, bases, ns)


class TestCase(absltest.TestCase, metaclass=_Meta):
  pass
