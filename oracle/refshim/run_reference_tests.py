"""Runs the REFERENCE's own unit tests (metrics_test.py, regions_test.py and the
ZonalEnergySpectrum tests of derived_variables_test.py) against the reference's
own code with `xarray` / `absl` resolved to the stand-ins of this directory.

  python oracle/refshim/run_reference_tests.py [-v] [pattern ...]

This is the fidelity check of the mini-xarray: the reference's tests hold its
known answers and identities, so they fail if the stand-in restates a piece of
xarray semantics wrongly.  Needs /root/reference (this container only); the
summary is committed as tests/golden/reference_selftest.txt.
TEST INFRASTRUCTURE ONLY.
"""
import os
import sys
import unittest

sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get('WB2_REFERENCE', '/root/reference')


def setup_path():
  for p in (REFERENCE, HERE):
    if p in sys.path:
      sys.path.remove(p)
  sys.path.insert(0, REFERENCE)
  sys.path.insert(0, HERE)  # xarray, absl -> the stand-ins


# derived_variables_test.py: only the ZonalEnergySpectrum tests are on the hot
# path; interpolate_spectral_frequencies is SURVEY section 2 "out of scope"
# (it needs groupby / rolling, which the stand-in does not restate).
OUT_OF_SCOPE = ('DerivedVariablesTest', 'test_interpolate_frequencies')

MODULES = ('weatherbench2.metrics_test', 'weatherbench2.regions_test',
           'weatherbench2.derived_variables_test')


def main(argv):
  setup_path()
  verbose = '-v' in argv
  patterns = [a for a in argv if not a.startswith('-')]
  import importlib
  loader = unittest.TestLoader()
  if patterns:
    loader.testNamePatterns = [f'*{p}*' for p in patterns]
  suite = unittest.TestSuite()
  skipped = []

  def add(tests):
    for t in tests:
      if isinstance(t, unittest.TestSuite):
        add(t)
        continue
      tid = t.id()
      if any(pat in tid for pat in OUT_OF_SCOPE):
        skipped.append(tid)
      else:
        suite.addTest(t)
  for m in MODULES:
    add(loader.loadTestsFromModule(importlib.import_module(m)))
  print(f'not run (outside SURVEY.md section 8: precipitation accumulation, '
        f'frequency interpolation / the other derived variables): '
        f'{len(skipped)} tests')
  res = unittest.TextTestRunner(verbosity=2 if verbose else 1).run(suite)
  print(f'reference tests on the mini-xarray: ran {res.testsRun}, '
        f'failures {len(res.failures)}, errors {len(res.errors)}, '
        f'skipped {len(res.skipped)}')
  return 0 if res.wasSuccessful() else 1


if __name__ == '__main__':
  sys.exit(main(sys.argv[1:]))
