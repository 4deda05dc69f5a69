"""Writes tests/golden/reference_vectors_v1.npz: outputs of the REFERENCE's own
code (/root/reference/weatherbench2/{metrics,regions,derived_variables}.py,
imported unmodified) on the seeded cases of reference_cases.py.

xarray is absent in this container, so the reference runs on the mini-xarray
of oracle/refshim/ (a NumPy restatement of the xarray semantics the reference
touches, checked by running the reference's own 82 unit tests on it:
oracle/refshim/run_reference_tests.py -> reference_selftest.txt).  Everything
else -- every metric formula, region rule, climatology selection, rank and
spectrum computation -- is the reference's real code.  The vectors pin the NumPy
oracle (CPU) and the HIP path (GPU): tests/test_reference_vectors.py.

Only runs where /root/reference exists (this container):
    python tests/golden/make_reference_vectors.py
"""
import io
import os
import sys

sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SHIM = os.path.join(ROOT, 'oracle', 'refshim')
REFERENCE = os.environ.get('WB2_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)
sys.path.insert(0, REFERENCE)
sys.path.insert(0, SHIM)  # `import xarray` -> the mini-xarray

import xarray as xr  # noqa: E402  (the stand-in)
import xarray_beam as xbeam  # noqa: E402  (import-only stand-in)
from weatherbench2 import config as ref_config  # noqa: E402
from weatherbench2 import derived_variables as ref_dv  # noqa: E402
from weatherbench2 import evaluation as ref_evaluation  # noqa: E402  (beam stubs)
from weatherbench2 import metrics as ref_metrics  # noqa: E402
from weatherbench2 import regions as ref_regions  # noqa: E402
from weatherbench2 import thresholds as ref_thresholds  # noqa: E402

from tests.golden import reference_cases as rc  # noqa: E402

assert 'wb2shim' in xr.__version__
assert ref_metrics.__file__.startswith(REFERENCE)


def to_dataset(case, key):
  arrays = case[key]
  used = set()
  for a in arrays.values():
    used |= set(a['dims'])
  source = case.get(f'coords_{key}', case['coords'])  # per-dataset labels
  coords = {k: v for k, v in source.items() if k in used}
  for k, a in case.get('extra_coords', {}).items():  # e.g. valid_time(time)
    if set(a['dims']) <= used:
      coords[k] = (a['dims'], a['data'])
  return xr.Dataset({k: (a['dims'], a['data']) for k, a in arrays.items()},
                    coords)


def context(case):
  ctx = {'th': ref_thresholds}
  for key in ('climatology', 'clim_q', 'clim_g'):
    if key in case:
      ctx[key] = to_dataset(case, key)
  if 'lsm' in case:
    ctx['lsm'] = xr.DataArray(
        case['lsm']['data'], dims=case['lsm']['dims'],
        coords={'latitude': case['coords']['latitude'],
                'longitude': case['coords']['longitude']})
  return ctx


def store(out, key, ds):
  for name in ds.data_vars:
    da = ds[name]
    out[f'{key}/{name}'] = np.asarray(da.data)  # dtype kept: part of the record
    out[f'{key}/{name}/dims'] = np.array(list(da.dims), dtype='U32')


def main():
  out = {}
  regions = rc.region_factories()
  table = dict(rc.case_table(), **rc.tier2_table(), **rc.layout_table(),
               **rc.ragged_table())
  for cname, (build, metrics, rlabels, skipna, mode) in table.items():
    case = build()
    ctx = context(case)
    forecast, truth = to_dataset(case, 'forecast'), to_dataset(case, 'truth')
    if 'truth_full' in case:
      # by-init: the truth the metrics see is what evaluation.py:474 selects
      selected = to_dataset(case, 'truth_full').sel(time=forecast.valid_time)
      for k in truth.data_vars:
        assert selected[k].dims == truth[k].dims
        np.testing.assert_array_equal(selected[k].data, truth[k].data)
      truth = selected
    for mlabel, mfac in metrics.items():
      metric = mfac(ref_metrics, ctx)
      for rlabel in rlabels:
        region = regions[rlabel](ref_regions, ctx)
        fn = metric.compute if mode == 'compute' else metric.compute_chunk
        res = fn(forecast, truth, region=region, skipna=skipna)
        store(out, f'{cname}/{mlabel}/{rlabel}', res)
  for cname, build in rc.SPECTRUM_CASES.items():
    case = build()
    ds = to_dataset(case, 'dataset')
    spec = ref_dv.ZonalEnergySpectrum('geopotential').compute(ds)
    out[f'{cname}/spectrum'] = np.asarray(spec.data)
    out[f'{cname}/spectrum/dims'] = np.array(list(spec.dims), dtype='U32')
    for c in ('frequency', 'wavelength', 'zonal_wavenumber'):
      out[f'{cname}/{c}'] = np.asarray(spec.coords[c].data)
      out[f'{cname}/{c}/dims'] = np.array(list(spec.coords[c].dims),
                                          dtype='U32')
  # the reference's own metric x region loop (evaluation.py:388-438)
  case = rc.loop_case(np.float32)
  ctx = context(case)
  regions = rc.region_factories()
  for temporal_mean in (True, False):
    cfg = ref_config.Eval(
        metrics={k: f(ref_metrics, ctx) for k, f in rc.LOOP_METRICS.items()},
        regions={r: regions[r](ref_regions, ctx) for r in rc.LOOP_REGIONS},
        temporal_mean=temporal_mean)
    res = ref_evaluation._metric_and_region_loop(
        to_dataset(case, 'forecast'), to_dataset(case, 'truth'), cfg,
        skipna=False)
    key = f'loop_f32/temporal_mean_{int(temporal_mean)}'
    store(out, key, res)
    out[f'{key}/coord/metric'] = np.array(list(res.coords['metric'].data),
                                          dtype='U32')
    out[f'{key}/coord/metric/dims'] = np.array(['metric'], dtype='U32')
    out[f'{key}/coord/region'] = np.array(list(res.coords['region'].data),
                                          dtype='U32')
    out[f'{key}/coord/region/dims'] = np.array(['region'], dtype='U32')
  path = os.environ.get('WB2_VECTORS_OUT') or os.path.join(
      HERE, 'reference_vectors_v1.npz')
  np.savez_compressed(path, **out)
  n_val = sum(v.size for k, v in out.items() if not k.endswith('/dims'))
  print(f'wrote {path}: {len(out) // 2} arrays, {n_val} values, '
        f'{os.path.getsize(path) / 1e3:.0f} kB')
  evalall()


def evalall_datasets(case):
  """The opened (forecast, truth, climatology) of an evalall case."""
  def ds(key, extra=None):
    coords = dict(case[f'coords_{key}'])
    for k, a in (extra or {}).items():
      coords[k] = (a['dims'], a['data'])
    return xr.Dataset({k: (a['dims'], a['data'])
                       for k, a in case[key].items()}, coords)
  return (ds('forecast', case['forecast_extra_coords']), ds('truth'),
          ds('climatology'), ds('acc_climatology'))


def evalall():
  """The reference's own _evaluate_all_metrics (evaluation.py:441-483) and the
  per-chunk functions of its Beam driver (:601-675), with the baseline
  switches of config.Eval, on in-memory datasets: only dataset OPENING and
  NetCDF WRITING are replaced (module attributes patched at run time; the
  reference's files are untouched)."""
  out = {}
  regions = rc.region_factories()
  captured = {}
  ref_evaluation._to_netcdf = lambda ds, fn: captured.__setitem__('result', ds)
  ref_evaluation._get_output_path = lambda *a, **k: 'unused'
  for cname, (build, switches, metrics, rlabels, skipna) in (
      rc.evalall_table().items()):
    case = build()
    forecast, truth, climatology, acc_clim = evalall_datasets(case)
    ctx = {'acc_climatology': acc_clim}
    cfg = ref_config.Eval(
        metrics={k: f(ref_metrics, ctx) for k, f in metrics.items()},
        regions={r: regions[r](ref_regions, ctx) for r in rlabels}, **switches)
    data_config = ref_config.Data(
        selection=None, paths=None, by_init=case['by_init'])
    if case['kind'].endswith('_chunk') or 'chunk' in case['kind']:
      # Beam driver, one chunk = the whole forecast: the forecast's variables
      # are dropped (:684-690), truth is selected per chunk (:601-616), the
      # baseline replaces the forecast chunk, then _evaluate_chunk (:583-599)
      t = ref_evaluation._EvaluateAllMetrics(
          'e', cfg, data_config, input_chunks={'init_time': 1}, skipna=skipna)
      variables = list(forecast.keys())
      key = xbeam.Key({'init_time': 0})
      bare = forecast.drop_vars(variables)
      key, chunks = t._sel_corresponding_truth_chunk(key, bare, truth=truth)
      if cfg.evaluate_climatology:
        key, chunks = t._climatology_like_forecast_chunk(
            key, chunks, climatology=climatology, variables=variables)
      if cfg.evaluate_persistence:
        key, chunks = t._persistence_like_forecast_chunk(
            key, chunks, truth=truth, variables=variables)
      key.with_offsets = lambda **kw: key
      _, res = t._evaluate_chunk(key, list(chunks))
    else:
      ref_evaluation.open_forecast_and_truth_datasets = (
          lambda dc, ec, use_dask=False: (forecast, truth, climatology))
      ref_evaluation._evaluate_all_metrics('e', cfg, data_config, skipna=skipna)
      res = captured.pop('result')
    store(out, cname, res)
    for c in ('metric', 'region'):
      out[f'{cname}/coord/{c}'] = np.array(list(res.coords[c].data),
                                           dtype='U32')
      out[f'{cname}/coord/{c}/dims'] = np.array([c], dtype='U32')
  path = os.environ.get('WB2_EVALALL_OUT') or os.path.join(
      HERE, 'reference_evalall_v1.npz')
  np.savez_compressed(path, **out)
  print(f'wrote {path}: {len(out) // 2} arrays, '
        f'{os.path.getsize(path) / 1e3:.0f} kB')


if __name__ == '__main__':
  main()
