"""_evaluate_all_metrics and its baseline substitutions.

tests/golden/reference_evalall_v1.npz holds what the reference's OWN
`_evaluate_all_metrics` (evaluation.py:441-483) and the per-chunk functions of
its Beam driver (:601-675) return for the cases of
tests/golden/reference_cases.py:evalall_table -- forecast := climatology by
valid time, the probabilistic climatology (years of the truth as members, day
366 missing in 2019), persistence in the by-valid and the by-init layout --
run on the stand-in xarray of oracle/refshim (make_reference_vectors.py).

  CPU   the NumPy oracle (oracle/evaluation_np.py) reproduces every vector;
        the product's gathers (SlabGather index tables) select exactly the
        slabs the oracle's eager np.take copies;
  GPU   the product's `_evaluate_all_metrics` / chunk functions reproduce them
        through the HIP path, with host AND device-resident datasets, and the
        deterministic AND ensemble passes read the gathers in place (no
        materialisation).
"""
import os
import types

import numpy as np
import pytest

from oracle import evaluation_np as oev
from oracle import metrics_np as om
from oracle import regions_np as oreg
from oracle.named import DS, NA
from tests import helpers
from tests.golden import reference_cases as rc

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TABLE = rc.evalall_table()
ORACLE_TOL = dict(rtol=1e-12, atol=1e-13)


@pytest.fixture(scope='module')
def vectors():
  return np.load(os.path.join(HERE, 'reference_evalall_v1.npz'))


def oracle_datasets(case):
  def ds(key, extra=None):
    coords = dict(case[f'coords_{key}'])
    for k, a in (extra or {}).items():
      coords[k] = NA(a['data'], a['dims'])
    return DS({k: NA(a['data'], a['dims']) for k, a in case[key].items()},
              coords)
  return (ds('forecast', case['forecast_extra_coords']), ds('truth'),
          ds('climatology'), ds('acc_climatology'))


def oracle_regions():
  return types.SimpleNamespace(
      SliceRegion=oreg.SliceRegion,
      ExtraTropicalRegion=oreg.ExtraTropicalRegion)


def oracle_run(cname):
  build, switches, metrics, rlabels, skipna = TABLE[cname]
  case = build()
  forecast, truth, climatology, acc_clim = oracle_datasets(case)
  ctx = {'acc_climatology': acc_clim}
  factories = rc.region_factories()
  mets = {k: f(om, ctx) for k, f in metrics.items()}
  regions = {r: factories[r](oracle_regions(), ctx) for r in rlabels}
  if 'chunk' in case['kind']:
    variables = list(forecast.keys())
    truth_chunk = oev.truth_at_valid_time(truth, forecast)
    if switches.get('evaluate_climatology'):
      forecast = oev.climatology_like_forecast_chunk(forecast, climatology,
                                                     variables)
    if switches.get('evaluate_persistence'):
      forecast = oev.persistence_like_forecast_chunk(forecast, truth, variables)
    return oev.metric_and_region_loop(forecast, truth_chunk, mets, regions,
                                      skipna, compute_chunk=True)
  return oev.evaluate_all_metrics(
      forecast, truth, climatology, mets, regions, skipna, case['by_init'],
      evaluate_climatology=switches.get('evaluate_climatology', False),
      evaluate_persistence=switches.get('evaluate_persistence', False),
      evaluate_probabilistic_climatology=switches.get(
          'evaluate_probabilistic_climatology', False),
      start_year=switches.get('probabilistic_climatology_start_year'),
      end_year=switches.get('probabilistic_climatology_end_year'),
      hour_interval=switches.get('probabilistic_climatology_hour_interval'))


def _check(vectors, cname, pick, tol):
  """pick(metric label, region label) -> (values, dims) of `geopotential`."""
  want = vectors[f'{cname}/geopotential']
  dims = list(vectors[f'{cname}/geopotential/dims'])
  mlabels = list(vectors[f'{cname}/coord/metric'])
  rlabels = list(vectors[f'{cname}/coord/region'])
  assert dims[:2] == ['metric', 'region']
  for mi, m in enumerate(mlabels):
    for ri, r in enumerate(rlabels):
      got, gdims = pick(m, r)
      assert list(gdims) == dims[2:], (cname, m, r, gdims, dims)
      helpers.assert_close(got, want[mi, ri], err_msg=f'{cname}/{m}/{r}',
                           **tol)


@pytest.mark.parametrize('cname', list(TABLE))
def test_oracle_reproduces_the_reference_driver(vectors, cname):
  res = oracle_run(cname)

  def pick(m, r):
    da = res[(m, r)]['geopotential']
    return da.data, da.dims
  _check(vectors, cname, pick, ORACLE_TOL)
  # the reference merges the metrics with a sorted outer join
  assert list(vectors[f'{cname}/coord/metric']) == sorted(TABLE[cname][2])


# --------------------------------------------------------------------------
# the product's gathers select what the oracle copies (host only: no GPU)
# --------------------------------------------------------------------------
def product_datasets(case, device=None):
  from weatherbench2_amd import xarray_lite as xl

  def ds(key, extra=None):
    coords = dict(case[f'coords_{key}'])
    for k, a in (extra or {}).items():
      coords[k] = xl.DataArray(a['data'], a['dims'])
    data = {}
    for k, a in case[key].items():
      arr = a['data']
      if device is not None:
        import torch
        arr = torch.as_tensor(arr).to(device)
      data[k] = xl.DataArray(arr, a['dims'])
    return xl.Dataset(data, coords)
  return (ds('forecast', case['forecast_extra_coords']), ds('truth'),
          ds('climatology'), ds('acc_climatology'))


@pytest.mark.parametrize('kind', ['clim_byinit', 'clim_byvalid',
                                  'persist_byvalid', 'probclim_byinit'])
def test_gathers_select_what_the_oracle_copies(kind):
  from weatherbench2_amd import evaluation
  from weatherbench2_amd import xarray_lite as xl
  case = rc.evalall_case(kind)
  of, ot, oc, _ = oracle_datasets(case)
  pf, pt, pc, _ = product_datasets(case)
  time_dim = 'valid_time' if case['by_init'] else 'time'
  if kind.startswith('clim'):
    want = oev.climatology_forecast(of, oc, time_dim)
    got = evaluation.climatology_like_forecast(pf, pc, time_dim)
  elif kind == 'persist_byvalid':
    want = oev.create_persistence_forecast(of, ot)
    got = evaluation.create_persistence_forecast(pf, pt)
  else:
    want = oev.climatology_forecast(
        of, oev.make_probabilistic_climatology(ot, 2019, 2020, 12), time_dim)
    prob = evaluation.make_probabilistic_climatology(pt, 2019, 2020, 12)
    assert prob['geopotential'].dims[:3] == ('hour', 'number', 'dayofyear')
    got = evaluation.climatology_like_forecast(pf, prob, time_dim)
    assert got['geopotential'].data.has_missing  # day 366 of 2019
  g = got['geopotential']
  assert isinstance(g.data, xl.SlabGather)  # nothing was copied
  assert g.dims == want['geopotential'].dims
  np.testing.assert_array_equal(g.values, want['geopotential'].data)
  # the gather reads the source array itself
  src = (pt if kind in ('persist_byvalid', 'probclim_byinit') else pc)
  assert g.data.base is src['geopotential'].data


def test_flags_without_their_dataset_raise():
  """A baseline switch is never silently ignored (round-2 review)."""
  from weatherbench2_amd import config, evaluation
  case = rc.evalall_case('clim_byinit')
  pf, pt, _, _ = product_datasets(case)
  cfg = config.Eval(metrics={}, evaluate_climatology=True)
  with pytest.raises(ValueError, match='climatology'):
    evaluation._evaluate_all_metrics('e', cfg, config.Data(by_init=True), False,
                                     forecast=pf, truth=pt)
  for flag in ('evaluate_climatology', 'evaluate_persistence',
               'evaluate_probabilistic_climatology'):
    with pytest.raises(ValueError, match=flag):
      evaluation.evaluate_chunks([(pf, pt)], config.Eval(metrics={},
                                                         **{flag: True}))


def test_eval_is_a_plain_dataclass_with_the_reference_fields():
  import dataclasses
  from weatherbench2_amd import config
  names = [f.name for f in dataclasses.fields(config.Eval)]
  assert names == [
      'metrics', 'regions', 'evaluate_persistence', 'evaluate_climatology',
      'evaluate_probabilistic_climatology',
      'probabilistic_climatology_start_year',
      'probabilistic_climatology_end_year',
      'probabilistic_climatology_hour_interval', 'against_analysis',
      'derived_variables', 'temporal_mean', 'output_format']
  e = config.Eval(metrics={})
  assert e.regions is None and e.temporal_mean is True
  assert e.derived_variables == {} and e.output_format == 'netcdf'


# --------------------------------------------------------------------------
# GPU: the product reproduces the reference driver's numbers
# --------------------------------------------------------------------------
def _product_run(cname, resident: bool):
  import torch
  from weatherbench2_amd import config, evaluation
  from weatherbench2_amd import metrics as gm
  from weatherbench2_amd import regions as gr
  build, switches, metrics, rlabels, skipna = TABLE[cname]
  case = build()
  device = torch.device('cuda', 0) if resident else None
  forecast, truth, climatology, acc_clim = product_datasets(case, device)
  ctx = {'acc_climatology': acc_clim}
  factories = rc.region_factories()
  cfg = config.Eval(
      metrics={k: f(gm, ctx) for k, f in metrics.items()},
      regions={r: factories[r](gr, ctx) for r in rlabels}, **switches)
  if 'chunk' in case['kind']:
    # the Beam driver's sequence on one chunk (evaluation.py:601-675, 583-599)
    variables = list(forecast.keys())
    truth_chunk = evaluation.select_truth_at_valid_time(
        truth, forecast, lead_dim='lead_time')
    if cfg.evaluate_climatology:
      forecast, _ = evaluation._climatology_like_forecast_chunk(
          forecast, truth_chunk, climatology, variables)
    if cfg.evaluate_persistence:
      forecast, _ = evaluation._persistence_like_forecast_chunk(
          forecast, truth_chunk, truth, variables)
    return evaluation._metric_and_region_loop(forecast, truth_chunk, cfg,
                                              skipna, compute_chunk=True)
  return evaluation._evaluate_all_metrics(
      'e', cfg, config.Data(by_init=case['by_init']), skipna,
      forecast=forecast, truth=truth, climatology=climatology)


@pytest.mark.gpu
@pytest.mark.parametrize('resident', [False, True], ids=['host', 'resident'])
@pytest.mark.parametrize('cname', list(TABLE))
def test_product_reproduces_the_reference_driver(vectors, cname, resident):
  res = _product_run(cname, resident)
  var = res['geopotential']
  mlabels = [str(m) for m in res.coords['metric']]
  rlabels = [str(r) for r in res.coords['region']]
  assert mlabels == list(vectors[f'{cname}/coord/metric'])
  assert rlabels == list(vectors[f'{cname}/coord/region'])
  values = var.values

  def pick(m, r):
    return values[mlabels.index(m), rlabels.index(r)], var.dims[2:]
  tol = (dict(rtol=2e-6, atol=2e-7) if 'probclim' in cname
         else dict(rtol=1e-9, atol=1e-12))
  _check(vectors, cname, pick, tol)


@pytest.mark.gpu
def test_deterministic_passes_read_the_gather_in_place(monkeypatch):
  """forecast := climatology reaches K1 as (resident base, slab table): the
  SlabGather is never materialised on the way."""
  import torch
  from weatherbench2_amd import xarray_lite as xl
  calls = []
  real = xl.SlabGather.materialize
  monkeypatch.setattr(xl.SlabGather, 'materialize',
                      lambda self, device=None: calls.append(1) or real(
                          self, device))
  _product_run('evalall_clim_byinit', resident=True)
  _product_run('evalall_persist_byvalid', resident=True)
  assert not calls
  # forecast := probabilistic climatology: K3 reads every member slab where it
  # lives (one address per (outer, member); holes -> a resident NaN slab)
  _product_run('evalall_probclim_byinit', resident=True)
  _product_run('evalall_probclim_byinit_skipna', resident=True)
  assert not calls
  torch.cuda.synchronize()


@pytest.mark.gpu
def test_evaluate_chunks_honours_the_baseline_switches(vectors):
  """evaluate_chunks (the sharded driver) with evaluate_persistence: one chunk
  per init time, the temporal mean of the per-chunk results == the mean of the
  reference's per-chunk vector over init_time."""
  import torch
  from weatherbench2_amd import config, evaluation
  from weatherbench2_amd import metrics as gm
  from weatherbench2_amd import regions as gr
  cname = 'evalall_persist_byinit_chunk'
  build, switches, metrics, rlabels, skipna = TABLE[cname]
  case = build()
  forecast, truth, _, acc_clim = product_datasets(case,
                                                  torch.device('cuda', 0))
  ctx = {'acc_climatology': acc_clim}
  factories = rc.region_factories()
  cfg = config.Eval(metrics={k: f(gm, ctx) for k, f in metrics.items()},
                    regions={r: factories[r](gr, ctx) for r in rlabels},
                    **switches)
  chunks = []
  for i in range(forecast.sizes['init_time']):
    f = forecast.isel(init_time=slice(i, i + 1))
    t = evaluation.select_truth_at_valid_time(truth, f, lead_dim='lead_time')
    chunks.append((f, t))
  res = evaluation.evaluate_chunks(chunks, cfg, skipna, truth=truth)
  want = vectors[f'{cname}/geopotential']  # (metric, region, lead, init, level)
  dims = list(vectors[f'{cname}/geopotential/dims'])
  want = want.mean(axis=dims.index('init_time'))
  got = res['geopotential']
  order = [d for d in dims if d != 'init_time']
  vals = np.transpose(got.values, [got.dims.index(d) for d in order])
  helpers.assert_close(vals, want, rtol=1e-9, atol=1e-12)
