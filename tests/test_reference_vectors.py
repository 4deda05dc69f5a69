"""Outputs of the REFERENCE's own code pin the oracle and the HIP path.

tests/golden/reference_vectors_v1.npz was written by
tests/golden/make_reference_vectors.py: the reference's unmodified
metrics.py / regions.py / derived_variables.py executed on the seeded cases of
tests/golden/reference_cases.py (xarray resolved to the mini-xarray of
oracle/refshim/, itself checked by the reference's own 82 unit tests).

  CPU   the NumPy oracle reproduces every vector (result dims included);
  GPU   the HIP path, through the Metric API, reproduces them too.
"""
import os
import types

import numpy as np
import pytest

from oracle import metrics_np as om
from oracle import regions_np as oreg
from oracle import spectrum_np
from oracle import thresholds_np as oth
from oracle.named import DS, NA
from tests import helpers
from tests.golden import reference_cases as rc

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def vectors():
  return np.load(os.path.join(HERE, 'reference_vectors_v1.npz'))


def oracle_dataset(case, key) -> DS:
  arrays = case[key]
  used = set()
  for a in arrays.values():
    used |= set(a['dims'])
  source = case.get(f'coords_{key}', case['coords'])  # per-dataset labels
  coords = {k: v for k, v in source.items() if k in used}
  for k, a in case.get('extra_coords', {}).items():  # e.g. valid_time(time)
    if set(a['dims']) <= used:
      coords[k] = NA(a['data'], a['dims'])
  return DS({k: NA(a['data'], a['dims']) for k, a in arrays.items()}, coords)


def oracle_context(case):
  ctx = {'th': oth}
  for key in ('climatology', 'clim_q', 'clim_g'):
    if key in case:
      ctx[key] = oracle_dataset(case, key)
  if 'lsm' in case:
    ctx['lsm'] = NA(case['lsm']['data'], case['lsm']['dims'])
  lat, lon = case['coords']['latitude'], case['coords']['longitude']
  # the oracle's LandRegion takes the mask's labels as separate arguments
  regions = types.SimpleNamespace(
      SliceRegion=oreg.SliceRegion,
      ExtraTropicalRegion=oreg.ExtraTropicalRegion,
      CombinedRegion=oreg.CombinedRegion,
      LandRegion=lambda land_sea_mask, threshold=None: oreg.LandRegion(
          land_sea_mask=land_sea_mask, latitude=lat, longitude=lon,
          threshold=threshold))
  return ctx, regions


TABLE = dict(rc.case_table(), **rc.tier2_table(), **rc.layout_table(),
             **rc.ragged_table())
CASES = list(TABLE)

# float64 sums on both sides, identical elementwise arithmetic: only the order
# of the spatial sum differs (einsum vs the oracle's / the kernels' order)
ORACLE_TOL = dict(rtol=1e-12, atol=1e-13)
# float32 latitude / longitude coordinates (0.25-degree ERA5): the reference's
# weights are float32, its spatial sums accumulate in float32 and it RETURNS
# float32 (recorded in the vectors' dtype).  The oracle mirrors that; the product
# uses the same float32-valued weights but sums and returns float64 (profiles/NOTES.md 4)
# -- both agree with the reference to float32 summation noise.
F32_COORD_TOL = dict(rtol=5e-5, atol=5e-6)
# The same at 721 x 1440 (a million points per sum): two float32 evaluations of
# the reference's expression that differ only in summation order -- the
# stand-in's dot (what the vectors hold) and the oracle's einsum -- are 1e-4
# apart (ACC / MSE, global region; measured: profiles/NOTES.md 4), while the float64
# sums of the product sit within 1e-6 of the stand-in's.  The reference's
# float32 number is only defined to that noise.
F32_COORD_TOL_ERA5 = dict(rtol=3e-4, atol=5e-6)


def _tolerance_gpu(cname):
  if cname.endswith('era5_coords32'):
    return F32_COORD_TOL_ERA5
  if cname.endswith('coords32'):
    return F32_COORD_TOL
  if cname.startswith('ens') or cname.startswith('spatial_ens'):
    # float32 member statistics: the kernel's paired rank sum / sequential
    # member sums differ from NumPy's pairwise float32 sums by float32 rounding
    return dict(rtol=2e-6, atol=2e-7) if 'f32' in cname else dict(
        rtol=1e-9, atol=1e-12)
  return dict(rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('cname', CASES)
def test_oracle_reproduces_the_reference(vectors, cname):
  build, metrics, rlabels, skipna, mode = TABLE[cname]
  case = build()
  ctx, rmod = oracle_context(case)
  forecast, truth = oracle_dataset(case, 'forecast'), oracle_dataset(case,
                                                                     'truth')
  regions = rc.region_factories()
  n = 0
  for mlabel, mfac in metrics.items():
    metric = mfac(om, ctx)
    for rlabel in rlabels:
      region = regions[rlabel](rmod, ctx)
      fn = metric.compute if mode == 'compute' else metric.compute_chunk
      res = fn(forecast, truth, region=region, skipna=skipna)
      prefix = f'{cname}/{mlabel}/{rlabel}/'
      want = [k for k in vectors.files
              if k.startswith(prefix) and not k.endswith('/dims')]
      assert sorted(k[len(prefix):] for k in want) == sorted(res.keys()), prefix
      for key in want:
        got = res[key[len(prefix):]]
        assert list(got.dims) == list(vectors[key + '/dims']), key
        tol = ORACLE_TOL
        if cname.endswith('coords32'):
          # slice regions keep the float32 weights (float32 result); a mask
          # region multiplies them by a float64 field (float64 result)
          f32 = rlabel in ('global', 'europe')
          assert vectors[key].dtype == (np.float32 if f32 else np.float64)
          tol = ((F32_COORD_TOL_ERA5 if 'era5' in cname else F32_COORD_TOL)
                 if f32 else ORACLE_TOL)
        helpers.assert_close(got.data, vectors[key], err_msg=key, **tol)
        n += 1
  assert n > 0


@pytest.mark.parametrize('cname', list(rc.SPECTRUM_CASES))
def test_oracle_spectrum_reproduces_the_reference(vectors, cname):
  case = rc.SPECTRUM_CASES[cname]()
  a = case['dataset']['geopotential']
  lat, lon = case['coords']['latitude'], case['coords']['longitude']
  spec, freq, wavelength = spectrum_np.zonal_energy_spectrum(
      a['data'], lat, lon, a['dims'].index('latitude'),
      a['dims'].index('longitude'))
  want = vectors[f'{cname}/spectrum']
  dims = list(vectors[f'{cname}/spectrum/dims'])
  assert dims == ['time', 'level', 'latitude', 'zonal_wavenumber']
  np.testing.assert_allclose(spec, want, rtol=1e-12, atol=0)
  assert list(vectors[f'{cname}/frequency/dims']) == ['zonal_wavenumber',
                                                      'latitude']
  np.testing.assert_allclose(freq, vectors[f'{cname}/frequency'], rtol=1e-14)
  with np.errstate(divide='ignore'):
    np.testing.assert_allclose(wavelength, vectors[f'{cname}/wavelength'],
                               rtol=1e-14)


# ---------------------------------------------------------------------------
# GPU: the product path through the Metric API
# ---------------------------------------------------------------------------
def _product_metric(mfac, ctx):
  from weatherbench2_amd import metrics as gm
  from weatherbench2_amd import thresholds as gth
  gctx = dict(ctx, th=gth)
  for key in ('climatology', 'clim_q', 'clim_g'):
    if key in ctx:
      gctx[key] = helpers.to_gpu_dataset(ctx[key])
  return mfac(gm, gctx)


@pytest.mark.gpu
@pytest.mark.parametrize('cname', CASES)
def test_hip_path_reproduces_the_reference(vectors, cname):
  from weatherbench2_amd import metrics as gm
  build, metrics, rlabels, skipna, mode = TABLE[cname]
  case = build()
  ctx, rmod = oracle_context(case)
  g = helpers.to_gpu_dataset
  forecast = g(oracle_dataset(case, 'forecast'))
  truth = g(oracle_dataset(case, 'truth'))
  factories = rc.region_factories()
  gregions = {r: helpers.to_gpu_region(factories[r](rmod, ctx))
              for r in rlabels}
  tol = _tolerance_gpu(cname)
  n = 0
  with gm.fused_regions(gregions):
    for mlabel, mfac in metrics.items():
      metric = _product_metric(mfac, ctx)
      for rlabel, region in gregions.items():
        fn = metric.compute if mode == 'compute' else metric.compute_chunk
        res = fn(forecast, truth, region=region, skipna=skipna)
        prefix = f'{cname}/{mlabel}/{rlabel}/'
        want = [k for k in vectors.files
                if k.startswith(prefix) and not k.endswith('/dims')]
        assert sorted(k[len(prefix):] for k in want) == sorted(
            res.data_vars), prefix
        for key in want:
          got = res[key[len(prefix):]]
          assert list(got.dims) == list(vectors[key + '/dims']), key
          use = tol
          if cname.endswith('coords32') and vectors[key].dtype != np.float32:
            use = dict(rtol=1e-9, atol=1e-12)  # float64 sums in the reference
          helpers.assert_close(np.asarray(got.values), vectors[key],
                               err_msg=key, **use)
          if mlabel in rc.DET_METRICS or mlabel in rc.WIND_METRICS:
            # the public API returns the dtype the reference returns: float32
            # for slice regions on float32 coordinates, float64 otherwise
            assert np.asarray(got.values).dtype == vectors[key].dtype, key
          n += 1
  assert n > 0


@pytest.mark.gpu
@pytest.mark.parametrize('cname', list(rc.SPECTRUM_CASES))
def test_hip_spectrum_reproduces_the_reference(vectors, cname):
  from weatherbench2_amd import derived_variables as dv
  case = rc.SPECTRUM_CASES[cname]()
  ds = helpers.to_gpu_dataset(oracle_dataset(case, 'dataset'))
  got = dv.ZonalEnergySpectrum('geopotential').compute(ds)
  want = vectors[f'{cname}/spectrum']
  assert list(got.dims) == list(vectors[f'{cname}/spectrum/dims'])
  f32 = case['dataset']['geopotential']['data'].dtype == np.float32
  # float32 rows: the reference's FFT is complex64 (pocketfft), ours a float32
  # Stockham transform: agreement relative to the row's total power
  scale = want.sum(axis=-1, keepdims=True)
  np.testing.assert_allclose(np.asarray(got.values) / scale, want / scale,
                             rtol=0, atol=2e-6 if f32 else 1e-12)
  np.testing.assert_allclose(np.asarray(got.coords['frequency'].values),
                             vectors[f'{cname}/frequency'], rtol=1e-12)


# ---------------------------------------------------------------------------
# The reference's own metric x region loop (evaluation.py:388-438)
# ---------------------------------------------------------------------------
def _loop_vectors(vectors, temporal_mean):
  key = f'loop_f32/temporal_mean_{int(temporal_mean)}'
  names = [k[len(key) + 1:] for k in vectors.files
           if k.startswith(key + '/') and not k.endswith('/dims')
           and '/coord/' not in k]
  return (key, names, [str(m) for m in vectors[f'{key}/coord/metric']],
          [str(r) for r in vectors[f'{key}/coord/region']])


@pytest.mark.parametrize('temporal_mean', [True, False])
def test_reference_loop_is_the_stack_of_the_oracles_pairs(vectors,
                                                          temporal_mean):
  """Layout facts of the reference's loop output, checked against the oracle's
  per-(metric, region) results: dims (metric, region, ...); the `metric`
  coordinate comes out SORTED (xr.merge joins it with an outer join) while
  regions keep the order given; variables a metric does not produce
  (wind_vector) are NaN-filled."""
  key, names, mlabels, rlabels = _loop_vectors(vectors, temporal_mean)
  assert mlabels == sorted(rc.LOOP_METRICS) != list(rc.LOOP_METRICS)
  assert rlabels == list(rc.LOOP_REGIONS)
  case = rc.loop_case(np.float32)
  ctx, rmod = oracle_context(case)
  forecast, truth = oracle_dataset(case, 'forecast'), oracle_dataset(case,
                                                                     'truth')
  regions = rc.region_factories()
  for mi, mlabel in enumerate(mlabels):
    metric = rc.LOOP_METRICS[mlabel](om, ctx)
    for ri, rlabel in enumerate(rlabels):
      fn = metric.compute if temporal_mean else metric.compute_chunk
      want = fn(forecast, truth, region=regions[rlabel](rmod, ctx))
      for name in names:
        have = vectors[f'{key}/{name}'][mi, ri]
        dims = list(vectors[f'{key}/{name}/dims'])
        assert dims[:2] == ['metric', 'region']
        if name in want:
          assert list(want[name].dims) == dims[2:]
          helpers.assert_close(want[name].data, have, err_msg=f'{mlabel}/'
                               f'{rlabel}/{name}', **ORACLE_TOL)
        else:
          assert np.isnan(have).all()


@pytest.mark.gpu
@pytest.mark.parametrize('temporal_mean', [True, False])
def test_hip_loop_reproduces_the_reference_loop(vectors, temporal_mean):
  """weatherbench2_amd.evaluation._metric_and_region_loop == the reference's
  evaluation._metric_and_region_loop: labels, dims and every number."""
  from weatherbench2_amd import config, evaluation
  from weatherbench2_amd import regions as gregions_mod  # noqa: F401
  key, names, mlabels, rlabels = _loop_vectors(vectors, temporal_mean)
  case = rc.loop_case(np.float32)
  ctx, rmod = oracle_context(case)
  g = helpers.to_gpu_dataset
  factories = rc.region_factories()
  cfg = config.Eval(
      metrics={k: _product_metric(f, ctx) for k, f in rc.LOOP_METRICS.items()},
      regions={r: helpers.to_gpu_region(factories[r](rmod, ctx))
               for r in rc.LOOP_REGIONS},
      temporal_mean=temporal_mean)
  got = evaluation._metric_and_region_loop(
      g(oracle_dataset(case, 'forecast')), g(oracle_dataset(case, 'truth')),
      cfg, skipna=False)
  assert [str(m) for m in got.coords['metric']] == mlabels
  assert [str(r) for r in got.coords['region']] == rlabels
  assert sorted(got.data_vars) == sorted(names)
  for name in names:
    assert list(got[name].dims) == list(vectors[f'{key}/{name}/dims']), name
    helpers.assert_close(np.asarray(got[name].values), vectors[f'{key}/{name}'],
                         rtol=1e-9, atol=1e-12, err_msg=name)
