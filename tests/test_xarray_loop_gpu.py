"""Integration level (i) of SURVEY.md 8(b): the product's Metric / Region objects
as values of `Eval.metrics` / `Eval.regions`, driven by a REFERENCE-STYLE loop
over xarray objects -- keyword calls `eval_fn(forecast=, truth=, region=,
skipna=)`, then `.expand_dims({'metric': ..., 'region': ...})`, `xr.concat(...,
'region')`, `xr.merge(...)` on what comes back, exactly the calls of
evaluation.py:408-437 -- and level (ii), the product's own loop, with xarray in
and xarray out.  `xarray` is the mini-xarray of oracle/refshim (real xarray is
not installable here; the stand-in passes the reference's own unit tests), put
on PYTHONPATH in a subprocess so that weatherbench2_amd.xarray_lite picks it up.
Both results must equal the output of the reference's own loop
(tests/golden/reference_vectors_v1.npz, key loop_f32/...).
"""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import numpy as np
    import xarray as xr                      # oracle/refshim/xarray
    assert 'wb2shim' in xr.__version__
    from weatherbench2_amd import xarray_lite as xl
    from weatherbench2_amd import config, evaluation
    from weatherbench2_amd import metrics as gm
    from weatherbench2_amd import regions as gr
    from tests.golden import reference_cases as rc
    assert xl._xr is xr

    vectors = np.load('tests/golden/reference_vectors_v1.npz')
    case = rc.loop_case(np.float32)

    def to_xr(key):
      arrays = case[key]
      used = set()
      for a in arrays.values():
        used |= set(a['dims'])
      return xr.Dataset({k: (a['dims'], a['data']) for k, a in arrays.items()},
                        {k: v for k, v in case['coords'].items() if k in used})

    ctx = {'climatology': to_xr('climatology'),
           'lsm': xr.DataArray(case['lsm']['data'], dims=case['lsm']['dims'],
                               coords={'latitude': case['coords']['latitude'],
                                       'longitude': case['coords']['longitude']})}
    factories = rc.region_factories()
    metrics = {k: f(gm, ctx) for k, f in rc.LOOP_METRICS.items()}
    regions = {r: factories[r](gr, ctx) for r in rc.LOOP_REGIONS}
    forecast, truth = to_xr('forecast'), to_xr('truth')
    print('PLUMBING-OK', flush=True)

    def check(res, temporal_mean, what):
      key = f'loop_f32/temporal_mean_{int(temporal_mean)}'
      assert isinstance(res, xr.Dataset), (what, type(res))
      assert [str(m) for m in res.coords['metric'].values] == [
          str(m) for m in vectors[key + '/coord/metric']], what
      assert [str(r) for r in res.coords['region'].values] == [
          str(r) for r in vectors[key + '/coord/region']], what
      names = [k[len(key) + 1:] for k in vectors.files
               if k.startswith(key + '/') and not k.endswith('/dims')
               and '/coord/' not in k]
      assert sorted(res.data_vars) == sorted(names), (what, list(res.data_vars))
      for name in names:
        assert list(res[name].dims) == list(vectors[f'{key}/{name}/dims']), (
            what, name, res[name].dims)
        np.testing.assert_allclose(np.asarray(res[name].values),
                                   vectors[f'{key}/{name}'], rtol=1e-9,
                                   atol=1e-12, equal_nan=True,
                                   err_msg=f'{what} {name}')

    for temporal_mean in (True, False):
      # (ii) the product's drop-in loop: xarray in, xarray out
      cfg = config.Eval(metrics=metrics, regions=regions,
                        temporal_mean=temporal_mean)
      res = evaluation._metric_and_region_loop(forecast, truth, cfg,
                                               skipna=False)
      check(res, temporal_mean, 'product loop')

      # (i) a reference-style loop over the product's metric objects
      results = []
      for name, metric in metrics.items():
        metric_dim = xr.DataArray([name], coords={'metric': [name]})
        eval_fn = metric.compute if temporal_mean else metric.compute_chunk
        tmp_results = []
        for region_name, region in regions.items():
          region_dim = xr.DataArray([region_name],
                                    coords={'region': [region_name]})
          tmp_result = eval_fn(forecast=forecast, truth=truth, region=region,
                               skipna=False)
          assert isinstance(tmp_result, xr.Dataset), type(tmp_result)
          tmp_results.append(tmp_result.expand_dims(
              {'metric': metric_dim, 'region': region_dim}))
        results.append(xr.concat(tmp_results, 'region'))
      check(xr.merge(results), temporal_mean, 'reference-style loop')
    print('XARRAY-LOOP-OK')
''')


def _run():
  env = dict(os.environ)
  env['PYTHONPATH'] = os.pathsep.join(
      [os.path.join(ROOT, 'oracle', 'refshim'), ROOT,
       env.get('PYTHONPATH', '')])
  return subprocess.run([sys.executable, '-c', SCRIPT], env=env, cwd=ROOT,
                        capture_output=True, text=True, timeout=600)


@pytest.mark.gpu
def test_reference_style_loop_over_product_metrics_with_xarray_objects():
  res = _run()
  assert res.returncode == 0 and 'XARRAY-LOOP-OK' in res.stdout, (
      res.stdout[-2000:] + res.stderr[-6000:])


def test_xarray_loop_plumbing_without_a_gpu():
  """Everything of the script up to the first kernel launch runs anywhere:
  imports with the mini-xarray as `xarray`, dataset / metric / region
  construction from xarray objects."""
  import torch
  if torch.cuda.is_available():
    pytest.skip('covered by the GPU test')
  res = _run()
  assert 'PLUMBING-OK' in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
  assert res.returncode != 0  # no CPU fallback: the first launch must raise
  assert 'XARRAY-LOOP-OK' not in res.stdout
