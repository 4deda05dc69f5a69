"""GPU parity: zonal energy spectrum vs the NumPy oracle (run with -m gpu).

The reference's own FFT is complex64 for float32 input (numpy >= 2), so float32
results are compared relative to each row's TOTAL power;
float64 results to 1e-10.
"""
import numpy as np
import pytest

from oracle import spectrum_np
from weatherbench2_amd import xarray_lite as xl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dv():
  import torch
  if not torch.cuda.is_available():
    pytest.fail('-m gpu tests need a HIP device')
  from weatherbench2_amd import derived_variables as dv
  return dv


def _dataset(x, dims, lat, lon, extra=None):
  coords = {'latitude': lat, 'longitude': lon}
  coords.update(extra or {})
  return xl.Dataset({'z': xl.DataArray(x, dims)}, coords)


def _row_rel_err(got, want):
  scale = np.abs(want).sum(axis=-1, keepdims=True)
  return np.max(np.abs(got - want) / np.where(scale > 0, scale, 1))


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-12), (np.float32, 2e-6)])
@pytest.mark.parametrize('n_lon', [36, 64, 240, 1440])
def test_matches_oracle(dv, dtype, tol, n_lon):
  rs = np.random.RandomState(n_lon)
  n_lat = 19
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  x = rs.standard_normal((3, 2, n_lat, n_lon)).astype(dtype)
  want, freq, wavelength = spectrum_np.zonal_energy_spectrum(
      x, lat, lon, lat_axis=2, lon_axis=3)
  got = dv.ZonalEnergySpectrum('z').compute(
      _dataset(x, ('time', 'level', 'latitude', 'longitude'), lat, lon))
  assert got.dims == ('time', 'level', 'latitude', 'zonal_wavenumber')
  assert got.values.dtype == np.float64
  # rows at the poles have ~zero circumference: compare on the power scale
  circ = spectrum_np.circumference(lat)
  keep = np.abs(circ) > 1.0
  assert _row_rel_err(got.values[:, :, keep], want[:, :, keep]) < tol
  np.testing.assert_allclose(got.coords['frequency'].values, freq)
  np.testing.assert_allclose(got.coords['wavelength'].values, wavelength)
  np.testing.assert_array_equal(got.coords['zonal_wavenumber'],
                                np.arange(n_lon // 2 + 1))


def test_mock_layout_lon_lat(dv):
  """(…, longitude, latitude) as in schema.mock_truth_data (schema.py:83)."""
  rs = np.random.RandomState(0)
  lat = np.linspace(-60, 60, 13)
  lon = np.linspace(0, 360, 72, endpoint=False)
  x = rs.standard_normal((4, 72, 13))
  want, _, _ = spectrum_np.zonal_energy_spectrum(x, lat, lon, lat_axis=2,
                                                 lon_axis=1)
  got = dv.ZonalEnergySpectrum('z').compute(
      _dataset(x, ('time', 'longitude', 'latitude'), lat, lon))
  assert got.dims == ('time', 'latitude', 'zonal_wavenumber')
  np.testing.assert_allclose(got.values, want, rtol=1e-10, atol=1e-6)


def test_parseval(dv):
  # derived_variables_test.py:415-435: sum_k S[k] == spacing * sum_l f^2
  rs = np.random.RandomState(3)
  lat = np.arange(-30.0, 31.0, 5.0)
  lon = np.linspace(0, 360, 72, endpoint=False)
  x = rs.standard_normal((2, len(lat), len(lon)))
  ds = _dataset(x, ('level', 'latitude', 'longitude'), lat, lon)
  zes = dv.ZonalEnergySpectrum('z')
  spec = zes.compute(ds)
  # White noise has power at Nyquist, which the reference doubles although it
  # has no negative-frequency twin (derived_variables.py:600): remove that half.
  lhs = spec.values.sum(axis=-1) - spec.values[..., -1] / 2
  rhs = (x ** 2).sum(axis=-1) * zes.lon_spacing_m(ds)[None]
  np.testing.assert_allclose(lhs, rhs, rtol=1e-10)


@pytest.mark.parametrize('lat0', [0.0, 30.0, 60.0])
def test_spectral_peak(dv, lat0):
  # derived_variables_test.py:290-321
  lon = np.linspace(0, 360, 36, endpoint=False)
  lat = np.array([lat0])
  x = (10 * np.cos(2 * np.pi * lon / 100))[None, :]
  spec = dv.ZonalEnergySpectrum('z').compute(
      _dataset(x, ('latitude', 'longitude'), lat, lon))
  k = int(np.argmax(spec.values[0]))
  wl = spec.coords['wavelength'].values[:, 0]
  expected = spectrum_np.circumference(lat)[0] * 100 / 360
  assert k == int(np.argmin(np.abs(wl[1:] - expected))) + 1


def test_last_bin_doubled_and_nonuniform_lon_raises(dv):
  lat = np.array([0.0])
  lon = np.array([0.0, 90.0, 180.0, 270.0])
  x = np.array([[1.0, -1.0, 1.0, -1.0]])
  spec = dv.ZonalEnergySpectrum('z').compute(
      _dataset(x, ('latitude', 'longitude'), lat, lon))
  c = spectrum_np.circumference(lat)[0]
  np.testing.assert_allclose(spec.values[0], [0, 0, 2.0 * c], atol=1e-9 * c)
  with pytest.raises(ValueError):
    dv.ZonalEnergySpectrum('z').compute(
        _dataset(x, ('latitude', 'longitude'), lat,
                 np.array([0.0, 90.0, 181.0, 270.0])))


def test_fused_time_mean(dv):
  # scripts/compute_zonal_energy_spectrum.py:234 (xbeam.Mean over time)
  rs = np.random.RandomState(9)
  lat = np.linspace(-80, 80, 9)
  lon = np.linspace(0, 360, 48, endpoint=False)
  x = rs.standard_normal((6, 2, 9, 48)).astype(np.float32)
  ds = _dataset(x, ('time', 'level', 'latitude', 'longitude'), lat, lon)
  zes = dv.ZonalEnergySpectrum('z')
  per_time = zes.compute(ds).values
  fused = zes.compute(ds, time_mean_dim='time')
  assert fused.dims == ('level', 'latitude', 'zonal_wavenumber')
  np.testing.assert_allclose(fused.values, per_time.mean(axis=0), rtol=1e-12)


@pytest.mark.parametrize('n_lon', [64, 256, 1440])
@pytest.mark.parametrize('skipna', [True, False])
def test_fused_kernel_time_mean(n_lon, skipna):
  """The LDS-FFT kernel with the time mean fused (sums in registers, one store
  per output row) == the mean of the per-time spectra it writes otherwise,
  incl. NaN rows with and without skipna."""
  import torch
  from weatherbench2_amd import engine
  dev = torch.device('cuda')
  gen = torch.Generator(device=dev).manual_seed(n_lon)
  n_time, n_lev, n_lat = 5, 3, 7
  x = torch.randn((n_time, n_lev, n_lat, n_lon), generator=gen, device=dev)
  x[1, 0, 2, 5] = float('nan')     # one NaN poisons that row's spectrum
  x[3, 2, :, :] = float('nan')     # a whole level at one time
  x[:, 1, 4, 0] = float('nan')     # a row that is NaN at every time
  lat = np.linspace(-75, 75, n_lat)
  circ = torch.as_tensor(spectrum_np.circumference(lat)).to(dev)
  per_time = engine.zonal_spectrum(x, circ, n_lat)
  fused = engine.zonal_spectrum(x, circ, n_lat, n_time=n_time, skipna=skipna)
  assert fused.shape == (n_lev, n_lat, n_lon // 2 + 1)
  want = (torch.nanmean(per_time, dim=0) if skipna else per_time.mean(0))
  torch.testing.assert_close(fused, want, rtol=1e-12, atol=0, equal_nan=True)
  assert torch.isnan(fused[1, 4]).all()
  assert torch.isnan(fused[0, 2]).all() != skipna


@pytest.mark.parametrize('skipna', [True, False])
@pytest.mark.parametrize('n_lon,n_lev', [(1440, 13), (1440, 5), (256, 13)])
def test_fused_time_mean_tail_is_bit_identical(n_lon, n_lev, skipna):
  """More output rows than resident waves: the last partial round of the time
  mean is taken apart into single transforms and averaged by a second kernel
  (spectrum_fused.hip launch_time_mean).  Whether or not the split applies on
  this device, every output row must be BIT-identical to the time-ordered fp64
  sum of the materialised spectra divided by the count."""
  import torch
  from weatherbench2_amd import engine
  dev = torch.device('cuda')
  gen = torch.Generator(device=dev).manual_seed(n_lon + n_lev)
  n_time, n_lat = 3, 721
  x = torch.randn((n_time, n_lev, n_lat, n_lon), generator=gen, device=dev)
  x[1, 0, 2, 5] = float('nan')
  x[2, n_lev - 1, 700:, :] = float('nan')    # rows that land in the tail
  x[:, n_lev - 1, 720, 3] = float('nan')     # NaN at every time
  lat = np.linspace(-90, 90, n_lat)
  circ = torch.as_tensor(spectrum_np.circumference(lat)).to(dev)
  per_time = engine.zonal_spectrum(x, circ, n_lat)
  fused = engine.zonal_spectrum(x, circ, n_lat, n_time=n_time, skipna=skipna)
  total = torch.zeros_like(per_time[0])
  count = torch.zeros_like(per_time[0])
  for t in range(n_time):
    keep = ~torch.isnan(per_time[t]) if skipna else torch.ones_like(
        per_time[t], dtype=torch.bool)
    total = total + torch.where(keep, per_time[t], torch.zeros_like(total))
    count = count + keep
  want = total / count
  assert torch.equal(torch.isnan(fused), torch.isnan(want))
  assert torch.equal(torch.nan_to_num(fused), torch.nan_to_num(want))
  assert torch.isnan(fused[n_lev - 1, 720]).all()


def test_full_size_unit_parseval_and_determinism():
  """BASELINE config 4: 13 x 721 x 1440 float32."""
  import torch
  from weatherbench2_amd import engine
  dev = torch.device('cuda')
  gen = torch.Generator(device=dev).manual_seed(5)
  x = torch.randn((13, 721, 1440), generator=gen, device=dev)
  lat = np.linspace(-90, 90, 721)
  circ = torch.as_tensor(spectrum_np.circumference(lat)).to(dev)
  s1 = engine.zonal_spectrum(x, circ, 721)
  s2 = engine.zonal_spectrum(x, circ, 721)
  assert torch.equal(s1, s2)
  assert s1.shape == (13, 721, 721) and s1.dtype == torch.float64
  # Parseval: sum_k S[k] = C/N * sum_l f^2  (norm='forward', one-sided x2)
  lhs = s1.sum(-1) - s1[..., -1] / 2  # Nyquist is doubled by the reference
  rhs = (x.double() ** 2).sum(-1) * (circ / 1440)[None]
  mask = (circ.abs() > 1.0)[None].expand_as(lhs)
  torch.testing.assert_close(lhs[mask], rhs[mask], rtol=2e-5, atol=0)
  # one row against numpy's complex64 rfft, on the row-power scale
  row = x[5, 300].cpu().numpy()
  want = spectrum_np.simple_power(row[None])[0] * float(circ[300])
  got = s1[5, 300].cpu().numpy()
  assert np.max(np.abs(got - want)) / want.sum() < 2e-6


@pytest.mark.parametrize('dtype,tol', [(np.float32, 1e-6), (np.float64, 1e-13)])
@pytest.mark.parametrize('n_lon', [96, 288, 320, 384, 480, 640, 768, 1280, 1800,
                                   2048, 2560, 2880, 3600])
def test_row_lengths_beyond_the_datasets_grids(n_lon, dtype, tol):
  """The plans added for other grids (fft_core.hpp): materialised, time-mean
  and latitude-mean modes against the fp64 NumPy spectrum of the same rows."""
  import torch
  from weatherbench2_amd import engine
  dev = torch.device('cuda')
  rs = np.random.RandomState(n_lon)
  n_time, n_lat = 3, 7
  x = rs.standard_normal((n_time, 2, n_lat, n_lon)).astype(dtype)
  lat = np.linspace(-75, 75, n_lat)
  circ_np = spectrum_np.circumference(lat)
  circ = torch.as_tensor(circ_np).to(dev)
  want = spectrum_np.simple_power(x.astype(np.float64)) * circ_np[None, None, :,
                                                                 None]
  xd = torch.as_tensor(x).to(dev)
  got = engine.zonal_spectrum(xd, circ, n_lat).cpu().numpy()
  assert _row_rel_err(got, want) < tol
  mean = engine.zonal_spectrum(xd, circ, n_lat, n_time, False).cpu().numpy()
  assert _row_rel_err(mean, want.mean(0)) < tol
  # the latitude mean fused into the kernel (configs[3]'s mode) == the
  # materialised spectrum reduced afterwards
  w = torch.as_tensor(np.cos(np.deg2rad(lat))).to(dev)
  fields = xd.reshape(-1, n_lat, n_lon)
  lat_mean = engine.zonal_spectrum_lat_mean(fields, circ, w, n_lat)
  two_pass = (engine.zonal_spectrum(fields, circ, n_lat) *
              w[None, :, None]).sum(1) / w.sum()
  torch.testing.assert_close(lat_mean, two_pass, rtol=1e-12,
                             atol=1e-12 * float(two_pass.abs().max()))


@pytest.mark.parametrize('n_lon', [64, 128, 240, 256, 360, 512, 720, 1024, 1440,
                                   96, 1800, 3600])
def test_fused_fft_matches_rocfft_path(n_lon):
  """The single-kernel LDS FFT and the rocFFT pipeline agree to fp32 noise."""
  import os
  import subprocess
  import sys
  import torch
  from weatherbench2_amd import engine
  dev = torch.device('cuda')
  gen = torch.Generator(device=dev).manual_seed(n_lon)
  n_lat = 9
  x = torch.randn((3, n_lat, n_lon), generator=gen, device=dev)
  lat = np.linspace(-80, 80, n_lat)
  circ = torch.as_tensor(spectrum_np.circumference(lat)).to(dev)
  fused = engine.zonal_spectrum(x, circ, n_lat).cpu().numpy()
  # fp64 numpy spectrum of the same float32 data as the common yardstick
  want = spectrum_np.simple_power(x.cpu().numpy().astype(np.float64)) * (
      spectrum_np.circumference(lat)[None, :, None])
  assert _row_rel_err(fused, want) < 1e-6
  # the rocFFT path in a child process (the backend is chosen at plan creation)
  code = (
      'import numpy as np, torch, sys\n'
      'sys.path.insert(0, %r)\n'
      'from weatherbench2_amd import engine\n'
      'from oracle import spectrum_np\n'
      'dev = torch.device("cuda")\n'
      'gen = torch.Generator(device=dev).manual_seed(%d)\n'
      'x = torch.randn((3, %d, %d), generator=gen, device=dev)\n'
      'lat = np.linspace(-80, 80, %d)\n'
      'circ = torch.as_tensor(spectrum_np.circumference(lat)).to(dev)\n'
      'np.save(sys.argv[1], engine.zonal_spectrum(x, circ, %d).cpu().numpy())\n'
  ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), n_lon,
       n_lat, n_lon, n_lat, n_lat)
  out = f'/tmp/rocfft_{n_lon}.npy'
  env = dict(os.environ, WB2HIP_SPECTRUM_BACKEND='rocfft')
  subprocess.run([sys.executable, '-c', code, out], check=True, env=env,
                 timeout=300)
  rocfft = np.load(out)
  assert _row_rel_err(fused, rocfft) < 2e-6


# ---- the rest of derived_variables_test.py:219-520 --------------------------
EARTH_RADIUS_M = 1000 * (6357 + 6378) / 2  # schema.py:59


def make_multispectral_dataset(spatial_resolution_in_degrees=5, latitude=None,
                               min_wavelength_lon=50, max_wavelength_lon=100,
                               constant_to_add=0.0):
  """derived_variables_test.py:47-80: smooth (in spectral space) data on the
  mock grid (prediction_timedelta, time, level, longitude, latitude)."""
  res = spatial_resolution_in_degrees
  lat = np.linspace(-90, 90, round(180 / res) + 1)
  lon = np.linspace(0, 360, round(360 / res), endpoint=False)
  level = np.array([500, 700, 850])
  if latitude is not None:
    if isinstance(latitude, slice):
      lat = lat[(lat >= latitude.start) & (lat <= latitude.stop)]
    else:
      lat = lat[np.isclose(lat, latitude)]
  assert res < min_wavelength_lon / 2
  x = np.zeros((2, 1, 3, len(lon), len(lat)))
  n_signals = 100
  for wl in np.linspace(min_wavelength_lon, max_wavelength_lon, num=n_signals):
    x += (np.cos(2 * np.pi * lon / wl)[None, None, None, :, None]
          * np.exp(-wl / max_wavelength_lon)
          * np.sin(level / 500)[None, None, :, None, None]
          * np.cos(lat / 100)[None, None, None, None, :]) / n_signals
  x += constant_to_add * np.abs(x).mean()
  dims = ('prediction_timedelta', 'time', 'level', 'longitude', 'latitude')
  return xl.Dataset({'geopotential': xl.DataArray(x, dims)},
                    {'latitude': lat, 'longitude': lon, 'level': level,
                     'prediction_timedelta': np.arange(2), 'time': np.arange(1)})


def test_lon_spacing_m_correct_at_equator(dv):
  # derived_variables_test.py:229-244
  res = 30
  lat = np.linspace(-90, 90, 7)
  lon = np.linspace(0, 360, 12, endpoint=False)
  ds = _dataset(np.zeros((7, 12)), ('latitude', 'longitude'), lat, lon)
  zes = dv.ZonalEnergySpectrum('z')
  circum = EARTH_RADIUS_M * 2 * np.pi
  np.testing.assert_allclose(zes._circumference(lat)[3], circum)
  np.testing.assert_allclose(zes.lon_spacing_m(ds)[3], circum * res / 360)


def test_data_has_right_shape_dims_and_coordinates(dv):
  # derived_variables_test.py:246-288
  ds = make_multispectral_dataset(spatial_resolution_in_degrees=10)
  ds = xl.Dataset({'geopotential': ds['geopotential']},
                  {k: v for k, v in ds.coords.items()})
  spectrum = dv.ZonalEnergySpectrum('geopotential').compute(ds)
  expected = dict(ds.sizes)
  expected['zonal_wavenumber'] = ds.sizes['longitude'] // 2 + 1
  del expected['longitude']
  assert dict(spectrum.sizes) == expected
  freq = spectrum.coords['frequency']
  assert freq.dims == ('zonal_wavenumber', 'latitude')
  # skip the poles (cos(90 deg) ~ 1e-17: the spacing collapses)
  f = freq.values[:, 1:-1]
  assert (np.diff(f, axis=0) > 0).all()
  np.testing.assert_array_equal(f[0], 0)
  with np.errstate(divide='ignore'):
    np.testing.assert_array_equal(spectrum.coords['wavelength'].values,
                                  1 / freq.values)
  mid = f.shape[1] // 2
  assert (np.diff(f[1:, mid:], axis=1) > 0).all()
  assert (np.diff(f[1:, :mid + 1], axis=1) < 0).all()


@pytest.mark.parametrize('add_constant', [False, True])
def test_resolved_frequencies_are_mostly_independent_of_discretization(
    dv, add_constant):
  # derived_variables_test.py:322-407: fails for a different DFT normalisation
  latitude = 30
  lo, hi = 50, 100
  kw = dict(latitude=latitude, min_wavelength_lon=lo, max_wavelength_lon=hi,
            constant_to_add=50 if add_constant else 0)
  zes = dv.ZonalEnergySpectrum('geopotential')
  spec = {}
  for res in (5, 20):
    s = zes.compute(make_multispectral_dataset(
        spatial_resolution_in_degrees=res, **kw))
    ax = s.dims.index('zonal_wavenumber')
    spec[res] = (s.coords['frequency'].values[:, 0],
                 np.moveaxis(np.asarray(s.values), ax, 0))
  wavelength_m = lambda wl: ((wl / 360) * (2 * np.pi * EARTH_RADIUS_M)
                             * np.cos(np.pi * latitude / 180))
  rs = np.random.RandomState(0)
  test_frequencies = sorted(1 / wavelength_m(
      rs.uniform(low=1.1 * lo, high=0.9 * hi, size=30)))
  test_frequencies.append(0)
  nearest = lambda fr, f: int(np.argmin(np.abs(fr - f)))
  f5, s5 = spec[5]
  f20, s20 = spec[20]
  for i, f in enumerate(test_frequencies):
    err = np.abs(s5[nearest(f5, f)] - s20[nearest(f20, f)])
    idx = nearest(f20, f)
    below, above = max(0, idx - 1), min(len(f20) - 1, idx + 1)
    bound = np.abs(s20[below] - s20[above])
    assert (err < bound).all(), f'Failed at {i=} {f=}'


def _sel_lat(da, lat_value, dim):
  """(frequency values, data[..., dim]) of one latitude, `dim` moved last."""
  j = int(np.argmin(np.abs(np.asarray(da.coords['latitude']) - lat_value)))
  v = np.moveaxis(np.asarray(da.values),
                  (da.dims.index('latitude'), da.dims.index(dim)), (-2, -1))
  return j, v[..., j, :]


def test_interpolate_frequencies_default_args(dv):
  # derived_variables_test.py:437-477
  ds = make_multispectral_dataset(spatial_resolution_in_degrees=5,
                                  latitude=slice(-30, 30))
  spectrum = dv.ZonalEnergySpectrum('geopotential').compute(ds)
  interpolated = dv.interpolate_spectral_frequencies(
      spectrum, wavenumber_dim='zonal_wavenumber')
  assert set(interpolated.dims) == (
      {'frequency'} | set(spectrum.dims) - {'zonal_wavenumber'})
  # latitude 0 has the narrowest frequency range = the default grid: unchanged
  j0, s0 = _sel_lat(spectrum, 0, 'zonal_wavenumber')
  _, i0 = _sel_lat(interpolated, 0, 'frequency')
  np.testing.assert_allclose(
      interpolated.coords['frequency'],
      spectrum.coords['frequency'].values[:, j0], rtol=1e-12)
  np.testing.assert_allclose(i0, s0, rtol=1e-9, atol=1e-12)
  _, s5 = _sel_lat(spectrum, 5, 'zonal_wavenumber')
  _, i5 = _sel_lat(interpolated, 5, 'frequency')
  np.testing.assert_allclose(i5, s5, rtol=0.15, atol=1e-12)
  with np.errstate(divide='ignore'):
    np.testing.assert_allclose(interpolated.coords['wavelength'].values,
                               1 / np.asarray(interpolated.coords['frequency']))


def test_interpolate_frequencies_use_5_degree_values(dv):
  # derived_variables_test.py:479-520
  ds = make_multispectral_dataset(spatial_resolution_in_degrees=1.0,
                                  latitude=slice(-30, 30))
  spectrum = dv.ZonalEnergySpectrum('geopotential').compute(ds)
  ref_lat, ks = 5, slice(3, 8)
  j, s_ref = _sel_lat(spectrum, ref_lat, 'zonal_wavenumber')
  freqs = spectrum.coords['frequency'].values[ks, j]
  interpolated = dv.interpolate_spectral_frequencies(
      spectrum, wavenumber_dim='zonal_wavenumber', frequencies=freqs)
  assert set(interpolated.dims) == (
      {'frequency'} | set(spectrum.dims) - {'zonal_wavenumber'})
  _, i_ref = _sel_lat(interpolated, ref_lat, 'frequency')
  np.testing.assert_allclose(i_ref, s_ref[..., ks], rtol=1e-9, atol=1e-12)
  _, s_next = _sel_lat(spectrum, ref_lat + 1, 'zonal_wavenumber')
  _, i_next = _sel_lat(interpolated, ref_lat + 1, 'frequency')
  np.testing.assert_allclose(i_next, s_next[..., ks], rtol=0.1, atol=1e-12)
  with pytest.raises(ValueError):
    dv.interpolate_spectral_frequencies(spectrum, 'zonal_wavenumber',
                                        frequencies=np.ones((2, 2)))


@pytest.mark.parametrize('n_lat,n_lon,n_field', [(721, 1440, 3), (181, 360, 5),
                                                 (9, 64, 40), (33, 1440, 2)])
def test_fused_latitude_mean_matches_oracle(n_lat, n_lon, n_field):
  """BASELINE configs[3]: the area-weighted latitude mean of the zonal energy
  spectrum in ONE kernel (every (field, latitude segment) reduced in registers,
  segments added in order) == the oracle's spectrum
  (derived_variables.py:592-626) averaged with the latitude weights of
  metrics.py:35-60, and == the materialise-then-reduce path."""
  import torch
  from weatherbench2_amd import engine, plan as plan_lib
  dev = torch.device('cuda')
  rs = np.random.RandomState(n_lat + n_lon)
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  x = rs.standard_normal((n_field, n_lat, n_lon)).astype(np.float32)
  w = plan_lib.get_lat_weights(lat)
  spec, _, _ = spectrum_np.zonal_energy_spectrum(x, lat, lon, 1, 2)
  want = (spec * w[None, :, None]).sum(1) / w.sum()
  circ = torch.as_tensor(spectrum_np.circumference(lat)).to(dev)
  xd = torch.as_tensor(x, device=dev)
  wd = torch.as_tensor(w, device=dev)
  got = engine.zonal_spectrum_lat_mean(xd, circ, wd, n_lat)
  assert got.shape == (n_field, n_lon // 2 + 1) and got.dtype == torch.float64
  # float32 transform on both sides (the reference's FFT is complex64):
  # relative to the largest bin of each field
  scale = np.abs(want).max(axis=1, keepdims=True)
  np.testing.assert_allclose(got.cpu().numpy() / scale, want / scale, atol=3e-6)
  # the two product paths agree to summation-order noise
  full = engine.zonal_spectrum(xd, circ, n_lat)
  two_pass = (full * wd[None, :, None]).sum(1) / wd.sum()
  torch.testing.assert_close(got, two_pass, rtol=1e-12, atol=1e-12 * float(
      scale.max()))
  # deterministic
  again = engine.zonal_spectrum_lat_mean(xd, circ, wd, n_lat)
  assert torch.equal(got, again)


def test_latitude_mean_of_other_lengths_and_float64():
  import torch
  from weatherbench2_amd import engine, plan as plan_lib
  dev = torch.device('cuda')
  rs = np.random.RandomState(5)
  for n_lon, dtype in ((36, np.float32), (64, np.float64)):
    n_lat = 7
    lat = np.linspace(-60, 60, n_lat)
    lon = np.linspace(0, 360, n_lon, endpoint=False)
    x = rs.standard_normal((4, n_lat, n_lon)).astype(dtype)
    w = plan_lib.get_lat_weights(lat)
    spec, _, _ = spectrum_np.zonal_energy_spectrum(x, lat, lon, 1, 2)
    want = (spec * w[None, :, None]).sum(1) / w.sum()
    got = engine.zonal_spectrum_lat_mean(
        torch.as_tensor(x, device=dev),
        torch.as_tensor(spectrum_np.circumference(lat)).to(dev),
        torch.as_tensor(w, device=dev), n_lat)
    tol = 3e-6 if dtype == np.float32 else 1e-12
    scale = np.abs(want).max()
    np.testing.assert_allclose(got.cpu().numpy() / scale, want / scale, atol=tol)


def test_area_mean_dataset_api(dv):
  """derived_variables.zonal_energy_spectrum_area_mean == the weighted mean of
  ZonalEnergySpectrum.compute over latitude (any dim order of the input)."""
  from weatherbench2_amd import plan as plan_lib
  rs = np.random.RandomState(21)
  lat = np.linspace(-90, 90, 19)
  lon = np.linspace(0, 360, 64, endpoint=False)
  x = rs.standard_normal((3, 64, 19, 2)).astype(np.float32)
  ds = _dataset(x, ('time', 'longitude', 'latitude', 'level'), lat, lon)
  got = dv.zonal_energy_spectrum_area_mean(ds, 'z')
  assert got.dims == ('time', 'level', 'zonal_wavenumber')
  spec = dv.ZonalEnergySpectrum('z').compute(ds)   # (time, latitude, level, k)
  w = plan_lib.get_lat_weights(lat)
  want = (spec.transpose('time', 'level', 'latitude', 'zonal_wavenumber').values
          * w[None, None, :, None]).sum(2) / w.sum()
  np.testing.assert_allclose(got.values, want, rtol=1e-9,
                             atol=1e-9 * np.abs(want).max())


# ---- float64 rows through the fused kernel (complex128 LDS FFT) -------------
@pytest.mark.parametrize('n_lon', [64, 128, 240, 256, 360, 512, 720, 1024, 1440])
def test_fused_float64_matches_numpy(n_lon):
  """float64 rows of every instantiated length: the single-kernel path against
  np.fft.rfft (complex128, what the reference computes for float64 data:
  derived_variables.py:592-626) to 1e-12 of each row's total power -- all
  three modes (materialised, time mean fused, latitude mean fused)."""
  import torch
  from weatherbench2_amd import engine, plan as plan_lib
  dev = torch.device('cuda')
  rs = np.random.RandomState(n_lon)
  n_time, n_lev, n_lat = 3, 2, 11
  lat = np.linspace(-85, 85, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  x = rs.standard_normal((n_time, n_lev, n_lat, n_lon))
  want, _, _ = spectrum_np.zonal_energy_spectrum(x, lat, lon, 2, 3)
  circ = torch.as_tensor(spectrum_np.circumference(lat)).to(dev)
  xd = torch.as_tensor(x, device=dev)
  got = engine.zonal_spectrum(xd, circ, n_lat)
  assert got.dtype == torch.float64 and got.shape == want.shape
  assert _row_rel_err(got.cpu().numpy(), want) < 1e-12
  mean = engine.zonal_spectrum(xd, circ, n_lat, n_time=n_time, skipna=False)
  assert _row_rel_err(mean.cpu().numpy(), want.mean(0)) < 1e-12
  w = plan_lib.get_lat_weights(lat)
  lat_mean = engine.zonal_spectrum_lat_mean(xd, circ, torch.as_tensor(
      w, device=dev), n_lat)
  want_lm = (want * w[None, None, :, None]).sum(2) / w.sum()
  scale = np.abs(want_lm).max(axis=-1, keepdims=True)
  np.testing.assert_allclose(lat_mean.cpu().numpy() / scale, want_lm / scale,
                             atol=1e-12)


def test_fused_float64_full_size_and_nan_handling():
  """A 13 x 721 x 1440 float64 unit (16 B per point in, fp64 spectra out):
  Parseval per row, NaN rows with and without skipna, determinism."""
  import torch
  from weatherbench2_amd import engine
  dev = torch.device('cuda')
  gen = torch.Generator(device=dev).manual_seed(3)
  n_time, n_lat, n_lon = 2, 721, 1440
  x = torch.randn((n_time, 4, n_lat, n_lon), generator=gen, device=dev,
                  dtype=torch.float64)
  x[1, 2, 100, 7] = float('nan')
  lat = np.linspace(-90, 90, n_lat)
  circ = torch.ones(n_lat, dtype=torch.float64, device=dev)
  spec = engine.zonal_spectrum(x, circ, n_lat)
  again = engine.zonal_spectrum(x, circ, n_lat)
  assert torch.equal(torch.nan_to_num(spec, nan=-1.0),
                     torch.nan_to_num(again, nan=-1.0))
  # Parseval: sum of the (doubled) bins = mean square of the row (+ the Nyquist
  # bin counted twice, derived_variables.py:600)
  row = x[0, 0, 300]
  f = np.fft.rfft(row.cpu().numpy(), norm='forward')
  want = (np.abs(f) ** 2) * np.r_[1.0, np.full(n_lon // 2, 2.0)]
  np.testing.assert_allclose(spec[0, 0, 300].cpu().numpy(), want, rtol=1e-10,
                             atol=1e-15)
  assert torch.isnan(spec[1, 2, 100]).all()
  assert torch.isfinite(spec[1, 2, 99]).all()
  for skipna in (False, True):
    mean = engine.zonal_spectrum(x, circ, n_lat, n_time=n_time, skipna=skipna)
    ref = torch.nanmean(spec, 0) if skipna else spec.mean(0)
    torch.testing.assert_close(mean, ref, rtol=1e-12, atol=0, equal_nan=True)
  del lat


def test_paired_last_pass_variant_against_the_oracle(tmp_path):
  """WB2HIP_FFT_PAIRED=1 (measured, not the default: profiles/
  r06_measured_not_kept.md): the reducing modes of the 1440-point rows on the
  plan 2 x 20 x 6 x 6 whose last pass runs the butterflies j and T - j of a
  lane side by side, the recombination in registers (fft_core.hpp:
  PairedLast).  Another plan rounds differently in float32, so it is held to
  the oracle's tolerance, not to the materialising kernel's bits."""
  import os
  import subprocess
  import sys
  n_lon, n_lat, n_time, n_lev = 1440, 9, 4, 2
  code = (
      'import numpy as np, torch, sys\n'
      'sys.path.insert(0, %r)\n'
      'from weatherbench2_amd import engine\n'
      'from oracle import spectrum_np\n'
      'dev = torch.device("cuda")\n'
      'gen = torch.Generator(device=dev).manual_seed(17)\n'
      'x = torch.randn((%d, %d, %d, %d), generator=gen, device=dev)\n'
      'x[1, 0, 2, 5] = float("nan")\n'
      'lat = np.linspace(-80, 80, %d)\n'
      'circ = torch.as_tensor(spectrum_np.circumference(lat)).to(dev)\n'
      'w = torch.as_tensor(np.cos(np.deg2rad(lat))).to(dev)\n'
      'np.savez(sys.argv[1], x=x.cpu().numpy(),\n'
      '         mean=engine.zonal_spectrum(x, circ, %d, n_time=%d,\n'
      '                                    skipna=True).cpu().numpy(),\n'
      '         lat=engine.zonal_spectrum_lat_mean(x[0], circ, w, %d)\n'
      '         .cpu().numpy())\n'
  ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), n_time,
       n_lev, n_lat, n_lon, n_lat, n_lat, n_time, n_lat)
  out = str(tmp_path / 'paired.npz')
  env = dict(os.environ, WB2HIP_FFT_PAIRED='1')
  subprocess.run([sys.executable, '-c', code, out], check=True, env=env,
                 timeout=300)
  got = np.load(out)
  lat = np.linspace(-80, 80, n_lat)
  x = got['x'].astype(np.float64)
  spec = spectrum_np.simple_power(x) * spectrum_np.circumference(lat)[:, None]
  with np.errstate(all='ignore'):
    import warnings
    with warnings.catch_warnings():
      warnings.simplefilter('ignore')
      want_mean = np.nanmean(spec, axis=0)
  ok = ~np.isnan(want_mean).any(axis=-1)
  assert np.isnan(got['mean'][~ok]).all() == np.isnan(want_mean[~ok]).all()
  assert _row_rel_err(got['mean'][ok], want_mean[ok]) < 2e-6
  w = np.cos(np.deg2rad(lat))
  want_lat = (spec[0] * w[None, :, None]).sum(axis=1) / w.sum()
  ok = ~np.isnan(want_lat).any(axis=-1)
  assert ok.sum() >= 1
  assert _row_rel_err(got['lat'][ok], want_lat[ok]) < 2e-6


def test_unaligned_rows_take_the_late_hipfft_plan():
  """A plan of a length the one-kernel transform handles creates no hipFFT plan
  (rocFFT compiles kernels at plan creation: seconds); a call whose rows do not
  start on a 16-byte boundary cannot take that kernel and makes the hipFFT plan
  then -- with buffers of its own, not the caller's workspace."""
  import torch
  from weatherbench2_amd import engine
  dev = torch.device('cuda')
  n_lat, n_lon = 5, 240
  gen = torch.Generator(device=dev).manual_seed(77)
  flat = torch.randn(3 * n_lat * n_lon + 1, generator=gen, device=dev)
  x = flat[1:].view(3, n_lat, n_lon)
  assert x.data_ptr() % 16 != 0 and x.is_contiguous()
  lat = np.linspace(-60, 60, n_lat)
  circ = torch.as_tensor(spectrum_np.circumference(lat)).to(dev)
  want = spectrum_np.simple_power(x.cpu().numpy().astype(np.float64)) * (
      spectrum_np.circumference(lat)[None, :, None])
  for _ in range(2):   # the second call finds the late plan
    got = engine.zonal_spectrum(x, circ, n_lat).cpu().numpy()
    assert _row_rel_err(got, want) < 2e-6
  mean = engine.zonal_spectrum(x, circ, n_lat, 3, True).cpu().numpy()
  assert _row_rel_err(mean, want.mean(0)) < 2e-6
  # the aligned copy of the same rows goes through the one kernel, same plan
  fused = engine.zonal_spectrum(x.clone(), circ, n_lat).cpu().numpy()
  assert _row_rel_err(fused, got) < 2e-6
