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


@pytest.mark.parametrize('n_lon', [64, 128, 240, 256, 360, 512, 720, 1024, 1440])
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
