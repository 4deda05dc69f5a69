"""Staging of pageable host chunks (csrc/staging.cpp): the copy pool on its own
(CPU), and on the GPU the uploader round trip plus evaluate_chunks fed with
NumPy forecast chunks -- the form in which the Beam pipeline hands chunks over
(/root/reference/weatherbench2/evaluation.py:583-599, 693-705) -- bit-identical
to the device-resident run."""
import ctypes

import numpy as np
import pytest


def test_host_copy_pool_copies_every_byte():
  from weatherbench2_amd import _lib
  lib = _lib.load()
  rs = np.random.default_rng(0)
  src = rs.integers(0, 255, (9 << 20) + 77, dtype=np.uint8)
  for n in (0, 1, 4095, 4096, 4097, (4 << 20) - 1, (4 << 20) + 123, src.size):
    for threads in (1, 3, 8):
      dst = np.zeros(src.size + 64, dtype=np.uint8)
      assert lib.wb2_host_copy(dst.ctypes.data, src.ctypes.data, n,
                               threads) == 0
      assert np.array_equal(dst[:n], src[:n])
      assert not dst[n:].any()
  # unaligned destinations and sources (the streaming stores align themselves)
  for d_off, s_off in ((3, 0), (0, 5), (17, 9), (31, 33)):
    dst = np.zeros(src.size + 64, dtype=np.uint8)
    n = (5 << 20) + 11
    assert lib.wb2_host_copy(dst.ctypes.data + d_off, src.ctypes.data + s_off,
                             n, 4) == 0
    assert np.array_equal(dst[d_off:d_off + n], src[s_off:s_off + n])
    assert not dst[:d_off].any() and not dst[d_off + n:].any()
  assert lib.wb2_host_copy(None, src.ctypes.data, 16, 2) != 0
  assert lib.wb2_host_copy(src.ctypes.data, src.ctypes.data, -1, 2) != 0
  assert lib.wb2_host_copy(src.ctypes.data, src.ctypes.data, 16, 0) != 0


def test_uploader_needs_a_device_and_says_so():
  import torch
  from weatherbench2_amd import _lib
  if torch.cuda.is_available():
    pytest.skip('a HIP device is present')
  lib = _lib.load()
  handle = ctypes.c_void_p()
  assert lib.wb2_uploader_create(4, 1 << 20, 3, ctypes.byref(handle)) != 0
  assert b'pinned ring' in lib.wb2_last_error()
  assert lib.wb2_uploader_create(0, 1 << 20, 3, ctypes.byref(handle)) != 0
  assert lib.wb2_uploader_destroy(None) == 0


@pytest.mark.gpu
def test_upload_round_trip():
  import torch
  from weatherbench2_amd import feeder
  dev = torch.device('cuda')
  rs = np.random.default_rng(1)
  # below one slice, exactly one slice, several slices + a ragged tail
  for n in (1, 1000, feeder._SLICE_BYTES // 4,
            5 * feeder._SLICE_BYTES // 4 + 12345):
    a = rs.standard_normal(n, dtype=np.float32)
    got = feeder.upload(a, dev)
    assert got.dtype == torch.float32 and tuple(got.shape) == a.shape
    assert np.array_equal(got.cpu().numpy(), a)
  a = rs.standard_normal((3, 5, 7))
  assert np.array_equal(feeder.upload(a, dev).cpu().numpy(), a)
  many = [rs.standard_normal(n, dtype=np.float32)
          for n in (7, 0, 2_000_000, 9_000_001)] + [rs.standard_normal((4, 9))]
  for a, d in zip(many, feeder.upload_many(many, dev)):
    assert tuple(d.shape) == a.shape and np.array_equal(d.cpu().numpy(), a)
  # many uploads back to back: the ring is reused, destinations stay intact
  arrays = [rs.standard_normal(3_000_000, dtype=np.float32) for _ in range(6)]
  devs = [feeder.upload(a, dev) for a in arrays]
  torch.cuda.synchronize()
  for a, d in zip(arrays, devs):
    assert np.array_equal(d.cpu().numpy(), a)


@pytest.mark.gpu
def test_download_round_trip(monkeypatch):
  """Results leave through the same ring (wb2_uploader_download): every byte of
  the tensor, behind its producer on the current stream, between uploads."""
  import torch
  from weatherbench2_amd import feeder
  dev = torch.device('cuda')
  g = torch.Generator(device=dev).manual_seed(3)
  sl = feeder._SLICE_BYTES
  # the small ones take torch's own copy; then one slice, a ring and a bit,
  # three rounds of the ring with a ragged tail
  monkeypatch.setattr(feeder, '_DOWNLOAD_MIN_BYTES', 1 << 16)
  for nbytes in (8, 1 << 16, sl, feeder._RING_SLOTS * sl + 4096,
                 3 * feeder._RING_SLOTS * sl // 2 + 12344):
    x = torch.randn(nbytes // 8, dtype=torch.float64, device=dev, generator=g)
    y = x * 2.0 + 1.0          # produced on the current stream, not yet done
    got = feeder.download(y)
    assert got.dtype == np.float64 and got.shape == (nbytes // 8,)
    assert np.array_equal(got, y.cpu().numpy())
    del x, y
  # a strided view leaves as its contiguous copy; float32; 3-D
  x = torch.randn((5, 300, 4100), dtype=torch.float32, device=dev, generator=g)
  view = x[:, ::2, 1:4097]
  got = feeder.download(view)
  assert got.dtype == np.float32 and got.shape == (5, 150, 4096)
  assert np.array_equal(got, view.cpu().numpy())
  # between two uploads: the slots of the upload are waited for, the ring goes
  # on where the download left it
  rs = np.random.default_rng(5)
  a = rs.standard_normal(3 * sl // 4 + 17)
  up = feeder.upload(a, dev)
  got = feeder.download(x)
  up2 = feeder.upload(a, dev)
  torch.cuda.synchronize()
  assert np.array_equal(got, x.cpu().numpy())
  assert np.array_equal(up.cpu().numpy(), a)
  assert np.array_equal(up2.cpu().numpy(), a)
  # host tensors pass through
  assert np.array_equal(feeder.download(torch.arange(5.0)), np.arange(5.0))


@pytest.mark.gpu
@pytest.mark.parametrize('batch', [1, 3, None])
def test_host_fed_chunks_equal_device_resident_chunks(batch):
  """The same chunk list once device-resident, once with every forecast
  variable a pageable NumPy array (staged by the fetch thread): identical
  results, bit for bit."""
  import torch
  from tests import helpers, official_chunks as oc
  from weatherbench2_amd import config, evaluation, metrics as gm
  forecast, truth, clim = oc.make(n_init=3, n_lead=2, n_lat=61, n_lon=120)
  lat, lon = forecast.coords['latitude'], forecast.coords['longitude']
  oregions = oc.oracle_regions(lat, lon, oc.land_sea_mask(len(lat), len(lon)))
  gregions = {k: helpers.to_gpu_region(v) for k, v in oregions.items()}
  hf, ht, hc = (helpers.to_gpu_dataset(x) for x in (forecast, truth, clim))
  gf, gt, gc = (evaluation.make_resident(x) for x in (hf, ht, hc))
  cfg = config.Eval(metrics=oc.product_metrics(gm, gc), regions=gregions)
  chunks = oc.chunk_pairs(gf, gt)
  # forecast chunks as NumPy arrays, truth chunks device-resident
  fed = [(hfc, tc) for (hfc, _), (_, tc) in zip(oc.chunk_pairs(hf, ht), chunks)]
  assert all(isinstance(v.data, np.ndarray)
             for v in fed[0][0].data_vars.values())
  kwargs = {} if batch is None else {'batch_chunks': batch}
  want = evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0, **kwargs)
  old = evaluation._STAGE_MIN_BYTES
  evaluation._STAGE_MIN_BYTES = 1024   # these test chunks are small
  try:
    got = evaluation.evaluate_chunks(fed, cfg, False, prefetch=2, **kwargs)
  finally:
    evaluation._STAGE_MIN_BYTES = old
  assert sorted(got.data_vars) == sorted(want.data_vars)
  for name in want.data_vars:
    a, b = np.asarray(got[name].values), np.asarray(want[name].values)
    assert a.dtype == b.dtype and a.shape == b.shape
    assert np.array_equal(a, b, equal_nan=True), name
