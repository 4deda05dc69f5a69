"""CPU-side checks of the C-ABI boundary and the host planning logic."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
  header = open(os.path.join(ROOT, 'include', 'wb2hip.h')).read()
  header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
  return sorted(set(re.findall(r'\b(wb2_[a-z0-9_]+)\s*\(', header)))


@pytest.fixture(scope='module')
def lib():
  from weatherbench2_amd import build
  build.build(verbose=False)
  from weatherbench2_amd import _lib
  return _lib


def test_library_loads_and_exports_every_declared_symbol(lib):
  handle = lib.load()
  declared = _declared_symbols()
  assert len(declared) >= 14
  for name in declared:
    assert hasattr(handle, name), f'{name} declared in wb2hip.h but not exported'
  assert sorted(lib.exported_symbols()) == declared
  assert handle.wb2_version() == 1


def test_pure_host_entry_points(lib):
  h = lib.load()
  assert h.wb2_num_slots(lib.MODE_DET, 0) == 3
  assert h.wb2_num_slots(lib.MODE_DET, 1) == 4
  assert h.wb2_num_slots(lib.MODE_DET_ACC, 0) == 6
  assert h.wb2_num_slots(lib.MODE_DET_ACC, 1) == 10
  assert h.wb2_num_slots(lib.MODE_WIND, 1) == 2
  assert h.wb2_ens_num_slots(0) == 6 and h.wb2_ens_num_slots(1) == 10
  assert h.wb2_tile_cols(lib.WB2_F32, 1440, 1) == 256
  # wide loads need one whole vector per row, not alignment (a row-end lane
  # shifts back): 721 latitudes last, unaligned views
  assert h.wb2_tile_cols(lib.WB2_F32, 1440, 0) == 256
  assert h.wb2_tile_cols(lib.WB2_F32, 721, 0) == 256
  assert h.wb2_tile_cols(lib.WB2_F32, 7, 1) == 256
  assert h.wb2_tile_cols(lib.WB2_F32, 3, 1) == 64
  assert h.wb2_tile_cols(lib.WB2_F64, 1, 1) == 64
  assert h.wb2_tile_cols(lib.WB2_F64, 1440, 1) == 128
  assert h.wb2_ens_tile_cols(1440) == 64
  assert h.wb2_num_slots(99, 0) < 0
  assert b'unknown mode' in h.wb2_last_error()


def test_argument_validation_without_a_gpu(lib):
  h = lib.load()
  # null pointers are rejected before anything touches the device
  rc = h.wb2_time_accumulate(None, 1, 1, 1, 0, None, None, None)
  assert rc < 0 and b'null pointer' in h.wb2_last_error()


def test_new_entry_points_validate_their_arguments(lib):
  """The round-5 entry points without a device: layout arithmetic, the struct
  and every precondition that is checked before anything is enqueued."""
  import ctypes
  h = lib.load()
  block, n_block, slots = (ctypes.c_int32(), ctypes.c_int32(),
                           ctypes.c_int32())
  ref = ctypes.byref
  # 50 members: 7 blocks of 8 (16 sums per lane; 32 with NaN skipping), blocks
  # of 4 when a weight field AND skipna double the sums twice
  assert h.wb2_energy_layout(50, 0, 0, ref(block), ref(n_block),
                             ref(slots)) == 0
  assert (block.value, n_block.value, slots.value) == (8, 7, 16)
  assert h.wb2_energy_layout(50, 1, 0, ref(block), ref(n_block),
                             ref(slots)) == 0
  assert (block.value, n_block.value, slots.value) == (8, 7, 32)
  assert h.wb2_energy_layout(50, 1, 1, ref(block), ref(n_block),
                             ref(slots)) == 0
  assert (block.value, n_block.value, slots.value) == (4, 13, 16)
  assert h.wb2_energy_layout(1, 0, 0, ref(block), ref(n_block),
                             ref(slots)) == 0 and n_block.value == 1
  assert h.wb2_energy_layout(0, 0, 0, ref(block), ref(n_block),
                             ref(slots)) < 0
  assert h.wb2_energy_layout(5, 0, 0, None, None, None) < 0
  # empty chunks are legal no-ops whatever the pointers are
  assert h.wb2_det_suite_step(ref(lib.PlanTables()), lib.MODE_DET, lib.WB2_F32,
                              0, None, None, 1, 0, None, None, 0, 0, 0, 0, None,
                              None, None, None) == 0
  assert h.wb2_energy_score(lib.WB2_F32, 0, None, None, None, None, 5, 0, 0,
                            ref(lib.PlanTables()), None, None, None, None) == 0
  assert h.wb2_gather_accumulate(None, None, None, 0, 3, 0, None, None,
                                 None) == 0
  # a null plan, missing outputs, a negative count
  rc = h.wb2_det_suite_step(None, lib.MODE_DET, lib.WB2_F32, 0, None, None, 1,
                            4, None, None, 0, 0, 0, 0, None, None, None, None)
  assert rc < 0 and b'null plan' in h.wb2_last_error()
  rc = h.wb2_det_suite_step(ref(lib.PlanTables()), lib.MODE_DET, lib.WB2_F32,
                            0, None, None, 1, 4, None, None, 0, 0, 0, 0, None,
                            None, None, None)
  assert rc < 0 and b'null pointer' in h.wb2_last_error()
  rc = h.wb2_energy_score(lib.WB2_F32, 0, None, None, None, None, 5, 0, 3,
                          ref(lib.PlanTables()), None, None, None, None)
  assert rc < 0 and b'null pointer' in h.wb2_last_error()
  rc = h.wb2_energy_score(7, 0, None, None, None, None, 5, 0, 3,
                          ref(lib.PlanTables()), None, None, None, None)
  assert rc < 0 and b'unknown dtype' in h.wb2_last_error()
  rc = h.wb2_gather_accumulate(None, None, None, 5, 1, 0, None, None, None)
  assert rc < 0 and b'null pointer' in h.wb2_last_error()
  assert h.wb2_gather_accumulate(None, None, None, -1, 1, 0, None, None,
                                 None) < 0
  assert h.wb2_uploader_upload(None, None, None, 16, None) < 0
  assert b'null uploader' in h.wb2_last_error()
  assert h.wb2_uploader_download(None, None, None, 16, None) < 0
  assert b'null uploader' in h.wb2_last_error()
  # the map entries: empty chunks are no-ops, the rest is checked up front
  assert h.wb2_spatial_accumulate_addr(lib.WB2_F32, 0, 1, None, None, 0, 5,
                                       100, None, None, None) == 0
  assert h.wb2_spatial_accumulate_addr(lib.WB2_F32, 0, 1, None, None, 2, 0,
                                       100, None, None, None) == 0
  rc = h.wb2_spatial_accumulate_addr(lib.WB2_F32, 0, 1, None, None, 2, 5, 100,
                                     None, None, None)
  assert rc < 0 and b'null pointer' in h.wb2_last_error()
  rc = h.wb2_spatial_accumulate_addr(9, 0, 1, None, None, 2, 5, 100, None,
                                     None, None)
  assert rc < 0 and b'unknown dtype' in h.wb2_last_error()
  rc = h.wb2_time_accumulate_runs(lib.WB2_F32, None, 2, 1, 8, 0, None, 0, None,
                                  None, None)
  assert rc < 0 and b'run=0' in h.wb2_last_error()
  assert h.wb2_time_accumulate_runs(lib.WB2_F32, None, 0, 1, 8, 0, None, 4,
                                    None, None, None) == 0
  # SEEPS maps of slabs given by address: same checks as wb2_seeps_map
  assert h.wb2_seeps_map_addr(lib.WB2_F32, None, 0, 100, None, 0.0, None,
                              None) == 0
  rc = h.wb2_seeps_map_addr(lib.WB2_F32, None, 4, 100, None, 0.0, None, None)
  assert rc < 0 and b'null pointer' in h.wb2_last_error()
  rc = h.wb2_seeps_map_addr(9, None, 4, 100, None, 0.0, None, None)
  assert rc < 0 and b'unknown dtype' in h.wb2_last_error()
  # ensemble slabs by address: no slabs is a no-op, tables are required
  assert h.wb2_ens_partials_addr(lib.WB2_F32, 0, None, None, 5, 100, 0, 9, 64,
                                 None, None, None, None, None, 8, 1, None,
                                 None, 1, 1, None, None) == 0
  rc = h.wb2_ens_partials_addr(lib.WB2_F32, 0, None, None, 5, 100, 3, 9, 64,
                               None, None, None, None, None, 8, 1, None, None,
                               1, 1, None, None)
  assert rc < 0 and b'null pointer' in h.wb2_last_error()
  # the struct is the header's: 8 int32, 3 pointers, 2 int32, pointer, double,
  # 9 pointers
  assert ctypes.sizeof(lib.PlanTables) == 8 * 4 + 3 * 8 + 2 * 4 + 8 + 8 + 9 * 8


def test_product_never_imports_the_oracle():
  pkg = os.path.join(ROOT, 'weatherbench2_amd')
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith('.py'):
        src = open(os.path.join(dirpath, f)).read()
        assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f


def test_no_gpu_means_loud_failure(lib):
  import torch
  if torch.cuda.is_available():
    pytest.skip('a GPU is visible')
  from weatherbench2_amd import engine
  with pytest.raises(lib.Wb2HipError):
    engine.require_gpu()


# ---------------------------------------------------------------------------
# host planning: region decomposition is index/mask work -> bit-exact vs oracle
# ---------------------------------------------------------------------------
def _oracle_region_weights(region, lat, lon):
  """Weights the reference would use, as a dense (lat, lon) array with
  multiplicities (rows/cols selected twice count twice)."""
  from oracle import metrics_np as om
  from oracle.named import DS, NA
  ds = DS({'x': NA(np.zeros((len(lat), len(lon))), ('latitude', 'longitude'))},
          {'latitude': lat, 'longitude': lon})
  w = om.get_lat_weights(lat)
  if region is None:
    full = np.broadcast_to(w.data[:, None], (len(lat), len(lon))).copy()
    return full
  ds2, w2 = region.apply(ds, w)
  lat2, lon2 = ds2.coord('latitude'), ds2.coord('longitude')
  ones = NA(np.ones((len(lat2), len(lon2))), ('latitude', 'longitude'))
  dense = (ones * w2).transpose('latitude', 'longitude').data
  full = np.zeros((len(lat), len(lon)))
  li = {v: i for i, v in enumerate(lat.tolist())}
  lj = {v: i for i, v in enumerate(lon.tolist())}
  for a, la in enumerate(lat2.tolist()):
    for b, lo in enumerate(lon2.tolist()):
      full[li[la], lj[lo]] += dense[a, b]
  return full


@pytest.mark.parametrize('res', [30.0, 5.625, 1.0])
def test_region_decomposition_matches_reference_semantics(res):
  from oracle import regions_np as oreg
  from oracle.named import NA
  from tests import helpers
  from weatherbench2_amd import plan as plan_lib
  from weatherbench2_amd import regions as greg
  n_lat = round(180 / res) + 1
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, round(360 / res), endpoint=False)
  rs = np.random.RandomState(1)
  lsm = np.clip(rs.rand(len(lat), len(lon)) * 1.4 - 0.2, 0, 1)
  oregions = dict(helpers.predefined_regions(oracle=True))
  oregions['none'] = None
  oregions['xt'] = oreg.ExtraTropicalRegion()
  oregions['overlap'] = oreg.SliceRegion(lat_slice=[slice(None, 10),
                                                    slice(-10, None)])
  oregions['land'] = oreg.LandRegion(NA(lsm, ('latitude', 'longitude')), lat,
                                     lon)
  oregions['tropics_land_thr'] = oreg.CombinedRegion([
      oreg.SliceRegion(lat_slice=slice(-20, 20)),
      oreg.LandRegion(NA(lsm, ('latitude', 'longitude')), lat, lon, 0.5)])
  land_only = {k: v for k, v in oregions.items()
               if k in ('land', 'tropics_land_thr')}
  plain = {k: v for k, v in oregions.items() if k not in land_only}
  for group in (plain, {**plain, 'land': oregions['land']},
                {'tropics_land_thr': oregions['tropics_land_thr']}):
    gregions = {k: helpers.to_gpu_region(v) for k, v in group.items()}
    for layout in (plan_lib.LATLON, plan_lib.LONLAT):
      pl = plan_lib.build_plan(lat, lon, layout, gregions, 'cpu',
                               rows_per_chunk=5)
      w_row = pl.w_row.numpy()
      w_col = np.ones(pl.n_col) if pl.w_col is None else pl.w_col.numpy()
      field = None if pl.wfield is None else pl.wfield.numpy()
      band_of_row = np.repeat(np.arange(pl.n_band), np.diff(pl.band_row0))
      seg_of_col = np.repeat(np.arange(pl.n_seg), np.diff(pl.seg_col0_host))
      cb, cs = pl.coef_band.numpy(), pl.coef_seg.numpy()
      for i, name in enumerate(pl.region_names):
        dense = (cb[i][band_of_row][:, None] * cs[i][seg_of_col][None, :]
                 * w_row[:, None] * w_col[None, :])
        if pl.region_wf.numpy()[i]:
          dense = dense * field
        if layout == plan_lib.LONLAT:
          dense = dense.T
        want = _oracle_region_weights(group[name], lat, lon)
        np.testing.assert_array_equal(dense != 0, want != 0, err_msg=name)
        np.testing.assert_allclose(dense, want, rtol=1e-15, atol=0,
                                   err_msg=name)
        np.testing.assert_allclose(pl.region_wsum_host[i], want.sum(),
                                   rtol=1e-12)
      # chunks tile every band exactly once, padded to a multiple of 8
      assert pl.n_chunk % 8 == 0
      rows = np.concatenate([np.arange(r, r + n) for r, n in
                             zip(pl.chunk_row0_host, pl.chunk_nrow_host)])
      np.testing.assert_array_equal(rows, np.arange(pl.n_row))


def test_lat_weights_follow_coordinate_dtype():
  from oracle import metrics_np as om
  from weatherbench2_amd import plan as plan_lib
  for dtype in (np.float64, np.float32):
    lat = np.linspace(-90, 90, 721).astype(dtype)
    got = plan_lib.get_lat_weights(lat)
    want = om.get_lat_weights(lat).data
    assert got.dtype == dtype
    np.testing.assert_array_equal(got, want)
  with pytest.raises(ValueError):
    plan_lib.get_lat_weights(np.array([10.0, 0.0, -10.0]))


def test_seg_entry_tables():
  from weatherbench2_amd import plan as plan_lib
  from tests import helpers
  lat = np.linspace(-90, 90, 721)
  lon = np.linspace(0, 360, 1440, endpoint=False)
  pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON,
                           helpers.predefined_regions(oracle=False), 'cpu')
  for tile in (64, 128, 256):
    eoff, n_ts = pl.seg_entries(tile)
    eoff = eoff.numpy()
    c = pl.seg_col0_host
    count = 0
    for s in range(pl.n_seg):
      tiles = set(range(c[s] // tile, (c[s + 1] - 1) // tile + 1))
      assert eoff[s + 1] - eoff[s] == len(tiles)
      count += len(tiles)
    assert n_ts == count


def test_lat_weights_export_matches_numpy():
  """wb2_lat_weights (host code, no GPU): metrics.py:35-60 in C for non-Python
  callers; the reference's known answer (metrics_test.py:63-82) and NumPy."""
  import ctypes
  from weatherbench2_amd import _lib, plan
  lib = _lib.load()
  lat = np.array([-75.0, -45.0, -15.0, 15.0, 45.0, 75.0])
  out = np.empty_like(lat)
  assert lib.wb2_lat_weights(_lib.WB2_F64, lat.ctypes.data, lat.size,
                             out.ctypes.data) == 0
  s3 = np.sqrt(3.0)
  np.testing.assert_allclose(
      out, 3 * np.array([1 - s3 / 2, (s3 - 1) / 2, 0.5, 0.5, (s3 - 1) / 2,
                         1 - s3 / 2]), rtol=1e-14)
  for lat in (np.linspace(-90, 90, 721), np.linspace(-87.1875, 87.1875, 32)):
    out = np.empty_like(lat)
    assert lib.wb2_lat_weights(_lib.WB2_F64, lat.ctypes.data, lat.size,
                               out.ctypes.data) == 0
    np.testing.assert_allclose(out, plan.get_lat_weights(lat), rtol=1e-13)
    lat32 = lat.astype(np.float32)
    out32 = np.empty_like(lat32)
    assert lib.wb2_lat_weights(_lib.WB2_F32, lat32.ctypes.data, lat32.size,
                               out32.ctypes.data) == 0
    assert out32.dtype == np.float32
    np.testing.assert_allclose(out32, plan.get_lat_weights(lat32), rtol=1e-4,
                               atol=1e-7)
  # decreasing latitudes are an error, like the reference's assertion
  dec = np.linspace(90, -90, 5)
  assert lib.wb2_lat_weights(_lib.WB2_F64, dec.ctypes.data, 5,
                             np.empty(5).ctypes.data) < 0
  assert b'not increasing' in lib.wb2_last_error()


def test_gather_pointers_address_every_slab_and_route_holes():
  """engine.gather_pointers (host logic of wb2_ens_partials_gather): slab i of
  a contiguous base sits at base + i * slab bytes; holes (-1) point at the
  resident NaN slab of that size -- checked here on a CPU tensor by reading the
  addresses back."""
  import ctypes as c
  import torch
  from weatherbench2_amd import engine
  n_row, n_col = 3, 5
  base = torch.arange(7 * n_row * n_col, dtype=torch.float32).reshape(7, n_row,
                                                                      n_col)
  index = np.array([[6, -1, 0], [2, 2, -1]], dtype=np.int64)
  ptrs = engine.gather_pointers(base, index, n_row * n_col)
  assert ptrs.shape == index.shape and ptrs.dtype == np.int64
  for (o, m), i in np.ndenumerate(index):
    got = np.ctypeslib.as_array(
        c.cast(int(ptrs[o, m]), c.POINTER(c.c_float)), shape=(n_row * n_col,))
    if i < 0:
      assert np.isnan(got).all()
    else:
      np.testing.assert_array_equal(got, base[i].numpy().ravel())
  # one NaN slab per (device, dtype, size), shared by every hole
  assert ptrs[0, 1] == ptrs[1, 2]


def test_gather_entry_point_validates_its_arguments(lib):
  h = lib.load()
  rc = h.wb2_ens_partials_gather(0, 0, None, None, None, 5, 1, 4, 4, None, None,
                                 None, None, None, 8, 1, None, None, 1, 1, None,
                                 None, None)
  assert rc < 0 and b'null pointer' in h.wb2_last_error()


def test_pair_and_program_entry_points_validate_without_a_gpu(lib):
  """wb2_pairs_supported is pure host logic; a chunk program (csrc/program.cpp)
  is host state until wb2_program_finalize: its builders check slots, ranges
  and order before any device call."""
  h = lib.load()
  # pair kernels exist at the full vector width of DET / DET_ACC only
  assert h.wb2_pairs_supported(lib.MODE_DET_ACC, lib.WB2_F32, 0, 1, 1440, 1) == 1
  assert h.wb2_pairs_supported(lib.MODE_DET, lib.WB2_F64, 1, 0, 721, 0) == 1
  assert h.wb2_pairs_supported(lib.MODE_DET, lib.WB2_F32, 0, 0, 3, 1) == 0
  assert h.wb2_pairs_supported(lib.MODE_WIND, lib.WB2_F32, 0, 0, 1440, 1) == 0
  assert h.wb2_pairs_supported(lib.MODE_SEEPS, lib.WB2_F32, 0, 0, 1440, 1) == 0
  # n_pair must fit the launch; null tables are rejected before the device
  assert h.wb2_stream_partials_pairs(
      lib.MODE_DET, lib.WB2_F32, 0, None, None, 1, 4, 1, 8, 16, None, None,
      None, lib.WB2_F64, None, None, 8, 1, None, None, 1, 1, None, None,
      None) < 0
  assert b'null pointer' in h.wb2_last_error()
  assert h.wb2_det_wind_suite_step(None, lib.MODE_DET, lib.WB2_F32, 0, None,
                                   None, 1, 4, 1, None, None, None, None,
                                   None) < 0
  assert b'null plan' in h.wb2_last_error()

  prog = ctypes.c_void_p()
  assert h.wb2_program_create(ctypes.byref(prog)) == 0 and prog.value
  assert h.wb2_program_create(None) < 0
  plan = lib.PlanTables(n_row=8, n_col=16, n_chunk=8, n_ctile=1, n_seg=1,
                        n_ts=1, n_band=1, n_region=1)
  slot = np.zeros(2 * 3, dtype=np.int32)
  rel = np.zeros(2 * 3, dtype=np.int64)
  scratch = ctypes.create_string_buffer(64)
  part = ctypes.addressof(scratch)
  # a gather needs a launch; a launch needs its tables and sane sizes
  assert h.wb2_program_add_gather(prog, 0, 0, 1, 0, 8, 0, slot.ctypes.data,
                                  rel.ctypes.data) < 0
  assert b'add a launch first' in h.wb2_last_error()
  assert h.wb2_program_add_launch(prog, ctypes.byref(plan), lib.MODE_DET,
                                  lib.WB2_F32, 0, 2, 3, 0, None, None, part,
                                  None, 0, 0) < 0
  assert h.wb2_program_add_launch(prog, ctypes.byref(plan), lib.MODE_DET,
                                  lib.WB2_F32, 0, 2, 3, 2, slot.ctypes.data,
                                  rel.ctypes.data, part, None, 0, 0) < 0
  assert b'bad sizes' in h.wb2_last_error()
  assert h.wb2_program_add_launch(prog, ctypes.byref(plan), lib.MODE_DET,
                                  lib.WB2_F32, 0, 2, 3, 1, slot.ctypes.data,
                                  rel.ctypes.data, part, None, 0, 0) < 0
  assert b'wind_partials' in h.wb2_last_error()
  slot[4] = -1
  assert h.wb2_program_add_launch(prog, ctypes.byref(plan), lib.MODE_DET,
                                  lib.WB2_F32, 0, 2, 3, 0, slot.ctypes.data,
                                  rel.ctypes.data, part, None, 0, 0) < 0
  assert b'negative pointer slot' in h.wb2_last_error()
  slot[4] = 5
  assert h.wb2_program_add_launch(prog, ctypes.byref(plan), lib.MODE_DET,
                                  lib.WB2_F32, 0, 2, 3, 0, slot.ctypes.data,
                                  rel.ctypes.data, part, None, 0, 0) == 0
  # an ensemble pass: member count, stride, tables and a float64 field
  eslot = np.zeros(2 * 3, dtype=np.int32)
  assert h.wb2_program_add_ens_launch(prog, ctypes.byref(plan), lib.WB2_F32, 0,
                                      0, 128, 3, eslot.ctypes.data,
                                      rel.ctypes.data, part, 0) < 0
  assert b'bad sizes' in h.wb2_last_error()
  assert h.wb2_program_add_ens_launch(prog, ctypes.byref(plan), lib.WB2_F32, 0,
                                      5, 128, 3, None, rel.ctypes.data, part,
                                      0) < 0
  assert b'null pointer' in h.wb2_last_error()
  field32 = lib.PlanTables(n_row=8, n_col=16, n_chunk=8, n_ctile=1, n_seg=1,
                           n_ts=1, n_band=1, n_region=1, wfield=part,
                           wfield_dtype=lib.WB2_F32)
  assert h.wb2_program_add_ens_launch(prog, ctypes.byref(field32), lib.WB2_F32,
                                      0, 5, 128, 3, eslot.ctypes.data,
                                      rel.ctypes.data, part, 0) < 0
  assert b'float64 weight field' in h.wb2_last_error()
  cell = np.array([0, 7], dtype=np.int32)
  base = np.zeros(2, dtype=np.int64)
  assert h.wb2_program_add_gather(prog, 1, 1, 2, 0, 8, 0, cell.ctypes.data,
                                  base.ctypes.data) == 0
  assert h.wb2_program_add_gather(prog, 2, 0, 1, 0, 8, 0, cell.ctypes.data,
                                  base.ctypes.data) < 0   # input 2 of 2
  assert h.wb2_program_add_gather(prog, 0, 2, 2, 0, 8, 0, cell.ctypes.data,
                                  base.ctypes.data) < 0   # past the launch
  assert h.wb2_program_add_sink(prog, None, None, 1, 1, 0, None, None, 0) < 0
  # finalize checks every slot and cell against what a replay will bring
  assert h.wb2_program_finalize(prog, part, 4, 8) < 0
  assert b'pointer slot 5' in h.wb2_last_error()
  assert h.wb2_program_finalize(prog, part, 6, 4) < 0
  assert b'gather cell' in h.wb2_last_error()
  # a replay before finalize is refused
  ptrs = np.zeros(6, dtype=np.int64)
  args = np.zeros(3, dtype=np.int64)
  assert h.wb2_program_replay(prog, ptrs.ctypes.data, 6, None, 0,
                              args.ctypes.data, None, 0, None) < 0
  assert b'finalize the program first' in h.wb2_last_error()
  assert h.wb2_program_destroy(prog) == 0
  assert h.wb2_program_destroy(None) == 0
