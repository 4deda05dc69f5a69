"""ctypes binding of libwb2hip.so (the C ABI declared in include/wb2hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails the
product path raises.  torch is imported first on purpose: it ships its own
libamdhip64.so.7 and the dynamic linker then resolves our DT_NEEDED entry to
that already-loaded runtime, so torch tensors' device pointers and our kernels
live in ONE HIP runtime / context.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

from weatherbench2_amd import build as _build

WB2_F32, WB2_F64 = 0, 1
MODE_DET, MODE_DET_ACC, MODE_WIND, MODE_ENS, MODE_GAUSS = 0, 1, 2, 3, 4
MODE_GAUSS_THR, MODE_ENS_THR, MODE_SEEPS = 5, 6, 7
GENERIC_KQ = {MODE_GAUSS: 2, MODE_GAUSS_THR: 3, MODE_ENS_THR: 4,
              MODE_SEEPS: 1}
NMETRIC = 5
NMETRIC_ENS = 8
METRIC_INDEX = {'mse': 0, 'rmse': 1, 'mae': 2, 'bias': 3, 'acc': 4}
ENS_METRIC_INDEX = {'crps': 0, 'crps_spread': 1, 'crps_skill': 2,
                    'ensemble_mean_mse': 3, 'ensemble_mean_rmse': 4,
                    'ensemble_variance': 5, 'ensemble_stddev': 6,
                    'debiased_ensemble_mean_mse': 7}

_c = ctypes
_vp, _i32, _i64, _int = _c.c_void_p, _c.c_int32, _c.c_int64, _c.c_int
_u64 = _c.c_uint64



class PlanTables(_c.Structure):
  """wb2_plan_tables of include/wb2hip.h (host struct of device pointers)."""
  _fields_ = [
      ('n_row', _i32), ('n_col', _i32), ('n_chunk', _i32), ('n_ctile', _i32),
      ('n_seg', _i32), ('n_ts', _i32), ('n_band', _i32), ('n_region', _i32),
      ('w_row', _vp), ('w_col', _vp), ('wfield', _vp),
      ('wfield_dtype', _i32), ('reserved', _i32), ('aux', _vp),
      ('scalar', _c.c_double), ('chunk_row0', _vp), ('chunk_nrow', _vp),
      ('seg_col0', _vp), ('seg_eoff', _vp), ('band_chunk0', _vp),
      ('coef_band', _vp), ('coef_seg', _vp), ('region_wf', _vp),
      ('region_wsum', _vp)]


_SIGNATURES = {
    'wb2_version': (_int, []),
    'wb2_last_error': (_c.c_char_p, []),
    'wb2_num_slots': (_int, [_int, _int]),
    'wb2_tile_cols': (_int, [_int, _int, _int]),
    'wb2_tile_cols_ex': (_int, [_int, _int, _int, _int, _int, _int]),
    'wb2_stream_partials': (_int, [
        _int, _int, _int, _c.POINTER(_vp), _c.POINTER(_vp), _i64, _i32, _i32,
        _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    'wb2_stream_partials_ex': (_int, [
        _int, _int, _int, _c.POINTER(_vp), _c.POINTER(_vp), _i64, _i32, _i32,
        _vp, _vp, _vp, _int, _vp, _c.c_double, _vp, _vp, _i32, _i32, _vp, _vp,
        _i32, _i32, _vp, _vp]),
    'wb2_stream_partials_addr': (_int, [
        _int, _int, _int, _c.POINTER(_vp), _int, _i64, _i32, _i32,
        _vp, _vp, _vp, _int, _vp, _c.c_double, _vp, _vp, _i32, _i32, _vp, _vp,
        _i32, _i32, _vp, _vp]),
    'wb2_pairs_supported': (_int, [_int, _int, _int, _int, _int, _int]),
    'wb2_stream_partials_pairs': (_int, [
        _int, _int, _int, _c.POINTER(_vp), _c.POINTER(_vp), _int, _i64, _i64,
        _i32, _i32, _vp, _vp, _vp, _int, _vp, _vp, _i32, _i32, _vp, _vp, _i32,
        _i32, _vp, _vp, _vp]),
    'wb2_det_wind_suite_step': (_int, [
        _c.POINTER(PlanTables), _int, _int, _int, _c.POINTER(_vp),
        _c.POINTER(_vp), _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    'wb2_det_combine': (_int, [
        _int, _int, _vp, _i64, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp,
        _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    'wb2_det_suite_step': (_int, [
        _c.POINTER(PlanTables), _int, _int, _int, _c.POINTER(_vp),
        _c.POINTER(_vp), _int, _i64, _vp, _vp, _i64, _i64, _i64, _int, _vp,
        _vp, _vp, _vp]),
    'wb2_gather_accumulate': (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp, _vp,
                                     _vp]),
    'wb2_gather_accumulate_rows': (_int, [_vp, _vp, _vp, _i64, _i64, _int,
                                          _vp, _vp, _vp, _vp, _vp, _vp]),
    'wb2_program_create': (_int, [_c.POINTER(_vp)]),
    'wb2_program_destroy': (_int, [_vp]),
    'wb2_program_add_launch': (_int, [
        _vp, _c.POINTER(PlanTables), _int, _int, _int, _i32, _i64, _i64, _vp,
        _vp, _vp, _vp, _i64, _int]),
    'wb2_program_add_ens_launch': (_int, [
        _vp, _c.POINTER(PlanTables), _int, _int, _i32, _i64, _i64, _vp, _vp,
        _vp, _i64]),
    'wb2_program_add_gather': (_int, [_vp, _i32, _i64, _i64, _i32, _i64, _i32,
                                      _vp, _vp]),
    'wb2_program_add_sink': (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp, _vp,
                                    _i64]),
    'wb2_program_finalize': (_int, [_vp, _vp, _i32, _i32]),
    'wb2_program_replay': (_int, [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _i32,
                                  _vp]),
    'wb2_program_stats': (_int, [_vp, _c.POINTER(_c.c_double),
                                 _c.POINTER(_i64)]),
    'wb2_energy_layout': (_int, [_i32, _int, _int, _c.POINTER(_i32),
                                 _c.POINTER(_i32), _c.POINTER(_i32)]),
    'wb2_energy_score': (_int, [
        _int, _int, _vp, _vp, _vp, _vp, _i32, _i64, _i64,
        _c.POINTER(PlanTables), _vp, _vp, _vp, _vp]),
    'wb2_time_accumulate': (_int, [_vp, _i64, _i64, _i64, _int, _vp, _vp,
                                   _vp]),
    'wb2_time_accumulate_scatter': (_int, [_int, _vp, _i64, _i64, _i64, _int,
                                           _vp, _vp, _vp, _vp]),
    'wb2_time_accumulate_runs': (_int, [_int, _vp, _i64, _i64, _i64, _int,
                                        _vp, _i64, _vp, _vp, _vp]),
    'wb2_ens_partials': (_int, [
        _int, _int, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _vp, _vp,
        _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    'wb2_ens_partials_maps': (_int, [
        _int, _int, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _vp, _vp,
        _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    'wb2_ens_partials_addr': (_int, [
        _int, _int, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp,
        _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    'wb2_ens_partials_gather': (_int, [
        _int, _int, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp,
        _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    'wb2_ens_threshold_partials': (_int, [
        _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32,
        _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    'wb2_ens_threshold_maps': (_int, [
        _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _vp,
        _vp]),
    'wb2_seeps_map': (_int, [_int, _c.POINTER(_vp), _c.POINTER(_vp), _i64, _i64,
                             _vp, _c.c_double, _vp, _vp]),
    'wb2_seeps_map_addr': (_int, [_int, _c.POINTER(_vp), _i64, _i64, _vp,
                                  _c.c_double, _vp, _vp]),
    'wb2_axis_moments_splits': (_int, [_i64, _i64, _i64, _i64]),
    'wb2_axis_moments': (_int, [_int, _vp, _i64, _i64, _i64, _vp, _i64, _int,
                                _int, _vp, _vp, _vp, _vp, _vp]),
    'wb2_rank_histogram': (_int, [
        _int, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _int, _u64, _vp,
        _vp, _vp]),
    'wb2_rank_histogram_seeded': (_int, [
        _int, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _i32,
        _c.POINTER(_u64), _vp, _c.POINTER(_i64), _vp, _vp, _vp]),
    'wb2_rank_histogram_mean': (_int, [
        _int, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i64, _i64, _i32, _i32,
        _int, _u64, _c.POINTER(_u64), _vp, _c.POINTER(_i64), _int, _vp, _vp]),
    'wb2_ens_combine': (_int, [
        _int, _vp, _i64, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp,
        _vp, _vp, _i32, _vp, _vp, _vp]),
    'wb2_ens_num_slots': (_int, [_int]),
    'wb2_ens_tile_cols': (_int, [_i32]),
    'wb2_spatial_maps': (_int, [_int, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp,
                                _vp, _vp]),
    'wb2_spatial_accumulate': (_int, [_int, _int, _vp, _vp, _vp, _vp, _i64,
                                      _i64, _i64, _vp, _vp, _vp]),
    'wb2_spatial_accumulate_addr': (_int, [_int, _int, _int, _vp, _vp, _i64,
                                           _i64, _i64, _vp, _vp, _vp]),
    'wb2_spectrum_plan_create': (_int, [_int, _i32, _i64, _c.POINTER(_vp)]),
    'wb2_spectrum_plan_destroy': (_int, [_vp]),
    'wb2_spectrum_plan_workspace': (_i64, [_vp]),
    'wb2_zonal_spectrum': (_int, [_vp, _vp, _vp, _i32, _i64, _int, _vp, _vp,
                                  _vp]),
    'wb2_zonal_spectrum_latmean_segments': (_int, [_vp, _i32]),
    'wb2_zonal_spectrum_latmean': (_int, [_vp, _vp, _vp, _i32, _i32,
                                          _c.c_double, _vp, _vp, _vp]),
    'wb2_lat_weights': (_int, [_int, _vp, _i64, _vp]),
    'wb2_uploader_create': (_int, [_i32, _i64, _i32, _c.POINTER(_vp)]),
    'wb2_uploader_destroy': (_int, [_vp]),
    'wb2_host_copy': (_int, [_vp, _vp, _i64, _i32]),
    'wb2_uploader_upload': (_int, [_vp, _vp, _vp, _i64, _vp]),
    'wb2_uploader_download': (_int, [_vp, _vp, _vp, _i64, _vp]),
    'wb2_uploader_upload_many': (_int, [_vp, _i32, _c.POINTER(_vp),
                                        _c.POINTER(_vp), _c.POINTER(_i64),
                                        _vp]),
    'wb2_comm_unique_id': (_int, [_vp]),
    'wb2_comm_init_rank': (_int, [_vp, _i32, _i32, _c.POINTER(_vp)]),
    'wb2_comm_destroy': (_int, [_vp]),
    'wb2_time_mean_allreduce': (_int, [_vp, _vp, _i64, _vp, _vp]),
}

_lib = None


class Wb2HipError(RuntimeError):
  pass


def lib_path() -> str:
  # WB2HIP_LIB selects an alternative build (kernel-tuning sweeps only).
  return os.environ.get('WB2HIP_LIB') or _build.LIB_PATH


def load():
  """Loads (once) and returns the ctypes handle; raises if unavailable."""
  global _lib
  if _lib is not None:
    return _lib
  path = lib_path()
  if not os.path.exists(path):
    raise Wb2HipError(
        f'{path} is missing: run `python -c "import __graft_entry__ as g; '
        'g.build()"` (hipcc --offload-arch=gfx950). There is no CPU fallback.')
  lib = ctypes.CDLL(path)
  for name, (restype, argtypes) in _SIGNATURES.items():
    try:
      fn = getattr(lib, name)
    except AttributeError as e:
      raise Wb2HipError(f'{path} does not export {name}: stale build? '
                        'rebuild with __graft_entry__.build()') from e
    fn.restype = restype
    fn.argtypes = argtypes
  if lib.wb2_version() != 1:
    raise Wb2HipError(f'{path}: unexpected ABI version {lib.wb2_version()}')
  _lib = lib
  return lib


def exported_symbols() -> list[str]:
  return sorted(_SIGNATURES)


def check(status: int, what: str):
  if status != 0:
    msg = load().wb2_last_error().decode(errors='replace')
    raise Wb2HipError(f'{what} failed ({status}): {msg}')


def ptr(t) -> int:
  """Device (or host) address of a torch tensor / None."""
  return 0 if t is None else t.data_ptr()


def ptr_array(tensors):
  """void*[n] of tensors' addresses (None -> NULL; plain ints are taken as
  addresses: slices of a table the caller keeps alive)."""
  arr = (_vp * len(tensors))()
  for i, t in enumerate(tensors):
    arr[i] = (t if isinstance(t, int) else ptr(t)) or None
  return arr
