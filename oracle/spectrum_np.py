"""NumPy restatement of ZonalEnergySpectrum (TEST INFRASTRUCTURE).

Follows /root/reference/weatherbench2/derived_variables.py:531-626.
"""
from __future__ import annotations

import numpy as np

EARTH_RADIUS_M = 1000 * (6357 + 6378) / 2  # schema.py:59


def circumference(latitude: np.ndarray) -> np.ndarray:
  """derived_variables.py:578-581."""
  circum_at_equator = 2 * np.pi * EARTH_RADIUS_M
  return np.cos(np.asarray(latitude) * np.pi / 180) * circum_at_equator


def lon_spacing_m(latitude: np.ndarray, longitude: np.ndarray) -> np.ndarray:
  """derived_variables.py:583-590."""
  diffs = np.diff(np.asarray(longitude))
  if np.max(np.abs(diffs - diffs[0])) > 1e-3:
    raise ValueError(f'Expected uniform longitude spacing. {longitude=}')
  return circumference(latitude) * diffs[0] / 360


def simple_power(f_x: np.ndarray) -> np.ndarray:
  """derived_variables.py:596-602 (last bin is doubled even for even N)."""
  f_k = np.fft.rfft(f_x, axis=-1, norm='forward')
  one_and_many_twos = np.concatenate(([1], [2] * (f_k.shape[-1] - 1)))
  return np.real(f_k * np.conj(f_k)) * one_and_many_twos


def zonal_energy_spectrum(x: np.ndarray, latitude: np.ndarray,
                          longitude: np.ndarray, lat_axis: int,
                          lon_axis: int):
  """Returns (spectrum, frequency[k, lat], wavelength[k, lat]).

  `spectrum` has x's dims with longitude moved LAST and renamed
  zonal_wavenumber (what xr.apply_ufunc does, derived_variables.py:604-609),
  multiplied by the circumference of each latitude circle (:624-626).
  """
  x = np.asarray(x)
  xm = np.moveaxis(x, lon_axis, -1)
  lat_ax_after = lat_axis if lat_axis < lon_axis else lat_axis - 1
  power = simple_power(xm)
  shape = [1] * power.ndim
  shape[lat_ax_after] = len(latitude)
  spectrum = power * circumference(latitude).reshape(shape)
  spacing = lon_spacing_m(latitude, longitude)
  with np.errstate(divide='ignore'):
    frequency = np.fft.rfftfreq(len(longitude))[:, None] / spacing[None, :]
    wavelength = 1 / frequency
  return spectrum, frequency, wavelength
