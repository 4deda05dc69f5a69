"""Evaluation configuration: field-compatible with weatherbench2/config.py.

`Eval` (config.py:96-137) is the plug-in point of the reference: its `metrics`
and `regions` dicts carry the operator objects into the metric x region loop.
The dataclasses below keep the reference's field names and defaults so that an
existing configuration can be built unchanged with GPU metric objects as the
values of `Eval.metrics`.  `Selection` / `Paths` / `Data` (config.py:28-93)
describe dataset opening, which stays on the host and is out of scope here;
they are mirrored only so `Data`-carrying code keeps importing.
"""
from __future__ import annotations

import dataclasses
import typing as t


@dataclasses.dataclass
class Selection:
  """Which variables / levels / times to evaluate (config.py:28-55)."""

  variables: t.Sequence[str]
  time_slice: slice
  levels: t.Optional[t.Sequence[int]] = None
  lat_slice: t.Optional[slice] = dataclasses.field(
      default_factory=lambda: slice(None, None))
  lon_slice: t.Optional[slice] = dataclasses.field(
      default_factory=lambda: slice(None, None))
  aux_variables: t.Optional[t.Sequence[str]] = None


@dataclasses.dataclass
class Paths:
  """Dataset locations (config.py:58-74)."""

  forecast: str
  obs: str
  output_dir: str
  output_file_prefix: t.Optional[str] = ''
  climatology: t.Optional[str] = None


@dataclasses.dataclass
class Data:
  """Data configuration (config.py:77-93)."""

  selection: Selection
  paths: Paths
  by_init: t.Optional[bool] = True
  rename_variables: t.Optional[t.Dict[str, str]] = None
  pressure_level_suffixes: t.Optional[bool] = False


@dataclasses.dataclass
class Eval:
  """Evaluation configuration (config.py:96-137).

  metrics: {name: Metric}; regions: optional {name: Region};
  derived_variables: {name: DerivedVariable} computed on the fly;
  temporal_mean: average metrics over time / init_time.
  """

  metrics: t.Dict[str, t.Any]
  regions: t.Optional[t.Dict[str, t.Any]] = None
  evaluate_persistence: t.Optional[bool] = False
  evaluate_climatology: t.Optional[bool] = False
  evaluate_probabilistic_climatology: t.Optional[bool] = False
  probabilistic_climatology_start_year: t.Optional[int] = None
  probabilistic_climatology_end_year: t.Optional[int] = None
  probabilistic_climatology_hour_interval: t.Optional[int] = None
  against_analysis: t.Optional[bool] = False
  derived_variables: t.Dict[str, t.Any] = dataclasses.field(
      default_factory=dict)
  temporal_mean: t.Optional[bool] = True
  output_format: str = 'netcdf'
