"""`config.Eval`: the plug-in point of the metric x region loop.

The reference's `Eval` (weatherbench2/config.py:96-137) carries the operator
objects -- `metrics: {name: Metric}`, `regions: {name: Region}`,
`derived_variables: {name: DerivedVariable}` -- into
`evaluation._metric_and_region_loop`, and the baseline switches
(`evaluate_climatology`, `evaluate_persistence`,
`evaluate_probabilistic_climatology`) into `evaluation._evaluate_all_metrics`
(evaluation.py:452-472).  The dataclass below has the same field names, order
and defaults, so an existing configuration is built unchanged with GPU metric
objects as the values of `metrics`; a reference `config.Eval` instance works
just as well (the loop only reads attributes).

Who reads what here:
  metrics, regions, derived_variables, temporal_mean
      evaluation._metric_and_region_loop / evaluate_chunks
  evaluate_climatology, evaluate_probabilistic_climatology (+ its three
  parameters), evaluate_persistence
      evaluation._evaluate_all_metrics (forecast replaced by a gather from the
      climatology / the truth, as zero-copy slab tables)
  against_analysis, output_format
      belong to dataset opening / writing (evaluation.py:208-334, 381-384),
      which is out of scope: carried, never read.

`Data` is the slice of the reference's `config.Data` (config.py:74-93) that the
compute path reads -- `by_init` -- for callers of `_evaluate_all_metrics`; the
path / selection half of it belongs to the IO layer.
"""
from __future__ import annotations

import dataclasses
import typing as t


@dataclasses.dataclass
class Eval:
  """Evaluation configuration (field-compatible with weatherbench2
  config.py:96-137).

  Attributes:
    metrics: {name: Metric} evaluated on every chunk.
    regions: optional {name: Region}; results get a `region` dim.
    evaluate_persistence: evaluate the persistence forecast (truth at the
      initialisation time) instead of the forecast.
    evaluate_climatology: evaluate the climatology, gathered by (dayofyear,
      hour) of the valid time, instead of the forecast.
    evaluate_probabilistic_climatology: evaluate an ensemble whose members are
      the years of the ground truth.
    probabilistic_climatology_start_year: first year of that ensemble.
    probabilistic_climatology_end_year: last year of that ensemble.
    probabilistic_climatology_hour_interval: hours between its `hour` labels.
    against_analysis: (dataset opening; carried, not read here).
    derived_variables: {name: DerivedVariable} computed on the fly and assigned
      into forecast / truth before the metrics.
    temporal_mean: average the metrics over time / init_time.
    output_format: (result writing; carried, not read here).
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


@dataclasses.dataclass
class Data:
  """What `_evaluate_all_metrics` reads of the reference's `config.Data`
  (config.py:74-93): the time convention of the forecast.  `selection` and
  `paths` are accepted for construction compatibility and not read (datasets
  arrive already opened)."""

  selection: t.Any = None
  paths: t.Any = None
  by_init: t.Optional[bool] = True
  rename_variables: t.Optional[t.Dict[str, str]] = None
  pressure_level_suffixes: t.Optional[bool] = False
