"""`config.Eval`: the plug-in point of the metric x region loop.

The reference's `Eval` (weatherbench2/config.py:96-137) carries the operator
objects -- `metrics: {name: Metric}`, `regions: {name: Region}`,
`derived_variables: {name: DerivedVariable}` -- into
`evaluation._metric_and_region_loop`.  This module provides a dataclass with the
same field names, order and defaults, so an existing configuration is built
unchanged with GPU metric objects as the values of `metrics`; a reference
`config.Eval` instance works just as well (the loop only reads attributes).
The dataset-opening half of the reference's config (`Selection`, `Paths`,
`Data`, config.py:28-93) belongs to the IO layer, which is out of scope.
"""
from __future__ import annotations

import dataclasses
import typing as t

_REQUIRED = dataclasses.MISSING

# (field, default) in the reference's order; only the first three and
# `temporal_mean` are read on this path, the rest are accepted and kept so that
# code written against the reference's Eval keeps constructing it.
_EVAL_FIELDS: tuple = (
    ('metrics', _REQUIRED),                       # {name: Metric}
    ('regions', None),                            # {name: Region} or None
    ('evaluate_persistence', False),
    ('evaluate_climatology', False),
    ('evaluate_probabilistic_climatology', False),
    ('probabilistic_climatology_start_year', None),
    ('probabilistic_climatology_end_year', None),
    ('probabilistic_climatology_hour_interval', None),
    ('against_analysis', False),
    ('derived_variables', dict),                  # factory: fresh dict
    ('temporal_mean', True),                      # average over (init_)time
    ('output_format', 'netcdf'),
)


def _field(default):
  if default is _REQUIRED:
    return dataclasses.field()
  if default is dict:
    return dataclasses.field(default_factory=dict)
  return dataclasses.field(default=default)


Eval = dataclasses.make_dataclass(
    'Eval', [(name, t.Any, _field(default)) for name, default in _EVAL_FIELDS])
Eval.__doc__ = (
    'Evaluation configuration (field-compatible with weatherbench2 '
    'config.py:96-137).\n\n'
    '  metrics            {name: Metric} evaluated on every chunk\n'
    '  regions            optional {name: Region}; results get a `region` dim\n'
    '  derived_variables  {name: DerivedVariable} computed on the fly and\n'
    '                     assigned into forecast / truth before the metrics\n'
    '  temporal_mean      average the metrics over time / init_time\n'
    'The remaining fields configure baselines and output of the reference\'s\n'
    'drivers and are carried along untouched.')
Eval.__module__ = __name__
