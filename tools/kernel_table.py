"""The kernel table of DESIGN.md section 3, printed from the COMMITTED rocprofv3
summaries (profiles/<round>_*_kernel_stats.csv) so that prose cannot drift from
the evidence:

  python tools/kernel_table.py r06            # markdown table on stdout

frac = algorithmic bytes per launch (SURVEY 8d x the units one launch of the
profiled command processes) / average duration / 8 TB/s.
"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PTS = 721 * 1440          # one slab
UNIT = 13 * PTS           # one (init, lead) unit of 13 levels
PEAK = 8.0e12

# (csv suffix, kernel-name substring, label, algorithmic bytes per launch)
ROWS = [
    ('deterministic', 'stream_partials_kernel<float, 4, 1, false, false',
     'K1 DET_ACC, 16 units x 13 slabs, 13 slice regions', 16 * UNIT * 12),
    ('ensemble', 'ens_partials_kernel<float, 64, 50',
     'K3, 50 members, 13 slabs', UNIT * 51 * 4),
    ('spectrum_materialized', 'fused_spectrum_kernel<720, 0, float',
     'K4f MATERIALISE, 16 units', 16 * (UNIT * 4 + 13 * 721 * 721 * 8)),
    ('spectrum', 'fused_spectrum_kernel<720, 2, float',
     'K4f LATSEG (configs[3]; + latseg_combine), 16 units', 865015424),
    ('spectrum', 'latseg_combine_kernel', 'latseg_combine', None),
    ('spectrum_mean', 'fused_spectrum_kernel<720, 1, float',
     'K4f TIME_MEAN, 16 units', 16 * UNIT * 4 + 13 * 721 * 721 * 8),
    ('spectrum_materialized_f64', 'fused_spectrum_kernel<720, 0, double',
     'K4f MATERIALISE float64, 16 units',
     16 * (UNIT * 8 + 13 * 721 * 721 * 8)),
    ('spectrum_mean_f64', 'fused_spectrum_kernel<720, 1, double',
     'K4f TIME_MEAN float64, 16 units', 16 * UNIT * 8 + 13 * 721 * 721 * 8),
    ('energy_score', 'energy_partials_kernel', 'K3e partials, 50 members',
     UNIT * 51 * 4),
    ('energy_score', 'energy_finalize_kernel', 'K3e finalize', None),
    ('official_chunk', 'stream_partials_kernel<float, 4, 1, false, true',
     'K1 DET_ACC + land-mask field, 24-chunk window (1368 slabs)',
     1368 * PTS * 12),
    ('official_chunk', 'stream_pair_kernel<float, 4, true, false, true',
     'K1p pairs + land-mask field, 24-chunk window (336 pairs)',
     672 * PTS * 12),
    ('official_chunk_by_chunk',
     'stream_partials_kernel<float, 4, 1, false, true',
     'K1 DET_ACC + field, one chunk (57 slabs)', 57 * PTS * 12),
    ('official_chunk_by_chunk',
     'stream_pair_kernel<float, 4, true, false, true',
     'K1p + field, one chunk (14 pairs)', 28 * PTS * 12),
    ('official_chunk_by_chunk', 'stream_partials_kernel<float, 4, 7',
     'K1 SEEPS, one slab, on a side stream BESIDE K1 (its duration is not '
     'its own)', None),
    # every K3 launch of `official_probabilistic.py --chunks 1024 --windows
    # default --only-windows`: (64 warm-up + 1024 timed + 2 x 32 trial) chunks
    # of 23 slabs x 51 arrays of 240 x 121 float32, each read once -- the bytes
    # of the whole run over its launches (2 per 32-chunk window: 576 and 160
    # slabs; the first window of a call goes variable by variable)
    ('official_probabilistic', 'ens_partials_kernel<float, 64, 50',
     'K3 by address, 32-chunk windows of the 240 x 121 ENS chunks (all '
     'launches of the run)', ('total', 1152 * 23 * 240 * 121 * 51 * 4)),
    ('official_spatial', 'spatial_accumulate_addr_kernel',
     'K5a map accumulate, 85 destinations x 8 steps (k = 8)',
     85 * PTS * (8 * 8 + 48)),
    ('official_spatial_by_chunk', 'spatial_accumulate_addr_kernel',
     'K5a map accumulate, 85 destinations x 1 step', 85 * PTS * 56),
]


def stats(path):
  out = []
  with open(path) as f:
    for r in csv.DictReader(f):
      out.append((r['Name'], int(r['Calls']), float(r['AverageNs'])))
  return out


def main():
  rnd = sys.argv[1] if len(sys.argv) > 1 else 'r06'
  print('| kernel (file) | calls | avg µs | algorithmic MB / launch | TB/s | frac of 8 TB/s |')
  print('|---|---|---|---|---|---|')
  for suffix, needle, label, nbytes in ROWS:
    path = os.path.join(ROOT, 'profiles', f'{rnd}_{suffix}_kernel_stats.csv')
    if not os.path.exists(path):
      continue
    hits = [s for s in stats(path) if needle in s[0]]
    if not hits:
      continue
    name, calls, avg = max(hits, key=lambda s: s[1] * s[2])
    if isinstance(nbytes, tuple):   # bytes of the whole run
      nbytes = nbytes[1] / calls
    if nbytes:
      tbps = nbytes / (avg * 1e-9) / 1e12
      tail = f'{nbytes / 1e6:.1f} | {tbps:.2f} | **{tbps * 1e12 / PEAK:.3f}**'
    else:
      tail = '— | — | —'
    print(f'| {label} (`{rnd}_{suffix}`) | {calls} | {avg / 1e3:.1f} | {tail} |')


if __name__ == '__main__':
  main()
