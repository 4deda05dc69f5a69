"""profiles/r03_pmc_raw.txt (+ the default bench line's live collection) ->
profiles/r03_pmc_traffic.json, the table bench.py's `roofline.traffic` of the
secondary workloads is looked up in.

  python tools/pmc_traffic_json.py [--units-per-launch 16]

FETCH_SIZE x 2048 B (gfx950: wide reads tallied at half, calibrated in round 1),
WRITE_SIZE x 1024 B; counter means per launch from tools/round3_profile.sh.
"""
import argparse
import collections
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--units-per-launch', type=int, default=16,
                  help='of the spectrum workloads in the raw file')
  args = ap.parse_args()
  prof = os.path.join(ROOT, 'profiles')
  raw = collections.defaultdict(lambda: collections.defaultdict(dict))
  for line in open(os.path.join(prof, 'r03_pmc_raw.txt')):
    m = re.match(r'(\S+) (FETCH_SIZE|WRITE_SIZE) \| (.*?) \| launches (\d+) '
                 r'mean ([\d.e+]+)', line)
    if m:
      w, c, k, _, mean = m.groups()
      raw[w][c][k] = float(mean)

  def entry(w, pick, extra):
    tot = lambda c: sum(v for k, v in raw[w][c].items()
                        if any(p in k for p in pick))
    f, wr = int(tot('FETCH_SIZE') * 2048), int(tot('WRITE_SIZE') * 1024)
    return dict(extra, fetch_bytes=f, write_bytes=wr, traffic_bytes=f + wr)

  out = {'_comment': (
      'HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE '
      '(separate passes, --kernel-trace only; tools/round3_profile.sh; the '
      'deterministic entry is the LIVE collection of the default bench run, '
      'tools/live_traffic.py), FETCH_SIZE x 2048 B (gfx950 correction, '
      'calibrated in round 1), WRITE_SIZE x 1024 B. Keyed by the bench '
      'workload and its per-launch size. Raw counter means: '
      'profiles/r03_pmc_raw.txt. Made by tools/pmc_traffic_json.py.')}
  line = json.loads(open(os.path.join(prof, 'r03_bench_default_line.json'))
                    .read().strip().splitlines()[-1])
  det = line['roofline'].get('traffic_detail') or {}
  if det:
    out['deterministic'] = {
        'units_per_launch': det.get('units_per_launch', 16),
        'regions': line['config'].get('regions', 13),
        'fetch_bytes': int(det['fetch_bytes']),
        'write_bytes': int(det['write_bytes']),
        'traffic_bytes': int(det['fetch_bytes']) + int(det['write_bytes'])}
  u = {'units_per_launch': args.units_per_launch}
  out['ensemble'] = entry('ensemble', ['ens_partials'],
                          {'slabs_per_launch': 13, 'members': 50})
  out['spectrum'] = entry('spectrum_materialized', ['fused_spectrum_kernel'], u)
  out['spectrum_latmean'] = entry(
      'spectrum', ['fused_spectrum_kernel', 'latseg_combine'], u)
  out['spectrum_mean'] = entry('spectrum_mean', ['fused_spectrum_kernel'], u)
  json.dump(out, open(os.path.join(prof, 'r03_pmc_traffic.json'), 'w'),
            indent=1)
  for k, v in out.items():
    if k != '_comment':
      print(k, v)


if __name__ == '__main__':
  main()
