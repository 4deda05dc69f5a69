"""HBM traffic of the benched K1 launch from PMC counters, collected NOW.

  python tools/live_traffic.py [--units 16 --pool 48 --rows-per-chunk 0]

Runs `bench.py --traffic-probe` (a few launches of exactly the benched K1
configuration: MODE_DET_ACC, float32, 13 regions, `units` units per launch
gathered through slab tables from `pool`-unit pools) twice under
`rocprofv3 --pmc <counter> --kernel-trace` -- FETCH_SIZE and WRITE_SIZE in
passes of their own, no other trace domain, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes -- and prints ONE JSON line:

  {"fetch_bytes", "write_bytes", "traffic_bytes", "algorithmic_bytes",
   "ratio", "launches", "kernel"}

Counter units and the gfx950 correction (guide, "HBM"): FETCH_SIZE is in KiB and
tallies a wide coalesced streaming read at HALF its bytes on gfx950 (calibrated
on a known 3 GiB read in round 1, profiles/r01_pmc_traffic.md) => x 2048 B;
WRITE_SIZE x 1024 B.  Used by bench.py (roofline.traffic) and by
tests/test_live_traffic_gpu.py, which asserts ratio <= 1.02: the figure is a
regression test, not a quotation.
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_LEV, N_LAT, N_LON = 13, 721, 1440
KERNEL = 'stream_partials_kernel'
SCALE = {'FETCH_SIZE': 2048.0, 'WRITE_SIZE': 1024.0}
VARIANT = ['deterministic']


def collect(counter: str, probe_args: list, timeout: float) -> tuple:
  """(mean counter value per launch of K1, launches)."""
  rocprof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
  out_dir = tempfile.mkdtemp(prefix=f'wb2_pmc_{counter}_')
  try:
    cmd = [rocprof, '--pmc', counter, '--kernel-trace', '--output-format',
           'csv', '-d', out_dir, '-o', 'run', '--', sys.executable,
           os.path.join(ROOT, 'bench.py'), '--traffic-probe', VARIANT[0],
           '--no-pmc',
           '--no-secondary', '--no-pcie', '--no-api', '--no-full-suite',
           '--no-cpu-baseline', '--warmup', '1', '--steps', '4',
           '--ramp-ms', '0'] + probe_args
    env = dict(os.environ, TMPDIR='/tmp')
    res = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=timeout)
    files = glob.glob(os.path.join(out_dir, '**', '*counter_collection.csv'),
                      recursive=True)
    if res.returncode != 0 or not files:
      raise RuntimeError(f'rocprofv3 --pmc {counter} failed '
                         f'(rc {res.returncode}): {res.stderr.strip()[-300:]}')
    values = collections.defaultdict(list)
    with open(files[0]) as f:
      for row in csv.DictReader(f):
        if row.get('Counter_Name') == counter and KERNEL in row['Kernel_Name']:
          values[row['Kernel_Name']].append(float(row['Counter_Value']))
    if not values:
      raise RuntimeError(f'no {KERNEL} launch in the {counter} pass')
    name, vals = max(values.items(), key=lambda kv: len(kv[1]))
    vals = vals[1:] if len(vals) > 2 else vals  # the first launch warms caches
    return sum(vals) / len(vals), len(vals), name
  finally:
    shutil.rmtree(out_dir, ignore_errors=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--units', type=int, default=16)
  ap.add_argument('--pool', type=int, default=48)
  ap.add_argument('--rows-per-chunk', type=int, default=0)
  ap.add_argument('--timeout', type=float, default=110.0)
  ap.add_argument('--variant', default='deterministic',
                  choices=['deterministic', 'official16_landmask', 'skipna'],
                  help='which K1 instantiation to probe (bench.k1_variants)')
  args = ap.parse_args()
  VARIANT[0] = args.variant
  probe = ['--units', str(args.units), '--pool', str(args.pool),
           '--rows-per-chunk', str(args.rows_per_chunk)]
  fetch, n_f, kernel = collect('FETCH_SIZE', probe, args.timeout)
  write, n_w, _ = collect('WRITE_SIZE', probe, args.timeout)
  fetch_b, write_b = fetch * SCALE['FETCH_SIZE'], write * SCALE['WRITE_SIZE']
  algorithmic = args.units * N_LEV * N_LAT * N_LON * 12.0
  print(json.dumps({
      'fetch_bytes': fetch_b, 'write_bytes': write_b,
      'traffic_bytes': fetch_b + write_b, 'algorithmic_bytes': algorithmic,
      'ratio': (fetch_b + write_b) / algorithmic,
      'launches': min(n_f, n_w), 'kernel': kernel[:80],
      'units_per_launch': args.units,
      'counters': 'FETCH_SIZE x 2048 B (gfx950: wide reads tallied at half), '
                  'WRITE_SIZE x 1024 B; separate rocprofv3 passes'}))


if __name__ == '__main__':
  main()
