"""HBM traffic of the benched kernels from PMC counters, collected NOW.

  python tools/live_traffic.py [--workload all|deterministic]
                               [--variant official16_landmask|skipna]
                               [--units 16 --pool 48 --rows-per-chunk 0]

Runs `bench.py --traffic-probe ...` (a few launches of exactly the benched
configurations) twice under `rocprofv3 --pmc <counter> --kernel-trace` --
FETCH_SIZE and WRITE_SIZE in passes of their own (they do not fit one pass:
/opt/skills/guides/MI355X_MICROARCH.md, TCC counters), no other trace domain --
and prints ONE JSON line.

  --workload deterministic   K1 only (or one of its variants): the flat object
        {"fetch_bytes", "write_bytes", "traffic_bytes", "algorithmic_bytes",
         "ratio", "launches", "kernel"}
  --workload all             K1, K3 (BASELINE configs[2]) and the three modes of
        the fused spectrum kernel (configs[3] = LATSEG + combine, MATERIALISE,
        TIME_MEAN) from the SAME two passes: {workload: that object}

Counter units and the gfx950 correction (guide, "HBM"): FETCH_SIZE is in KiB and
tallies a wide coalesced streaming read at HALF its bytes on gfx950 (calibrated
on a known 3 GiB read in round 1, profiles/r01_pmc_traffic.md) => x 2048 B;
WRITE_SIZE x 1024 B.  Used by bench.py (roofline.traffic of every leg) and by
tests/test_live_traffic_gpu.py, which asserts ratio <= 1.02 (configs[3]: 1.05,
its per-segment partial spectra are scratch traffic): the figures are
regression tests, not quotations.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_LEV, N_LAT, N_LON = 13, 721, 1440
N_BINS = N_LON // 2 + 1
SCALE = {'FETCH_SIZE': 2048.0, 'WRITE_SIZE': 1024.0}
SPECTRUM_MODES = {'0': 'spectrum_materialized', '1': 'spectrum_mean',
                  '2': 'spectrum'}


def classify(kernel_name: str):
  """Workload a kernel launch belongs to (None: not a benched kernel)."""
  if 'stream_partials_kernel' in kernel_name:
    return 'deterministic'
  if 'ens_partials_kernel' in kernel_name:
    return 'ensemble'
  m = re.search(r'fused_spectrum_kernel<\d+,\s*\(?[^0-9>]*(\d)', kernel_name)
  if m:
    return SPECTRUM_MODES.get(m.group(1))
  if 'latseg_combine_kernel' in kernel_name:
    return 'spectrum+combine'
  if 'spatial_accumulate_addr_kernel' in kernel_name:
    return 'map_accumulate'
  return None


def collect(counter: str, probe: str, probe_args: list, timeout: float) -> dict:
  """{workload: (mean counter value per launch, launches, kernel name)}."""
  rocprof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
  out_dir = tempfile.mkdtemp(prefix=f'wb2_pmc_{counter}_')
  try:
    cmd = [rocprof, '--pmc', counter, '--kernel-trace', '--output-format',
           'csv', '-d', out_dir, '-o', 'run', '--', sys.executable]
    if probe == 'map_accumulate':
      # the fused kernel of map_suite.py alone (tools/map_accumulate_bench.py)
      cmd += [os.path.join(ROOT, 'tools', 'map_accumulate_bench.py'),
              '--reps', '6', '--pool', '6']
    else:
      cmd += [os.path.join(ROOT, 'bench.py'), '--traffic-probe', probe,
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
    values = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(files[0]) as f:
      for row in csv.DictReader(f):
        if row.get('Counter_Name') != counter:
          continue
        what = classify(row['Kernel_Name'])
        if what:
          values[what][row['Kernel_Name']].append(float(row['Counter_Value']))
    out = {}
    for what, by_name in values.items():
      # several instantiations may have run (K1 variants): the most frequent
      name, vals = max(by_name.items(), key=lambda kv: len(kv[1]))
      vals = vals[1:] if len(vals) > 2 else vals  # the first launch warms caches
      out[what] = (sum(vals) / len(vals), len(vals), name)
    if not out:
      raise RuntimeError(f'no benched kernel in the {counter} pass')
    return out
  finally:
    shutil.rmtree(out_dir, ignore_errors=True)


def algorithmic_bytes(workload: str, units: int, spectrum_units: int = 16,
                      members: int = 50) -> float:
  """SURVEY.md 8(d), strictly, per launch."""
  pts_unit = N_LEV * N_LAT * N_LON
  if workload == 'map_accumulate':
    # 85 slabs: forecast + truth read, three float64 sums read and written
    return 85 * N_LAT * N_LON * 56.0
  if workload == 'deterministic':
    return units * pts_unit * 12.0
  if workload == 'ensemble':
    return pts_unit * (members + 1) * 4.0
  pts = spectrum_units * pts_unit
  rows = spectrum_units * N_LEV * N_LAT
  if workload == 'spectrum_materialized':
    return pts * 4.0 + rows * N_BINS * 8.0
  if workload == 'spectrum_mean':
    return pts * 4.0 + (rows // spectrum_units) * N_BINS * 8.0
  if workload == 'spectrum':  # one reduced spectrum per field
    return pts * 4.0 + spectrum_units * N_LEV * N_BINS * 8.0
  raise ValueError(workload)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--units', type=int, default=16)
  ap.add_argument('--pool', type=int, default=48)
  ap.add_argument('--rows-per-chunk', type=int, default=0)
  ap.add_argument('--timeout', type=float, default=240.0)
  ap.add_argument('--workload', default='deterministic',
                  choices=['deterministic', 'all', 'map_accumulate'])
  ap.add_argument('--variant', default='deterministic',
                  choices=['deterministic', 'official16_landmask', 'skipna'],
                  help='which K1 instantiation to probe (bench.k1_variants)')
  args = ap.parse_args()
  probe = ('all' if args.workload == 'all' else 'map_accumulate'
           if args.workload == 'map_accumulate' else args.variant)
  probe_args = ['--units', str(args.units), '--pool', str(args.pool),
                '--rows-per-chunk', str(args.rows_per_chunk)]
  fetch = collect('FETCH_SIZE', probe, probe_args, args.timeout)
  write = collect('WRITE_SIZE', probe, probe_args, args.timeout)
  note = ('FETCH_SIZE x 2048 B (gfx950: wide reads tallied at half), '
          'WRITE_SIZE x 1024 B; separate rocprofv3 passes')
  result = {}
  for what in fetch:
    if what == 'spectrum+combine' or what not in write:
      continue
    f_b = fetch[what][0] * SCALE['FETCH_SIZE']
    w_b = write[what][0] * SCALE['WRITE_SIZE']
    if what == 'spectrum' and 'spectrum+combine' in fetch:
      # configs[3] = the LATSEG launch + its combine launch
      f_b += fetch['spectrum+combine'][0] * SCALE['FETCH_SIZE']
      w_b += write['spectrum+combine'][0] * SCALE['WRITE_SIZE']
    alg = algorithmic_bytes(what, args.units)
    result[what] = {
        'fetch_bytes': f_b, 'write_bytes': w_b, 'traffic_bytes': f_b + w_b,
        'algorithmic_bytes': alg, 'ratio': (f_b + w_b) / alg,
        'launches': min(fetch[what][1], write[what][1]),
        'kernel': fetch[what][2][:80], 'counters': note}
  if args.workload == 'map_accumulate':
    print(json.dumps(result['map_accumulate']))
  elif args.workload == 'deterministic':
    flat = result['deterministic']
    flat['units_per_launch'] = args.units
    print(json.dumps(flat))
  else:
    print(json.dumps(result))


if __name__ == '__main__':
  main()
