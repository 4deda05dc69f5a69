import glob, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(2):
  for lib in sorted(glob.glob(os.path.join(ROOT, 'build', 'variants', 'ens_*.so'))):
    env = dict(os.environ, WB2HIP_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'ensemble', '--steps', '8', '--warmup', '2'],
                       env=env, capture_output=True, text=True)
    try:
      d = json.loads(r.stdout.strip().splitlines()[-1])
      print(os.path.basename(lib), round(d['roofline']['achieved']), 'GB/s', round(d['roofline']['kernel_ms'], 4), 'ms')
    except Exception as e:
      print(os.path.basename(lib), 'FAILED', r.stderr[-300:])
