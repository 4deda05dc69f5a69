"""roofline.traffic as a regression test: the HBM bytes of the benched K1
launch, from PMC counters collected now (tools/live_traffic.py: two rocprofv3
passes around `bench.py --traffic-probe`), must stay within 2 % of the
algorithmic 12 B per grid point -- a kernel change that re-reads data fails
here instead of hiding behind a number quoted from profiles/."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_k1_traffic_is_the_algorithmic_bytes():
  if not (shutil.which('rocprofv3') or os.path.exists('/opt/rocm/bin/rocprofv3')):
    pytest.skip('rocprofv3 not installed')
  res = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'tools', 'live_traffic.py'),
       '--units', '16', '--pool', '24'],
      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
      timeout=400)
  assert res.returncode == 0, res.stderr[-2000:]
  out = json.loads(res.stdout.strip().splitlines()[-1])
  assert out['launches'] >= 2
  assert 'stream_partials_kernel' in out['kernel']
  assert out['algorithmic_bytes'] == 16 * 13 * 721 * 1440 * 12
  # reads: every input byte once; writes: the partials (~0.2 %)
  assert 0.98 <= out['fetch_bytes'] / out['algorithmic_bytes'] <= 1.015, out
  assert out['ratio'] <= 1.02, out
