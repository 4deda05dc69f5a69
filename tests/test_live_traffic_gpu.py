"""roofline.traffic as a regression test: the HBM bytes of the benched launches
-- K1, K3 and the three modes of the fused spectrum kernel --, from PMC counters
collected now (tools/live_traffic.py: two rocprofv3 passes around
`bench.py --traffic-probe all`), must stay within 2 % of the algorithmic bytes
of SURVEY.md 8(d) (configs[3]: 5 %, its per-segment partial spectra are
scratch) -- a kernel change that re-reads data fails here instead of hiding
behind a number quoted from profiles/."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PTS = 13 * 721 * 1440


@pytest.mark.gpu
def test_traffic_of_every_benched_kernel_is_the_algorithmic_bytes():
  if not (shutil.which('rocprofv3') or os.path.exists('/opt/rocm/bin/rocprofv3')):
    pytest.skip('rocprofv3 not installed')
  res = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'tools', 'live_traffic.py'),
       '--workload', 'all', '--units', '16', '--pool', '24'],
      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
      timeout=900)
  assert res.returncode == 0, res.stderr[-2000:]
  out = json.loads(res.stdout.strip().splitlines()[-1])
  want = {
      'deterministic': (16 * PTS * 12, 'stream_partials_kernel', 1.02),
      'ensemble': (PTS * 51 * 4, 'ens_partials_kernel', 1.02),
      'spectrum_materialized': (16 * PTS * 4 + 16 * 13 * 721 * 721 * 8,
                                'fused_spectrum_kernel', 1.02),
      'spectrum_mean': (16 * PTS * 4 + 13 * 721 * 721 * 8,
                        'fused_spectrum_kernel', 1.02),
      'spectrum': (16 * PTS * 4 + 16 * 13 * 721 * 8, 'fused_spectrum_kernel',
                   1.05),
  }
  assert set(out) == set(want), sorted(out)
  for name, (nbytes, kernel, limit) in want.items():
    got = out[name]
    assert got['launches'] >= 2, (name, got)
    assert kernel in got['kernel'], (name, got)
    assert got['algorithmic_bytes'] == nbytes, (name, got)
    # reads: every input byte once
    assert got['fetch_bytes'] >= 0.97 * min(nbytes, 16 * PTS * 4 if
                                            'spectrum' in name else nbytes), (
        name, got)
    assert got['ratio'] <= limit, (name, got)


@pytest.mark.gpu
def test_the_map_accumulate_kernel_moves_56_bytes_per_point():
  """wb2_spatial_accumulate_addr over one official chunk (85 slabs): forecast
  + truth read once (8 B/pt), three float64 running sums read and written
  (48 B/pt) -- nothing else crosses HBM."""
  if not (shutil.which('rocprofv3') or os.path.exists('/opt/rocm/bin/rocprofv3')):
    pytest.skip('rocprofv3 not installed')
  res = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'tools', 'live_traffic.py'),
       '--workload', 'map_accumulate'],
      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
      timeout=600)
  assert res.returncode == 0, res.stderr[-2000:]
  got = json.loads(res.stdout.strip().splitlines()[-1])
  pts = 85 * 721 * 1440
  assert 'spatial_accumulate_addr_kernel' in got['kernel'], got
  assert got['algorithmic_bytes'] == pts * 56, got
  assert 0.97 * pts * 32 <= got['fetch_bytes'] <= 1.03 * pts * 32, got
  assert 0.97 * pts * 24 <= got['write_bytes'] <= 1.03 * pts * 24, got
  assert got['ratio'] <= 1.02, got
