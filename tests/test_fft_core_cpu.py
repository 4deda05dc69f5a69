"""The per-lane code of the LDS real FFT (weatherbench2_amd/csrc/fft_core.hpp)
is __host__ __device__: tools/fft_host_check.hip runs it lane by lane on the
CPU -- index maps, twiddle tables, composite in-register butterflies and the
real-FFT recombination of every instantiated row length -- against a direct
O(N^2) DFT in double (np.fft.rfft(norm='forward') semantics,
/root/reference/weatherbench2/derived_variables.py:597-602).  No GPU needed;
hipcc compiles the host side only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hipcc():
  for cand in ('/opt/rocm/bin/hipcc', shutil.which('hipcc')):
    if cand and os.path.exists(cand):
      return cand
  return None


@pytest.mark.skipif(_hipcc() is None, reason='hipcc not installed')
def test_fft_core_host_emulation(tmp_path):
  exe = str(tmp_path / 'fft_host_check')
  subprocess.run(
      [_hipcc(), '--cuda-host-only', '-O2', '-std=c++17',
       '-I' + os.path.join(ROOT, 'weatherbench2_amd', 'csrc'),
       os.path.join(ROOT, 'tools', 'fft_host_check.hip'), '-o', exe],
      check=True, cwd=str(tmp_path))
  out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
  lines = [l.split() for l in out.strip().splitlines()]
  sizes = {int(l[1]): float(l[3]) for l in lines if l[0] == 'N'}
  assert set(sizes) == {64, 128, 240, 256, 360, 512, 720, 1024, 1440, 96, 288,
                        320, 384, 480, 640, 768, 1280, 1800, 2048, 2560, 2880,
                        3600}
  # float32 transform: error relative to the row's total power
  assert max(sizes.values()) < 1e-6, sizes
  # the paired last pass (20 x 6 x 6: two butterflies and the recombination in
  # one lane; every bin produced exactly once)
  paired = {int(l[1]): float(l[3]) for l in lines if l[0] == 'P'}
  assert set(paired) == {1440} and max(paired.values()) < 1e-6, paired
