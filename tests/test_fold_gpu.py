"""The halving-tree epilogue of K1 / K3 (csrc/reduce_common.hpp): every sum it
produces must be bit-identical to a plain xor-butterfly of that value alone --
whatever the number of sums folded together -- so that MSE from a DET pass and
from a DET_ACC pass agree bit for bit, and the gfx950 exchange primitives it is
built from (v_permlane32_swap / v_permlane16_swap, DPP row_ror / quad_perm,
ds_swizzle) must move lanes the way the code assumes.  tools/fold_check.hip is
a standalone HIP program (one wave) that checks exactly that on the device."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_halving_tree_equals_the_butterfly_bit_for_bit():
  exe = os.path.join(ROOT, 'build', 'fold_check')
  src = os.path.join(ROOT, 'tools', 'fold_check.hip')
  deps = [src, os.path.join(ROOT, 'weatherbench2_amd', 'csrc',
                            'reduce_common.hpp')]
  if not os.path.exists(exe) or any(
      os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(
        ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3',
         '-ffp-contract=off', '-Wno-unused-result',
         '-I' + os.path.join(ROOT, 'weatherbench2_amd', 'csrc'),
         '-I' + os.path.join(ROOT, 'include'), '-o', exe, src], check=True)
  out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
  assert out.returncode == 0, out.stdout[-3000:]
  lines = out.stdout.strip().splitlines()
  assert lines[-1] == 'OK'
  for n in (1, 3, 6, 10, 12, 20):
    assert f'N={n} writers={n} mismatches=0 slots_not_written_once=0' in lines
  for vec in (4, 2, 1):
    assert f'fold VEC={vec}: K=3 vs K=6 mismatches=0' in lines
