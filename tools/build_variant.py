"""Builds an A/B variant of libwb2hip.so for kernel tuning on the GPU box.

  python tools/build_variant.py NAME SOURCE.hip [-DFLAG=V ...]

compiles SOURCE with the extra flags and links it with the default objects of
every other translation unit into build/variants/libwb2hip_NAME.so (git-ignored,
travels with gpurun).  Select it with WB2HIP_LIB=<path> (see _lib.lib_path).
SOURCE = ensemble_exact.hip recompiles every per-member-count unit of K3.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from weatherbench2_amd import build as b  # noqa: E402


def main():
  name, src = sys.argv[1], sys.argv[2]
  extra = sys.argv[3:]
  b.build(force=False, verbose=False)  # default objects in build/obj
  obj_dir = os.path.join(ROOT, 'build', 'obj')
  var_dir = os.path.join(ROOT, 'build', 'variants')
  os.makedirs(var_dir, exist_ok=True)
  flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
           '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + b.CSRC]
  import concurrent.futures
  jobs, objs = [], []
  for s, oname, defines in b.translation_units():
    if os.path.basename(s) == src:
      obj = os.path.join(var_dir, f'{name}_{oname}')
      jobs.append([b._hipcc()] + flags + extra + defines +
                  ['-c', s, '-o', obj])
      objs.append(obj)
    else:
      objs.append(os.path.join(obj_dir, oname))
  with concurrent.futures.ThreadPoolExecutor(max_workers=8) as pool:
    list(pool.map(lambda cmd: subprocess.run(cmd, check=True), jobs))
  out = os.path.join(var_dir, f'libwb2hip_{name}.so')
  subprocess.run([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC',
                  '-o', out] + objs + ['-L/opt/rocm/lib', '-lhipfft', '-ldl', '-pthread'],
                 check=True)
  print(out)


if __name__ == '__main__':
  main()
