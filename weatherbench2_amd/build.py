"""Builds libwb2hip.so in-tree with hipcc for gfx950 (no JIT cache, no pip)."""
from __future__ import annotations

import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, 'csrc')
LIB_PATH = os.path.join(PKG_DIR, 'libwb2hip.so')
SOURCES = ('common.cpp', 'comm.cpp', 'staging.cpp', 'program.cpp', 'stream_reduce.hip', 'ensemble.hip', 'energy_score.hip',
           'spectrum.hip', 'spectrum_fused.hip', 'spatial_maps.hip',
           'rank_histogram.hip', 'axis_reduce.hip')
# compiled once per member count listed in sort3_networks.inc (WB2_SORT3_SIZES)
EXACT_SOURCE = 'ensemble_exact.hip'


def _hipcc() -> str:
  for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
    if cand and (os.path.isabs(cand) and os.path.exists(cand)
                 or not os.path.isabs(cand)):
      return cand
  raise RuntimeError('hipcc not found')


def sources() -> list[str]:
  return [os.path.join(CSRC, s) for s in SOURCES
          if os.path.exists(os.path.join(CSRC, s))]


def exact_sizes() -> list[tuple]:
  """[(members, padded count)] of WB2_SORT3_SIZES in sort3_networks.inc."""
  import re
  text = open(os.path.join(CSRC, 'sort3_networks.inc')).read()
  block = text[text.index('#define WB2_SORT3_SIZES(X)'):]
  # the macro body: the continuation lines (a blank line need not follow)
  lines = block.split('\n')
  body = [lines[0]]
  for line in lines[1:]:
    if not body[-1].rstrip().endswith('\\'):
      break
    body.append(line)
  return [(int(a), int(b)) for a, b in re.findall(r'X\((\d+),\s*(\d+)\)',
                                                  '\n'.join(body))]


def translation_units() -> list[tuple]:
  """[(source path, object name, extra flags)] of the whole library."""
  units = [(s, os.path.basename(s) + '.o', []) for s in sources()]
  exact = os.path.join(CSRC, EXACT_SOURCE)
  for m, npad in exact_sizes():
    # -fno-slp-vectorize: merged into packed adds, the hosted kernels' sum /
    # |t - x| chains cost register pairs and copies (146 -> 108 VGPRs, + 5-10 %
    # measured); the exact kernels compile to the same code either way
    units.append((exact, f'ensemble_exact_{m}.o',
                  [f'-DWB2_ENS_M={m}', f'-DWB2_ENS_NPAD={npad}',
                   '-fno-slp-vectorize']))
  return units


def _headers() -> list[str]:
  """Every header a translation unit may include: all of csrc/*.hpp, *.inc
  and the public header (a glob, not a hand-kept list: a new or regenerated
  header -- gauss_tables.inc, a sorter program -- can never leave a stale
  library behind)."""
  import glob
  return sorted(glob.glob(os.path.join(CSRC, '*.hpp')) +
                glob.glob(os.path.join(CSRC, '*.inc')) +
                [os.path.join(ROOT, 'include', 'wb2hip.h')])


def needs_rebuild() -> bool:
  if not os.path.exists(LIB_PATH):
    return True
  t = os.path.getmtime(LIB_PATH)
  deps = sources() + [os.path.join(CSRC, EXACT_SOURCE)] + _headers()
  return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_fold_check(verbose: bool = True) -> None:
  """build/fold_check: the standalone device check of the fold primitives
  (tools/fold_check.hip, run by tests/test_fold_gpu.py)."""
  src = os.path.join(ROOT, 'tools', 'fold_check.hip')
  exe = os.path.join(ROOT, 'build', 'fold_check')
  deps = [src, os.path.join(CSRC, 'reduce_common.hpp'),
          os.path.join(CSRC, 'common.hpp')]
  if not os.path.exists(src):
    return
  if os.path.exists(exe) and all(
      os.path.getmtime(d) <= os.path.getmtime(exe) for d in deps):
    return
  os.makedirs(os.path.dirname(exe), exist_ok=True)
  cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-ffp-contract=off',
         '-Wno-unused-result', '-I' + CSRC, '-I' + os.path.join(ROOT, 'include'),
         '-o', exe, src]
  if verbose:
    print('[wb2hip build]', ' '.join(cmd), file=sys.stderr)
  subprocess.run(cmd, check=True)


def build(force: bool = False, verbose: bool = True) -> str:
  """Compiles every HIP source for gfx950 into weatherbench2_amd/libwb2hip.so.

  One hipcc process per translation unit (in parallel), then one link step.
  """
  if not force and not needs_rebuild():
    build_fold_check(verbose)
    return LIB_PATH
  import concurrent.futures
  obj_dir = os.path.join(ROOT, 'build', 'obj')
  os.makedirs(obj_dir, exist_ok=True)
  flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
           '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]
  extra = os.environ.get('WB2HIP_CXXFLAGS', '').split()

  stamp = ' '.join(flags + extra)
  search = [CSRC, os.path.join(ROOT, 'include')]

  def newest_dep(path, seen=None) -> float:
    """mtime of the newest file `path` includes (quoted includes, followed
    through csrc/ and include/), itself included."""
    import re
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
      return 0.0
    seen.add(path)
    newest = os.path.getmtime(path)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(),
                          re.M):
      for d in [os.path.dirname(path)] + search:
        cand = os.path.join(d, inc)
        if os.path.exists(cand):
          newest = max(newest, newest_dep(cand, seen))
          break
    return newest

  def compile_one(unit):
    src, name, defines = unit
    obj = os.path.join(obj_dir, name)
    # an object newer than its source and every header, built with the same
    # flags (recorded beside it), is reused: a one-file edit recompiles one
    # unit (force=True recompiles everything)
    flag_file = obj + '.flags'
    mine = stamp + ' ' + ' '.join(defines)
    if (not force and os.path.exists(obj) and os.path.exists(flag_file)
        and open(flag_file).read() == mine
        and os.path.getmtime(obj) > newest_dep(src)):
      return obj
    cmd = [_hipcc()] + flags + extra + defines + ['-c', src, '-o', obj]
    if verbose:
      print('[wb2hip build]', ' '.join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    with open(flag_file, 'w') as f:
      f.write(mine)
    return obj

  # the largest units first: the longest compile bounds the wall time
  units = sorted(translation_units(),
                 key=lambda u: -int(u[2][0].split('=')[1]) if u[2] else -1000)
  with concurrent.futures.ThreadPoolExecutor(
      max_workers=min(8, os.cpu_count() or 4)) as pool:
    objs = list(pool.map(compile_one, units))
  link = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH
          ] + objs
  if any(s.endswith('spectrum.hip') for s in sources()):
    link += ['-L/opt/rocm/lib', '-lhipfft']
  link += ['-ldl', '-pthread']
  if verbose:
    print('[wb2hip build]', ' '.join(link), file=sys.stderr)
  subprocess.run(link, check=True)
  build_fold_check(verbose)
  return LIB_PATH


if __name__ == '__main__':
  build(force='--force' in sys.argv)
