"""Builds (gcc) and runs the pure-C caller of libwb2hip.so: the boundary is a C
ABI, not a Python extension (tests/c_abi/c_abi_example.c)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, 'build', 'c_abi_example')


def _build():
  from weatherbench2_amd import build
  build.build(verbose=False)
  os.makedirs(os.path.dirname(EXE), exist_ok=True)
  subprocess.run(
      ['gcc', '-std=c11', '-O2', '-I', os.path.join(ROOT, 'include'),
       '-I', '/opt/rocm/include',
       os.path.join(ROOT, 'tests', 'c_abi', 'c_abi_example.c'),
       '-L', os.path.join(ROOT, 'weatherbench2_amd'), '-lwb2hip',
       '-L', '/opt/rocm/lib', '-lamdhip64', '-lm',
       '-Wl,-rpath,' + os.path.join(ROOT, 'weatherbench2_amd'),
       '-Wl,-rpath,/opt/rocm/lib', '-o', EXE], check=True)


def test_c_caller_compiles_and_links():
  _build()
  assert os.path.exists(EXE)


@pytest.mark.gpu
def test_c_caller_matches_reference_loop():
  _build()
  out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
  assert out.returncode == 0, out.stdout + out.stderr
  assert 'c_abi_example ok' in out.stdout
