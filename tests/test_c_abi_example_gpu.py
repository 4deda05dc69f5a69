"""Builds (gcc) and runs the pure-C callers of libwb2hip.so: the boundary is a C
ABI, not a Python extension (tests/c_abi/c_abi_example.c: the kernels call by
call; tests/c_abi/c_abi_program.c: chunks replayed with one call each)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, 'build', 'c_abi_example')
EXE_PROGRAM = os.path.join(ROOT, 'build', 'c_abi_program')


def _build(name='c_abi_example', exe=EXE):
  from weatherbench2_amd import build
  build.build(verbose=False)
  os.makedirs(os.path.dirname(exe), exist_ok=True)
  subprocess.run(
      ['gcc', '-std=c11', '-O2', '-Wall', '-I', os.path.join(ROOT, 'include'),
       '-I', '/opt/rocm/include',
       os.path.join(ROOT, 'tests', 'c_abi', name + '.c'),
       '-L', os.path.join(ROOT, 'weatherbench2_amd'), '-lwb2hip',
       '-L', '/opt/rocm/lib', '-lamdhip64', '-lm',
       '-Wl,-rpath,' + os.path.join(ROOT, 'weatherbench2_amd'),
       '-Wl,-rpath,/opt/rocm/lib', '-o', exe], check=True)


def test_c_caller_compiles_and_links():
  _build()
  assert os.path.exists(EXE)


@pytest.mark.gpu
def test_c_caller_matches_reference_loop():
  _build()
  out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
  assert out.returncode == 0, out.stdout + out.stderr
  assert 'c_abi_example ok' in out.stdout


def test_c_program_caller_compiles_and_links():
  _build('c_abi_program', EXE_PROGRAM)
  assert os.path.exists(EXE_PROGRAM)


@pytest.mark.gpu
def test_c_caller_replays_chunks_with_one_call_each():
  """wb2_program_* from C: three chunks (two variables + a wind pair) replayed
  by ONE call each give the bits of wb2_det_wind_suite_step per chunk and of
  the host's running sums."""
  _build('c_abi_program', EXE_PROGRAM)
  out = subprocess.run([EXE_PROGRAM], capture_output=True, text=True,
                       timeout=120)
  assert out.returncode == 0, out.stdout + out.stderr
  assert 'c_abi_program ok' in out.stdout
