#!/bin/bash
# K4f development loop on the CPU box: host emulation check, gfx950 compile,
# register / instruction report of the N = 1440 kernels.
#   tools/k4_iter.sh [extra hipcc flags, e.g. -DWB2_FFT_MIN_WAVES=4]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/k4 && cd /tmp/k4
/opt/rocm/bin/hipcc --cuda-host-only -O2 -std=c++17 -I$ROOT/weatherbench2_amd/csrc \
    $ROOT/tools/fft_host_check.hip -o /tmp/k4/fft_host_check
/tmp/k4/fft_host_check | tail -1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" \
    -I$ROOT/include -I$ROOT/weatherbench2_amd/csrc \
    -c $ROOT/weatherbench2_amd/csrc/spectrum_fused.hip -o sf.o --save-temps 2>&1 | grep -v warning || true
S=/tmp/k4/spectrum_fused-hip-amdgcn-amd-amdhsa-gfx950.s
for v in 2 1 0; do
  echo "== fused_spectrum_kernel<720, MODE=$v> (0 materialise, 1 time mean, 2 latitude segments)"
  grep "fused_spectrum_kernelILi720ELi${v}E.*\(num_vgpr\|numbered_sgpr\|private_seg_size\)," $S | sed 's/.*\.\(num_vgpr\|numbered_sgpr\|private_seg_size\)/  \1/'
  python $ROOT/tools/isa_hist.py $S "fused_spectrum_kernelILi720ELi${v}E" --loop --dump /tmp/k4/new720_$v.s | head -${LINES_SHOWN:-14}
done
