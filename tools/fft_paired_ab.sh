cd $GRAFT_REPO_ROOT
WB2HIP_FFT_PAIRED=1 timeout 600 python -m pytest tests/test_spectrum_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for w in spectrum spectrum_mean; do
  for v in 0 1; do
    echo -n "$w paired=$v default-lib: "; WB2HIP_FFT_PAIRED=$v timeout 200 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['frac'],4), round(d['roofline']['kernel_ms'],4))"
  done
  echo -n "$w paired=1 w3-lib: "; WB2HIP_LIB=build/variants/libwb2hip_fftp_w3.so WB2HIP_FFT_PAIRED=1 timeout 200 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['frac'],4), round(d['roofline']['kernel_ms'],4))"
done; done
