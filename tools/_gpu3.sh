cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_ens_exact_gpu.py -x -q -m gpu > gpurun_out/s3/pytest.txt 2>&1 ) 2>&1 | grep real; tail -4 gpurun_out/s3/pytest.txt
WB2HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libwb2hip_hosted_hybrid.so timeout 600 python -m pytest tests/test_ens_exact_gpu.py -x -q -m gpu -k "hosted or match_oracle" > gpurun_out/s3/pytest_hybrid.txt 2>&1; tail -3 gpurun_out/s3/pytest_hybrid.txt
ONLY=members51,members64,members100,members7_hosted,members13_hosted,members24_hosted,members33_hosted,members44_hosted,members47_hosted,members63_hosted,members77_hosted
timeout 400 python tools/k3_variants.py --reps 3 --only $ONLY > gpurun_out/s3/k3_flags.json 2> gpurun_out/s3/k3_flags.err
WB2HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libwb2hip_hosted_hybrid.so timeout 400 python tools/k3_variants.py --reps 3 --only $ONLY > gpurun_out/s3/k3_hybrid.json 2> gpurun_out/s3/k3_hybrid.err
python - <<'EOF'
import json
for n in ('flags','hybrid'):
    try:
        d=json.load(open(f'gpurun_out/s3/k3_{n}.json'))
        print(n, {k: round(v['frac'],3) for k,v in d.items()})
    except Exception as e: print(n,'ERR',e)
EOF
