cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_ens_exact_gpu.py tests/test_staging.py -x -q -m gpu > gpurun_out/s2/pytest.txt 2>&1 ) 2>&1 | grep real; tail -5 gpurun_out/s2/pytest.txt
ONLY=members8,members16,members32,members51,members64,members100,members7_hosted,members13_hosted,members24_hosted,members33_hosted,members44_hosted,members47_hosted,members63_hosted,members77_hosted
timeout 400 python tools/k3_variants.py --reps 2 --only $ONLY > gpurun_out/s2/k3_flags.json 2> gpurun_out/s2/k3_flags.err
WB2HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libwb2hip_hosted_branchy.so timeout 400 python tools/k3_variants.py --reps 2 --only $ONLY > gpurun_out/s2/k3_branchy.json 2> gpurun_out/s2/k3_branchy.err
WB2HIP_ENS_HOSTED=0 timeout 400 python tools/k3_variants.py --reps 2 --only members7_hosted,members13_hosted,members24_hosted,members33_hosted,members44_hosted,members47_hosted,members63_hosted,members77_hosted > gpurun_out/s2/k3_padded.json 2> gpurun_out/s2/k3_padded.err
python - <<'EOF'
import json
for n in ('flags','branchy','padded'):
    try:
        d=json.load(open(f'gpurun_out/s2/k3_{n}.json'))
        print(n, {k: round(v['frac'],3) for k,v in d.items()})
    except Exception as e: print(n,'ERR',e)
EOF
for th in 8 16 32 64; do WB2HIP_COPY_THREADS=$th timeout 120 python tools/upload_sweep.py 2>/dev/null | tail -1; done | tee gpurun_out/s2/upload_sweep.txt
WB2HIP_COPY_THREADS=16 WB2HIP_STAGE_MEMCPY=1 timeout 120 python tools/upload_sweep.py 2>/dev/null | tail -1 | tee -a gpurun_out/s2/upload_sweep.txt
WB2HIP_COPY_THREADS=32 WB2HIP_STAGE_SLICE_MIB=16 WB2HIP_STAGE_SLOTS=6 timeout 120 python tools/upload_sweep.py 2>/dev/null | tail -1 | tee -a gpurun_out/s2/upload_sweep.txt
timeout 300 python tools/official_chunk.py --chunks 48 --pool 24 --batch 1,default --host-fed > gpurun_out/s2/official_host.json 2> gpurun_out/s2/official_host.err
python -c "
import json
d=json.load(open('gpurun_out/s2/official_host.json'))
print(json.dumps(d.get('host_fed'),indent=0)[:1500])
print({k:(v['value'],v['host_ms_per_chunk']) for k,v in d['by_batch_chunks'].items()})
"
