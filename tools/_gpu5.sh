cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_tier2_gpu.py tests/test_reference_vectors.py tests/test_ens_gpu.py -x -q -m gpu > gpurun_out/s5/pytest.txt 2>&1 ) 2>&1 | grep real; tail -15 gpurun_out/s5/pytest.txt
timeout 300 python tools/tier2_variants.py --only energy_score,ens_thresholds > gpurun_out/s5/tier2.json 2> gpurun_out/s5/tier2.err; cat gpurun_out/s5/tier2.json | head -c 1500; tail -3 gpurun_out/s5/tier2.err
ONLY=members36,members45,members48,members72,members80,members90,members13_hosted,members24_hosted,members33_hosted,members44_hosted,members47_hosted,members63_hosted,members77_hosted
timeout 400 python tools/k3_variants.py --reps 3 --only $ONLY > gpurun_out/s5/k3.json 2> gpurun_out/s5/k3.err
python - <<'EOF'
import json
try:
    d=json.load(open('gpurun_out/s5/k3.json'))
    print({k: round(v['frac'],3) for k,v in d.items()})
except Exception as e: print('ERR',e)
EOF
cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_e -o run -- python $GRAFT_REPO_ROOT/tools/tier2_variants.py --only energy_score --reps 1 > /dev/null 2>&1
python - <<'EOF'
import csv,glob,collections
f=glob.glob('/tmp/pmc_e/**/*counter_collection.csv',recursive=True)
v=collections.defaultdict(list)
for row in csv.DictReader(open(f[0])):
    if row['Counter_Name']=='FETCH_SIZE': v[row['Kernel_Name'][:60]].append(float(row['Counter_Value']))
for k,x in v.items():
    if 'energy' in k or 'combine' in k: print(k, len(x), sum(x)/len(x)*2048/1e9,'GB fetch per launch; algorithmic', 13*721*1440*51*4/1e9)
EOF
