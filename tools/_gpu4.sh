cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s4
export TMPDIR=/tmp
ONLY=members16,members32,members51,members64,members100,members13_hosted,members24_hosted,members33_hosted,members44_hosted,members47_hosted,members63_hosted,members77_hosted
for v in hosted_hybrid hybrid_noslp; do
WB2HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libwb2hip_$v.so timeout 400 python tools/k3_variants.py --reps 3 --only $ONLY > gpurun_out/s4/k3_$v.json 2> gpurun_out/s4/k3_$v.err
done
python - <<'EOF'
import json
for n in ('hosted_hybrid','hybrid_noslp'):
    try:
        d=json.load(open(f'gpurun_out/s4/k3_{n}.json'))
        print(n, {k: round(v['frac'],3) for k,v in d.items()})
    except Exception as e: print(n,'ERR',e)
EOF
