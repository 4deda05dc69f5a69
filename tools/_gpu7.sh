cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { timeout 200 python tools/tier2_variants.py --only energy_score --reps 3 2>&1 | tail -1 | python -c "
import json,sys
l=sys.stdin.readline()
try:
  d=json.loads(l); print('$1', round(d['energy_score']['frac'],3), round(d['energy_score']['ms_per_call'],4))
except Exception as e: print('$1','ERR',l[:300])"; }
for rep in 1 2; do
for rows in 12 16 24; do WB2HIP_ENERGY_ROWS_PER_CHUNK=$rows run "default(b8,u4) rows=$rows"; done
for v in r2 r8 b10r4 b12r4; do WB2HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libwb2hip_energy_$v.so run "$v rows=16"; done
done
timeout 600 python -m pytest tests/test_tier2_gpu.py -x -q -m gpu -k energy 2>&1 | tail -2
