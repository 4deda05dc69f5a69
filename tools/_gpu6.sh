cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s6
export TMPDIR=/tmp
run() { timeout 200 python tools/tier2_variants.py --only energy_score --reps 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$1', round(d['energy_score']['frac'],3), round(d['energy_score']['ms_per_call'],4))"; }
for rows in 5 8 16 32; do WB2HIP_ENERGY_ROWS_PER_CHUNK=$rows run "default(b8,u2) rows=$rows"; done
for v in r1 r4 b10 b13 b10r1; do WB2HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libwb2hip_energy_$v.so run "$v rows=16"; done
WB2HIP_ENERGY_ROWS_PER_CHUNK=32 WB2HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libwb2hip_energy_b10.so run "b10 rows=32"
WB2HIP_ENERGY_ROWS_PER_CHUNK=32 WB2HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libwb2hip_energy_b13.so run "b13 rows=32"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_e -o run -- python $GRAFT_REPO_ROOT/tools/tier2_variants.py --only energy_score --reps 1 > /dev/null 2>&1
f=$(find /tmp/st_e -name '*kernel_stats.csv' | head -1); head -8 $f | cut -c1-200
