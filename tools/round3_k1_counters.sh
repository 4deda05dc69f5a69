# Round 3: where do the waves of K1's weight-field / skipna instantiations wait,
# next to the headline instantiation?  (rocprofv3 --pmc, --kernel-trace only)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/k1_counters.txt
: > $OUT
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_LDS"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
 "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
)
for v in deterministic official16_landmask skipna; do
  i=0
  for counters in "${PASSES[@]}"; do
    i=$((i + 1))
    d=$GRAFT_REPO_ROOT/gpurun_out/k1pmc_${v}_$i
    (cd /tmp && timeout 240 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $d -o run -- \
        python $GRAFT_REPO_ROOT/bench.py --traffic-probe $v --no-pmc --no-secondary --no-pcie --no-api --no-full-suite --no-cpu-baseline --warmup 1 --steps 4 --ramp-ms 0 > /dev/null 2>&1)
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python - "$f" $v <<'PY' | tee -a $OUT
import csv, sys, collections
f, v = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    k = row['Kernel_Name']
    if 'stream_partials_kernel' in k:
        acc[k[:95]][row['Counter_Name']].append(float(row['Counter_Value']))
best = max(acc.items(), key=lambda kv: max(len(x) for x in kv[1].values()))
k, d = best
print(v, '|', k, '|', ' '.join(f'{c}={sum(x[1:]) / max(len(x) - 1, 1):.5g}' for c, x in sorted(d.items())))
PY
    rm -rf $d
  done
done
