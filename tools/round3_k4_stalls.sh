# Round-3 starting point: where do the waves of the
# K4f kernels wait?  Counter passes (rocprofv3 --pmc with --kernel-trace only,
# <= 8 SQ counters per pass; TA / TCP blocks in passes of their own) over the
# three spectrum workloads; prints per-kernel averages and the derived shares.
#   gpurun --timeout 600 -- 'bash tools/round3_k4_stalls.sh'
# Counter names checked against /opt/rocm/share/rocprofiler-sdk/counter_defs.yaml
# (gfx950).  Output also lands in gpurun_out/k4_stalls.txt.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/k4_stalls.txt
: > $OUT
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SALU"
 "SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL"
 "SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES"
 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_WAVES"
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL"
 "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
 "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum"
)
for w in spectrum spectrum_mean spectrum_materialized; do
  i=0
  for counters in "${PASSES[@]}"; do
    i=$((i + 1))
    d=$GRAFT_REPO_ROOT/gpurun_out/k4pmc_${w}_$i
    (cd /tmp && timeout 240 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $d -o run -- \
        python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 1 --ramp-ms 0 > /dev/null 2>&1)
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python - "$f" $w <<'PY' | tee -a $OUT
import csv, sys, collections
f, w = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    k = row['Kernel_Name']
    if 'fused_spectrum_kernel' in k or 'latseg' in k:
        acc[k[:70]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in acc.items():
    print(w, '|', k, '|', ' '.join(f'{c}={sum(v)/len(v):.5g}' for c, v in sorted(d.items())))
PY
    rm -rf $d
  done
done
python - $OUT <<'PY'
# shares per kernel: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES
import re, sys, collections
vals = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    w, k, rest = [x.strip() for x in line.split('|')]
    for m in re.finditer(r'(\w+)=([0-9.e+-]+)', rest):
        vals[(w, k)][m.group(1)] = float(m.group(2))
for (w, k), v in vals.items():
    wc = v.get('SQ_WAVE_CYCLES')
    if not wc:
        continue
    print(f'== {w} {k[:50]}')
    for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU',
              'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC'):
        if c in v:
            print(f'   {c:24s} {v[c] / wc:6.1%} of wave cycles')
    if 'SQ_LDS_IDX_ACTIVE' in v and 'SQ_BUSY_CU_CYCLES' in v:
        print(f'   LDS array active          {v["SQ_LDS_IDX_ACTIVE"] / v["SQ_BUSY_CU_CYCLES"]:6.1%} of CU busy cycles;'
              f' bank conflicts {v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v["SQ_LDS_IDX_ACTIVE"], 1):6.1%} of them')
PY
