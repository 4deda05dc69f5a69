# Round 2, GPU call 2: where does K4f's time go (diagnostic variants + SQ
# counters), K3 with double-buffered rows.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2
O=gpurun_out/c2
export TMPDIR=/tmp
run() {  # name workload lib
  local lib=""; [ -n "$3" ] && lib="$GRAFT_REPO_ROOT/build/variants/libwb2hip_$3.so"
  WB2HIP_LIB=$lib timeout 200 python bench.py --workload $2 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/$1.json
  python - <<PY
import json
try:
  d = json.load(open('$O/$1.json'))
  r = d['roofline']
  print('%-22s %-14s kernel_ms %.4f  GB/s %.0f  frac %.3f  value %.4g' % ('$1', '$2', r['kernel_ms'], r['achieved'], r['frac'], d['value']))
except Exception as e:
  print('$1 FAILED', e)
PY
}
{
timeout 300 python -m pytest tests/test_ens_gpu.py tests/test_spectrum_gpu.py -m gpu -x -q 2>&1 | tail -3
for v in "" k3_r01 k3_pf_scalar k3_pf_mw3 k3_nopf_mw4; do
  run ens_${v:-main} ensemble "$v"
done
for v in "" k4_nopf k4_mw4 k4_mw4_nopf k4_d1 k4_d2 k4_d3 k4_d4 k4_d7 k4_d8 k4_d15; do
  run specmean_${v:-main} spectrum_mean "$v"
done
for v in "" k4_mw4 k4_mw4_nopf k4_d16 k4_d20 k4_d4 k4_d7 k4_d15; do
  run spec_${v:-main} spectrum "$v"
done
} 2>&1 | tee $O/summary.txt
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" | sort -u > $O/counters.txt; wc -l $O/counters.txt
pmc() {  # tag workload counters...
  local tag=$1 w=$2; shift 2
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$tag -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $w > /dev/null 2>&1)
  f=$(find $O/pmc_$tag -name '*counter_collection.csv' | head -1)
  python - "$f" $tag <<'PY'
import csv, sys, collections
f, w = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
try:
  for row in csv.DictReader(open(f)):
    k = row['Kernel_Name'][:60]
    if 'wb2' in k:
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
  for k, d in acc.items():
    print(w, '|', k, '|', ' '.join(f'{c}={sum(v)/len(v):.4g}' for c, v in sorted(d.items())))
except Exception as e:
  print(w, 'FAILED', e)
PY
  rm -rf $O/pmc_$tag
}
{
for w in spectrum_mean spectrum ensemble; do
pmc A_$w $w SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS
pmc B_$w $w SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
pmc C_$w $w SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS
pmc D_$w $w GRBM_GUI_ACTIVE GRBM_COUNT
done
} 2>&1 | tee $O/pmc.txt
