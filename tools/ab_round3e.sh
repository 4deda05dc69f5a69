# Round 3, A/B 5 (K1): which of the instruction-count changes pays on the GPU.
#   old  the committed kernel (v_lshl_add_u64 per load, compiler-chosen packing)
#   a00  new source, plain addressing, scalar elementwise stage
#   a01  plain addressing + packed column pairs
#   a10  SGPR row pointers + scalar elementwise stage
#   default  SGPR row pointers + packed column pairs
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab3e
mkdir -p $O
: > $O/summary.txt
V=$GRAFT_REPO_ROOT/build/variants
for rep in 1 2; do
  for n in old a00 a01 a10 default; do
    lib=""; [ "$n" != default ] && lib=$V/libwb2hip_$n.so
    WB2HIP_LIB=$lib timeout 200 python bench.py --variants-only 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-8s' % '$n' + ' '.join('%s=%.4f' % (k[:10], v['kernel_ms']) for k, v in d.items()))" | tee -a $O/summary.txt
  done
done
