# Round 3, A/B 6 (K1): occupancy / batching variants of the new source
#   default  plain addressing + packed pairs
#   u1    half the rows per batch            pipe  + double-buffered batches
#   w5    heavy instantiations at 5 waves    v4    heavy at 4 columns per lane
#   h5    every instantiation at >= 5 waves  sg    SGPR row pointers
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab3f
mkdir -p $O
: > $O/summary.txt
V=$GRAFT_REPO_ROOT/build/variants
line() {
  python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-10s' % '$1' + ' '.join('%s=%.4f' % (k[:10], v['kernel_ms']) for k, v in d.items()))" | tee -a $O/summary.txt
}
for rep in 1 2; do
  for n in default u1 pipe w5 v4 h5 sg; do
    lib=""; [ "$n" != default ] && lib=$V/libwb2hip_$n.so
    WB2HIP_LIB=$lib timeout 200 python bench.py --variants-only 2>/dev/null | tail -1 | line $n
  done
done
for r in 16 24 48 64; do
  timeout 200 python bench.py --variants-only --rows-per-chunk $r 2>/dev/null | tail -1 | line rows$r
done
