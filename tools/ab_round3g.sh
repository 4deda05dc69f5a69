# Round 3, A/B 7 (K1): committed kernel vs the new source vs new + 4 columns per
# lane for the heavy instantiations, same box, interleaved
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab3g
mkdir -p $O
: > $O/summary.txt
V=$GRAFT_REPO_ROOT/build/variants
line() {
  python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-10s' % '$1' + ' '.join('%s=%.4f' % (k[:10], v['kernel_ms']) for k, v in d.items()))" | tee -a $O/summary.txt
}
for rep in 1 2 3; do
  for n in old default v4; do
    lib=""; [ "$n" != default ] && lib=$V/libwb2hip_$n.so
    WB2HIP_LIB=$lib timeout 60 python bench.py --variants-only 2>/dev/null | tail -1 | line $n
  done
done
