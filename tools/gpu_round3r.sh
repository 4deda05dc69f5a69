cd $GRAFT_REPO_ROOT
O=gpurun_out/r3r
mkdir -p $O
timeout 300 python tools/k3_variants.py 2>$O/err.txt | tail -1 | tee $O/k3_variants.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print('%-22s %.4f ms  frac %.3f' % (k, v['kernel_ms'], v['frac']))"
tail -3 $O/err.txt
