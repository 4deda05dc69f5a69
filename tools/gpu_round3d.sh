# Round 3: K1 production variants A/B (2 columns per lane for the register-heavy
# instantiations = default, vs 4 = vec4heavy) + the whole GPU suite.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d
mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
for rep in 1 2; do
for n in default vec4heavy; do
  lib=""; [ "$n" != default ] && lib=$V/libwb2hip_$n.so
  WB2HIP_LIB=$lib timeout 300 python - <<PY | tee -a $O/variants.txt
import json, sys, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda', 0)
gen = torch.Generator(device=dev).manual_seed(1)
pool = 48
mk = lambda: torch.randn((pool * 13, 721, 1440), generator=gen, device=dev)
f, t, c = mk(), mk(), mk()
out = bench.k1_variants(dev, f, t, c, 16, pool)
print('$n', ' '.join('%s=%.4f(%.3f)' % (k, v['kernel_ms'], v['frac']) for k, v in out.items()))
PY
done
done
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -12 | tee $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
