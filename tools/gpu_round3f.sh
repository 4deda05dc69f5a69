# Round 3: K1 weight-field / skipna instantiations, more variants (kernel ms)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f
mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
for n in default v4of pipe u1 wg4 wg3 default; do
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
res = []
for only in ('official16_landmask', 'skipna'):
  out = bench.k1_variants(dev, f, t, c, 16, pool, only=only)
  res += ['%s=%.4f(%.3f)' % (k, v['kernel_ms'], v['frac']) for k, v in out.items()]
print('$n', ' '.join(res))
PY
done
