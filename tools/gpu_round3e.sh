# Round 3: the 2-D weight field of K1 -- is it re-fetched through the fabric?
# default = outer slab fastest (WB2_WF_OUTER_FASTEST=1), wf_chunkfast = old order
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e
mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
for n in default wf_chunkfast; do
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
for rep in range(2):
  out = bench.k1_variants(dev, f, t, c, 16, pool, only='official16_landmask')
  print('$n', ' '.join('%s=%.4f(%.3f)' % (k, v['kernel_ms'], v['frac']) for k, v in out.items()))
PY
  WB2HIP_LIB=$lib timeout 300 python tools/live_traffic.py --variant official16_landmask | sed "s/^/$n official16 /" | tee -a $O/traffic.txt
done
timeout 300 python tools/live_traffic.py --variant skipna | sed "s/^/default skipna /" | tee -a $O/traffic.txt
timeout 300 python -m pytest -x -q -m gpu tests/test_det_gpu.py tests/test_eval_gpu.py tests/test_edge_gpu.py 2>&1 | tail -3
