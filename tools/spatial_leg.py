"""`deterministic_spatial` legs of tools/official_chunk.py alone (+ --profile:
cProfile of the windowed host path)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, ROOT)
import torch
import official_chunk as oc

dev = torch.device('cuda', 0)
chunks, cfg = oc.build(dev, 512, 32)
if '--profile' in sys.argv:
  import cProfile, pstats
  from weatherbench2_amd import evaluation
  scfg = oc.spatial_config(cfg)
  evaluation.evaluate_chunks(chunks[:40], scfg, False, prefetch=0, batch_chunks=32)
  pr = cProfile.Profile()
  pr.enable()
  evaluation.evaluate_chunks(chunks[:256], scfg, False, prefetch=0, batch_chunks=32)
  pr.disable()
  pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
  sys.exit(0)
if '--window' in sys.argv or '--chunk-by-chunk' in sys.argv:
  # one window size alone (for rocprofv3 --stats: one launch shape per file)
  from weatherbench2_amd import evaluation
  scfg = oc.spatial_config(cfg)
  batch = 32 if '--window' in sys.argv else 1
  evaluation.evaluate_chunks(chunks[:256], scfg, False, prefetch=0,
                             batch_chunks=batch)
  torch.cuda.synchronize()
  sys.exit(0)
r = oc.measure_spatial(chunks, cfg)
for k, v in r['by_window'].items():
  print(k, round(v['value'] / 1e9, 1), 'G steady', round(v['steady_ms_per_chunk'], 3),
        'host', round(v['host_ms_per_chunk'], 3), 'kernel',
        round(v['roofline']['kernel_ms_per_chunk'], 3), 'frac',
        round(v['roofline']['frac'], 3), 'launches', v['fused_launches'])
print(json.dumps(r))
