"""cProfile of `deterministic_temporal` through evaluate_chunks (default
windows, or `python tools/temporal_profile.py 1` chunk by chunk)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
  import torch
  import official_chunk as oc
  from weatherbench2_amd import evaluation
  dev = torch.device('cuda', 0)
  batch = int(sys.argv[1]) if len(sys.argv) > 1 else None
  chunks, cfg = oc.build(dev, 1536, 32)
  cfg_t = oc.temporal_config(cfg)
  evaluation.evaluate_chunks(chunks[:96], cfg_t, False, prefetch=0,
                             batch_chunks=batch)
  torch.cuda.synchronize()
  pr = cProfile.Profile()
  pr.enable()
  evaluation.evaluate_chunks(chunks, cfg_t, False, prefetch=0,
                             batch_chunks=batch)
  torch.cuda.synchronize()
  pr.disable()
  st = pstats.Stats(pr)
  st.sort_stats('tottime').print_stats(18)
  st.sort_stats('cumulative').print_stats(30)
  st.print_callees('result')


if __name__ == '__main__':
  main()
