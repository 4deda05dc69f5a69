"""cProfile of the host side of tools/official_probabilistic.py's replayed leg
(where do the microseconds of a chunk go once the GPU work is one call)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
  import torch
  import official_probabilistic as op
  from weatherbench2_amd import config, evaluation
  dev = torch.device('cuda:0')
  chunks, metrics, regions, _, _ = op.build(dev, 512, 240, 121)
  cfg = config.Eval(metrics=metrics, regions=regions)
  batch = None if len(sys.argv) > 1 and sys.argv[1] == 'default' else int(
      sys.argv[1]) if len(sys.argv) > 1 else 1
  evaluation.evaluate_chunks(chunks[:64], cfg, False, prefetch=0,
                             batch_chunks=batch)
  torch.cuda.synchronize()
  pr = cProfile.Profile()
  pr.enable()
  evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0,
                             batch_chunks=batch)
  torch.cuda.synchronize()
  pr.disable()
  st = pstats.Stats(pr)
  st.sort_stats('tottime').print_stats(22)
  st.sort_stats('cumulative').print_stats(22)


if __name__ == '__main__':
  main()
