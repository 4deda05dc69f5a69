"""`deterministic_temporal` in evaluate_chunks' default windows, for a kernel
trace (rocprofv3 --kernel-trace --stats -- python tools/temporal_trace.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
  import torch
  import official_chunk as oc
  from weatherbench2_amd import evaluation
  dev = torch.device('cuda', 0)
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 240
  batch = None if len(sys.argv) < 3 or sys.argv[2] == 'default' else int(
      sys.argv[2])
  chunks, cfg = oc.build(dev, n, 32)
  cfg_t = oc.temporal_config(cfg)
  which = cfg_t if len(sys.argv) < 4 else cfg
  for _ in range(2):
    evaluation.evaluate_chunks(chunks, which, False, prefetch=0,
                               batch_chunks=batch)
    torch.cuda.synchronize()


if __name__ == '__main__':
  main()
