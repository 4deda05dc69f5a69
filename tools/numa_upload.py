"""Does the NUMA placement of a pageable source explain a slow staged upload?
Source arrays first-touched on node S, copy threads (and the pinned ring)
created by a thread bound to node T; reports the uploader's rate for every
(S, T).  tools/upload_sweep.py measures the default placement."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpus_of(node):
  out = []
  for part in open(f'/sys/devices/system/node/node{node}/cpulist').read().strip().split(','):
    a, _, b = part.partition('-')
    out += list(range(int(a), int(b or a) + 1))
  return out


def main():
  import threading
  import torch
  from weatherbench2_amd import feeder
  dev = torch.device('cuda', 0)
  nodes = sorted(int(d[4:]) for d in os.listdir('/sys/devices/system/node')
                 if d.startswith('node') and d[4:].isdigit())
  everything = os.sched_getaffinity(0)
  for s in nodes:
    os.sched_setaffinity(0, cpus_of(s))
    arrays = [np.ones((13, 721, 1440), dtype=np.float32) for _ in range(6)]
    nbytes = sum(a.nbytes for a in arrays)
    for t in nodes:
      result = {}

      def work():
        os.sched_setaffinity(0, cpus_of(t))   # this thread + the pool it makes
        for _ in range(2):
          keep = feeder.upload_many(arrays, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
          keep = feeder.upload_many(arrays, dev)
        torch.cuda.synchronize()
        result['rate'] = 5 * nbytes / (time.perf_counter() - t0) / 1e9
        feeder.close_thread_uploaders()
      th = threading.Thread(target=work)
      th.start()
      th.join()
      print(json.dumps({'source_node': s, 'copy_threads_node': t,
                        'threads': feeder.copy_threads(),
                        'GBps': round(result['rate'], 1)}))
    os.sched_setaffinity(0, everything)


if __name__ == '__main__':
  main()
