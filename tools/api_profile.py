import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import runpy
import cProfile, pstats
# reuse the tool's setup by exec'ing it up to the steady loop
src = open(os.path.join(sys.path[0], 'tools', 'api_throughput.py')).read()
src = src.split("import cProfile, pstats")[0]
ns = {'__file__': os.path.join(sys.path[0], 'tools', 'api_throughput.py')}
exec(compile(src, 'api', 'exec'), ns)
gm, evaluation, torch = ns['gm'], ns['evaluation'], ns['torch']
variants, truth, cfg = ns['variants'], ns['truth'], ns['cfg']
pr = cProfile.Profile(); pr.enable()
for _ in range(4):
  for fi in variants:
    evaluation._metric_and_region_loop(fi, truth, cfg, False, compute_chunk=True)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
