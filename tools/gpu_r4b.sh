cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest -x -q -m gpu tests/test_chunk_batching_gpu.py > gpurun_out/r4b/pytest_new.txt 2>&1; tail -15 gpurun_out/r4b/pytest_new.txt
timeout 600 python tools/official_chunk.py --chunks 128 --batch 1,16,32,64 > gpurun_out/r4b/official.json 2> gpurun_out/r4b/official.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r4b/official.json'))
for k,v in d['by_batch_chunks'].items():
  r=v.get('roofline',{})
  print(k, f"{v['value']/1e9:.1f} G  wall {v['wall_ms_per_chunk']:.3f} host {v['host_ms_per_chunk']:.3f} launches {v['k1_launches_per_chunk']} k1ms/chunk {r.get('k1_ms_per_chunk')} det {r.get('det_acc',{}).get('frac')} wind {r.get('wind',{}).get('frac')}")
PY
tail -3 gpurun_out/r4b/official.err
timeout 600 python tools/official_chunk.py --chunks 64 --batch 16 --profile > gpurun_out/r4b/profile16.txt 2>&1; head -120 gpurun_out/r4b/profile16.txt
timeout 1500 python -m pytest -x -q -m gpu tests > gpurun_out/r4b/pytest_full.txt 2>&1; tail -5 gpurun_out/r4b/pytest_full.txt
