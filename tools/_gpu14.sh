cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/s14
timeout 900 python -m pytest tests/test_chunk_program_gpu.py tests/test_thresholds_gpu.py tests/test_tier2_gpu.py -x -q -m gpu 2>&1 | tail -12
timeout 600 python tools/official_chunk.py --batch 1,default > gpurun_out/s14/official.json 2> gpurun_out/s14/official.err
python -c "
import json
d=json.load(open('gpurun_out/s14/official.json'))
print({k:(round(v['value']/1e9,1),round(v['host_ms_per_chunk'],3),round(v['wall_ms_per_chunk'],3), v['k1_launches_per_chunk'], round(v['roofline']['frac'],3), round(v['roofline']['frac_of_bytes_read'],3)) for k,v in d['by_batch_chunks'].items()})
print(json.dumps(d.get('host_fed',{}).get('by_window')))
" ; tail -3 gpurun_out/s14/official.err
timeout 300 python tools/official_chunk.py --chunks 256 --batch 1 --sections 2>&1 | tail -22
python - <<'EOF'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import torch, official_chunk
from weatherbench2_amd import evaluation, program
dev=torch.device('cuda',0)
chunks,cfg=official_chunk.build(dev,16,8)
evaluation.evaluate_chunks(chunks,cfg,False,prefetch=0,batch_chunks=1)
print('REASONS', program.REASONS)
EOF
