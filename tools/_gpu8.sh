cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s8
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_chunk_program_gpu.py tests/test_chunk_batching_gpu.py tests/test_staging.py tests/test_tier2_gpu.py -x -q -m gpu > gpurun_out/s8/pytest.txt 2>&1 ) 2>&1 | grep real; tail -25 gpurun_out/s8/pytest.txt
timeout 600 python tools/official_chunk.py --chunks 128 --batch 1,default --host-fed > gpurun_out/s8/official.json 2> gpurun_out/s8/official.err
python -c "
import json
d=json.load(open('gpurun_out/s8/official.json'))
print({k:(round(v['value']/1e9,1),round(v['host_ms_per_chunk'],3),round(v['wall_ms_per_chunk'],3)) for k,v in d['by_batch_chunks'].items()})
print(json.dumps(d.get('host_fed',{}).get('by_window')))
" ; tail -3 gpurun_out/s8/official.err
