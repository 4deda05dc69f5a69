cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/s11
timeout 600 python tools/official_chunk.py --batch 1,default --host-fed > gpurun_out/s11/official.json 2> gpurun_out/s11/official.err
python -c "
import json
d=json.load(open('gpurun_out/s11/official.json'))
print({k:(round(v['value']/1e9,1),round(v['host_ms_per_chunk'],3),round(v['wall_ms_per_chunk'],3)) for k,v in d['by_batch_chunks'].items()})
print(json.dumps(d.get('host_fed',{}).get('by_window')))
print(d.get('host_fed',{}).get('uploader_alone_GBps'))
" ; tail -3 gpurun_out/s11/official.err
timeout 300 python tools/official_chunk.py --chunks 512 --batch 1 --profile > gpurun_out/s11/profile.txt 2>&1; grep -A28 "cumulative" gpurun_out/s11/profile.txt | head -40
