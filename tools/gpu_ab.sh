# One-off GPU checks of a kernel change (edit freely; not part of the product):
cd $GRAFT_REPO_ROOT
timeout 600 python tools/official_chunk.py --chunks 256 --batch 1,8,16,32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d['by_batch_chunks'].items(): print(k, round(v['value']/1e9,1), 'wall', round(v['wall_ms_per_chunk'],3), 'host', round(v['host_ms_per_chunk'],3))"
