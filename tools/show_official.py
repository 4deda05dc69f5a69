import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
for k,v in d['by_batch_chunks'].items():
    r=v.get('roofline',{})
    print(k, round(v['value']/1e9,1), 'wall', round(v['wall_ms_per_chunk'],4), 'host', round(v['host_ms_per_chunk'],4), 'k1', round(r.get('k1_ms_per_chunk',0),4), 'frac', round(r.get('frac',0),3))
