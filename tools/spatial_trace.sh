# kernel timeline of deterministic_spatial in windows (gpurun_out/spatial/)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/spatial
rm -rf /tmp/kt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o run -- python $GRAFT_REPO_ROOT/tools/spatial_leg.py --window > /tmp/kt.log 2>&1)
python - $(find /tmp/kt -name "*kernel_trace.csv" | head -1) /nonexistent <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
try:
    cp=list(csv.DictReader(open(sys.argv[2])))
    for r in cp: r['Kernel_Name']='COPY '+r.get('Direction','')
    rows+=cp
except Exception as e: print('no copies', e)
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=max(i for i,r in enumerate(rows) if 'spatial_accumulate_addr' in r['Kernel_Name'])
sel=rows[max(0,last-70):last+1]
t0=int(sel[0]['Start_Timestamp'])
out=open('gpurun_out/spatial/trace_tail.txt','w')
for r in sel:
    nme=r['Kernel_Name'].replace('wb2::(anonymous namespace)::','').replace('void ','')
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    out.write(f"{(st-t0)/1e3:9.1f} us  +{(en-st)/1e3:8.1f}  q{r.get('Queue_Id','?')} {nme[:70]}\n")
PY
