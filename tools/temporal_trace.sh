# kernel timeline of deterministic_temporal in default windows (gpurun_out/temporal/)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/temporal
for which in temporal det; do
  extra=""; [ $which = det ] && extra="x"
  rm -rf /tmp/kt
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/kt -o run -- python $GRAFT_REPO_ROOT/tools/temporal_trace.py 240 default $extra > /tmp/kt.log 2>&1)
  cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) gpurun_out/temporal/${which}_kernel_stats.csv
  python - $(find /tmp/kt -name '*kernel_trace.csv' | head -1) $(find /tmp/kt -name '*memory_copy_trace.csv' | head -1) $which <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows: r['k']='K'
try:
    cp=list(csv.DictReader(open(sys.argv[2])))
    for r in cp:
        r['k']='C'; r['Kernel_Name']='COPY '+r.get('Direction','')+' '+r.get('Bytes','')
    rows+=cp
except Exception as e:
    print('no copies', e)
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=max(i for i,r in enumerate(rows) if 'gather_accumulate' in r['Kernel_Name'])
sel=rows[max(0,last-45):last+1]
t0=int(sel[0]['Start_Timestamp'])
out=open(f'gpurun_out/temporal/{sys.argv[3]}_tail.txt','w')
for r in sel:
    nme=r['Kernel_Name'].replace('wb2::(anonymous namespace)::','').replace('void ','')
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    out.write(f"{(st-t0)/1e3:9.1f} us  +{(en-st)/1e3:8.1f}  q{r.get('Queue_Id','?')} {nme[:80]}\n")
PY
done
