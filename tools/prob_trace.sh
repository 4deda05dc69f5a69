# kernel timeline of the probabilistic run at its official chunking
# (gpurun_out/ens/): bash tools/prob_trace.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/ens
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o run -- python $GRAFT_REPO_ROOT/tools/official_probabilistic.py --chunks 256 --windows default > /tmp/kt.log 2>&1)
tail -2 /tmp/kt.log
f=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) gpurun_out/ens/prob_kernel_stats.csv
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if 'at::native' not in r['Kernel_Name']]
last=max(i for i,r in enumerate(rows) if 'gather_accumulate' in r['Kernel_Name'])
sel=rows[max(0,last-24):last+1]
t0=int(sel[0]['Start_Timestamp'])
out=open('gpurun_out/ens/trace_tail.txt','w')
for r in sel:
    nme=r['Kernel_Name']
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    out.write(f"{(st-t0)/1e3:9.1f} us  +{(en-st)/1e3:8.1f}  q{r.get('Queue_Id','?')} {nme[:90]}\n")
PY
