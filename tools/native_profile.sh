# kernel timeline of the official run (gpurun_out/native/): bash tools/native_profile.sh [batch]
B=${1:-1}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/native
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o run -- python $GRAFT_REPO_ROOT/tools/official_chunk.py --batch $B --chunks 192 --sections > /tmp/kt.log 2>&1)
f=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
python - "$f" $B <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if 'at::native' not in r['Kernel_Name']]
n=40 if sys.argv[2]=='1' else 60
# find the last gather_accumulate and print the n kernels before it
last=max(i for i,r in enumerate(rows) if 'gather_accumulate' in r['Kernel_Name'])
sel=rows[max(0,last-n):last+1]
t0=int(sel[0]['Start_Timestamp'])
out=open(f'gpurun_out/native/trace_tail_b{sys.argv[2]}.txt','w')
prev_end=None
for r in sel:
    nme=r['Kernel_Name']
    for key in ('stream_pair','stream_partials','det_combine','gather_accumulate','copyBuffer'):
        if key in nme: nme=key+('<'+nme.split('<')[1][:28] if '<' in nme else ''); break
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    out.write(f"{(st-t0)/1e3:9.1f} us  +{(en-st)/1e3:8.1f}  q{r.get('Queue_Id','?')} {nme[:70]}\n")
PY
