# host profile + kernel timeline of the chunk-by-chunk official run (gpurun_out/native/)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/native
timeout 300 python tools/official_chunk.py --batch 1,default --chunks 512 2>/dev/null | tail -1 > gpurun_out/native/official.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o run -- python $GRAFT_REPO_ROOT/tools/official_chunk.py --batch 1 --chunks 96 --sections > /tmp/kt.log 2>&1)
f=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if 'at::native' not in r['Kernel_Name']]
t0=int(rows[-60]['Start_Timestamp'])
out=open('gpurun_out/native/trace_tail.txt','w')
for r in rows[-60:]:
    n=r['Kernel_Name']
    for key in ('stream_pair','stream_partials','det_combine','gather_accumulate','copyBuffer'):
        if key in n: n=key+('<'+n.split('<')[1][:28] if '<' in n else ''); break
    out.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}  q{r.get('Queue_Id','?')} {n[:70]}\n")
PY
