# Round 3, A/B 12: non-temporal vs plain stores in the map-writing kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab3m
mkdir -p $O
: > $O/summary.txt
V=$GRAFT_REPO_ROOT/build/variants
for rep in 1 2; do
  for n in default mps emps; do
    lib=""; [ "$n" != default ] && lib=$V/libwb2hip_$n.so
    echo -n "$n " | tee -a $O/summary.txt
    WB2HIP_LIB=$lib timeout 100 python tools/maps_store_bench.py 2>&1 | tail -1 | tee -a $O/summary.txt
  done
done
