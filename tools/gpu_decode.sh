#!/bin/bash
# decode bench lines (batch 1 x beam 3 and the reference's batches) + the per-token kernel timeline of one of them
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${TAG:-r04}
for W in ${WORKLOADS:-decode_llama7b_128 decode_llama7b_b6x3}; do
  (timeout 600 python bench.py --workload $W --steps 3 --warmup 1 2>&1 | tail -1) > $O/${TAG}_bench_$W.json
  cut -c1-420 $O/${TAG}_bench_$W.json; echo
done
if [ "${PROF:-1}" = "1" ]; then
  W=${PROF_WORKLOAD:-decode_llama7b_b6x3}
  cd /tmp && export TMPDIR=/tmp
  P=/tmp/prof_dec; mkdir -p $P
  timeout 600 rocprofv3 --kernel-trace --stats -d $P/st -o r -- python $R/bench.py --workload $W --steps 1 --warmup 1 > $P/st.log 2>&1
  cd $R
  python tools/decode_timeline.py $P/st/r_results.db 40 2>&1 | cut -c1-150 > $O/${TAG}_decode_timeline_$W.txt
  cat $O/${TAG}_decode_timeline_$W.txt
fi
