#!/bin/bash
# default bench line + kernel-trace stats of the default workload
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${TAG:-bench}
(timeout 900 python bench.py ${BENCH_ARGS:-} 2>&1 | tail -1) > $O/${TAG}_bench_default.json
if [ "${PROF:-1}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  P=/tmp/prof_$TAG; mkdir -p $P
  timeout 600 rocprofv3 --kernel-trace --stats -d $P/st -o r -- python $R/bench.py --workload arm_pretrain_large_1024 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $P/st.log 2>&1
  cd $R
  python tools/rocpd_summary.py $P/st/r_results.db 2>&1 | head -70 | cut -c1-170 > $O/${TAG}_pretrain_stats.txt
fi
cut -c1-900 $O/${TAG}_bench_default.json; echo; cat $O/${TAG}_pretrain_stats.txt 2>/dev/null | head -60
