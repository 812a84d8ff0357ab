#!/bin/bash
# Dev tool: kernel-trace stats of one bench workload (one gpurun call); summary -> gpurun_out/prof_<workload>_stats.txt
#   bash tools/gpu_step_prof.sh <workload> [steps]
W=$1; S=${2:-4}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; P=/tmp/prof_$W; mkdir -p $P $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $P/k -o r -- python $R/bench.py --workload $W --steps $S --warmup 2 --no-cpu-baseline > $P/run.log 2>&1
cd $R
python tools/rocpd_summary.py $P/k/r_results.db 2>&1 | grep -v 'naive_conv\|kernel_grouped_conv\|batched_gemm_xdlops_bwd_weight\|kernel_group' | head -140 | cut -c1-200 > $O/prof_${W}_stats.txt
tail -2 $P/run.log | cut -c1-600 >> $O/prof_${W}_stats.txt
