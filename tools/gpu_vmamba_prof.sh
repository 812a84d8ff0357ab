#!/bin/bash
# Dev tool: kernel-trace stats of the VMamba-base training step (one gpurun call); summary -> gpurun_out/prof_vmamba_stats.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; P=/tmp/prof_vm; mkdir -p $P $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $P/vm -o r -- python $R/bench.py --workload vmamba_base_224 --steps 4 --warmup 2 --no-cpu-baseline > $P/vm.log 2>&1
cd $R
python tools/rocpd_summary.py $P/vm/r_results.db 2>&1 | grep -v 'naive_conv\|kernel_grouped_conv\|batched_gemm_xdlops_bwd_weight\|kernel_group' | head -130 | cut -c1-200 > $O/prof_vmamba_stats.txt
tail -3 $P/vm.log >> $O/prof_vmamba_stats.txt
