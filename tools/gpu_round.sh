#!/bin/bash
# Dev helper run ON the GPU box via gpurun: tests, bench lines, rocprofv3 kernel stats + PMC passes.
# Everything lands in gpurun_out/ (scratch); summaries worth keeping are copied to profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
TAG=${1:-r01}
(timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > $O/pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $O/smoke.log
for w in scan_fwd_target scan_fwd_cfg2 scan_fwd_target_bf16; do
  (timeout 600 python bench.py --workload $w 2>&1 | tail -1) > $O/bench_$w.json
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stats_$TAG -o scan -- python $R/bench.py --workload scan_fwd_target --steps 50 --warmup 5 --no-cpu-baseline > $O/prof_stats_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch_$TAG -o scan -- python $R/bench.py --workload scan_fwd_target --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_fetch_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_write_$TAG -o scan -- python $R/bench.py --workload scan_fwd_target --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_write_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $O/prof_sq_$TAG -o scan -- python $R/bench.py --workload scan_fwd_target --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_sq_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/prof_sq2_$TAG -o scan -- python $R/bench.py --workload scan_fwd_target --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_sq2_$TAG.log 2>&1
cd $R
find $O -name "*.csv" | head -50 > $O/csv_list.txt
cat $O/pytest_gpu.log $O/smoke.log $O/bench_*.json
