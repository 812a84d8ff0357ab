#!/bin/bash
# Dev helper: LDS bank-conflict counters of every kernel of the headline step (one PMC pass over 2 steps).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_steplds; mkdir -p $P
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS -d $P/sq -o r -- python $R/bench.py --steps 1 --warmup 1 --no-secondary --no-cpu-baseline > $P/sq.log 2>&1
cd $R
python tools/rocpd_summary.py $P/sq/r_results.db 2>&1 | cut -c1-170 > $O/step_lds.txt
grep "SQ_LDS_BANK_CONFLICT" $O/step_lds.txt | sort -t= -k3 -g -r | head -30
