#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
(timeout 600 python -m pytest tests/test_scan_gpu.py tests/test_mixer_gpu.py -q -m gpu -x 2>&1 | tail -2) > $O/lds_pytest.log
cat $O/lds_pytest.log
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_lds; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $P/fwd_sq -o r -- python $R/bench.py --workload scan_fwd_target --steps 10 --warmup 2 --no-cpu-baseline > $P/fwd.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $P/bwd_sq -o r -- python $R/bench.py --workload scan_bwd_pretrain --steps 5 --warmup 2 --no-cpu-baseline > $P/bwd.log 2>&1
cd $R
for n in fwd_sq bwd_sq; do python tools/rocpd_summary.py $P/$n/r_results.db 2>&1 | head -60 | cut -c1-170 > $O/prof_lds_$n.txt; done
grep "SQ_LDS" $O/prof_lds_fwd_sq.txt | grep "scan_fwd_stream"; grep "SQ_LDS" $O/prof_lds_bwd_sq.txt | grep "scan_bwd"
