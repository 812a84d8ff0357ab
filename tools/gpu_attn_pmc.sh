#!/bin/bash
# PMC passes of the attention kernels at the pre-training decoder shape (tools/attn_bench.py); summaries to gpurun_out/${TAG}_attn_sq*.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${TAG:-attn}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_$TAG; mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU -d $P/sq -o r -- python $R/tools/attn_bench.py > $P/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/sq2 -o r -- python $R/tools/attn_bench.py > $P/sq2.log 2>&1
cd $R
python tools/rocpd_summary.py $P/sq/r_results.db 2>&1 | grep -i "attn\|kernel  " | cut -c1-150 > $O/${TAG}_attn_sq.txt
python tools/rocpd_summary.py $P/sq2/r_results.db 2>&1 | grep -i "attn\|kernel  " | cut -c1-150 > $O/${TAG}_attn_sq2.txt
cat $O/${TAG}_attn_sq.txt $O/${TAG}_attn_sq2.txt
