#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-attn}
cd $R
python tools/attn_bench.py 16 8 4080 64 block_causal --sdpa 2>&1 | grep -v Warn > $O/${TAG}_bench.txt
python tools/attn_bench.py 32 16 401 32 none --sdpa 2>&1 | grep -v Warn >> $O/${TAG}_bench.txt
python tools/attn_bench.py 64 16 197 64 none --sdpa 2>&1 | grep -v Warn >> $O/${TAG}_bench.txt
cat $O/${TAG}_bench.txt
if [ "${2:-}" = pmc ]; then
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_$TAG; mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $P/sq -o r -- python $R/tools/attn_bench.py > $P/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $P/sq2 -o r -- python $R/tools/attn_bench.py > $P/sq2.log 2>&1
cd $R
for n in sq sq2; do python tools/rocpd_summary.py $P/$n/r_results.db 2>&1 | grep "attn\|kernel " | cut -c1-170 > $O/${TAG}_$n.txt; done
cat $O/${TAG}_sq.txt $O/${TAG}_sq2.txt
tail -3 $P/sq2.log
fi
