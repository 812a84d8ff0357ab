#!/bin/bash
# Round-2 GPU call A (run ON the box via gpurun): full -m gpu suite, smoke, scan_bwd A/B, default bench line (with the
# decode secondary), PMC traffic + SQ passes of the headline kernel (scan_bwd at the pre-training shape), step profile.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r02a}
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -q -m gpu --maxfail=25 2>&1 | tail -80) > $O/${TAG}_pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $O/${TAG}_smoke.log
(timeout 300 python tools/bwd_bench.py 2>&1 | grep -v Warning | tail -16) > $O/${TAG}_bwd_bench.txt
(timeout 900 python bench.py --steps 10 --warmup 3 2>&1 | tail -1) > $O/${TAG}_bench_default.json
(timeout 300 python bench.py --workload scan_bwd_pretrain 2>&1 | tail -1) > $O/${TAG}_bench_scan_bwd_pretrain.json
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_$TAG; mkdir -p $P
prof() { local name=$1; shift; timeout 600 rocprofv3 "$@" > $P/$name.log 2>&1; }
BW="python $R/bench.py --workload scan_bwd_pretrain --steps 5 --warmup 2 --no-cpu-baseline"
prof bwd_stats --kernel-trace --stats -d $P/bwd_stats -o r -- python $R/bench.py --workload scan_bwd_pretrain --steps 50 --warmup 5 --no-cpu-baseline
prof bwd_fetch --kernel-trace --pmc FETCH_SIZE -d $P/bwd_fetch -o r -- $BW
prof bwd_write --kernel-trace --pmc WRITE_SIZE -d $P/bwd_write -o r -- $BW
prof bwd_sq --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $P/bwd_sq -o r -- $BW
prof bwd_sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/bwd_sq2 -o r -- $BW
prof pretrain_stats --kernel-trace --stats -d $P/pretrain_stats -o r -- python $R/bench.py --workload arm_pretrain_large_1024 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
cd $R
for n in bwd_stats bwd_fetch bwd_write bwd_sq bwd_sq2; do
  python tools/rocpd_summary.py $P/$n/r_results.db 2>&1 | head -40 | cut -c1-170 > $O/prof_${TAG}_$n.txt
done
python tools/rocpd_summary.py $P/pretrain_stats/r_results.db 2>&1 | head -90 | cut -c1-170 > $O/prof_${TAG}_pretrain_stats.txt
python tools/pmc_traffic.py scan_bwd_pretrain scan_bwd_kernel $O/prof_${TAG}_bwd_fetch.txt $O/prof_${TAG}_bwd_write.txt $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.log 2>&1
tail -25 $O/${TAG}_pytest_gpu.log; cat $O/${TAG}_smoke.log $O/${TAG}_bwd_bench.txt; cut -c1-600 $O/${TAG}_bench_default.json; echo; cat $O/${TAG}_pmc_traffic.log
