#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r02c}
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -q -m gpu --maxfail=25 2>&1 | tail -60) > $O/${TAG}_pytest_gpu.log
(timeout 900 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2>&1 | tail -1) > $O/${TAG}_bench_default.json
(timeout 600 python bench.py --workload mae_vit_large_1280 2>&1 | tail -1) > $O/${TAG}_bench_mae.json
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_$TAG; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats -d $P/pretrain_stats -o r -- python $R/bench.py --workload arm_pretrain_large_1024 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $P/pretrain.log 2>&1
cd $R
python tools/rocpd_summary.py $P/pretrain_stats/r_results.db 2>&1 | head -60 | cut -c1-170 > $O/prof_${TAG}_pretrain_stats.txt
tail -15 $O/${TAG}_pytest_gpu.log; cut -c1-400 $O/${TAG}_bench_default.json; echo; cut -c1-400 $O/${TAG}_bench_mae.json; echo; grep -i "attn\|scan_bwd" $O/prof_${TAG}_pretrain_stats.txt | cut -c1-150
