#!/bin/bash
# round 5, call E: per-token kernel timelines of the wide decode workloads + the MAE step's kernel stats (after the patch-embedding im2col kernel)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_e; mkdir -p $P
prof() { local name=$1; shift; timeout 600 rocprofv3 "$@" > $P/$name.log 2>&1; }
prof decq --kernel-trace --stats -d $P/decq -o r -- python $R/bench.py --workload decode_qwen1p8b_b16x5 --steps 1 --warmup 1 --no-cpu-baseline
prof dec48 --kernel-trace --stats -d $P/dec48 -o r -- python $R/bench.py --workload decode_llama7b_b16x3 --steps 1 --warmup 1 --no-cpu-baseline
prof dec1 --kernel-trace --stats -d $P/dec1 -o r -- python $R/bench.py --workload decode_llama7b_128 --steps 1 --warmup 1 --no-cpu-baseline
prof mae --kernel-trace --stats -d $P/mae -o r -- python $R/bench.py --workload mae_vit_large_1280 --steps 3 --warmup 1 --no-cpu-baseline
cd $R
python tools/decode_timeline.py $P/decq/r_results.db 40 2>&1 | cut -c1-150 > $O/e_decode_timeline_qwen_b16x5.txt
python tools/decode_timeline.py $P/dec48/r_results.db 40 2>&1 | cut -c1-150 > $O/e_decode_timeline_llama_b16x3.txt
python tools/decode_timeline.py $P/dec1/r_results.db 40 2>&1 | cut -c1-150 > $O/e_decode_timeline_llama_128.txt
python tools/rocpd_summary.py $P/mae/r_results.db 2>&1 | head -60 | cut -c1-170 > $O/e_mae_stats.txt
head -45 $O/e_decode_timeline_qwen_b16x5.txt
