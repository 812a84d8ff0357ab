#!/bin/bash
# Dev helper run ON the GPU box: re-tune the library GEMM table for the default workload, then bench with the old and new tables.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
(timeout 600 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2>&1 | tail -1) > $O/tune_bench_before.json
cp medical_image_analysis_amd/tuned/tunableop_gfx950.csv /tmp/old_table.csv
(MXVL_TUNE_MS=60 MXVL_TUNE_ITERS=40 timeout 1500 python tools/tune_gemms.py arm_pretrain_large_1024 2>&1 | tail -3) > $O/tune.log
cp medical_image_analysis_amd/tuned/tunableop_gfx950.csv $O/tunableop_gfx950.csv
(timeout 600 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2>&1 | tail -1) > $O/tune_bench_after.json
cat $O/tune.log; cut -c1-220 $O/tune_bench_before.json; echo; cut -c1-220 $O/tune_bench_after.json; echo; wc -l $O/tunableop_gfx950.csv
