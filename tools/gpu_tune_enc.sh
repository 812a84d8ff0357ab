#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
(timeout 600 python tools/tune_gemms.py arm_encoder_large_224 2>&1 | tail -2) > $O/tune_enc.log
cp medical_image_analysis_amd/tuned/tunableop_gfx950.csv $O/tunableop_gfx950_enc.csv
(timeout 600 python bench.py --workload arm_encoder_large_224 2>&1 | tail -1 | cut -c1-200) > $O/tune_enc_after.txt
(timeout 600 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200) >> $O/tune_enc_after.txt
cat $O/tune_enc.log $O/tune_enc_after.txt; wc -l $O/tunableop_gfx950_enc.csv
