#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
(timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15) > $O/r02_pytest_gpu_tail.log
(timeout 600 python bench.py --workload arm_encoder_large_224 2>&1 | tail -1) > $O/r02_bench_arm_encoder_large_224.json
tail -3 $O/r02_pytest_gpu_tail.log; cut -c1-200 $O/r02_bench_arm_encoder_large_224.json
