#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
(timeout 900 python bench.py 2>&1 | tail -1) > $O/r02_bench_default.json
for w in scan_bwd_pretrain scan_fwd_target scan_fwd_cfg2 scan_fwd_target_bf16 arm_encoder_large_224; do
  (timeout 600 python bench.py --workload $w 2>&1 | tail -1) > $O/r02_bench_$w.json
done
for f in $O/r02_bench_default.json $O/r02_bench_scan_*.json $O/r02_bench_arm_encoder_large_224.json; do echo "== $f"; cut -c1-200 $f; done
