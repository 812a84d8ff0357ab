#!/bin/bash
# A/B of library builds (build.py --exp N): EXPS="16 32 ..." WHAT=ab_bwd|ab_fwd
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 900 python tools/scan_r03_bench.py ${WHAT:-ab_bwd} ${ROUNDS:-3} ${EXPS:-} 2>&1 | grep -v amdgpu.ids) > $O/${TAG:-ab}.txt
cat $O/${TAG:-ab}.txt
