#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python tools/step_ops.py 2>&1 | grep -v "Warn\|warn" | tail -75 > gpurun_out/${1:-ops}_step_ops.txt
cat gpurun_out/${1:-ops}_step_ops.txt | cut -c1-190
