#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python tools/wgrad_bench.py 65280 2>&1 | grep -v "Warn\|warn\|amdgpu.ids" > gpurun_out/wgrad_bench.txt
cat gpurun_out/wgrad_bench.txt | cut -c1-250
