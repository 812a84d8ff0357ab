#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python tools/fwd_variant_bench.py 2>&1 | grep -v "Warn\|warn\|amdgpu.ids" | tee gpurun_out/fwd_variants.txt
