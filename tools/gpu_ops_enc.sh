#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python tools/step_ops_encoder.py 2>&1 | grep -v "Warn\|warn\|amdgpu.ids" | tail -64 > gpurun_out/enc_step_ops.txt
cut -c1-200 gpurun_out/enc_step_ops.txt
