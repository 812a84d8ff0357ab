#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python tools/step_ops_vmamba.py 2>&1 | grep -v "Warn\|warn\|amdgpu.ids" | tail -62 > gpurun_out/vm_step_ops.txt
cut -c1-180 gpurun_out/vm_step_ops.txt | head -45
