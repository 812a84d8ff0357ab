#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in 1 0; do
  (MXVL_MIXER_NODE=$v timeout 600 python bench.py --workload arm_encoder_large_224 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('node=$v', round(d['value'],1),'img/s', round(d['ms_per_step'],1),'ms')") 2>&1 | tail -1
done | tee gpurun_out/enc_ab.txt
timeout 600 python tools/step_ops_encoder.py 2>&1 | grep -v "Warn\|warn\|amdgpu.ids" | tail -45 > gpurun_out/enc_step_ops.txt
cut -c1-170 gpurun_out/enc_step_ops.txt | head -36
