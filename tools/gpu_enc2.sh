#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
(timeout 900 python -m pytest tests/test_mixer_gpu.py tests/test_models_gpu.py tests/test_lora_clip_golden.py tests/test_mambaxray_vl.py -q -m gpu -x 2>&1 | tail -3) > gpurun_out/enc2_pytest.log
cat gpurun_out/enc2_pytest.log
for v in 1 0 1; do
  (MXVL_MIXER_NODE=$v timeout 600 python bench.py --workload arm_encoder_large_224 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('node=$v', round(d['value'],1),'img/s', round(d['ms_per_step'],1),'ms')") 2>&1 | tail -1
done | tee gpurun_out/enc_ab.txt
