#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
(timeout 900 python -m pytest tests/test_mixer_gpu.py tests/test_models_gpu.py -q -m gpu -x 2>&1 | tail -3) > gpurun_out/enc3_pytest.log
cat gpurun_out/enc3_pytest.log
(timeout 600 python bench.py --workload arm_encoder_large_224 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'img/s', round(d['ms_per_step'],1),'ms')") 2>&1 | tail -1
timeout 600 python tools/step_ops_encoder.py 2>&1 | grep -i "conv1d\|total device" | cut -c1-150
