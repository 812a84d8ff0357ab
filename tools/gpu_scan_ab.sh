#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
(timeout 600 python -m pytest tests/test_scan_gpu.py tests/test_mixer_gpu.py tests/test_vmamba_gpu.py -q -m gpu -x 2>&1 | tail -4) > $O/${1:-ab}_pytest.log
cat $O/${1:-ab}_pytest.log
for w in scan_fwd_target scan_fwd_target_bf16 scan_fwd_cfg2 scan_bwd_pretrain; do
  (timeout 300 python bench.py --workload $w --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step']*1e3,1),'us frac',round(d['roofline']['frac'],4))") 2>&1 | tail -1
done | tee $O/${1:-ab}_bench.txt
