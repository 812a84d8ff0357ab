#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
(timeout 600 python -m pytest tests/test_scan_gpu.py tests/test_mixer_gpu.py -q -m gpu -x 2>&1 | tail -15) > $O/${1:-bwd2}_pytest.log
cat $O/${1:-bwd2}_pytest.log
(timeout 300 python tools/bwd_bench.py 2>&1 | grep -v Warning | grep bwd) > $O/${1:-bwd2}_bench.txt
cat $O/${1:-bwd2}_bench.txt
