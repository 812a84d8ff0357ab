#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r02b}
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_attention_gpu.py -q --maxfail=60 -x 2>&1 | tail -30) > $O/${TAG}_pytest_attn.log
tail -8 $O/${TAG}_pytest_attn.log
bash tools/gpu_attn_prof.sh $TAG ${2:-}
