#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r02b}
mkdir -p $O
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/tools/ubench/tr_probe.hip -o tr_probe 2>/dev/null && ./tr_probe > $O/${TAG}_tr_probe.txt 2>&1
cd $R
(timeout 900 python -m pytest tests/test_attention_gpu.py -q --maxfail=60 -x 2>&1 | tail -60) > $O/${TAG}_pytest_attn.log
tail -5 $O/${TAG}_tr_probe.txt; tail -60 $O/${TAG}_pytest_attn.log
