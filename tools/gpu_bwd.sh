#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
(timeout 300 python tools/bwd_bench.py 2>&1 | grep -v Warning | grep bwd) > $O/${1:-bwd}_bench.txt
cat $O/${1:-bwd}_bench.txt
