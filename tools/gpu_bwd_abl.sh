#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
: > $O/${1:-abl}_bench.txt
for a in 0 1 2 8 16 24 27 4; do
  (MXVL_BWD_ABLATE=$a timeout 300 python tools/bwd_bench.py 16 1024 4080 16 bfloat16 2>&1 | grep "bwd B") >> $O/${1:-abl}_bench.txt
done
cat $O/${1:-abl}_bench.txt
