#!/bin/bash
# Dev helper: run the given pytest selection on the GPU box, full output tail into gpurun_out/t.log
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest "$@" -q -m gpu -x 2>&1 | tail -40 > gpurun_out/t.log
cat gpurun_out/t.log
