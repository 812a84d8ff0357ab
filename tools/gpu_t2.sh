#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest "$@" -q -m gpu -x -v 2>&1 | grep -v "^  File\|PASSED" | tail -40 > gpurun_out/t2.log
cat gpurun_out/t2.log | cut -c1-250
