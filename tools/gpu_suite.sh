#!/bin/bash
# the whole -m gpu suite + smoke on the box; tail of the log to gpurun_out/${TAG}_pytest_gpu_tail.log
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 1200 python -m pytest tests -q -m gpu ${PYTEST_ARGS:-} 2>&1 | tail -${TAILN:-30}) > $O/${TAG:-suite}_pytest_gpu_tail.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3) > $O/${TAG:-suite}_smoke.log
cat $O/${TAG:-suite}_pytest_gpu_tail.log $O/${TAG:-suite}_smoke.log
