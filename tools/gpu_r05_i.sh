#!/bin/bash
# round 5, call I: step A/B of the w1|w2 dgrad layout (NT on a transposed weight copy vs NN)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for mode in nt nn nt nn; do
  timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --mlp-dgrad $mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', round(d['value'],2), 'img/s', round(d['ms_per_step'],2), 'ms')"
done
