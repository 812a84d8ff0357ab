#!/bin/bash
# round 5, call M: same-box A/B of the default line, round-4 tree (commit 4833c6c, checked out under _r04_tmp) vs this tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for i in 1 2; do
  for t in _r04_tmp .; do
    (cd $t && timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', round(d['value'],2), 'img/s', round(d['ms_per_step'],2), 'ms')")
  done
done
for i in 1 2; do
  for t in _r04_tmp .; do
    (cd $t && timeout 900 python bench.py --workload decode_llama7b_b6x3 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t 6x3', round(d['value'],1), 'tok/s')")
  done
done
