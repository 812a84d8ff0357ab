#!/bin/bash
# round 5, call K: K-split choice of down_proj at 33..80 rows (8 x 128-column workgroups vs 4 x 64), prologue grid
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_report_decoder.py -m gpu -q -k "batched or qwen_width or split_k or wide_and_batched" 2>&1 | tail -3
for w in decode_llama7b_b16x5 decode_llama7b_b16x3; do
  for mode in wide r4 wide r4; do
    timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --decode-splits $mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $mode', round(d['value'],1), 'tok/s', round(d['roofline']['kernel_ms'],3), 'ms/token', round(d['roofline']['frac'],3))"
  done
done
