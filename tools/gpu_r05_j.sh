#!/bin/bash
# round 5, call J: frozen-LLM autocast shadows A/B on the fine-tuning steps; default bench on another box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_report_decoder.py tests/test_mambaxray_vl.py -m gpu -q -k "frozen_decoder or downstream" 2>&1 | tail -3
for w in finetune_stage3_llama7b r2gencsr_step; do
  for mode in on off on off; do
    timeout 900 python bench.py --workload $w --steps 6 --warmup 2 --llm-shadows $mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $mode', round(d['value'],2), 'studies/s', round(d['ms_per_step'],2), 'ms', 'loss', round(d['config']['final_loss'],4))"
  done
done
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/j_bench_default.json
python -c "
import json
d = json.loads(open('gpurun_out/j_bench_default.json').read().strip().splitlines()[-1])
print('default', round(d['value'],2), 'img/s', round(d['ms_per_step'],2), 'ms | secondary', round(d['secondary']['value'],1), 'tok/s', round(d['secondary']['roofline']['frac'],3), '| north_star', round(d['north_star_kernel']['roofline']['kernel_ms']*1e3,1), 'us', round(d['north_star_kernel']['roofline']['frac'],3))
"
