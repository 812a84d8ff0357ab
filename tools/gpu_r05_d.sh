#!/bin/bash
# round 5, call D: SwiGLU backward fused into the dgrad GEMM -- tests, kernel A/B, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_fused_ops_gpu.py tests/test_models_gpu.py -m gpu -q -x > gpurun_out/d_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/d_pytest.log
tail -12 gpurun_out/d_pytest.log
python tools/gemm_swiglu_bwd_bench.py > gpurun_out/d_gemm_swiglu_bwd_bench.txt 2>&1
cat gpurun_out/d_gemm_swiglu_bwd_bench.txt | tail -4
for mode in fused unfused fused unfused; do
  timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --mlp-bwd $mode > gpurun_out/d_bench_default_$mode.json 2> gpurun_out/d_bench_default_$mode.err
  python - "$mode" <<'PY'
import json, sys
m = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/d_bench_default_{m}.json").read().strip().splitlines()[-1])
    print(m, round(d["value"], 2), "img/s", round(d["ms_per_step"], 2), "ms", d["roofline"]["kernel"], round(d["roofline"]["frac"], 4))
except Exception as e:
    print(m, "FAILED", e)
PY
done
