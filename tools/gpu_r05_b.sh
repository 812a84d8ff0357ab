#!/bin/bash
# round 5, call B: fused-norm decode projections (tests + A/B against the round-4 split / fold path), Qwen-width golden, fp16 default decode
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_report_decoder.py tests/test_mambaxray_vl.py tests/test_abi.py tests/test_hybrid_decoder.py -m gpu -q > gpurun_out/b_pytest_decode.log 2>&1
echo "pytest decode rc=$?" >> gpurun_out/b_pytest_decode.log
tail -8 gpurun_out/b_pytest_decode.log
python -m pytest tests/test_models_gpu.py -m gpu -q -k "second_consumer or column_sums" > gpurun_out/b_pytest_models.log 2>&1
tail -3 gpurun_out/b_pytest_models.log
for w in decode_llama7b_128 decode_llama7b_b6x3 decode_llama7b_b16x3 decode_llama7b_b16x5 decode_qwen1p8b_b16x5 decode_qwen1p8b_b1x5; do
  for mode in fused split; do
    timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --decode-norm $mode > gpurun_out/b_bench_${w}_$mode.json 2> gpurun_out/b_bench_${w}_$mode.err
    python - "$w" "$mode" <<'PY'
import json, sys
w, mode = sys.argv[1:]
try:
    d = json.loads(open(f"gpurun_out/b_bench_{w}_{mode}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f'{w:26s} {mode:5s} {d["value"]:8.1f} tok/s  {r["kernel_ms"]:.3f} ms/token  frac {r["frac"]:.3f}')
except Exception as e:
    print(w, mode, "FAILED", e)
PY
  done
done
