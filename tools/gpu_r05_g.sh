#!/bin/bash
# round 5, call G: wide kernel at 17..32 rows (A/B), decode tests after the dispatch heuristic, Qwen check
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_report_decoder.py -m gpu -q -k "rows_9_to_80 or split_k or batched or qwen_width or wide_and_batched" > gpurun_out/g_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/g_pytest.log
tail -4 gpurun_out/g_pytest.log
for w in decode_llama7b_b6x3 decode_llama7b_b8x3; do
  for mode in wide2 wide wide2 wide; do
    timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --decode-gemm $mode > gpurun_out/g_bench_${w}_$mode.json 2> gpurun_out/g_bench_${w}_$mode.err
    python - "$w" "$mode" <<'PY'
import json, sys
w, mode = sys.argv[1:]
try:
    d = json.loads(open(f"gpurun_out/g_bench_{w}_{mode}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f'{w:26s} {mode:6s} {d["value"]:8.1f} tok/s  {r["kernel_ms"]:.3f} ms/token  frac {r["frac"]:.3f}')
except Exception as e:
    print(w, mode, "FAILED", e)
PY
  done
done
for mode in wide ksplit; do
  timeout 600 python bench.py --workload decode_qwen1p8b_b16x5 --steps 3 --warmup 1 --no-cpu-baseline --decode-gemm $mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('qwen16x5 $mode', round(d['value'],1), round(d['roofline']['kernel_ms'],3))"
done
