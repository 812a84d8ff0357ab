#!/bin/bash
# round 5, call F: wide-row decode GEMM (waves split N, LDS-shared activations) -- tests + A/B against the K-split kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_report_decoder.py -m gpu -q -x -k "rows_9_to_80 or split_k or batched or qwen_width or wide_and_batched or beam_step" > gpurun_out/f_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/f_pytest.log
tail -6 gpurun_out/f_pytest.log
(timeout 300 python tools/decode_gemm_bench.py 48 80 2>&1 | grep -v amdgpu.ids) > gpurun_out/f_decode_gemm_bench.txt
cat gpurun_out/f_decode_gemm_bench.txt
for w in decode_llama7b_b16x3 decode_llama7b_b16x5 decode_qwen1p8b_b16x5; do
  for mode in wide ksplit; do
    timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --decode-gemm $mode > gpurun_out/f_bench_${w}_$mode.json 2> gpurun_out/f_bench_${w}_$mode.err
    python - "$w" "$mode" <<'PY'
import json, sys
w, mode = sys.argv[1:]
try:
    d = json.loads(open(f"gpurun_out/f_bench_{w}_{mode}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f'{w:26s} {mode:6s} {d["value"]:8.1f} tok/s  {r["kernel_ms"]:.3f} ms/token  frac {r["frac"]:.3f}')
except Exception as e:
    print(w, mode, "FAILED", e)
PY
  done
done
