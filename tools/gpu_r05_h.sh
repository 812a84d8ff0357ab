#!/bin/bash
# round 5, call H: the MFMA kernel as a plain NT GEMM against hipBLASLt at the headline step's shapes; decode 6x3 after the norm fix
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(python tools/gemm_nt_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/h_gemm_nt_bench.txt
(python tools/gemm_nt_bench.py --tuned 2>&1 | grep -v amdgpu.ids) > gpurun_out/h_gemm_nt_bench_tuned.txt
cat gpurun_out/h_gemm_nt_bench.txt; echo "--- tuned"; cat gpurun_out/h_gemm_nt_bench_tuned.txt
for w in decode_llama7b_b6x3 decode_llama7b_b16x5; do
  timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['value'],1), round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],3))"
done
