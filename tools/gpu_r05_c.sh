#!/bin/bash
# round 5, call C: re-run of the decode tests after the rows<=8 fused-norm / deterministic split planes change; stage-3 / R2GenCSR step lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_report_decoder.py tests/test_mambaxray_vl.py tests/test_hybrid_decoder.py -m gpu -q > gpurun_out/c_pytest_decode.log 2>&1
echo "pytest decode rc=$?" >> gpurun_out/c_pytest_decode.log
tail -8 gpurun_out/c_pytest_decode.log
for w in finetune_stage3_llama7b r2gencsr_step; do
  timeout 900 python bench.py --workload $w --steps 4 --warmup 2 > gpurun_out/c_bench_$w.json 2> gpurun_out/c_bench_$w.err
  echo "$w rc=$?"; tail -c 1500 gpurun_out/c_bench_$w.json; tail -5 gpurun_out/c_bench_$w.err
done
for w in decode_llama7b_128 decode_llama7b_b6x3; do
  timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/c_bench_$w.json 2> gpurun_out/c_bench_$w.err
  head -c 300 gpurun_out/c_bench_$w.json; echo
done
