#!/bin/bash
# round 5, call A: fp16 decode kernels + vocabulary-independent beam step + full-size cfg#3 test, first fp16 / Qwen bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_report_decoder.py tests/test_mambaxray_vl.py tests/test_abi.py -m gpu -x -q -k "not qwen_width" > gpurun_out/a_pytest_decode.log 2>&1
echo "pytest decode rc=$?" >> gpurun_out/a_pytest_decode.log
tail -5 gpurun_out/a_pytest_decode.log
python -m pytest tests/test_models_gpu.py -m gpu -x -q -k full_size > gpurun_out/a_pytest_fullsize.log 2>&1
tail -3 gpurun_out/a_pytest_fullsize.log
for w in decode_llama7b_128 decode_llama7b_128_fp16 decode_llama7b_b6x3 decode_llama7b_b6x3_fp16 decode_llama7b_b16x3 decode_qwen1p8b_b16x5 decode_qwen1p8b_b1x5; do
  timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/a_bench_$w.json 2> gpurun_out/a_bench_$w.err
  echo "$w rc=$?"; head -c 400 gpurun_out/a_bench_$w.json; echo
done
