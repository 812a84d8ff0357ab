#!/bin/bash
# The A/B runs of round 5, each as it was run in ONE gpurun call (the results are the profiles/r05_*_ab.txt / *_bench.txt files that name
# their section here).   gpurun -- 'bash tools/gpu_r05_ab.sh <section>'
#   norm       RMSNorm fused into the consuming decode projection vs K-split + explicit fold       -> profiles/r05_decode_norm_ab.txt
#   wide       17..80-row projections: waves split N + LDS-shared activations vs K-split kernels    -> profiles/r05_decode_gemm_wide_ab.txt (first version, 33..80 rows), r05_decode_gemm_wide_rows_ab.txt
#   ring       the wide kernel's LDS ring: as deep as 150 KB hold (5..8 stages) vs 3 stages          -> profiles/r05_decode_gemm_ring_ab.txt
#   probe      what one CU / the chip streams from HBM by load path and pattern (tools/cu_stream_probe.hip) -> profiles/r05_cu_stream_probe.txt
#   mlp        SwiGLU backward inside w3's dgrad GEMM vs the two-kernel backward (kernel + step)    -> profiles/r05_gemm_swiglu_bwd_bench.txt
#   gemm       the MFMA kernel as a plain NT GEMM vs hipBLASLt at the step's shapes                 -> profiles/r05_gemm_nt_bench.txt
#   shadows    cached autocast copies of the frozen LLM's weights vs per-call casts                 -> profiles/r05_llm_shadows_ab.txt
#   secondary  the default line's decode leg vs the standalone decode line                          -> profiles/r05_secondary_warmup.txt
#   timelines  per-token kernel timelines of the wide decode workloads + the MAE step's stats       -> profiles/r05_decode_timeline_*, r05_mae_stats.txt
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
line() {   # workload, then extra bench.py flags -> one summary line
  local w=$1; shift
  timeout 900 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$w', '$*', round(d['value'], 1), d['unit'], round(r.get('kernel_ms', d['ms_per_step']), 3), 'ms', round(r['frac'], 3))"
}
case "$1" in
  norm)
    for w in decode_llama7b_128 decode_llama7b_b6x3 decode_llama7b_b16x3 decode_llama7b_b16x5 decode_qwen1p8b_b16x5 decode_qwen1p8b_b1x5; do
      for m in fused split; do line $w --decode-norm $m; done; done ;;
  wide)
    (python tools/decode_gemm_bench.py 48 80 2>&1 | grep -v amdgpu.ids)
    for w in decode_llama7b_b16x3 decode_llama7b_b16x5 decode_qwen1p8b_b16x5 decode_llama7b_b6x3; do
      for m in wide ksplit; do line $w --decode-gemm $m; done; done ;;
  ring)
    timeout 600 python -m pytest tests/test_report_decoder.py -q -m gpu -k "wide or decode_gemm or qwen_width" 2>&1 | tail -4
    for w in decode_llama7b_b16x3 decode_llama7b_b16x5 decode_qwen1p8b_b16x5 decode_llama7b_b6x3; do
      for m in wide wide_nw4 wide_pf3 wide wide_nw4 wide_pf3; do line $w --decode-gemm $m; done; done ;;
  probe)
    hipcc --offload-arch=gfx950 -O3 -o /tmp/cu_stream_probe tools/cu_stream_probe.hip 2>/dev/null || exit 1
    timeout 200 /tmp/cu_stream_probe
    for m in ROWS_ONLY SWZ_ONLY VROWS_ONLY SHORT_ONLY MALL_ONLY; do echo "## $m"; env $m=1 timeout 200 /tmp/cu_stream_probe; done ;;
  mlp)
    (python tools/gemm_swiglu_bwd_bench.py 2>&1 | grep -v amdgpu.ids)
    for m in fused unfused fused unfused; do
      timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --mlp-bwd $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', round(d['value'],2), 'img/s', round(d['ms_per_step'],2), 'ms')"
    done ;;
  gemm)
    (python tools/gemm_nt_bench.py 2>&1 | grep -v amdgpu.ids); echo "--- tuned"; (python tools/gemm_nt_bench.py --tuned 2>&1 | grep -v amdgpu.ids) ;;
  shadows)
    for w in finetune_stage3_llama7b r2gencsr_step; do for m in on off on off; do
      timeout 900 python bench.py --workload $w --steps 6 --warmup 2 --llm-shadows $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $m', round(d['value'],2), 'studies/s', round(d['ms_per_step'],2), 'ms', 'loss', round(d['config']['final_loss'],4))"
    done; done ;;
  secondary)
    line decode_llama7b_128
    for n in 1 3 8; do
      timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --secondary-warmup $n 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('decode leg after the training leg, $n warm-up generate():', round(d['secondary']['value'],1), 'tok/s | images/s', round(d['value'],2))"
    done ;;
  timelines)
    P=/tmp/prof_ab; mkdir -p $P; R=$(pwd)
    for w in decode_qwen1p8b_b16x5 decode_llama7b_b16x3 decode_llama7b_128; do
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $P/$w -o r -- python $R/bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline > $P/$w.log 2>&1)
      python tools/decode_timeline.py $P/$w/r_results.db 40 2>&1 | cut -c1-150 > gpurun_out/ab_decode_timeline_$w.txt
    done
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $P/mae -o r -- python $R/bench.py --workload mae_vit_large_1280 --steps 3 --warmup 1 --no-cpu-baseline > $P/mae.log 2>&1)
    python tools/rocpd_summary.py $P/mae/r_results.db 2>&1 | head -60 | cut -c1-170 > gpurun_out/ab_mae_stats.txt ;;
  *) echo "usage: $0 norm|wide|ring|probe|mlp|gemm|shadows|secondary|timelines"; exit 2 ;;
esac
