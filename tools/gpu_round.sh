#!/bin/bash
# Dev helper run ON the GPU box via gpurun: tests, bench lines, rocprofv3 kernel stats + PMC passes.
# Only text summaries land in gpurun_out/ (it is size-capped); raw rocprof databases stay in /tmp on the box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
TAG=${1:-r01}
MODE=${2:-full}     # full: every workload + PMC passes;  lite: tests, smoke, the benches named in $LITE_WORKLOADS, two kernel-stats profiles
LITE_WORKLOADS=${LITE_WORKLOADS:-"arm_encoder_large_224 vmamba_base_224"}
(timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6) > $O/pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $O/smoke.log
(timeout 900 python bench.py 2>&1 | tail -1) > $O/bench_default.json
ALL="scan_fwd_target scan_fwd_cfg2 scan_fwd_target_bf16 arm_pretrain_base_192 decode_llama7b_128 mae_vit_large_1280 vmamba_base_224 arm_encoder_large_224"
secondary() {
  for w in $1; do
    (timeout 600 python bench.py --workload $w 2>&1 | tail -1) > $O/bench_$w.json
  done
}
[ "$MODE" != lite ] && secondary "$ALL"
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_$TAG; mkdir -p $P
prof() { # name, rocprof args..., -- cmd
  local name=$1; shift
  timeout 900 rocprofv3 "$@" > $P/$name.log 2>&1
}
prof scan_stats --kernel-trace --stats -d $P/scan_stats -o r -- python $R/bench.py --workload scan_fwd_target --steps 100 --warmup 10 --no-cpu-baseline
if [ "$MODE" != lite ]; then
prof scan_fetch --kernel-trace --pmc FETCH_SIZE -d $P/scan_fetch -o r -- python $R/bench.py --workload scan_fwd_target --steps 10 --warmup 2 --no-cpu-baseline
prof scan_write --kernel-trace --pmc WRITE_SIZE -d $P/scan_write -o r -- python $R/bench.py --workload scan_fwd_target --steps 10 --warmup 2 --no-cpu-baseline
prof scan_sq --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $P/scan_sq -o r -- python $R/bench.py --workload scan_fwd_target --steps 10 --warmup 2 --no-cpu-baseline
prof scan_sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/scan_sq2 -o r -- python $R/bench.py --workload scan_fwd_target --steps 10 --warmup 2 --no-cpu-baseline
fi
prof pretrain_stats --kernel-trace --stats -d $P/pretrain_stats -o r -- python $R/bench.py --workload arm_pretrain_large_1024 --steps 3 --warmup 1 --no-cpu-baseline
[ "$MODE" != lite ] && prof decode_stats --kernel-trace --stats -d $P/decode_stats -o r -- python $R/bench.py --workload decode_llama7b_128 --steps 1 --warmup 1
cd $R
NAMES="scan_stats scan_fetch scan_write scan_sq scan_sq2 pretrain_stats decode_stats"
[ "$MODE" = lite ] && NAMES="scan_stats pretrain_stats"
for n in $NAMES; do
  python tools/rocpd_summary.py $P/$n/r_results.db 2>&1 | head -45 | cut -c1-170 > $O/prof_${TAG}_$n.txt
done
[ "$MODE" != lite ] && python tools/pmc_traffic.py scan_fwd_target scan_fwd_stream_kernel $O/prof_${TAG}_scan_fetch.txt $O/prof_${TAG}_scan_write.txt $O/${TAG}_pmc_traffic.json > /dev/null 2>&1
if [ "$MODE" = lite ]; then   # lowest priority last: optional GEMM tuning of secondary workloads, then their bench lines
  if [ -n "${TUNE_WORKLOADS:-}" ]; then
    (timeout 300 python tools/tune_gemms.py $TUNE_WORKLOADS 2>&1 | tail -1) > $O/tune.log
    cp medical_image_analysis_amd/tuned/tunableop_gfx950.csv $O/tunableop_gfx950.csv
  fi
  secondary "$LITE_WORKLOADS"
fi
cat $O/pytest_gpu.log $O/smoke.log; for f in $O/bench_*.json; do echo "== $f"; cut -c1-260 $f; done
