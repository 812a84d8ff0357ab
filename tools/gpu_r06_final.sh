#!/bin/bash
# Round-6 evidence run ON the GPU box (same structure as gpu_r05_final.sh) (via gpurun): the full -m gpu suite, smoke(), PMC traffic passes (FETCH / WRITE in separate
# rocprofv3 passes) of both scan kernels and of the decode projection kernel, kernel-trace stats of the default workload, per-token
# kernel timelines of the decode workloads, the default bench line (with secondary and cpu_baseline) and the secondary workloads.
# Only text summaries land in gpurun_out/ ; what is to be judged is copied into profiles/.
# PARTS=decode (third pass of round 5, after the decode projections changed): the suite, smoke(), and the decode-side evidence only.
set -u
PARTS=${PARTS:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r06}
mkdir -p $O
cd $R
(timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -15) > $O/${TAG}_pytest_gpu_tail.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3) > $O/${TAG}_smoke.log
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_$TAG; mkdir -p $P
BW="python $R/bench.py --workload scan_bwd_pretrain --steps 5 --warmup 2 --no-cpu-baseline"
FW="python $R/bench.py --workload scan_fwd_target --steps 10 --warmup 2 --no-cpu-baseline"
DG="python $R/tools/decode_gemm_bench.py 18"
prof() { local name=$1; shift; timeout 600 rocprofv3 "$@" > $P/$name.log 2>&1; }
if [ $PARTS = all ]; then
prof bwd_fetch --kernel-trace --pmc FETCH_SIZE -d $P/bwd_fetch -o r -- $BW
prof bwd_write --kernel-trace --pmc WRITE_SIZE -d $P/bwd_write -o r -- $BW
prof fwd_fetch --kernel-trace --pmc FETCH_SIZE -d $P/fwd_fetch -o r -- $FW
prof fwd_write --kernel-trace --pmc WRITE_SIZE -d $P/fwd_write -o r -- $FW
fi
prof dg_fetch --kernel-trace --pmc FETCH_SIZE -d $P/dg_fetch -o r -- $DG
prof dg_write --kernel-trace --pmc WRITE_SIZE -d $P/dg_write -o r -- $DG
prof dg_sq --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY -d $P/dg_sq -o r -- $DG
[ $PARTS = all ] && prof pretrain_mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $P/pretrain_mfma -o r -- python $R/bench.py --workload arm_pretrain_large_1024 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
if [ $PARTS = all ]; then
prof tn_fetch --kernel-trace --pmc FETCH_SIZE -d $P/tn_fetch -o r -- python $R/bench.py --workload arm_pretrain_large_1024 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
prof tn_write --kernel-trace --pmc WRITE_SIZE -d $P/tn_write -o r -- python $R/bench.py --workload arm_pretrain_large_1024 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
fi
[ $PARTS = all ] && prof pretrain_stats --kernel-trace --stats -d $P/pretrain_stats -o r -- python $R/bench.py --workload arm_pretrain_large_1024 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
prof dec1 --kernel-trace --stats -d $P/dec1 -o r -- python $R/bench.py --workload decode_llama7b_128 --steps 1 --warmup 1 --no-cpu-baseline
prof dec18 --kernel-trace --stats -d $P/dec18 -o r -- python $R/bench.py --workload decode_llama7b_b6x3 --steps 1 --warmup 1 --no-cpu-baseline
prof dec80 --kernel-trace --stats -d $P/dec80 -o r -- python $R/bench.py --workload decode_llama7b_b16x5 --steps 1 --warmup 1 --no-cpu-baseline
prof decq --kernel-trace --stats -d $P/decq -o r -- python $R/bench.py --workload decode_qwen1p8b_b16x5 --steps 1 --warmup 1 --no-cpu-baseline
cd $R
for n in bwd_fetch bwd_write fwd_fetch fwd_write dg_fetch dg_write dg_sq pretrain_stats tn_fetch tn_write; do
  [ -f $P/$n/r_results.db ] || continue
  case $n in
    tn_*) python tools/rocpd_summary.py $P/$n/r_results.db 2>&1 | grep -E "^==|^kernel|gemm_tn" | cut -c1-170 > $O/prof_${TAG}_$n.txt ;;   # a whole step: keep this kernel's rows
    *) python tools/rocpd_summary.py $P/$n/r_results.db 2>&1 | head -70 | cut -c1-170 > $O/prof_${TAG}_$n.txt ;;
  esac
done
[ -f $P/pretrain_mfma/r_results.db ] && python tools/mfma_busy.py $P/pretrain_mfma/r_results.db 30 2>&1 | cut -c1-170 > $O/${TAG}_pretrain_mfma_busy.txt
python tools/decode_timeline.py $P/dec1/r_results.db 40 2>&1 | cut -c1-150 > $O/${TAG}_decode_timeline_decode_llama7b_128.txt
python tools/decode_timeline.py $P/dec18/r_results.db 40 2>&1 | cut -c1-150 > $O/${TAG}_decode_timeline_decode_llama7b_b6x3.txt
python tools/decode_timeline.py $P/dec80/r_results.db 40 2>&1 | cut -c1-150 > $O/${TAG}_decode_timeline_decode_llama7b_b16x5.txt
python tools/decode_timeline.py $P/decq/r_results.db 40 2>&1 | cut -c1-150 > $O/${TAG}_decode_timeline_decode_qwen1p8b_b16x5.txt
if [ $PARTS = all ]; then
rm -f $O/${TAG}_pmc_traffic.json
python tools/pmc_traffic.py scan_bwd_pretrain scan_bwd_kernel $O/prof_${TAG}_bwd_fetch.txt $O/prof_${TAG}_bwd_write.txt $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.log 2>&1
python tools/pmc_traffic.py scan_fwd_target scan_fwd_stream_kernel $O/prof_${TAG}_fwd_fetch.txt $O/prof_${TAG}_fwd_write.txt $O/${TAG}_pmc_traffic.json >> $O/${TAG}_pmc_traffic.log 2>&1
python tools/pmc_traffic.py gemm_tn_pretrain gemm_tn_kernel $O/prof_${TAG}_tn_fetch.txt $O/prof_${TAG}_tn_write.txt $O/${TAG}_pmc_traffic.json >> $O/${TAG}_pmc_traffic.log 2>&1
cp $O/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json     # bench.py reads the record from profiles/
(timeout 900 python bench.py 2>&1 | tail -1) > $O/${TAG}_bench_default.json
fi
[ $PARTS = decode ] && WORKLOADS=${WORKLOADS:-decode_llama7b_b6x3 decode_llama7b_b6x3_fp16 decode_llama7b_b8x3 decode_llama7b_b16x3 decode_llama7b_b16x5 decode_qwen1p8b_b16x5 decode_llama7b_128 decode_qwen1p8b_b1x5}
for w in ${WORKLOADS:-scan_bwd_pretrain scan_fwd_target scan_fwd_cfg2 scan_fwd_target_bf16 decode_llama7b_128 decode_llama7b_128_fp16 decode_llama7b_b6x3 decode_llama7b_b6x3_fp16 decode_llama7b_b8x3 decode_llama7b_b16x3 decode_llama7b_b16x5 decode_qwen1p8b_b16x5 decode_qwen1p8b_b1x5 finetune_stage3_llama7b r2gencsr_step mae_vit_large_1280 arm_encoder_large_224 vmamba_base_224 arm_pretrain_base_192}; do
  (timeout 600 python bench.py --workload $w 2>&1 | tail -1) > $O/${TAG}_bench_$w.json
done
for w in ${GRAPH_WORKLOADS:-arm_pretrain_base_192 vmamba_base_224}; do
  (timeout 600 python bench.py --workload $w --graph 2>&1 | tail -1) > $O/${TAG}_bench_${w}_graph.json
done
(timeout 300 python tools/decode_gemm_bench.py 3 18 24 48 80 2>&1 | grep -v amdgpu.ids) > $O/${TAG}_decode_gemm_bench.txt
if [ $PARTS = all ]; then
(timeout 300 python tools/gemm_swiglu_bwd_bench.py 2>&1 | grep -v amdgpu.ids) > $O/${TAG}_gemm_swiglu_bwd_bench.txt
(timeout 300 python tools/wgrad_tn_bench.py --tuned 2>&1 | grep -v amdgpu.ids) > $O/${TAG}_wgrad_tn_bench_final.txt
(timeout 300 python tools/scan_r03_bench.py all 2 2>&1 | grep -v amdgpu.ids) > $O/${TAG}_scan_variants.txt
(timeout 300 python tools/scan_r03_bench.py n1 3 2>&1 | grep -v amdgpu.ids) > $O/${TAG}_scan_n1_final.txt
(timeout 300 python tools/step_eager.py 16 pretrain 2>&1 | grep -v "amdgpu.ids\|arn") > $O/${TAG}_step_eager_pretrain.txt
(timeout 300 python tools/step_eager.py 32 vmamba 2>&1 | grep -v "amdgpu.ids\|arn") > $O/${TAG}_step_eager_vmamba.txt
(timeout 300 python tools/cross_bench.py 2>&1 | grep -v amdgpu.ids) > $O/${TAG}_cross_dwconv_bench.txt
(timeout 600 python tools/step_eager.py 0 r2gencsr 2>&1 | grep -v "amdgpu.ids\|arn" | cut -c1-150 | head -60) > $O/${TAG}_step_eager_r2gencsr_after.txt
fi
cat $O/${TAG}_pytest_gpu_tail.log $O/${TAG}_smoke.log $O/${TAG}_pmc_traffic.log; for f in $O/${TAG}_bench_*.json; do echo "== $f"; cut -c1-330 $f; done
