#!/bin/bash
# Round-3 evidence run ON the GPU box (via gpurun): the full -m gpu suite, smoke(), the PMC traffic passes of both scan kernels
# (FETCH / WRITE in separate rocprofv3 passes, as MI355X_MICROARCH.md prescribes), SQ counters, kernel-trace stats of the default
# workload, the default bench line (with secondary and cpu_baseline) and the secondary workloads.
# Only text summaries land in gpurun_out/ ; copy what is to be judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r03}
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15) > $O/${TAG}_pytest_gpu_tail.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3) > $O/${TAG}_smoke.log
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_$TAG; mkdir -p $P
BW="python $R/bench.py --workload scan_bwd_pretrain --steps 5 --warmup 2 --no-cpu-baseline"
FW="python $R/bench.py --workload scan_fwd_target --steps 10 --warmup 2 --no-cpu-baseline"
prof() { local name=$1; shift; timeout 600 rocprofv3 "$@" > $P/$name.log 2>&1; }
prof bwd_fetch --kernel-trace --pmc FETCH_SIZE -d $P/bwd_fetch -o r -- $BW
prof bwd_write --kernel-trace --pmc WRITE_SIZE -d $P/bwd_write -o r -- $BW
prof fwd_fetch --kernel-trace --pmc FETCH_SIZE -d $P/fwd_fetch -o r -- $FW
prof fwd_write --kernel-trace --pmc WRITE_SIZE -d $P/fwd_write -o r -- $FW
prof bwd_sq --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $P/bwd_sq -o r -- $BW
prof bwd_sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/bwd_sq2 -o r -- $BW
prof fwd_sq --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $P/fwd_sq -o r -- $FW
prof gemm_sq --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY -d $P/gemm_sq -o r -- python $R/tools/gemm_swiglu_bench.py 1
prof pretrain_stats --kernel-trace --stats -d $P/pretrain_stats -o r -- python $R/bench.py --workload arm_pretrain_large_1024 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
prof encoder_stats --kernel-trace --stats -d $P/encoder_stats -o r -- python $R/bench.py --workload arm_encoder_large_224 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
cd $R
for n in bwd_fetch bwd_write fwd_fetch fwd_write bwd_sq bwd_sq2 fwd_sq gemm_sq pretrain_stats encoder_stats; do
  python tools/rocpd_summary.py $P/$n/r_results.db 2>&1 | head -70 | cut -c1-170 > $O/prof_${TAG}_$n.txt
done
rm -f $O/${TAG}_pmc_traffic.json
python tools/pmc_traffic.py scan_bwd_pretrain scan_bwd_kernel $O/prof_${TAG}_bwd_fetch.txt $O/prof_${TAG}_bwd_write.txt $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.log 2>&1
python tools/pmc_traffic.py scan_fwd_target scan_fwd_stream_kernel $O/prof_${TAG}_fwd_fetch.txt $O/prof_${TAG}_fwd_write.txt $O/${TAG}_pmc_traffic.json >> $O/${TAG}_pmc_traffic.log 2>&1
cp $O/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json     # bench.py reads the record from profiles/
(timeout 900 python bench.py 2>&1 | tail -1) > $O/${TAG}_bench_default.json
for w in ${WORKLOADS:-scan_bwd_pretrain scan_fwd_target scan_fwd_cfg2 scan_fwd_target_bf16 decode_llama7b_128 mae_vit_large_1280 arm_encoder_large_224 vmamba_base_224 arm_pretrain_base_192}; do
  (timeout 600 python bench.py --workload $w 2>&1 | tail -1) > $O/${TAG}_bench_$w.json
done
(timeout 300 python tools/scan_r03_bench.py all 2 2>&1 | grep -v amdgpu.ids) > $O/${TAG}_scan_variants.txt
(timeout 300 python tools/gemm_swiglu_bench.py 3 2>&1 | grep -v amdgpu.ids) > $O/${TAG}_gemm_swiglu_bench.txt
(timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids) > $O/${TAG}_attn_bench.txt
(timeout 300 python tools/dir_perm_bench.py 2 2>&1 | grep -v amdgpu.ids) > $O/${TAG}_dir_perm_bench.txt
cat $O/${TAG}_pytest_gpu_tail.log $O/${TAG}_smoke.log $O/${TAG}_pmc_traffic.log; for f in $O/${TAG}_bench_*.json; do echo "== $f"; cut -c1-330 $f; done
