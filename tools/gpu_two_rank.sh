#!/bin/bash
# Dev check of the multi-rank bench path on a 1-GPU box: `python bench.py --gpus 2` launches its own 2 ranks (torch.distributed.run,
# 127.0.0.1); MXVL_BENCH_ONE_GPU=1 puts both on cuda:0 with gloo collectives.  Not a scaling number -- the two ranks share one GPU.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
(MXVL_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 --no-secondary --no-cpu-baseline 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -4) > gpurun_out/two_rank_one_gpu.log
cut -c1-900 gpurun_out/two_rank_one_gpu.log
