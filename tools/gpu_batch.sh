#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for b in 16 24 32; do
  (timeout 600 python bench.py --batch $b --steps 6 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b', round(d['value'],2),'img/s', round(d['ms_per_step'],1),'ms', 'mem GB', round(__import__('torch').cuda.max_memory_allocated()/2**30,1) if False else '')") 2>&1 | tail -1
done | tee gpurun_out/batch_sweep.txt
