#!/bin/bash
# Dev helper run ON the GPU box via gpurun: the two HBM-traffic PMC passes of the scan roofline kernel (separate --pmc runs, as
# MI355X_MICROARCH.md prescribes), the per-launch traffic record, then the scan bench line that reports it.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r01}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_$TAG; mkdir -p $P
for c in FETCH_SIZE WRITE_SIZE; do
  n=scan_$(echo $c | tr A-Z a-z | sed 's/_size//')
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $P/$n -o r -- python $R/bench.py --workload scan_fwd_target --steps 10 --warmup 2 --no-cpu-baseline > $P/$n.log 2>&1
  (cd $R && python tools/rocpd_summary.py $P/$n/r_results.db 2>&1 | head -45 | cut -c1-170 > $O/prof_${TAG}_$n.txt)
done
cd $R
python tools/pmc_traffic.py scan_fwd_target scan_fwd_stream_kernel $O/prof_${TAG}_scan_fetch.txt $O/prof_${TAG}_scan_write.txt $O/${TAG}_pmc_traffic.json
cp $O/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
(timeout 300 python bench.py --workload scan_fwd_target 2>&1 | tail -1) > $O/bench_scan_fwd_target.json
cut -c1-200 $O/prof_${TAG}_scan_fetch.txt | tail -3; cut -c1-200 $O/prof_${TAG}_scan_write.txt | tail -3; cat $O/bench_scan_fwd_target.json
