cd /tmp && export TMPDIR=/tmp
for w in ${WORKLOADS:-vmamba_base_224 arm_encoder_large_224 arm_pretrain_base_192}; do
  rm -rf /tmp/hb_$w
  timeout 600 rocprofv3 --kernel-trace -d /tmp/hb_$w -o r -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline ${EXTRA:-} > /tmp/hb_$w.log 2>&1
  python - <<PY
import sqlite3, glob, json
db = glob.glob('/tmp/hb_$w/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select start, end from kernels order by start").fetchall()
# the last 10 steps: take kernels in the last (10/13) of the trace by time is fuzzy; use totals instead
line = [l for l in open('/tmp/hb_$w.log') if l.startswith('{')][-1]
d = json.loads(line)
win = d['ms_per_step'] * 10 * 1e6
t1 = rows[-1][1]
sel = [(s, e) for s, e in rows if s >= t1 - win]
tot = sum(e - s for s, e in sel)
print("$w: %.2f ms/step under the tracer; last 10 steps: %d kernels per step, GPU busy %.2f ms per step = %.2f of the step" % (d['ms_per_step'], len(sel) // 10, tot / 1e7, tot / win))
PY
done
