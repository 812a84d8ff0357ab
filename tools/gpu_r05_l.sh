#!/bin/bash
# round 5, call L: why is the decode leg of the default line ~4 % below the standalone decode line?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python - <<'PY'
import json, subprocess, sys, time
def run(args):
    out = subprocess.run([sys.executable, "bench.py"] + args, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    return json.loads(out)
d = run(["--workload", "decode_llama7b_128", "--steps", "5", "--warmup", "1", "--no-cpu-baseline"])
print("standalone            ", round(d["value"], 1))
d = run(["--steps", "6", "--warmup", "2", "--no-cpu-baseline"])
print("after training (as is)", round(d["secondary"]["value"], 1), "| images/s", round(d["value"], 2))
d = run(["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--secondary-idle", "3"])
print("after training + 3 s idle", round(d["secondary"]["value"], 1))
d = run(["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--secondary-warmup", "3"])
print("after training, 3 warm-up generates", round(d["secondary"]["value"], 1))
PY
