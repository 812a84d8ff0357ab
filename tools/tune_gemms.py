"""Re-tune the library GEMM solutions (hipBLASLt / rocBLAS) the pre-training step uses, via PyTorch TunableOp.
Run ON an MI355X:  [MXVL_TUNE_BUDGET_S=150] python tools/tune_gemms.py [workload ...]   -> medical_image_analysis_amd/tuned/tunableop_gfx950.csv
The product only READS that file (pretrain_engine.enable_tuned_gemms); shapes that are not in it use the library default."""
import os, subprocess, sys, glob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "medical_image_analysis_amd", "tuned", "tunableop_gfx950.csv")
argv = sys.argv[1:]
batch = []
if "--batch" in argv:                      # per-GPU batch override (GEMM shapes follow the token count)
    i = argv.index("--batch")
    batch = ["--batch", argv[i + 1]]
    del argv[i:i + 2]
workloads = argv or ["arm_pretrain_large_1024", "arm_pretrain_base_192"]
tmp = "/tmp/mxvl_tune"
os.makedirs(tmp, exist_ok=True)
lines, header = {}, []
if os.path.exists(OUT):
    for ln in open(OUT):
        (header if ln.startswith("Validator") else lines.setdefault(",".join(ln.split(",")[:2]), ln) and [])
for w in workloads:
    base = os.path.join(tmp, f"tune_{w}.csv")
    for f in glob.glob(base.replace(".csv", "*.csv")):
        os.remove(f)
    env = dict(os.environ, PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_FILENAME=base,
               PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=os.environ.get("MXVL_TUNE_MS", "15"), PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=os.environ.get("MXVL_TUNE_ITERS", "10"), MXVL_TUNED_GEMMS="0")
    # MXVL_TUNE_BUDGET_S bounds one workload's session: at the deadline the child gets SIGINT, so the interpreter exits normally and
    # TunableOp still writes the shapes it has finished (a `timeout` SIGTERM around this script loses the whole session's results)
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--steps", "2", "--warmup", "1",
                              "--no-cpu-baseline"] + batch, env=env, cwd=ROOT)
    try:
        rc = child.wait(timeout=float(os.environ.get("MXVL_TUNE_BUDGET_S", "1e9")))
        if rc != 0:
            raise subprocess.CalledProcessError(rc, child.args)
    except subprocess.TimeoutExpired:
        import signal
        child.send_signal(signal.SIGINT)
        try:
            child.wait(timeout=60)
        except subprocess.TimeoutExpired:
            child.kill()
        print(f"{w}: tuning budget reached; keeping the shapes finished so far")
    for f in glob.glob(base.replace(".csv", "*.csv")):
        for ln in open(f):
            if ln.startswith("Validator"):
                if ln not in header:
                    header.append(ln)
            else:
                lines[",".join(ln.split(",")[:2])] = ln
with open(OUT, "w") as f:
    f.writelines(header)
    f.writelines(lines[k] for k in sorted(lines))
print(f"wrote {OUT}: {len(lines)} tuned GEMM shapes")
