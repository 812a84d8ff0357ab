"""Dev tool (round 3): static audit of the compiled kernels for waits that defeat a prefetch.  No GPU needed.

    python tools/isa_wait_audit.py [file.hip ...]            # default: every csrc/*.hip; compiles with -save-temps into /tmp/mxvl_isa_audit
    python tools/isa_wait_audit.py --trace scan_bwd.hip <kernel-name-substring>     # loads / waits / barriers / MFMA blocks in order

hipcc (SIInsertWaitcnts) places `s_waitcnt vmcnt(N)` statically.  Three source patterns made it wait for a load right where the
load was issued, on every kernel of this repo that tried to keep the next tile in flight:
  (1) a load inside a per-lane `if` whose other side defines the same registers (`x = ok ? load : 0`): waited for at the join;
  (2) a run-time-uniform branch that loads into registers the common path uses right after the join (the fp32-dout branch of
      scan_bwd): the wait sits at that use whichever way the branch goes;
  (3) loads issued before a loop and first used inside it: if ANY path reaches the loop header with them pending, the wait sits
      in front of their first use in the loop BODY and runs every iteration -- behind the next tile's requests.
The audit reports, per kernel: vector-memory loads, loads followed within two instructions by `s_waitcnt vmcnt(0)`, and loop
bodies whose first MFMA is preceded by a `vmcnt(0)` with loads issued earlier in the same body.
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "medical_image_analysis_amd", "csrc")
TMP = os.environ.get("MXVL_ISA_TMP", "/tmp/mxvl_isa_audit")     # outside the repository: -save-temps dumps are 40+ MB and every gpurun pushes the tree


def compile_s(src):
    os.makedirs(TMP, exist_ok=True)
    base = os.path.basename(src)[:-4]
    out = os.path.join(TMP, f"{base}-hip-amdgcn-amd-amdhsa-gfx950.s")
    import glob
    newest = max(os.path.getmtime(f) for f in [src] + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    if not os.path.exists(out) or os.path.getmtime(out) < newest:       # (the kernels live in headers as often as in the .hip file)
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-munsafe-fp-atomics",
                               "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-save-temps=obj", "-c", src,
                               "-o", os.path.join(TMP, base + ".o")], cwd=TMP, stderr=subprocess.DEVNULL)
    return out


def kernels(path):
    """yield (name, [instruction strings with labels kept as 'LABEL:'])"""
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+|mxvl\w*):\s*(;.*)?$", lines[i])
        if m and ".amdhsa_kernel" not in lines[i]:
            name, body = m.group(1), []
            i += 1
            while i < len(lines) and not lines[i].startswith(".Lfunc_end"):
                t = lines[i].strip()
                if re.match(r"^\.LBB\d+_\d+:", lines[i]):
                    body.append(lines[i].split()[0])
                elif t and not t.startswith(";") and not t.startswith("."):
                    body.append(t)
                i += 1
            if any("s_endpgm" in b for b in body):
                yield name, body
        i += 1


def demangle(n):
    try:
        return subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], text=True).strip()
    except Exception:
        return n


def audit(path):
    rows = []
    for name, body in kernels(path):
        ins = [b for b in body if not b.endswith(":")]
        loads = sum(1 for b in ins if re.match(r"(global|flat|buffer)_load", b) and "_lds_" not in b)
        dma = sum(1 for b in ins if "global_load_lds" in b)
        imm = 0
        for k, b in enumerate(ins):
            if re.match(r"(global|flat|buffer)_load", b) and any("s_waitcnt vmcnt(0)" in x for x in ins[k + 1:k + 3]):
                imm += 1
        # loop bodies: label L ... s_cbranch* L ; first MFMA preceded by vmcnt(0) with a load earlier in the body
        pos = {b[:-1]: k for k, b in enumerate(body) if b.endswith(":")}
        blocked = 0
        for k, b in enumerate(body):
            m = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)", b)
            if m and m.group(1) in pos and pos[m.group(1)] < k:
                seg = body[pos[m.group(1)]:k]
                fm = next((q for q, x in enumerate(seg) if x.startswith("v_mfma")), None)
                if fm is None:
                    continue
                w = [q for q, x in enumerate(seg[:fm]) if "vmcnt(0)" in x]
                ld = [q for q, x in enumerate(seg[:fm]) if re.match(r"(global|flat|buffer)_load", x)]
                if w and ld and min(ld) < max(w):
                    blocked += 1
        rows.append((demangle(name), loads, dma, imm, blocked))
    return rows


def trace(path, sub):
    for name, body in kernels(path):
        d = demangle(name)
        if sub not in d and sub not in name:
            continue
        print("==", d)
        n, mf = 0, 0
        for b in body:
            if b.endswith(":"):
                continue
            n += 1
            if b.startswith("v_mfma"):
                mf += 1
                continue
            if re.search(r"global_load|global_store|global_atomic|vmcnt|s_barrier|scratch_|buffer_", b):
                if mf:
                    print(f"          ... {mf} mfma")
                    mf = 0
                print(f"{n:7d}  {b[:110]}")
        if mf:
            print(f"          ... {mf} mfma")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--trace":
        trace(compile_s(os.path.join(CSRC, sys.argv[2])), sys.argv[3])
        sys.exit(0)
    files = [os.path.join(CSRC, f) for f in sys.argv[1:]] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    print(f"{'kernel':100s} loads  dma  load+wait0  loops blocked before the first MFMA")
    for f in files:
        for d, loads, dma, imm, blocked in audit(compile_s(f)):
            if loads + dma == 0:
                continue
            flag = "  <--" if blocked or (imm >= 4 and imm * 2 >= loads) else ""
            print(f"{d[:100]:100s} {loads:5d} {dma:4d} {imm:10d}  {blocked}{flag}")
