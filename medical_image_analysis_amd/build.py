"""Build libmxvl.so (all HIP kernels + the C-ABI) in-tree with hipcc for gfx950.

    python -m medical_image_analysis_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the gpurun snapshot.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmxvl.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


ABLATE_LIB = os.path.join(PKG, "build", "libmxvl_ablate.so")


def build(force: bool = False, verbose: bool = False, ablate: bool = False) -> str:
    """ablate=True: a SEPARATE measurement library (build/libmxvl_ablate.so, -DMXVL_ABLATE) whose kernels honour the
    MXVL_*_ABLATE / MXVL_BWD_WAVES environment switches; the product library has none of that code (mxvl_common.h)."""
    if ablate:
        return _build_ablate(verbose)
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(PKG, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(PKG, "build", os.path.basename(src).replace(".hip", ".o"))
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(
                [os.path.getmtime(src)] + [os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h"))]
                + [os.path.getmtime(h) for h in glob.glob(os.path.join(ROOT, "include", "*.h"))]):
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-munsafe-fp-atomics",
               "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_exp(exp: int, files=("scan_fwd.hip", "scan_bwd.hip"), verbose: bool = False) -> str:
    """A/B measurement build: build/libmxvl_exp<exp>.so = the product objects, with `files` recompiled under -DMXVL_EXP=<exp>
    (csrc experiments are `#if MXVL_EXP & bit` blocks that exist only while an experiment is open).  tools/*_bench.py load it
    beside libmxvl.so in ONE process so both arms run on the same box, interleaved."""
    build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out_dir = os.path.join(PKG, "build", f"exp{exp}")
    os.makedirs(out_dir, exist_ok=True)
    objs, procs = [], []
    for src in sources():
        base = os.path.basename(src)
        if base in files:
            obj = os.path.join(out_dir, base.replace(".hip", ".o"))
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-munsafe-fp-atomics",
                   f"-DMXVL_EXP={exp}", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
        else:
            obj = os.path.join(PKG, "build", base.replace(".hip", ".o"))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    lib = os.path.join(PKG, "build", f"libmxvl_exp{exp}.so")
    subprocess.check_call([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def _build_ablate(verbose: bool) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.join(PKG, "build"), exist_ok=True)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-munsafe-fp-atomics",
           "-DMXVL_ABLATE", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-shared", "-o", ABLATE_LIB] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return ABLATE_LIB


if __name__ == "__main__":
    if "--exp" in sys.argv:
        print(build_exp(int(sys.argv[sys.argv.index("--exp") + 1]), verbose="--verbose" in sys.argv))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, ablate="--ablate" in sys.argv))
