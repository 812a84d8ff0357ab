"""Build libmxvl.so (all HIP kernels + the C-ABI) in-tree with hipcc for gfx950.

    python -m medical_image_analysis_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the gpurun snapshot; the objects
(build/obj/*.o) do not (.gpurunignore).  Staleness is decided by CONTENT, not by mtime: every object carries the sha256 of its
source + every header + the compile line (build/obj/<name>.o.sha256), the library the sha256 of all of that
(libmxvl.so.sha256) -- a snapshot copy, a checkout or a touched file neither forces nor hides a rebuild.  `build()` reports what
it did in `LAST_BUILD` ("up to date" / the objects it recompiled); MXVL_BUILD_FORCE=1 or --force recompiles everything.
"""
from __future__ import annotations

import glob
import hashlib
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


FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-munsafe-fp-atomics"]
# per-source additions (behind FLAGS: the later flag wins).  llm_ops.hip restates torch expressions whose products and sums are separate
# rounding steps: under -ffp-contract=fast the backend fuses them (v_fma_f16 / v_fma_mixlo_f16) whatever the source says
FILE_FLAGS = {"llm_ops.hip": ["-ffp-contract=off"]}


def flags_for(src: str) -> list:
    return FLAGS + FILE_FLAGS.get(os.path.basename(src), [])
OBJ_DIR = os.path.join(PKG, "build", "obj")
LAST_BUILD = {"state": "not run", "compiled": []}


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")))


def _digest(paths, extra=()) -> str:
    h = hashlib.sha256()
    for e in extra:
        h.update(str(e).encode() + b"\0")
    for path in paths:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def local_includes(src: str) -> list:
    """Transitive `#include "x.h"` closure of one source over csrc/ and include/ (what its object really depends on)."""
    import re
    seen, todo = [], [src]
    while todo:
        path = todo.pop()
        with open(path) as f:
            text = f.read()
        for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, re.M):
            for base in (os.path.dirname(path), CSRC, os.path.join(ROOT, "include")):
                cand = os.path.normpath(os.path.join(base, name))
                if os.path.exists(cand):
                    if cand not in seen:
                        seen.append(cand)
                        todo.append(cand)
                    break
    return sorted(seen)


def source_digest() -> str:
    """sha256 over every .hip, every header and the compile line: the identity of a libmxvl.so build."""
    return _digest(sources() + headers(), FLAGS + [f"{k}:{' '.join(v)}" for k, v in sorted(FILE_FLAGS.items())])


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _stale() -> bool:
    return not os.path.exists(LIB) or _read(LIB + ".sha256") != source_digest()


ABLATE_LIB = os.path.join(PKG, "build", "libmxvl_ablate.so")


def build(force: bool = False, verbose: bool = False, ablate: bool = False) -> str:
    """ablate=True: a SEPARATE measurement library (build/libmxvl_ablate.so, -DMXVL_ABLATE) whose kernels honour the
    MXVL_*_ABLATE / MXVL_BWD_WAVES environment switches; the product library has none of that code (mxvl_common.h)."""
    if ablate:
        return _build_ablate(verbose)
    force = force or os.environ.get("MXVL_BUILD_FORCE") == "1"
    if not force and not _stale():
        LAST_BUILD.update(state="up to date (content hash matches)", compiled=[])
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(OBJ_DIR, exist_ok=True)
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src).replace(".hip", ".o"))
        objs.append(obj)
        want = _digest([src] + local_includes(src), flags_for(src))
        if not force and os.path.exists(obj) and _read(obj + ".sha256") == want:
            continue
        cmd = [hipcc] + flags_for(src) + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, obj, want, subprocess.Popen(cmd)))
    for src, obj, want, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
        with open(obj + ".sha256", "w") as f:
            f.write(want + "\n")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(LIB + ".sha256", "w") as f:
        f.write(source_digest() + "\n")
    LAST_BUILD.update(state="forced rebuild" if force else "rebuilt", compiled=[os.path.basename(t[0]) for t in procs])
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
            cmd = [hipcc] + flags_for(src) + [f"-DMXVL_EXP={exp}", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
        else:
            obj = os.path.join(OBJ_DIR, base.replace(".hip", ".o"))
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
    cmd = [hipcc] + FLAGS + ["-DMXVL_ABLATE", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-shared", "-o", ABLATE_LIB] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return ABLATE_LIB


if __name__ == "__main__":
    if "--exp" in sys.argv:
        print(build_exp(int(sys.argv[sys.argv.index("--exp") + 1]), verbose="--verbose" in sys.argv))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, ablate="--ablate" in sys.argv))
    print(LAST_BUILD["state"], LAST_BUILD["compiled"])
