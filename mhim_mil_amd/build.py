"""Build libmhimx.so (HIP, gfx950 only) in-tree with hipcc.

    python -m mhim_mil_amd.build        # or __graft_entry__.build()

The library is built next to this file so it travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).  hipcc cross-compiles
for gfx950 without a GPU present.
"""
from __future__ import annotations

import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libmhimx.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]
FLAGS += os.environ.get("MHIMX_EXTRA_FLAGS", "").split()          # experiments only (e.g. -DMHIMX_DBG_NOCOMPUTE)
# The SLP vectorizer pairs scalar fp32 FMAs into v_pk_fma_f32 and, where the two lanes want the ODD register of a pair, sets op_sel
# to swizzle it into the low half.  That form dropped its term in lanes 48..63 about once per 500 launches of merge2_grads1 when a second
# process shared the GPU (tools/exp_merge_forensic.py, DESIGN section 5: 38 events in 30 000 passes, 0 in 60 000 without it), so every
# file is built without SLP except the ones listed here, whose code has no such instruction and whose softmax VALU work profits from the
# packed forms (c3: 2 %).  tests/test_isa_lint_cpu.py disassembles the library and fails on any op_sel-swizzled packed fp32 instruction.
SLP_OK = {"nys_flash.hip", "nys_flash_tok.hip"}


def flags_for(src):
    return FLAGS if os.path.basename(src) in SLP_OK or "-fno-slp-vectorize" in FLAGS else FLAGS + ["-fno-slp-vectorize"]
if os.environ.get("MHIMX_LIB_NAME"):
    LIB = os.path.join(HERE, os.environ["MHIMX_LIB_NAME"])
    OBJ = os.path.join(HERE, "build_" + os.environ["MHIMX_LIB_NAME"].replace(".", "_"))


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, lib_name: str | None = None, extra_flags=()) -> str:
    """lib_name / extra_flags: a side build (its own object directory build_<name>/), e.g. the stamped profile library
    ``build(lib_name="libmhimx_prof.so", extra_flags=["-DPW_PROF=2", "-DWG_PROF"])`` that tools/measure_clock.sh and bench.py's
    ``roofline.sustained_clock_GHz`` read the in-kernel shader clock from."""
    LIB, OBJ = globals()["LIB"], globals()["OBJ"]
    if lib_name:
        LIB = os.path.join(HERE, lib_name)
        OBJ = os.path.join(HERE, "build_" + lib_name.replace(".", "_"))
    extra_flags = list(extra_flags)
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "mhimx.h"))
    jobs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        t0 = time.time()
        r = subprocess.run([HIPCC, *flags_for(src), *extra_flags, "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose:
            print(f"[mhimx build] {os.path.basename(src)} ({time.time() - t0:.1f}s)", file=sys.stderr)
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in sources()]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[mhimx build] linked {LIB}", file=sys.stderr)
    return LIB


PROF_LIB, PROF_FLAGS = "libmhimx_prof.so", ["-DPW_PROF=2", "-DWG_PROF", "-DMHIMX_FT_PROF"]


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--prof" in sys.argv:
        build(force="--force" in sys.argv, lib_name=PROF_LIB, extra_flags=PROF_FLAGS)
