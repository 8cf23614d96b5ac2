"""ISA lint of the built library (CPU container and GPU box alike: only llvm-objdump is needed).

A packed fp32 instruction whose op_sel swizzles the ODD register of a source pair into the low half (what the SLP vectorizer emits for
pairs of scalar FMAs) lost its term in lanes 48..63 about once per 500 launches of merge2_grads1 when a second process shared the GPU
(tools/exp_merge_forensic.py).  mhim_mil_amd/build.py therefore builds without SLP outside the files that have no such
instruction; this test disassembles every gfx950 code object of libmhimx.so and fails if one comes back."""
import os
import re
import shutil
import subprocess

import pytest

from mhim_mil_amd import _lib

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PACKED = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")
SWIZZLE = re.compile(r"op_sel:\[[01,]*1[01,]*\]")


def test_no_swizzled_packed_fp32_instruction(tmp_path):
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not installed")
    assert os.path.exists(_lib.LIB_PATH), "libmhimx.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    lib = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    r = subprocess.run([OBJDUMP, "--offloading", str(lib)], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    objs = sorted(p for p in os.listdir(tmp_path) if "gfx950" in p)
    assert len(objs) >= 10, f"expected one gfx950 code object per translation unit, found {objs}"
    bad, packed = [], 0
    for o in objs:
        d = subprocess.run([OBJDUMP, "-d", str(tmp_path / o)], capture_output=True, text=True)
        assert d.returncode == 0, d.stderr
        kernel = "?"
        for line in d.stdout.splitlines():
            if line.endswith(">:"):
                kernel = line.split("<")[-1][:-2]
            elif PACKED.search(line):
                packed += 1
                if SWIZZLE.search(line):
                    bad.append(f"{kernel}: {line.strip()[:120]}")
    assert packed > 0, "the disassembly shows no packed fp32 instruction at all: the lint is not looking at the device code"
    assert not bad, "op_sel-swizzled packed fp32 instructions (build the file without SLP, mhim_mil_amd/build.py):\n" + "\n".join(bad[:20])


ASM_LOAD_SOURCES = ("bag_project.hip", "bag_project_ws.hip", "wgrad.hip", "scorer_fused.hip")      # the files with inline-asm vector loads


def test_asm_loads_keep_their_registers(tmp_path):
    """tools/asm_lint.py on the assembly of every source with inline-asm vector loads: no instruction may read or overwrite the
    destination of such a load before the wait that covers it (the compiler does not know the data is still in flight).  This is how
    bag_wgrad_ws_kernel faulted under GPU sharing: its last wait named 6 of the 24 prefetch registers."""
    import sys
    from concurrent.futures import ThreadPoolExecutor
    from mhim_mil_amd import build
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import asm_lint
    if not os.path.exists(build.HIPCC):
        pytest.skip("hipcc not installed")
    srcs = [s for s in build.sources() if os.path.basename(s) in ASM_LOAD_SOURCES]
    assert len(srcs) == len(ASM_LOAD_SOURCES)
    # (every inline-asm vector load of the library lives in these files)
    for s in build.sources():
        if os.path.basename(s) not in ASM_LOAD_SOURCES:
            assert "global_load_dword" not in open(s).read(), f"{s}: inline-asm vector loads - add the file to ASM_LOAD_SOURCES"

    def cc(src):
        out = tmp_path / (os.path.basename(src)[:-4] + ".s")
        flags = [f for f in build.flags_for(src) if f != "-fPIC"]
        r = subprocess.run([build.HIPCC, *flags, "--cuda-device-only", "-S", src, "-o", str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return str(out)

    with ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        outs = list(ex.map(cc, srcs))
    for o in outs:
        text = open(o).read()
        assert ";;#ASMSTART" in text, f"{o}: no inline asm found - the lint is not looking at the right thing"
        bad = asm_lint.lint(o)
        assert not bad, f"{o}: registers of in-flight asm loads are touched:\n" + "\n".join(f"{b[0]} line {b[1]} (load at {b[2]}): {b[3]}" for b in bad[:10])
