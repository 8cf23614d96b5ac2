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
