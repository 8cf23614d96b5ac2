"""Per-kernel resource table of a build: VGPRs, SGPRs, scratch (private segment), LDS, kernarg bytes - from the metadata notes of the code
objects in a build directory's .o files.   python tools/kernel_meta.py mhim_mil_amd/build [other/build]  (two: only the kernels that differ)"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"


def meta(bdir):
    txt = ""
    with tempfile.TemporaryDirectory() as td:
        for f in sorted(os.listdir(bdir)):
            if not f.endswith(".o"):
                continue
            fb, co = os.path.join(td, "fb"), os.path.join(td, "co")
            subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", os.path.join(bdir, f), fb], check=True)
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fb}", f"--output={co}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True)
            if r.returncode == 0:
                txt += subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out = {}
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        out[name] = dict(vgpr=g("vgpr_count"), agpr=blk.split()[0], sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"),
                         lds=g("group_segment_fixed_size"), kernarg=g("kernarg_segment_size"))
    return out


a = meta(sys.argv[1])
b = meta(sys.argv[2]) if len(sys.argv) > 2 else None
for k in sorted(a):
    if b is None:
        print(k[:70], a[k])
    elif k in b and {x: a[k][x] for x in ("vgpr", "agpr", "scratch", "lds")} != {x: b[k][x] for x in ("vgpr", "agpr", "scratch", "lds")}:
        print(k[:70], "\n   A", a[k], "\n   B", b[k])
