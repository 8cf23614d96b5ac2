"""Helpers to read tests/golden/*.npz (written by oracle/gen_golden.py from the reference import)."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return meta, {k: z[k] for k in z.files if k != "meta"}


def names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def tagged(arrs, tag):
    """Keys 'tag:name|kind' -> {name: {kind: array}}."""
    out = {}
    for k, v in arrs.items():
        if not k.startswith(tag + ":"):
            continue
        name, kind = k[len(tag) + 1:].rsplit("|", 1)
        out.setdefault(name, {})[kind] = v
    return out


def check_compact(got, exp, rtol, atol, what=""):
    """Compare a tensor with its compact fixture form (full, or norm/sum/strided sample)."""
    g = np.asarray(got, dtype=np.float64)
    if "full" in exp:
        np.testing.assert_allclose(g, exp["full"].astype(np.float64).reshape(g.shape), rtol=rtol, atol=atol,
                                   err_msg=what)
        return
    flat = g.reshape(-1)
    stride = int(exp["stride"])
    samp = flat[::stride][:exp["sample"].shape[0]]
    scale = float(exp["norm"]) / np.sqrt(flat.size) + 1e-30
    np.testing.assert_allclose(samp, exp["sample"].astype(np.float64), rtol=rtol, atol=atol + rtol * scale,
                               err_msg=what + " (sample)")
    assert abs(np.linalg.norm(flat) - float(exp["norm"])) <= rtol * float(exp["norm"]) + atol, what + " (norm)"
