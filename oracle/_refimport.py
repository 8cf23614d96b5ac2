"""Import the reference's hot-path modules — BUILD CONTAINER ONLY.

/root/reference does not exist on the GPU box and never travels there.  This
helper is used by oracle/check_against_reference.py and oracle/gen_golden.py to
(1) validate the restatement in mhim_oracle.py and (2) produce the committed
fixtures under tests/golden/.  Nothing under tests/ -m gpu, smoke() or bench.py
may call it.

The reference package __init__ pulls in torchvision (modules/__init__.py:4 ->
modules/abmil.py:4), which is not installed; a bare package stub for ``modules``
side-steps the __init__ and empty ``torchvision`` modules satisfy the import
(recipe: SURVEY.md Appendix C).
"""
from __future__ import annotations

import os
import sys
import types

REF_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "modules"))


def load():
    """Returns a namespace with the reference classes/functions on the hot path."""
    if not available():
        raise RuntimeError("reference tree not present (this helper only works in the build container)")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if "modules" not in sys.modules or not hasattr(sys.modules["modules"], "__path__"):
        pkg = types.ModuleType("modules")
        pkg.__path__ = [os.path.join(REF_ROOT, "modules")]
        sys.modules["modules"] = pkg
    for name in ("torchvision", "torchvision.models"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]

    ns = types.SimpleNamespace()
    from modules.mhim import MHIM
    from modules.mhim_modules.masking import select_mask_fn, mask_fn
    from modules.mhim_modules.scoring import get_pseudo_score, get_pseudo_score_trans
    from modules.mhim_modules import baseline as ref_baseline
    from modules.mhim_modules.merge import Merge, MCA
    from modules.mhim_modules.losses import SoftTargetCrossEntropy
    from modules.nystrom_attention import NystromAttention, moore_penrose_iter_pinv
    from modules.emb_position import PPEG
    ns.MHIM, ns.select_mask_fn, ns.mask_fn = MHIM, select_mask_fn, mask_fn
    ns.get_pseudo_score, ns.get_pseudo_score_trans = get_pseudo_score, get_pseudo_score_trans
    ns.baseline, ns.Merge, ns.MCA = ref_baseline, Merge, MCA
    ns.SoftTargetCrossEntropy = SoftTargetCrossEntropy
    ns.NystromAttention, ns.pinv = NystromAttention, moore_penrose_iter_pinv
    ns.PPEG = PPEG
    try:
        from modules import abmil as ref_abmil
        from modules import transmil as ref_transmil
        ns.abmil, ns.transmil = ref_abmil, ref_transmil
    except Exception as e:  # pragma: no cover - informational only
        ns.abmil = ns.transmil = None
        ns.abmil_error = repr(e)
    # engines/common_mil.py has no imports at all: exec it to get CommonMIL.
    g = {}
    with open(os.path.join(REF_ROOT, "engines", "common_mil.py")) as f:
        exec(compile(f.read(), "common_mil.py", "exec"), g)
    ns.CommonMIL = g["CommonMIL"]
    return ns


def build_mhim(ns, sd_numpy: dict, **kw):
    """Construct the reference MHIM and load a numpy state dict (reference key names)."""
    import torch
    m = ns.MHIM(**kw)
    sd_numpy = dict(sd_numpy)
    if "merge.global_q_mm" in sd_numpy:      # the reference registers the same Parameter under two names
        sd_numpy.setdefault("merge.global_q", sd_numpy["merge.global_q_mm"])
    missing, unexpected = m.load_state_dict({k: torch.as_tensor(v) for k, v in sd_numpy.items()}, strict=False)
    assert not unexpected, unexpected
    assert not missing, missing
    return m


def zero_aux_dropouts(m):
    """Parity runs: zero every dropout the constructor does not expose."""
    if hasattr(m.merge, "attn"):
        m.merge.attn.dropout.p = 0.0
        m.merge.attn.to_out[1].p = 0.0
    enc = m.online_encoder
    for l in ("layer1", "layer2"):
        if hasattr(enc, l):
            getattr(enc, l).attn.to_out[1].p = 0.0
    return m
