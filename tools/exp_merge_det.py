"""merge_fwd + merge_bwd (projection-free form) many times on the same inputs: are the gradients bit-reproducible (also under contention)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mhim_mil_amd import synth, ops
R, k, runs = int(sys.argv[1]), 5, int(sys.argv[2])
E = 512
sd = synth.mhim_state(7, input_dim=64, merge_k=k)
f32 = lambda a: torch.as_tensor(a).float().contiguous().cuda()
p = {kk: f32(v) for kk, v in sd.items() if kk.startswith("merge.")}
tr = (ops.transpose(p["merge.attn.to_kv.weight"]), ops.transpose(p["merge.attn.to_q.weight"]), ops.transpose(p["merge.attn.to_out.0.weight"]))
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.randn(R, E, device="cuda", generator=g).abs() * 0.7
dz = torch.randn(k, E, device="cuda", generator=g) * 0.1
def run():
    mw = ops.MergeW(p["merge.global_q_mm"].reshape(k, E).clone(), p["merge.norm.weight"], p["merge.norm.bias"], p["merge.attn.to_kv.weight"],
                    p["merge.attn.to_q.weight"], p["merge.attn.to_out.0.weight"], p["merge.attn.to_out.0.bias"], 0.9999, prec="bf16x3", transposes=tr)
    zd, qn, ws = ops.merge_fwd(mw, X)
    gr = ops.merge_bwd(mw, X, dz, ws, splits=4)
    torch.cuda.synchronize()
    out = {kk: v.cpu().numpy().copy() for kk, v in gr.items() if torch.is_tensor(v)}
    out["~ws"] = ws.view(torch.uint8).cpu().numpy().copy().view(np.float32) if ws.dtype != torch.float32 else ws.cpu().numpy().copy()
    return out
ref = run()
bad = {}
for i in range(runs):
    o = run()
    for kk in ref:
        if not np.array_equal(o[kk], ref[kk]):
            rr = np.nonzero((o[kk] != ref[kk]).reshape(ref[kk].shape[0], -1).any(1))[0] if o[kk].ndim > 1 else np.zeros(0, dtype=int)
            bad.setdefault(kk, []).append((i, float(np.nanmax(np.abs(o[kk] - ref[kk]))), rr[:6].tolist()))
            if kk == "~ws":
                idx = np.nonzero(o[kk].reshape(-1) != ref[kk].reshape(-1))[0]
                regs = {}
                for n_, a_, b_ in layout(R, k):
                    c_ = int(((idx >= a_) & (idx < b_)).sum())
                    if c_:
                        regs[n_] = (c_, int(idx[(idx >= a_) & (idx < b_)][0] - a_))
                print("run", i, "workspace regions that differ (count, first offset):", regs)
# workspace regions (mca2_prep.hpp merge2_ws_layout; 256-byte granules)
def layout(R, k):
    T = -(-R // 32); JP, Ed, I = 48, 512, 512
    names = [("gq", k * Ed), ("gmean", k), ("grstd", k), ("Q", k * I), ("aq/U", JP * Ed), ("aqf", 3 * 16 * 64 * 8), ("gtf_aq", 32 * 2 * 64 * 8), ("mean", R),
             ("rstd", R), ("S", R * JP), ("pm", T * JP), ("pl", T * JP), ("psd", T * JP), ("ypart", T * JP * Ed), ("stats", JP * 2), ("Y", JP * Ed),
             ("O", k * I), ("dO", k * I), ("dyf", 3 * 16 * 64 * 8), ("gtf_dy", 32 * 2 * 64 * 8), ("dpart", JP * 8), ("upart", T * JP * Ed),
             ("lnpart", T * 2 * Ed), ("dQ", k * I)]
    off, out = 0, []
    for n, cnt in names:
        out.append((n, off, off + cnt)); off += -(-cnt * 4 // 256) * 64
    return out
if "~ws" in bad:
    for (i, _, _) in bad["~ws"][:2]:
        pass
lay = layout(R, k)
for i in range(0):
    pass
print("R", R, "differing:", {k_: v[:3] for k_, v in bad.items()} if bad else "none", "in", runs, "runs; keys", list(ref)[:12])
