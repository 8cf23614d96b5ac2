"""Soak forensic for the rare deviation in the V half of d_wkv: one merge_fwd, then many merge_bwd on the SAME workspace.
On a deviating pass: which rows / columns, and which operand of  d_wkv[I + d, e] = sum_i dO[i, d] Y[h k + i, e]  explains the wrong values
(a wrong dO[i, d] seen by some lanes gives delta[e] proportional to Y[h k + i, e]; a wrong Y gives delta proportional to nothing per row).
Run next to a second process that loads the GPU (tools/soak_forensic.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mhim_mil_amd import synth, ops

R, k, runs = int(sys.argv[1]), 5, int(sys.argv[2])
out_dir = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out"
E = I = 512
sd = synth.mhim_state(7, input_dim=64, merge_k=k)
f32 = lambda a: torch.as_tensor(a).float().contiguous().cuda()
p = {kk: f32(v) for kk, v in sd.items() if kk.startswith("merge.")}
tr = (ops.transpose(p["merge.attn.to_kv.weight"]), ops.transpose(p["merge.attn.to_q.weight"]), ops.transpose(p["merge.attn.to_out.0.weight"]))
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.randn(R, E, device="cuda", generator=g).abs() * 0.7
dz = torch.randn(k, E, device="cuda", generator=g) * 0.1
mw = ops.MergeW(p["merge.global_q_mm"].reshape(k, E).clone(), p["merge.norm.weight"], p["merge.norm.bias"], p["merge.attn.to_kv.weight"],
                p["merge.attn.to_q.weight"], p["merge.attn.to_out.0.weight"], p["merge.attn.to_out.0.bias"], 0.9999, prec="bf16x3", transposes=tr)
zd, qn, ws = ops.merge_fwd(mw, X)


def layout(R, k):
    T = -(-R // 32); JP, Ed = 48, 512
    names = [("gq", k * Ed), ("gmean", k), ("grstd", k), ("Q", k * I), ("aq/U", JP * Ed), ("aqf", 3 * 16 * 64 * 8), ("gtf_aq", 32 * 2 * 64 * 8), ("mean", R),
             ("rstd", R), ("S", R * JP), ("pm", T * JP), ("pl", T * JP), ("psd", T * JP), ("ypart", T * JP * Ed), ("stats", JP * 2), ("Y", JP * Ed),
             ("O", k * I), ("dO", k * I), ("dyf", 3 * 16 * 64 * 8), ("gtf_dy", 32 * 2 * 64 * 8), ("dpart", JP * 8), ("upart", T * JP * Ed),
             ("lnpart", T * 2 * Ed), ("dQ", k * I)]
    off, out = 0, {}
    for n, cnt in names:
        out[n] = (off, off + cnt); off += -(-cnt * 4 // 256) * 64
    return out


lay = layout(R, k)
grads = {"d_wkv": torch.empty(2 * I, E, device="cuda")}


def bwd():
    gr = ops.merge_bwd(mw, X, dz, ws, splits=4, grads=dict(grads))
    torch.cuda.synchronize()
    return gr["d_wkv"].cpu().numpy().copy()


ref = bwd()
wsf = ws.view(torch.float32).cpu().numpy() if ws.dtype == torch.float32 else ws.view(torch.uint8).cpu().numpy().view(np.float32)
Y = wsf[lay["Y"][0]:lay["Y"][1]].reshape(48, E)
dO = wsf[lay["dO"][0]:lay["dO"][1]].reshape(k, I)
Q = wsf[lay["Q"][0]:lay["Q"][1]].reshape(k, I)
# the host restatement of the V half (fp32 arithmetic in the kernel's order is not needed: we only explain DIFFERENCES)
events = []
for it in range(runs):
    o = bwd()
    if np.array_equal(o, ref):
        continue
    rr, cc = np.nonzero(o != ref)
    ev = {"pass": it, "rows": sorted(set(rr.tolist())), "cols": (int(cc.min()), int(cc.max()), int(len(cc)))}
    for r in ev["rows"]:
        cols = cc[rr == r]
        delta = (o[r, cols].astype(np.float64) - ref[r, cols].astype(np.float64))
        half = "V" if r >= I else "K"
        d = r - I if r >= I else r
        h = d // 64
        expl = {}
        if half == "V":
            for i in range(k):
                y = Y[h * k + i, cols].astype(np.float64)
                c = float((delta * y).sum() / max((y * y).sum(), 1e-30))
                res = float(np.abs(delta - c * y).max())
                expl[f"dO[{i}]"] = (c, res, float(dO[i, d]))
        ev[f"row{r}"] = {"half": half, "cols": cols.tolist()[:40], "delta_max": float(np.abs(delta).max()), "explain(c,resid,dO)": expl,
                         "dO_col": dO[:, d].tolist(), "Q_col": (Q[:, d] * 1.0).tolist(),
                         "dO_neighbours": {dd: dO[:, dd].tolist() for dd in range(max(0, d - 2), min(I, d + 3))}}
    events.append(ev)
    brief = []
    for r in ev["rows"]:
        e = ev[f"row{r}"]
        best = min(e["explain(c,resid,dO)"].items(), key=lambda kv: kv[1][1]) if e["explain(c,resid,dO)"] else None
        brief.append((r, e["half"], e["cols"][0], e["cols"][-1], e["delta_max"], best))
    print("EVENT", it, brief, flush=True)
    if len(events) <= 2:
        np.savez(os.path.join(out_dir, f"merge_forensic_{it}.npz"), bad_rows=o[ev["rows"]], ref_rows=ref[ev["rows"]], rows=np.array(ev["rows"]), Y=Y, dO=dO, Q=Q)
print("R", R, "passes", runs, "events", len(events))
