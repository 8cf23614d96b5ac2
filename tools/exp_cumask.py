"""Probe: does partitioning the chip with CU-masked streams let several bags' steps truly overlap?
S host threads, each with its own trainer (own models and flat buffers) and its own HIP stream created with
hipExtStreamCreateWithCUMask over a disjoint slice of the 256 CUs, each running complete eager train steps through mhimx_step_run
(one C call per bag: the GIL is released inside ctypes, so the threads enqueue in parallel).  Reports aggregate ms per bag."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import synth
from mhim_mil_amd.mhim import MHIM
from mhim_mil_amd.engine import FusedTrainer

N, D = 10000, 1024
CFG = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True, merge_k=5,
           merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.25)
dev = torch.device("cuda", 0)
base = synth.mhim_state(7, input_dim=D, merge_k=5)
hip = None
for name in ("libamdhip64.so", os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")):
    try:
        hip = C.CDLL(name); break
    except OSError:
        pass


def mk():
    m = MHIM(input_dim=D, n_classes=2, baseline="attn", **CFG)
    sd = dict(base); sd["merge.global_q"] = sd["merge.global_q_mm"]
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    return m.to(dev).train()


def masked_stream(lo, hi):
    """A stream restricted to CUs [lo, hi) (bit i of the mask = CU i in the runtime's enumeration)."""
    words = (C.c_uint32 * 8)()
    for cu in range(lo, hi):
        words[cu // 32] |= 1 << (cu % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


def run(S, masked, steps=60):
    g = torch.Generator(device=dev); g.manual_seed(5)
    trainers = [FusedTrainer(mk(), mk(), aux_alpha=0.5) for _ in range(S)]
    bags = [[torch.randn(N, D, device=dev, generator=g).abs_() for _ in range(2)] for _ in range(S)]
    label = torch.tensor([1], device=dev)
    per = 256 // S
    streams = [masked_stream(j * per, (j + 1) * per) if masked else torch.cuda.Stream() for j in range(S)]
    torch.cuda.synchronize()
    bar = threading.Barrier(S + 1)

    def work(j):
        with torch.cuda.stream(streams[j]):
            for i in range(6):
                trainers[j].train_step(bags[j][i % 2], label)
            streams[j].synchronize()
            bar.wait()
            for i in range(steps):
                trainers[j].train_step(bags[j][i % 2], label)
            streams[j].synchronize()

    th = [threading.Thread(target=work, args=(j,)) for j in range(S)]
    for t in th: t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"S={S} streams ({'CU-masked, %d CUs each' % per if masked else 'plain'}): {dt / (S * steps) * 1e3:.4f} ms per bag aggregate", flush=True)


for S, masked in ((1, False), (2, False), (4, False), (2, True), (4, True), (8, True)):
    try:
        run(S, masked)
    except Exception as e:  # noqa
        print(f"S={S} masked={masked}: {type(e).__name__}: {str(e)[:200]}")
