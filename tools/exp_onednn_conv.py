"""Host-library check (CPU only): torch-CPU's oneDNN fp32 weight gradient of the depth-wise 33-tap convolution
(nystrom_attention.py:65-68 res_conv) at T = 256 tokens disagrees with fp64 and with ATen's native path; T >= 512 agrees.
oracle/gen_golden.py and the oracle tests switch oneDNN off for bags padded to exactly 256 tokens because of this."""
import warnings

import torch

warnings.filterwarnings("ignore")
torch.manual_seed(0)
for T in (256, 512, 768, 1024):
    v, g = torch.randn(1, 8, T, 64), torch.randn(1, 8, T, 64)
    vp = torch.nn.functional.pad(v.double(), (0, 0, 16, 16))
    ref = torch.stack([(g.double() * vp[:, :, t:t + T]).sum((0, 2, 3)) for t in range(33)], 1)
    for mk in (True, False):
        w = torch.randn(8, 1, 33, 1, requires_grad=True)
        with torch.backends.mkldnn.flags(enabled=mk):
            torch.nn.functional.conv2d(v, w, padding=(16, 0), groups=8).backward(g)
        print(f"T={T:5d} oneDNN={mk!s:5s} max |dW - fp64| = {float((w.grad.reshape(8, 33).double() - ref).abs().max()):.3e}"
              f"  (scale {float(ref.abs().max()):.1f})")
