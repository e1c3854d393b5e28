"""Timing of the 4-channel ends of the UNet (conv_in, conv_out, conv_out dgrad) at the bench shape [8][128][128][320].
PEA_CONV_OUT_DIRECT=1 in the environment selects the one-wave-per-pixel conv_out for an A/B."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pea_diffusion_amd import ops

def timeit(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for (B, H, C, Cout) in [(8, 128, 320, 4), (4, 128, 320, 4), (1, 1024, 128, 3)]:
    g = torch.Generator().manual_seed(0)
    h = torch.randn(B, H, H, C, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(Cout, C, 3, 3, generator=g) * 0.02).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    wp = ops.pack_conv_out(w)
    t = timeit(lambda: ops.conv_out(h, wp, b))
    print(f"conv_out [{B}][{H}][{H}][{C}] -> {Cout}: {t:.1f} us  ({h.numel() * 2 / t / 1e6:.2f} TB/s of input)")
    if Cout == 4:
        x = torch.randn(B, 4, H, H, generator=g).cuda()
        wi = (torch.randn(C, 4, 3, 3, generator=g) * 0.2).cuda()
        bi = torch.randn(C, generator=g).cuda()
        t = timeit(lambda: ops.conv_in(x, wi, bi))
        print(f"conv_in  [{B}][4][{H}][{H}] -> {C}: {t:.1f} us")
        dy = torch.randn(B, 4, H, H, generator=g).cuda()
        t = timeit(lambda: ops.conv_out_dgrad(dy, wp, C))
        print(f"conv_out dgrad [{B}][4][{H}][{H}] -> {C}: {t:.1f} us")
