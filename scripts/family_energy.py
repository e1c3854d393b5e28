"""Energy per unit of work of the step's kernel families, one family at a time: each representative launch runs back to back for
~1.5 s on operands rotated through a few buffers while the card's hwmon power is sampled every 20 ms (the first 0.4 s are
dropped: the SMU's average lags).  Prints W, us per launch, TFLOP/s or GB/s, and J per TFLOP (MFMA families) / per GB (HBM
families).  The step's own figure (bench.py: energy_j_per_tflop) is the mixture of these plus the launch gaps."""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
BF = torch.bfloat16

def power_file():
    """hwmon power file of THIS process's card (a box exposes all its cards: match the PCI address, as bench.GpuSampler does)"""
    import bench
    pci = None
    try:
        pr = torch.cuda.get_device_properties(0)
        pci = f"{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:
        pass
    gs = bench.GpuSampler(0, pci)
    if not gs.dev:
        return None
    for f in glob.glob(gs.dev + "/hwmon/hwmon*/power1_average") + glob.glob(gs.dev + "/hwmon/hwmon*/power1_input"):
        print(f"power from {f} (matched by pci: {gs.matched})", flush=True)
        return f
    return None
torch.zeros(1, device="cuda")
PF = power_file()

def run(name, fn, work, unit, seconds=1.5):
    for _ in range(5): fn(0)
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()
    def sampler():
        while not stop.is_set():
            try: samples.append((time.perf_counter(), int(open(PF).read()) / 1e6))
            except Exception: pass
            stop.wait(0.02)
    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    if PF: th.start()
    n = 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    while time.perf_counter() - t0 < seconds:
        for i in range(50): fn(n + i)
        n += 50
        torch.cuda.synchronize()
    e.record(); torch.cuda.synchronize()
    stop.set()
    us = s.elapsed_time(e) / n * 1e3
    late = [w for (t, w) in samples if t - t0 > 0.4]
    W = sum(late) / len(late) if late else float("nan")
    rate = work / (us * 1e-6)
    j_per = W / rate * (1e12 if unit == "TFLOP" else 1e9)
    print(f"{name:44s} {us:8.1f} us  {rate / (1e12 if unit == 'TFLOP' else 1e9):8.1f} {unit}/s  {W:7.1f} W  {j_per:7.3f} J/{unit}", flush=True)

NB = 4
def gemm_case(M, N, K, res=False):
    a = [torch.randn(M, K, device="cuda").to(BF) for _ in range(NB)]
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    r = [torch.randn(M, N, device="cuda").to(BF) for _ in range(NB)] if res else None
    out = torch.empty(M, N, device="cuda", dtype=BF)
    run(f"gemm {M}x{N}x{K}" + (" +res" if res else ""), lambda i: ops.gemm(a[i % NB], w, res=r[i % NB] if res else None, out=out), 2.0 * M * N * K, "TFLOP")
gemm_case(8192, 10240, 1280)
gemm_case(8192, 1280, 5120)
gemm_case(8192, 1280, 1280, res=True)
gemm_case(4096, 1280, 1280)
gemm_case(32768, 640, 640, res=True)
# conv
x = torch.randn(8, 64, 64, 640, device="cuda").to(BF)
wp = ops.pack_conv((torch.randn(640, 640, 3, 3, device="cuda") * (9 * 640) ** -0.5).to(BF).float())
run("conv3x3 8x64x64 640->640", lambda i: ops.conv3x3(x, wp), 2.0 * 8 * 64 * 64 * 640 * 9 * 640, "TFLOP")
# attention
for (B, H, S) in [(8, 20, 1024), (8, 10, 4096)]:
    C = H * 64
    q = (torch.randn(B, S, C, device="cuda") * 0.18).to(BF); k = torch.randn(B, S, C, device="cuda").to(BF); v = torch.randn(B, S, C, device="cuda").to(BF)
    o, lse = ops.attention_fwd(q, k, v, H, q_prescaled=True)
    run(f"self-attention fwd B{B} H{H} S{S}", lambda i: ops.attention_fwd(q, k, v, H, q_prescaled=True), 4.0 * B * H * S * S * 64, "TFLOP")
    Bb = B // 2
    do = torch.randn_like(o[:Bb])
    run(f"self-attention bwd B{Bb} H{H} S{S} (5 products)", lambda i: ops.attention_bwd(q[:Bb], k[:Bb], v[:Bb], o[:Bb], do, lse[:Bb], H, q_prescaled=True),
        10.0 * Bb * H * S * S * 64, "TFLOP")
B, H, S = 4, 20, 1024
C = H * 64
q = (torch.randn(B, S, C, device="cuda") * 0.18).to(BF); k = torch.randn(B, 77, C, device="cuda").to(BF); v = torch.randn(B, 77, C, device="cuda").to(BF)
o, lse = ops.attention_fwd(q, k, v, H, q_prescaled=True); do = torch.randn_like(o)
run("cross-attention bwd B4 H20 S1024 x 77 (bytes)", lambda i: ops.attention_bwd(q, k, v, o, do, lse, H, q_prescaled=True), 4.0 * B * S * C * 2, "GB")
# norms
xs = [torch.randn(8192, 1280, device="cuda").to(BF) for _ in range(NB)]
g = torch.ones(1280, device="cuda"); bta = torch.zeros(1280, device="cuda")
run("LayerNorm fwd 8192x1280 (bytes)", lambda i: ops.layernorm_fwd(xs[i % NB], g, bta), 2.0 * 8192 * 1280 * 2, "GB")
xg = [torch.randn(8, 16384, 320, device="cuda").to(BF) for _ in range(2)]
gg = torch.ones(320, device="cuda"); bg = torch.zeros(320, device="cuda")
run("GroupNorm fwd 8x16384x320 (min. bytes)", lambda i: ops.groupnorm_fwd(xg[i % 2], gg, bg, 32, 1e-5, True), 2.0 * 8 * 16384 * 320 * 2, "GB")
xg2 = [torch.randn(8, 1024, 1280, device="cuda").to(BF) for _ in range(NB)]
gg2 = torch.ones(1280, device="cuda"); bg2 = torch.zeros(1280, device="cuda")
run("GroupNorm fwd 8x1024x1280, one kernel (bytes)", lambda i: ops.groupnorm_fwd(xg2[i % NB], gg2, bg2, 32, 1e-5, True), 2.0 * 8 * 1024 * 1280 * 2, "GB")
