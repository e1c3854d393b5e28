"""GEMM / conv variant microbenchmark on the shapes of the SDXL KD step (B=4).  Interleaved rounds in one
process; torch.matmul (hipBLASLt) timed beside as a known-good reference ceiling on the same data."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from pea_diffusion_amd import ops
from pea_diffusion_amd._lib import lib

L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pea_diffusion_amd", "libpea_hip.so"))
BF = torch.bfloat16
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5,6,7".split(","))]
NAMES = {0: "128x128 w2x2 S2", 1: "128x128 w2x2 S3", 2: "128x128 w2x2 S4", 3: "256x128 w4x2 S2", 4: "256x128 w4x2 S3",
         5: "256x256 w2x4 S2", 6: "128x256 w2x4 S3", 7: "256x128 w2x2 S3", 8: "P128x160 w4x1 S4", 9: "P256x160 w4x1 S3",
         10: "P256x128 w4x2 S3", 11: "P128x128 w2x2 S3", 12: "P256x256 w2x4 S2", 13: "P128x160 w4x1 S3", 14: "LC128x160 c4+l4 S4", 15: "LC256x160 c4+l4 S3",
         16: "LC256x128 c8+l4 S3", 18: "LC128x128 c4+l4 S4", 19: "LC128x160 c4+l4 S3", 21: "LC16 128x160 2x2 S3",
         22: "LC16 128x160 2x2 S4", 23: "LC16 128x128 2x2 S4", 24: "LC16 256x160 c8(4x2)+l4 S3", 25: "LC16 128x160 c8(4x2)+l4 S3", 26: "LC16 128x160 c8(4x2)+l4 S4",
         27: "persistent LC16 256x160 c8(4x2)+l4 S3", 28: "persistent LC16 128x160 c8(4x2)+l4 S3", 29: "persistent LC16 128x160 c4(2x2)+l4 S4",
         30: "persistent LC16 128x128 c4(2x2)+l4 S4", 31: "persistent LC16 64x160 c4(2x2)+l4 S4", 33: "persistent LC16 256x128 c8(4x2)+l4 S3", 34: "persistent LC16 128x160 c8+l4+store4 S3, staged epilogue", 35: "persistent LC16 128x160 c8+l4 S3, deferred epilogue",
         36: "persistent LC16 128x160 c4(2x2)+l2 S2, TWO workgroups per CU", 37: "persistent LC16 128x128 c4(2x2)+l2 S2, two workgroups per CU"}

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

MERGED = len(sys.argv) > 2 and sys.argv[2] == "merged"      # forward shapes of the merged passes (2B = 8 samples)
shapes = [(8192, 1280, 1280), (8192, 1280, 5120), (8192, 10240, 1280), (4096, 1280, 1280), (4096, 1280, 5120), (4096, 3840, 1280), (4096, 10240, 1280), (16384, 640, 640),
          (16384, 640, 2560), (16384, 1920, 640), (16384, 5120, 640), (308, 2560, 2048), (4096, 1280, 2560), (8192, 8192, 8192)]
if MERGED:
    shapes = [(8192, 1280, 1280), (8192, 1280, 5120), (8192, 3840, 1280), (8192, 10240, 1280), (32768, 640, 640), (32768, 640, 2560),
              (32768, 1920, 640), (32768, 5120, 640), (616, 2560, 2048)]
print("variants:", {v: NAMES[v] for v in variants})
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    ref = a @ w.T
    tref = timeit(lambda: a @ w.T)
    line = f"gemm M{M} N{N} K{K}: torch {2*M*N*K/tref/1e12:7.1f} TF |"
    for v in variants:
        L.pea_debug_set_gemm_variant(v)
        out = ops.gemm(a, w)
        err = (out.float() - ref.float()).abs().max().item()
        t = timeit(lambda: ops.gemm(a, w))
        line += f" v{v} {2*M*N*K/t/1e12:7.1f}{'' if err < 0.5 else ' ERR%.2g' % err}"
    print(line, flush=True)
convs = [(8, 128, 320, 320), (8, 128, 960, 320), (8, 64, 640, 640), (8, 64, 1920, 640), (8, 32, 1280, 1280), (8, 32, 2560, 1280)] if MERGED else [(4, 128, 320, 320), (4, 128, 960, 320), (4, 64, 640, 640), (4, 64, 1920, 640), (4, 32, 1280, 1280), (4, 32, 2560, 1280)]
for (B, H, Ci, Co) in convs:
    x = torch.randn(B, H, H, Ci, device="cuda").to(BF)
    w = (torch.randn(Co, Ci, 3, 3, device="cuda") * (9 * Ci) ** -0.5)
    wp = ops.pack_conv(w.to(BF).float())
    xn = x.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last); wn = w.to(BF).contiguous(memory_format=torch.channels_last)
    ref = F.conv2d(xn, wn, padding=1)
    tref = timeit(lambda: F.conv2d(xn, wn, padding=1), 5)
    fl = 2.0 * B * H * H * Co * 9 * Ci
    line = f"conv B{B} H{H} {Ci}->{Co}: torch(MIOpen) {fl/tref/1e12:7.1f} TF |"
    for v in variants:
        L.pea_debug_set_gemm_variant(v)
        out = ops.conv3x3(x, wp)
        err = (out.float() - ref.permute(0, 2, 3, 1).float()).abs().max().item()
        t = timeit(lambda: ops.conv3x3(x, wp), 10)
        line += f" v{v} {fl/t/1e12:7.1f}{'' if err < 0.5 else ' ERR%.2g' % err}"
    print(line, flush=True)
L.pea_debug_set_gemm_variant(-1)
