"""Where the in-situ penalty of a one-round GEMM comes from (round 5).

A chain that looks like the block backward at the 32 x 32 level:  for l in layers:  LayerNorm-forward(x_l -> A)  ->  GEMM(A, W_l -> C_l)
with every buffer at its own address (activations from a pool larger than the Infinity Cache, as in the 23 GB arena of the step).
Arms (all in one process, per-GEMM time = (chain - the same chain without the GEMMs) / layers):
  hot        the same W every layer (weights come from the Infinity Cache)
  cold       W_l from a pool of `nw` matrices (> 256 MiB in all): what the step sees
  prefetch   cold + pea_op_prefetch(W_{l+1}) on a side stream while layer l runs (next-op weight prefetch)
  a_old      cold, and the GEMM's A operand was written long ago (not by the kernel right in front of it)
  inkernel   cold + the previous GEMM's DMA waves touch W_{l+1} behind their last K-step (GemmP::pf_ptr: what the tapes do)
usage: python scripts/chain_probe.py [M N K]..."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
from pea_diffusion_amd._lib import lib, check, ptr
BF = torch.bfloat16
L = lib()


def run(M, N, K, layers=48):
    dev = "cuda"
    nw = max(layers, int(400e6 / (N * K * 2)) + 1)
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(BF) for _ in range(nw)]
    npool = max(8, int(600e6 / (M * (N + 2 * K) * 2)) + 1)
    xs = [torch.randn(M, K, device=dev).to(BF) for _ in range(npool)]
    As = [torch.empty(M, K, device=dev, dtype=BF) for _ in range(npool)]
    Cs = [torch.empty(M, N, device=dev, dtype=BF) for _ in range(npool)]
    gamma = torch.ones(K, device=dev); beta = torch.zeros(K, device=dev)
    stats = torch.empty(M, 2, device=dev)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()

    def ln(i):
        check(L.pea_op_layernorm_fwd(ptr(xs[i % npool]), ptr(gamma), ptr(beta), ptr(As[i % npool]), ptr(stats), M, K, 1e-5, ctypes.c_void_p(main.cuda_stream)))

    def chain(mode, gemms=True):
        for l in range(layers):
            ln(l)
            if mode == "prefetch" and gemms and l + 1 < layers:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                w = Ws[(l + 1) % nw]
                check(L.pea_op_prefetch(ptr(w), w.numel() * 2, ctypes.c_void_p(side.cuda_stream)))
            if gemms:
                if mode == "inkernel":
                    wn = Ws[(l + 1) % nw]
                    L.pea_debug_set_gemm_prefetch(ptr(wn), wn.numel() * 2)
                w = Ws[0] if mode == "hot" else Ws[l % nw]
                a = As[(l + npool // 2) % npool] if mode == "a_old" else As[l % npool]
                ops.gemm(a, w, out=Cs[l % npool])

    def t(mode, gemms=True, reps=6):
        chain(mode, gemms); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); chain(mode, gemms); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3 / layers)
        ts.sort()
        return ts[len(ts) // 2]

    base = t("cold", gemms=False)
    fl = 2.0 * M * N * K
    line = f"M{M} N{N} K{K}: LN alone {base:5.1f} us/layer |"
    for mode in ("hot", "cold", "prefetch", "inkernel", "a_old", "hot", "cold", "inkernel"):
        g = t(mode) - base
        line += f" {mode} {g:5.1f} us ({fl / g / 1e6:4.0f} TF)"
    print(line, flush=True)


if __name__ == "__main__":
    shapes = [(4096, 1280, 1280), (8192, 1280, 1280), (4096, 1280, 3840), (8192, 3840, 1280), (8192, 10240, 1280), (8192, 1280, 5120), (4096, 5120, 1280)]
    if len(sys.argv) > 3:
        v = [int(x) for x in sys.argv[1:]]
        shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
    for s in shapes:
        run(*s, layers=48 if s[1] * s[2] < 8e6 else 24)
