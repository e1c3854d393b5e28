"""Probe: the B = 4 KD step as ONE chain (the bench's form) against TWO half-batch chains (B = 2 each) on two HIP streams,
every GEMM grid limited to half the chip (PEA_CU_LIMIT=128, set by this script for the two-chain run) so that the chains
co-run and one chain's HBM-bound epilogues / small launches overlap the other's K-loops.  Runs each mode in a child
process (the CU limit is read once per process).  usage: python scripts/dual_chain_probe.py [steps]"""
import os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)

def run(mode, steps):
    import torch
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    sys.path.insert(0, root)
    import bench
    dev = torch.device("cuda", 0)
    cfg = pc.sdxl_config()
    hw = cfg.sample_size
    nch = 2 if mode != "one" else 1
    B = 4 // nch
    chains = []
    first = None
    for c in range(nch):
        student = HipUNet(cfg, B, hw, hw, 77, needs_grad=True, share_weights_from=first)
        if first is None:
            student.init_random(seed=7)
            first = student
        teacher = HipUNet(cfg, B, hw, hw, 77, needs_grad=False, share_weights_from=first)
        torch.manual_seed(7)
        adapter = PEAAdapter(1024, cfg.pooled_dim, 1280, cfg.cross_attention_dim, False).to(dev)
        trainer = PEATrainer(adapter, student, teacher)
        batch = bench.synthetic_batch(cfg, B, 77, 1024, hw, dev, seed=100 + c)
        chains.append((trainer, batch, torch.cuda.Stream(device=dev)))
    def step():
        for tr, b, st in chains:
            with torch.cuda.stream(st):
                tr.training_step(b)
                tr.optimizer_step()
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{mode}: {dt*1e3:.2f} ms per 4 images -> {4/dt:.2f} images/s (PEA_CU_LIMIT={os.environ.get('PEA_CU_LIMIT')})", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        run(sys.argv[2], int(sys.argv[3]))
    else:
        steps = sys.argv[1] if len(sys.argv) > 1 else "12"
        for mode, lim in (("one", None), ("two", "128"), ("two", None), ("two", "160"), ("one", None)):
            env = dict(os.environ)
            if lim: env["PEA_CU_LIMIT"] = lim
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode, steps], env=env, capture_output=True, text=True)
            print("\n".join(l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l)[-2000:], flush=True)
