"""How much of the step is launch gaps?  Captures one KD step (~2300 launches) into a HIP graph through torch and
replays it against the eager step (same process, interleaved rounds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import config as pc
from pea_diffusion_amd.adapter import PEAAdapter
from pea_diffusion_amd.train import PEATrainer
from pea_diffusion_amd.unet import HipUNet
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda")
cfg = pc.sdxl_config(); B = 4; hw = 128
student = HipUNet(cfg, B, hw, hw, 77, needs_grad=True); student.init_random(7)
teacher = HipUNet(cfg, B, hw, hw, 77, share_weights_from=student)
ad = PEAAdapter(1024, 1280, 1024, 2048, False).to(dev)
tr = PEATrainer(ad, student, teacher)
batch = bench.synthetic_batch(cfg, B, 77, 1024, hw, dev, 100)
batch = {k: v for k, v in batch.items()}
for _ in range(3): tr.training_step(batch)
torch.cuda.synchronize()
def timeit(fn, n=8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): tr.training_step(batch)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            tr.training_step(batch)
        ok = True
    except Exception as e:
        print("capture failed:", repr(e)[:400]); ok = False
for r in range(3):
    print(f"round {r}: eager {timeit(lambda: tr.training_step(batch)):.2f} ms" + (f" | graph replay {timeit(g.replay):.2f} ms" if ok else ""), flush=True)
