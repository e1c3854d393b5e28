"""N>1 data-parallel path on CPU: world_size-2 `gloo` processes.  Each rank runs the CPU ORACLE step on
its shard of a global batch (the HIP step cannot run without a GPU), then the product's DP module
(pea_diffusion_amd/dist.py: ONE all-reduce of the flat adapter gradient, averaged) must reproduce the
gradient of the oracle step on the whole global batch."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import torch.distributed as dist
    from oracle.step_ref import AdapterRef, synthetic_batch, training_step_ref
    from oracle.unet_ref import UNet2DConditionRef, UNetConfig, cast_hook_ref
    from pea_diffusion_amd import dist as pdist
    assert pdist.init_from_env("gloo") == world
    cfg = UNetConfig(sample_size=8, block_out_channels=(32, 64), down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
                     up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 1),
                     num_attention_heads=(1, 1), cross_attention_dim=32, addition_time_embed_dim=8,
                     projection_class_embeddings_input_dim=16 + 48, layers_per_block=1, name="dp-toy")
    torch.manual_seed(0)
    us, ut = UNet2DConditionRef(cfg), UNet2DConditionRef(cfg)
    for p in list(us.parameters()) + list(ut.parameters()):
        p.requires_grad_(False)
    ad = AdapterRef(24, 16, 20, 32, False)
    flat0 = torch.cat([p.detach().reshape(-1) for p in ad.parameters()])
    if rank == 1:
        flat0 = flat0 + 1.0                       # diverged replica: broadcast must repair it
    pdist.broadcast_params_(flat0, src=0)
    o = 0
    with torch.no_grad():
        for p in ad.parameters():
            p.copy_(flat0[o:o + p.numel()].view_as(p))
            o += p.numel()
    gb = synthetic_batch(cfg, 4, L=5, enc_dim=24, seed=3)
    local = pdist.shard_batch(gb, rank, world)
    assert local["latents"].shape[0] == 2
    out = training_step_ref(ad, us, ut, local, cast_hook_ref)
    out["loss"].backward()
    flat_grad = torch.cat([p.grad.reshape(-1) for p in ad.parameters()])
    pdist.allreduce_mean_(flat_grad)
    loss_t = out["loss"].detach().clone()
    dist.all_reduce(loss_t)
    if rank == 0:
        for p in ad.parameters():
            p.grad = None
        ref = training_step_ref(ad, us, ut, gb, cast_hook_ref)
        ref["loss"].backward()
        g_ref = torch.cat([p.grad.reshape(-1) for p in ad.parameters()])
        q.put((float((flat_grad - g_ref).abs().max()), float(g_ref.abs().max()),
               float(loss_t / world), float(ref["loss"])))
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_gloo_allreduce_matches_global_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, scale, loss_dp, loss_ref = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err <= 1e-5 * max(scale, 1e-6) + 1e-9, (err, scale)
    assert abs(loss_dp - loss_ref) <= 1e-5 * abs(loss_ref)


def test_shard_batch_and_lr_schedule():
    sys.path.insert(0, ROOT)
    from pea_diffusion_amd import dist as pdist
    from pea_diffusion_amd.train import polynomial_lr
    b = {"x": torch.arange(8).view(8, 1), "s": 3}
    assert pdist.shard_batch(b, 1, 4)["x"].flatten().tolist() == [2, 3] and pdist.shard_batch(b, 0, 2)["s"] == 3
    # transformers polynomial decay with warmup (utils/model_utils.py:136-138): lr 1e-5, 100 warmup, end 5e-8
    assert polynomial_lr(0, 1e-5, 100, 2232142, 5e-8) == 0.0
    assert abs(polynomial_lr(50, 1e-5, 100, 2232142, 5e-8) - 5e-6) < 1e-12
    assert abs(polynomial_lr(100, 1e-5, 100, 2232142, 5e-8) - 1e-5) < 1e-12
    assert polynomial_lr(3_000_000, 1e-5, 100, 2232142, 5e-8) == 5e-8
    mid = polynomial_lr(1116121, 1e-5, 100, 2232142, 5e-8)
    assert 4.9e-6 < mid < 5.1e-6
