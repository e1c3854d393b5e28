"""-m gpu: dead-row elimination of the merged pass (PEATrainer.skip_dead_teacher_rows).  A sample with zh_or_not == 1 carries
KD weight (1 - zh_or_not) = 0 (train_sdxl_zh.py:402-441): its teacher row is never read by the loss, so the merged forward
runs over B + n_t samples instead of 2B.  The step must give the SAME four scalars and the same adapter gradient as the full
merged step (same kernels on the surviving rows: eps of the student rows and of the live teacher rows is compared bit for
bit), for every mask pattern incl. n_t = 0 and n_t = B, and against the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu
from test_model_gpu import gpu, lib_merge_state, rel_l2, round_weights_bf16_  # noqa: E402,F401


def _rows(tr):
    from pea_diffusion_amd._lib import lib
    return lib().pea_trainer_get_option(tr._h, b"merged_rows")


@pytest.mark.parametrize("L", [77, 12])          # 12: a shorter student context merged with the 77-token teacher (per-sample key counts)
def test_dead_rows_match_full_merged_step_tiny(gpu, L):
    from oracle import unet_ref as ou
    from oracle.step_ref import AdapterRef, synthetic_batch, training_step_ref
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    B = 4
    cfg = ou.tiny_config()
    torch.manual_seed(3)
    us = ou.UNet2DConditionRef(cfg)
    round_weights_bf16_(us)
    for p in us.parameters():
        p.requires_grad_(False)
    import copy
    ut = copy.deepcopy(us)
    ad_ref = AdapterRef(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False)
    ad = PEAAdapter(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False)
    ad.load_state_dict(ad_ref.state_dict())
    ad = ad.cuda()
    round_weights_bf16_(ad_ref)
    hs = HipUNet(pc.tiny_config(), B, 16, 16, L, needs_grad=True)
    hs.load_state_dict(us.state_dict())
    ht = HipUNet(pc.tiny_config(), B, 16, 16, 77, share_weights_from=hs)
    tr = PEATrainer(ad, hs, ht)
    base = synthetic_batch(cfg, B, L=L, enc_dim=128, seed=4)
    for zh in ([1, 0, 0, 1], [0, 0, 0, 0], [1, 1, 1, 1], [0, 1, 1, 1], [1, 1, 0, 1]):
        batch = dict(base, zh_or_not=torch.tensor(zh, dtype=torch.int64))
        tr.skip_dead_teacher_rows = False
        full = {k: float(v) for k, v in tr.training_step(batch, 0, sync=True).items()}
        assert lib_merge_state(tr) == 1 and _rows(tr) == 2 * B
        g_full, eps_s, eps_t = ad.flat_grad.clone(), tr.export("eps_student"), tr.export("eps_teacher")
        tr.skip_dead_teacher_rows = True
        out = {k: float(v) for k, v in tr.training_step(batch, 0, sync=True).items()}
        n_t = zh.count(0)
        assert _rows(tr) == B + n_t, (zh, _rows(tr))
        assert torch.equal(tr.export("eps_student"), eps_s), zh          # the surviving rows run the same arithmetic
        et = tr.export("eps_teacher")
        for i, z in enumerate(zh):
            if z == 0:
                assert torch.equal(et[i], eps_t[i]), (zh, i)
            else:
                assert torch.isnan(et[i]).all()                         # never computed
        for k in tr.LOG_KEYS:
            assert abs(out[k] - full[k]) <= 1e-6 * max(1.0, abs(full[k])), (zh, k, out[k], full[k])
        e = rel_l2(ad.flat_grad, g_full) if float(g_full.abs().max()) > 0 else float(ad.flat_grad.abs().max())
        assert e < 1e-4, (zh, e)       # (cross-attention dK / dV partial sums may be split differently at another batch size)
    # ... and against the oracle for the mixed mask
    batch = dict(base, zh_or_not=torch.tensor([1, 0, 0, 1], dtype=torch.int64))
    tr.skip_dead_teacher_rows = True
    out = tr.training_step(batch, 0, sync=True)
    bq = dict(batch)
    for k in ("enc", "enc_uncond", "teacher_ehs", "teacher_neg", "teacher_pooled"):
        bq[k] = batch[k].to(torch.bfloat16).float()
    ref = training_step_ref(ad_ref, us, ut, bq, ou.cast_hook_ref)
    ref["loss"].backward()
    total = abs(float(ref["loss"]))
    for k in tr.LOG_KEYS:
        assert abs(float(out[k]) - float(ref[k])) <= 2e-2 * abs(float(ref[k])) + 5e-3 * total, k
    g_ref = torch.cat([p.grad.reshape(-1) for p in ad_ref.parameters()])
    assert rel_l2(ad.flat_grad, g_ref) < 4e-2
    # switching back restores the full pass
    tr.skip_dead_teacher_rows = False
    tr.training_step(batch, 0, sync=True)
    assert _rows(tr) == 2 * B


def test_dead_rows_sdxl_1024_bench_workload(gpu):
    """the bench workload (SDXL 1024x1024, batch 4, half of the samples native captions): 6 rows instead of 8 in the merged pass,
    identical eps / losses, gradient equal to rounding of the split reductions"""
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    import bench
    cfg, hw, L, B = pc.sdxl_config(), 128, 77, 4
    student = HipUNet(cfg, B, hw, hw, L, needs_grad=True)
    student.init_random(3)
    teacher = HipUNet(cfg, B, hw, hw, L, share_weights_from=student)
    torch.manual_seed(0)
    ad = PEAAdapter(1024, 1280, 1024, 2048, False).cuda()
    tr = PEATrainer(ad, student, teacher)
    batch = bench.synthetic_batch(cfg, B, L, 1024, hw, torch.device("cuda"), seed=100)
    assert batch["zh_or_not"].tolist() == [1, 1, 0, 0]
    full = {k: float(v) for k, v in tr.training_step(batch, 0, sync=True).items()}
    g_full, eps_s, eps_t = ad.flat_grad.clone(), tr.export("eps_student"), tr.export("eps_teacher")
    tr.skip_dead_teacher_rows = True
    out = {k: float(v) for k, v in tr.training_step(batch, 0, sync=True).items()}
    assert _rows(tr) == 6
    assert torch.equal(tr.export("eps_student"), eps_s) and torch.equal(tr.export("eps_teacher")[2:], eps_t[2:])
    for k in tr.LOG_KEYS:
        assert abs(out[k] - full[k]) <= 1e-6 * max(1.0, abs(full[k])), (k, out[k], full[k])
    e = rel_l2(ad.flat_grad, g_full)
    print(f"[dead rows, SDXL 1024 B=4] merged rows 8 -> 6; adapter gradient vs the full pass rel_l2={e:.2e}; loss {out['loss']:.6f}")
    assert e < 1e-4


def test_dead_rows_behind_the_bucketed_trainer(gpu):
    """dead-row elimination through BucketedTrainer: every latent shape has its own trainer context (and its own B + n_t
    contexts behind it); releasing a context (memory cap 0) frees the borrowed arenas' donor and the borrowers together"""
    from oracle.step_ref import synthetic_batch
    from pea_diffusion_amd.train import BucketedTrainer
    from test_buckets_gpu import _tiny_models
    B, L = 2, 52
    shapes = [(16, 16), (24, 8)]
    cfg, us, ut, ad_ref, ad, hs, ht = _tiny_models(B, L, shapes[0])
    tr = BucketedTrainer(ad, hs, ht)
    for cap in (1 << 40, 0):
        tr.max_resident_bytes = cap
        for hw in shapes + shapes[::-1]:
            batch = dict(synthetic_batch(cfg, B, L=L, enc_dim=128, seed=7, latent_hw=hw), zh_or_not=torch.tensor([1, 0]))
            tr.skip_dead_teacher_rows = False
            full = {k: float(v) for k, v in tr.training_step(batch, 0, sync=True).items()}
            g_full = ad.flat_grad.clone()
            tr.skip_dead_teacher_rows = True
            out = {k: float(v) for k, v in tr.training_step(batch, 0, sync=True).items()}
            assert _rows(tr) == B + 1, (hw, _rows(tr))
            for k in tr.LOG_KEYS:
                assert abs(out[k] - full[k]) <= 1e-6 * max(1.0, abs(full[k])), (hw, cap, k)
            assert rel_l2(ad.flat_grad, g_full) < 1e-4, (hw, cap)


def test_dead_rows_sdxl_1024_batch8(gpu):
    """per-GPU batch 8 (BASELINE configs[2] per rank) with half of the teacher rows dead: 16 -> 12 merged rows.  (Round 4: the
    12-sample context's attention-backward scratch was sized for 12 samples' split count while the backward of its 8 student
    samples chooses more splits -- a memory fault; Tape::ensure_acts now sizes for both.)"""
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    import bench
    cfg, hw, L, B = pc.sdxl_config(), 128, 77, 8
    student = HipUNet(cfg, B, hw, hw, L, needs_grad=True)
    student.init_random(3)
    teacher = HipUNet(cfg, B, hw, hw, L, share_weights_from=student)
    torch.manual_seed(0)
    ad = PEAAdapter(1024, 1280, 1024, 2048, False).cuda()
    tr = PEATrainer(ad, student, teacher)
    batch = bench.synthetic_batch(cfg, B, L, 1024, hw, torch.device("cuda"), seed=100)
    full = {k: float(v) for k, v in tr.training_step(batch, 0, sync=True).items()}
    g_full, eps_s = ad.flat_grad.clone(), tr.export("eps_student")
    tr.skip_dead_teacher_rows = True
    for zh in ([1, 1, 1, 1, 0, 0, 0, 0], [0, 1, 1, 1, 1, 1, 1, 1]):
        b2 = dict(batch, zh_or_not=torch.tensor(zh))
        tr.skip_dead_teacher_rows = False
        full = {k: float(v) for k, v in tr.training_step(b2, 0, sync=True).items()}
        g_full, eps_s = ad.flat_grad.clone(), tr.export("eps_student")
        tr.skip_dead_teacher_rows = True
        out = {k: float(v) for k, v in tr.training_step(b2, 0, sync=True).items()}
        assert _rows(tr) == B + zh.count(0)
        assert torch.equal(tr.export("eps_student"), eps_s)
        for k in tr.LOG_KEYS:
            assert abs(out[k] - full[k]) <= 1e-6 * max(1.0, abs(full[k])), (zh, k)
        assert rel_l2(ad.flat_grad, g_full) < 1e-4, zh
    del tr, student, teacher
    torch.cuda.empty_cache()
