"""Model-level parity (-m gpu): the HIP UNet tape (forward + data-gradient backward), the adapter and
the fused KD training step, called through the C ABI, against the CPU oracle (oracle/*.py) on the same
seeded inputs and the same (bf16-rounded) weights.

Tolerances: single kernels are held to 1 bf16 ulp in tests/test_ops_gpu.py.  Through a whole UNet
(~10^2 chained bf16-stored ops) element-wise 1e-3 is not meaningful (SURVEY 7 "tolerance vs depth"), so
end-to-end checks use relative L2 error: <= 2e-2 for forward tensors, <= 4e-2 for gradients that went
through the full forward + backward, and rtol 1e-2 for the loss scalars (fp32 reductions of bf16 data).
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# check_against_storage_floor: the HIP path may deviate from the fp32 oracle by this factor times what bf16 storage alone does to
# the same quantity (the oracle in bf16-storage mode, oracle/bf16_store.py); measured ratios: profiles/r05_new_parity_tests.log
STORAGE_FLOOR_FACTOR = 1.5      # measured: 0.97-1.03 (eps), 0.97-1.22 (flat gradient), 1.06 (full SDXL) - 1.31 (tiny) worst single layer
# 20-step trajectory (test_training_trajectory_vs_oracle): final parameters / accumulated update against the oracle's
TRAJ_W_LIM, TRAJ_D_LIM, TRAJ_LOSS_LIM = 7e-3, 4e-2, 2e-3      # measured 2.2e-3 / 1.1e-2 / 2.4e-4 (weights move by 20 %)


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X")
    return torch.device("cuda")


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).pow(2).sum().sqrt() / (b.pow(2).sum().sqrt() + 1e-30)).item()


def round_weights_bf16_(module):
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() >= 2:
                p.copy_(p.to(torch.bfloat16).float())


def layer_grad_hooks(unet):
    """ORACLE side of the per-layer gradient check: the gradient w.r.t. the OUTPUT of every cross-attention `attn2.to_k` /
    `attn2.to_v` projection and of every `ResnetBlock2D.time_emb_proj` (the two routes by which a loss reaches the adapter:
    SURVEY 3.1 gradient routes a / b, graph of train_sdxl_zh.py:397), keyed by the module's state-dict weight key."""
    store, handles = {}, []
    for name, m in unet.named_modules():
        if name.endswith("attn2.to_k") or name.endswith("attn2.to_v") or name.endswith("time_emb_proj"):
            def fwd(mod, inp, out, key=name + ".weight"):
                if out.requires_grad:
                    out.register_hook(lambda g, key=key: store.__setitem__(key, g.detach().clone()))
            handles.append(m.register_forward_hook(fwd))
    return store, handles


def hip_layer_grads(tr):
    """HIP side: the per-layer column blocks of the two stacked projections' gradients after the trainer's last backward pass
    (pea_unet_stacked_grad / _layout), keyed by the same diffusers weight keys; K|V blocks as [rows][C], time_emb_proj as [B][C]"""
    from pea_diffusion_amd._lib import check, lib, ptr, stream_ptr
    L_ = lib()
    ctx = ctypes.c_void_p()
    check(L_.pea_trainer_backward_context(tr._h, ctypes.byref(ctx)))
    res = {}
    for which in (0, 1):
        rows, cols = ctypes.c_longlong(), ctypes.c_int()
        check(L_.pea_unet_stacked_grad(ctx, which, None, ctypes.byref(rows), ctypes.byref(cols), stream_ptr()))
        buf = torch.empty(rows.value, cols.value, device="cuda")
        rc = L_.pea_unet_stacked_grad(ctx, which, ptr(buf), ctypes.byref(rows), ctypes.byref(cols), stream_ptr())
        if which == 1 and rc == -5:              # PEA_E_NOTFOUND: the time embedding receives no gradient on this graph (SD1.5)
            continue
        check(rc)
        torch.cuda.synchronize()
        i = 0
        while True:
            name = ctypes.create_string_buffer(256)
            off, n = ctypes.c_int(), ctypes.c_int()
            if L_.pea_unet_stacked_layout(ctx, which, i, name, 256, ctypes.byref(off), ctypes.byref(n)) != 0:
                break
            res[name.value.decode()] = buf[:, off.value:off.value + n.value].cpu()
            i += 1
    return res


def _unpad_heads(h, C):
    """HIP stores heads whose width is not a multiple of 64 zero-padded (SD1.5: 40 / 80 / 160 -> 64 / 128 / 192): drop the
    padding columns of a [rows][heads * dp] block so that it lines up with the oracle's [rows][heads * d]"""
    Cp = h.shape[-1]
    if Cp == C:
        return h
    for heads in (8, 5, 10, 20, 4, 2, 1, 16):
        if C % heads == 0 and Cp % heads == 0 and Cp // heads == 64 * ((C // heads + 63) // 64):
            d, dp = C // heads, Cp // heads
            pad = h.reshape(h.shape[0], heads, dp)
            assert float(pad[:, :, d:].abs().max()) == 0.0, "gradient in the padding columns of a padded head"
            return pad[:, :, :d].reshape(h.shape[0], C)
    raise AssertionError((Cp, C))


def check_layer_grads(tr, store, B, limit, tag="", kv_only=False):
    """every cross-attention layer's dK and dV and every resnet's time-embedding gradient, EACH against the oracle's: a layer
    that contributes nothing (or twice) to d(encoder_hidden_states) / d(text_embeds) fails here although it would move the
    flat adapter gradient by a percent or two only"""
    hip = hip_layer_grads(tr)
    if kv_only:                                  # (a model whose time embedding receives no gradient: SD1.5)
        hip = {k: v for k, v in hip.items() if not k.endswith("time_emb_proj.weight")}
    assert sorted(hip.keys()) == sorted(store.keys()), (sorted(set(hip) ^ set(store))[:6], len(hip), len(store))
    worst = {"kv": (0.0, ""), "temb": (0.0, "")}
    for k, g in store.items():
        h = hip[k]
        if k.endswith("time_emb_proj.weight"):
            want, got, fam = g.reshape(B, -1), h, "temb"
        else:
            want = g.reshape(B, -1, g.shape[-1])
            got = _unpad_heads(h, g.shape[-1])
            got = got.reshape(B, -1, got.shape[-1])[:, :want.shape[1]]          # (a merged context is as long as the teacher's)
            fam = "kv"
        e = rel_l2(got, want)
        if e > worst[fam][0]:
            worst[fam] = (e, k)
        assert e < limit, (k, e, float(want.norm()))
    n_kv = sum(1 for k in store if not k.endswith("time_emb_proj.weight"))
    print(f"   [{tag}] per-layer gradients: {n_kv} K / V projections worst {worst['kv'][0]:.2e} ({worst['kv'][1]}), "
          f"{len(store) - n_kv} time_emb_proj worst {worst['temb'][0]:.2e} ({worst['temb'][1]}); limit {limit:.1e} each")
    return worst


def check_against_storage_floor(tag, B, hip_eps, hip_grad, hip_layers, ref32, st16, factor=None):
    """The adaptive form of the end-to-end tolerance.  `ref32` / `st16`: (eps_student, eps_teacher, flat adapter gradient,
    per-layer gradient dict) of the fp32 oracle and of the SAME oracle in bf16-storage mode (oracle/bf16_store.py: values and
    gradients rounded wherever the HIP path stores a tensor).  The distance between those two is what bf16 storage alone does
    to each quantity -- the noise floor of this model, batch and size, measured in the run.  The HIP path may deviate from the
    fp32 oracle by `factor` x that floor, no more: a missing or doubled term the size of the floor (one of 70 cross-attention
    layers is 1.4 % of d ehs) lifts the HIP error to 1.4-2 x the floor and fails, where a fixed 2e-2 limit let it pass.
    (Comparing HIP with the bf16-storage oracle directly does not sharpen anything: two bf16 realisations of ~900 chained ops
    are as far from each other as each is from fp32 -- measured in round 5, profiles/EXPERIMENTS.md.)"""
    factor = factor or STORAGE_FLOOR_FACTOR
    rows = []
    for name, h, a, b in (("eps_student", hip_eps[0], ref32[0], st16[0]), ("eps_teacher", hip_eps[1], ref32[1], st16[1]),
                          ("flat adapter gradient", hip_grad, ref32[2], st16[2])):
        rows.append((name, rel_l2(h, a), rel_l2(b, a)))
    worst = (0.0, "", 0.0, 0.0)
    for k, g32 in ref32[3].items():
        h = hip_layers[k]
        if k.endswith("time_emb_proj.weight"):
            want, got, st = g32.reshape(B, -1), h, st16[3][k].reshape(B, -1)
        else:
            want = g32.reshape(B, -1, g32.shape[-1])
            got = h.reshape(B, -1, h.shape[-1])[:, :want.shape[1]]
            st = st16[3][k].reshape(want.shape)
        e, f = rel_l2(got, want), rel_l2(st, want)
        if e / max(f, 1e-12) > worst[0]:
            worst = (e / max(f, 1e-12), k, e, f)
        assert e <= factor * f + 1e-3, (tag, k, e, f)
    print(f"   [{tag}] error vs fp32 oracle / bf16-storage noise floor: " +
          "; ".join(f"{n} {e:.2e} / {f:.2e} = {e / f:.2f}" for n, e, f in rows) +
          f"; worst single layer {worst[2]:.2e} / {worst[3]:.2e} = {worst[0]:.2f} ({worst[1]}); limit {factor:.2f} x floor")
    for n, e, f in rows:
        assert e <= factor * f + 1e-3, (tag, n, e, f)


def make_pair(cfg_fn, B, L, needs_grad, seed=0, hw=None):
    from oracle.unet_ref import UNet2DConditionRef
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.unet import HipUNet
    ocfg = cfg_fn()
    torch.manual_seed(seed)
    ref = UNet2DConditionRef(ocfg)
    round_weights_bf16_(ref)
    hw = hw or ocfg.sample_size
    hip = HipUNet(getattr(pc, cfg_fn.__name__)(), B, hw, hw, L, needs_grad=needs_grad)
    missing, unexpected = hip.load_state_dict(ref.state_dict())
    assert not missing and not unexpected
    return ocfg, ref, hip


def cond_inputs(cfg, B, L, hw, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, hw, hw, generator=g)
    t = torch.tensor([10, 250, 500, 999][:B])
    ehs = torch.randn(B, L, cfg.cross_attention_dim, generator=g)
    added = None
    if cfg.addition_embed_type == "text_time":
        added = {"text_embeds": torch.randn(B, cfg.pooled_dim, generator=g),
                 "time_ids": torch.tensor([[hw * 8, hw * 8, 0, 0, hw * 8, hw * 8]] * B)}
    return x, t, ehs, added


def test_weight_table_matches_oracle_state_dict(gpu):
    from oracle.unet_ref import UNet2DConditionRef, tiny_config
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.unet import HipUNet
    with torch.device("meta"):
        ref = UNet2DConditionRef(tiny_config())
    hip = HipUNet(pc.tiny_config(), 1, 16, 16, 7)
    table = hip.weight_table()
    sd = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert set(table) == set(sd)
    for k, shp in table.items():
        assert int(np.prod(shp)) == int(np.prod(sd[k])), k


@pytest.mark.parametrize("B,L", [(2, 7), (1, 77)])
def test_unet_forward_tiny(gpu, B, L):
    from oracle.unet_ref import cast_hook_ref, tap_names, tiny_config
    cfg, ref, hip = make_pair(tiny_config, B, L, needs_grad=False)
    x, t, ehs, added = cond_inputs(cfg, B, L, cfg.sample_size)
    taps = {}
    cast_hook_ref(ref, taps)
    with torch.no_grad():
        eref = ref(x, t, ehs.to(torch.bfloat16).float(), added_cond_kwargs=added)[0]
    # reference-style hooks on the HIP UNet (train_sdxl_zh.py:69-84 semantics)
    got = {}
    n = len(cfg.block_out_channels)
    for i in range(n):
        hip.down_blocks[i].register_forward_hook(lambda m, inp, out, k=f"d{i}": got.__setitem__(k, out[0]))
        hip.up_blocks[i].register_forward_hook(lambda m, inp, out, k=f"u{i}": got.__setitem__(k, out))
    hip.mid_block.register_forward_hook(lambda m, inp, out: got.__setitem__("m", out))
    eps = hip(x.cuda(), t.cuda(), ehs.cuda(), added_cond_kwargs={k: v.cuda() for k, v in added.items()})[0]
    e = rel_l2(eps, eref)
    print(f"[unet fwd tiny B{B} L{L}] eps rel_l2={e:.3e}")
    assert set(got) == set(tap_names(cfg))
    for k in tap_names(cfg):
        ek = rel_l2(got[k], taps[k])
        print(f"   tap {k}: shape {tuple(got[k].shape)} rel_l2={ek:.3e}")
        assert got[k].shape == taps[k].shape and ek < 2e-2, k
    assert e < 2e-2
    # storage layout behind the raw tap pointers: an up block that ends in an upsampler hands out the sub-pixel conv's
    # depth-to-space tensor (1) unless PEA_UPCONV_SUBPIXEL=0; everything else is NHWC (0).  The export above hid it.
    import os
    subpixel = os.environ.get("PEA_UPCONV_SUBPIXEL", "1") != "0"
    names = tap_names(cfg)
    if subpixel:
        # raw pointers of a depth-to-space tap are refused until the caller has asked for the layout (the shape alone does not
        # reveal the storage order); shapes without pointers, and NHWC taps, are always served
        import ctypes
        from pea_diffusion_amd._lib import lib
        d, Bc = ctypes.c_void_p(), ctypes.c_int()
        iu = names.index("u0")
        assert lib().pea_unet_tap_info(hip._h, iu, ctypes.byref(d), None, ctypes.byref(Bc), None, None, None) == -4
        assert b"depth-to-space" in lib().pea_last_error()
        assert lib().pea_unet_tap_info(hip._h, iu, None, None, ctypes.byref(Bc), None, None, None) == 0 and Bc.value == B
        assert lib().pea_unet_tap_info(hip._h, names.index("d0"), ctypes.byref(d), None, None, None, None, None) == 0 and d.value
    for i, k in enumerate(names):
        ends_in_upsampler = k.startswith("u") and int(k[1:]) != n - 1
        assert hip.tap_layout(i) == (1 if (ends_in_upsampler and subpixel) else 0), (k, hip.tap_layout(i))
    assert hip.tap_layout(len(names)) == -1
    dptr, gptr, shape, layout = hip.tap_pointers(names.index("u0"))     # (asks for the layout itself)
    assert dptr and shape[0] == B and layout == (1 if subpixel else 0)


@pytest.mark.parametrize("geglu_bwd_fused", [1, 0])
def test_unet_backward_tiny(gpu, geglu_bwd_fused):
    """geglu_bwd_fused 1 (default): the GEGLU backward runs in the epilogue of the FF output projection's dgrad GEMM
    (d y is never stored); 0: as its own kernel.  Both against the fp32 oracle."""
    from oracle.unet_ref import cast_hook_ref, tap_names, tiny_config
    from pea_diffusion_amd._lib import check, lib, ptr, stream_ptr
    lib().pea_debug_set_geglu_bwd_fused(geglu_bwd_fused)
    try:
        _unet_backward_tiny(geglu_bwd_fused)
    finally:
        lib().pea_debug_set_geglu_bwd_fused(1)


_bwd_tiny_grads = {}


def _unet_backward_tiny(mode):
    from oracle.unet_ref import cast_hook_ref, tap_names, tiny_config
    from pea_diffusion_amd._lib import check, lib, ptr, stream_ptr
    B, L = 2, 9
    cfg, ref, hip = make_pair(tiny_config, B, L, needs_grad=True)
    for p in ref.parameters():
        p.requires_grad_(False)
    x, t, ehs, added = cond_inputs(cfg, B, L, cfg.sample_size)
    ehs_r = ehs.to(torch.bfloat16).float().requires_grad_(True)
    te_r = added["text_embeds"].to(torch.bfloat16).float().requires_grad_(True)
    taps = {}
    cast_hook_ref(ref, taps)
    eref = ref(x, t, ehs_r, added_cond_kwargs={"text_embeds": te_r, "time_ids": added["time_ids"]})[0]
    g = torch.Generator().manual_seed(5)
    names = tap_names(cfg)
    seeds = {k: torch.randn(taps[k].shape, generator=g) * 0.1 for k in names}
    d_eps = torch.randn(eref.shape, generator=g)
    loss = (eref * d_eps).sum() + sum((taps[k] * seeds[k]).sum() for k in names)
    loss.backward()
    hip(x.cuda(), t.cuda(), ehs.cuda(), added_cond_kwargs={k: v.cuda() for k, v in added.items()})
    mask = 0
    for i, k in enumerate(names):
        hip.set_tap_grad(i, seeds[k])        # (an upsampler tap is stored depth-to-space: the import hides the layout)
        mask |= 1 << i
    d_ehs, d_text = hip.backward(d_eps.cuda(), mask)
    e1, e2 = rel_l2(d_ehs, ehs_r.grad), rel_l2(d_text, te_r.grad)
    print(f"[unet bwd tiny, geglu bwd fused={mode}] d_ehs rel_l2={e1:.3e} |ref|={ehs_r.grad.norm():.3e}  d_text rel_l2={e2:.3e} "
          f"|ref|={te_r.grad.norm():.3e}")
    assert e1 < 4e-2 and e2 < 4e-2
    _bwd_tiny_grads[mode] = d_ehs.clone()
    if len(_bwd_tiny_grads) == 2:      # the two forms differ by one bf16 rounding of d y: close, and NOT bit-identical
        d = rel_l2(_bwd_tiny_grads[1], _bwd_tiny_grads[0])
        print(f"[unet bwd tiny] fused vs own-kernel GEGLU backward: rel_l2={d:.3e}")
        assert 0 < d < 3e-2, "fused path not taken (bit-identical) or far from the unfused one"


def test_adapter_module_matches_reference_golden(gpu, golden_dir):
    """PEAAdapter vs golden vectors captured from the reference's own MLP (small dims are multiples of 64
    only for `sd15_full`/`sdxl_6M`/...; the small goldens use dims the MFMA tiles do not accept, so the
    full-size seeded goldens are used here)."""
    import os
    from pea_diffusion_amd.adapter import PEAAdapter
    # sdxl_in2048 / sdxl_in768: the constructors of the other student encoders (train_sdxl_zh.py:113,124,134)
    for tag in ["sdxl_6M", "sdxl_11M", "sdxl_in2048", "sdxl_in768", "sd15_full"]:
        g = np.load(os.path.join(golden_dir, f"mlp_{tag}.npz"))
        args = [int(a) for a in g["args"]]
        torch.manual_seed(int(g["seed"]))
        m = PEAAdapter(*args[:3], None, False) if len(args) == 3 else PEAAdapter(args[0], args[1], args[2], args[3], bool(args[4]))
        wsum = float(sum(v.double().abs().sum().item() for v in m.state_dict().values()))
        assert abs(wsum - float(g["wsum"])) < 1e-6 * float(g["wsum"])
        assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
        m = m.cuda()
        out = m(torch.from_numpy(g["x"]).cuda())
        outs = out if isinstance(out, tuple) else (out,)
        for i, o in enumerate(outs):
            e = rel_l2(o, torch.from_numpy(g[f"out{i}"]))
            print(f"[adapter golden {tag}] out{i} rel_l2={e:.3e}")
            assert e < 1e-2
        # backward (round 4): the reference MLP's parameter gradients for the fixture's seeded output gradients -- norm and
        # a strided sample of every gradient -- against pea_adapter_backward (dgrad + wgrad in HIP, bf16 operands)
        torch.autograd.backward(outs, [torch.from_numpy(g[f"gout{i}"]).cuda() for i in range(len(outs))])
        worst = 0.0
        for k, p_ in m.named_parameters():
            flat = p_.grad.reshape(-1).float().cpu()
            want = torch.from_numpy(g["gsample." + k])
            got = flat[::int(g["gstride." + k])]
            e = rel_l2(got, want)
            en = abs(float(flat.double().norm()) - float(g["gnorm." + k])) / float(g["gnorm." + k])
            worst = max(worst, e)
            assert e < 2e-2 and en < 1e-2, (tag, k, e, en)
        print(f"[adapter golden {tag}] parameter gradients vs the reference MLP: worst sample rel_l2={worst:.3e}")


def test_adapter_half_surface(gpu):
    """`proj = MLP(1024, 1280, 2048, 2048, use_residual=False).to(DEVICE).half()` then
    `proj.load_state_dict(torch.load(proj_path, map_location="cpu"))` and `x1, x2 = proj(text_embeddings)` on fp16
    encoder states (tests/test_sdxl_zh.py:92,153,207): fp16 parameters are presented, fp16 outputs come back, the fp32
    master copy is built once (not per call) and follows a later load_state_dict."""
    from oracle.step_ref import AdapterRef
    from pea_diffusion_amd.adapter import PEAAdapter
    torch.manual_seed(5)
    ref = AdapterRef(1024, 1280, 2048, 2048, False)
    proj = PEAAdapter(1024, 1280, 2048, 2048, use_residual=False).to("cuda").half()
    proj.load_state_dict(ref.state_dict())
    assert all(p.dtype == torch.float16 for p in proj.parameters())
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.half().float())                       # what the fp16 module holds
    x = torch.randn(2, 77, 1024)
    with torch.no_grad():
        x1, x2 = proj(x.cuda().half())
        flat_id = proj.flat_param.data_ptr()
        y1, y2 = proj(x.cuda().half())
        r1, r2 = ref(x.half().float())
    assert proj.flat_param.data_ptr() == flat_id, "fp32 master buffer was rebuilt on the second call"
    assert x1.dtype == torch.float16 and x2.dtype == torch.float16 and x1.shape == (2, 1280) and x2.shape == (2, 77, 2048)
    assert torch.equal(x1, y1) and torch.equal(x2, y2)
    assert rel_l2(x1, r1) < 1e-2 and rel_l2(x2, r2) < 1e-2
    sd2 = {k: v * 0.5 for k, v in ref.state_dict().items()}
    proj.load_state_dict(sd2)                               # in-place copy_: the presented parameters' versions move
    with torch.no_grad():
        z1, z2 = proj(x.cuda().half())
    assert not torch.equal(z2, x2)


@pytest.mark.parametrize("args", [(128, 192, 256, 128, False), (128, 128, 64, 192, True), (128, 64, 192, None, False)])
def test_adapter_forward_backward_vs_oracle(gpu, args):
    from oracle.step_ref import AdapterRef
    from pea_diffusion_amd.adapter import PEAAdapter
    torch.manual_seed(3)
    ref = AdapterRef(*args)
    hip = PEAAdapter(*args)
    hip.load_state_dict(ref.state_dict())
    hip = hip.cuda()
    round_weights_bf16_(ref)
    B, L = 3, 11
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, L, args[0], generator=g).to(torch.bfloat16).float()
    out_r = ref(x)
    out_h = hip(x.cuda())
    outs_r = out_r if isinstance(out_r, tuple) else (out_r,)
    outs_h = out_h if isinstance(out_h, tuple) else (out_h,)
    gs = [torch.randn(o.shape, generator=g) for o in outs_r]
    for i, (a, b) in enumerate(zip(outs_h, outs_r)):
        e = rel_l2(a, b)
        print(f"[adapter {args}] out{i} rel_l2={e:.3e}")
        assert e < 1e-2
    torch.autograd.backward(outs_r, gs)
    torch.autograd.backward(outs_h, [t.cuda() for t in gs])
    for (k, p), (_, q) in zip(hip.named_parameters(), ref.named_parameters()):
        e = rel_l2(p.grad, q.grad)
        print(f"   grad {k}: rel_l2={e:.3e}")
        assert e < 2e-2, k


def _train_pair(B, L, seed=0):
    from oracle.step_ref import AdapterRef, synthetic_batch
    from oracle.unet_ref import UNet2DConditionRef, tiny_config
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    cfg = tiny_config()
    torch.manual_seed(seed)
    us, ut = UNet2DConditionRef(cfg), UNet2DConditionRef(cfg)
    round_weights_bf16_(us)
    round_weights_bf16_(ut)
    for p in list(us.parameters()) + list(ut.parameters()):
        p.requires_grad_(False)
    ad_ref = AdapterRef(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False)
    ad_hip = PEAAdapter(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False)
    ad_hip.load_state_dict(ad_ref.state_dict())
    ad_hip = ad_hip.cuda()
    round_weights_bf16_(ad_ref)
    hs = HipUNet(pc.tiny_config(), B, 16, 16, L, needs_grad=True)
    ht = HipUNet(pc.tiny_config(), B, 16, 16, 77, needs_grad=False)
    hs.load_state_dict(us.state_dict())
    ht.load_state_dict(ut.state_dict())
    batch = synthetic_batch(cfg, B, L=L, enc_dim=128, seed=seed)
    return cfg, us, ut, ad_ref, ad_hip, hs, ht, batch, PEATrainer(ad_hip, hs, ht)


@pytest.mark.parametrize("B,L", [(4, 12), (2, 77)])
def test_training_step_vs_oracle(gpu, B, L):
    from oracle.step_ref import training_step_ref
    from oracle.unet_ref import cast_hook_ref
    cfg, us, ut, ad_ref, ad_hip, hs, ht, batch, tr = _train_pair(B, L)
    bq = dict(batch)
    for k in ("enc", "enc_uncond", "teacher_ehs", "teacher_neg", "teacher_pooled"):
        bq[k] = batch[k].to(torch.bfloat16).float()
    store, handles = layer_grad_hooks(us)
    out_r = training_step_ref(ad_ref, us, ut, bq, cast_hook_ref)
    out_r["loss"].backward()
    out_h = tr.training_step(batch, 0, sync=True)
    for k in tr.LOG_KEYS:
        r, h = float(out_r[k]), float(out_h[k])
        print(f"[train step B{B} L{L}] {k}: hip={h:.6f} oracle={r:.6f}")
        assert abs(h - r) <= 1e-2 * max(abs(r), 1e-3), k
    print(f"   eps_student rel_l2={rel_l2(tr.export('eps_student'), out_r['noise_pred']):.3e} "
          f"eps_teacher rel_l2={rel_l2(tr.export('eps_teacher'), out_r['noise_pred_teacher']):.3e}")
    for (k, p), (_, q) in zip(ad_hip.named_parameters(), ad_ref.named_parameters()):
        e = rel_l2(p.grad, q.grad)
        print(f"   adapter grad {k}: rel_l2={e:.3e} |ref|={q.grad.norm():.3e}")
        assert e < 4e-2, k
    check_layer_grads(tr, store, B, 4e-2, f"B{B} L{L} vs fp32 oracle")
    # the adaptive tolerance: what bf16 storage alone does to each of these quantities (the same oracle in bf16-storage mode) is
    # the noise floor; the HIP path may deviate from the fp32 oracle by STORAGE_FLOOR_FACTOR x that, no more
    from oracle.bf16_store import bf16_storage
    ref32 = (out_r["noise_pred"].detach().clone(), out_r["noise_pred_teacher"].detach().clone(),
             torch.cat([p.grad.reshape(-1) for p in ad_ref.parameters()]).clone(), dict(store))
    hip_layers = hip_layer_grads(tr)
    hip_eps = (tr.export("eps_student").cpu(), tr.export("eps_teacher").cpu())
    g_hip = ad_hip.flat_grad.float().cpu().clone()
    store.clear()
    ad_ref.zero_grad()
    with bf16_storage():
        out_q = training_step_ref(ad_ref, us, ut, bq, cast_hook_ref)
        out_q["loss"].backward()
    for h in handles:
        h.remove()
    st16 = (out_q["noise_pred"].detach(), out_q["noise_pred_teacher"].detach(),
            torch.cat([p.grad.reshape(-1) for p in ad_ref.parameters()]), dict(store))
    check_against_storage_floor(f"tiny B{B} L{L}", B, hip_eps, g_hip, hip_layers, ref32, st16)


def test_training_step_repeatable_and_optimizer(gpu):
    cfg, us, ut, ad_ref, ad_hip, hs, ht, batch, tr = _train_pair(2, 12)
    a = tr.training_step(batch, 0, sync=True)
    g1 = ad_hip.flat_grad.clone()
    l1 = tr.losses.clone()
    b = tr.training_step(batch, 0, sync=True)
    # every reduction has a fixed order (no atomics anywhere on the path): bit-reproducible
    assert torch.equal(l1, tr.losses) and torch.equal(g1, ad_hip.flat_grad), "training step is not bit-reproducible"
    w0 = ad_hip.flat_param.clone()
    tr.lr, tr.warmup_steps = 1e-3, 0     # (with warm-up the first update runs at lr = lambda(0) = 0, as in the reference)
    tr.optimizer_step()
    assert not torch.equal(w0, ad_hip.flat_param)
    c = tr.training_step(batch, 0, sync=True)
    assert float(c["loss"]) != float(a["loss"])   # bf16 working copies were refreshed after the update


def test_training_trajectory_vs_oracle(gpu):
    """K consecutive steps of the reference's loop -- step -> AdamW -> step (train_sdxl_zh.py:449 under Lightning,
    utils/model_utils.py:45-81: FusedAdam in AdamW mode + polynomial schedule) -- on the HIP path against the oracle step +
    torch.optim.AdamW: what one-step tests cannot see (stale bf16 working copies of the weights after an update, optimizer state
    handed on wrongly, a schedule off by one) shows as a drifting loss curve or a different end point.  lr 1e-3, no warm-up,
    four batches in rotation; the oracle keeps fp32 master weights and runs on bf16-rounded working copies, as the product does."""
    import copy
    from oracle.step_ref import AdapterRef, synthetic_batch, training_step_ref
    from oracle.unet_ref import cast_hook_ref
    from pea_diffusion_amd.train import polynomial_lr
    K, B, L = 20, 2, 12
    cfg, us, ut, _, ad_hip, hs, ht, _, tr = _train_pair(B, L)
    torch.manual_seed(11)
    master = AdapterRef(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False)
    ad_hip.load_state_dict(master.state_dict())
    ad_hip.mark_updated()
    tr.lr, tr.warmup_steps, tr.total_steps, tr.lr_end = 1e-3, 0, 1000, 0.0
    opt = torch.optim.AdamW(master.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    w0 = torch.cat([p.detach().reshape(-1) for p in master.parameters()]).clone()
    assert torch.equal(ad_hip.flat_param.float().cpu(), w0)
    batches = [synthetic_batch(cfg, B, L=L, enc_dim=128, seed=40 + j) for j in range(4)]
    lh, lr_ = [], []
    for i in range(K):
        batch = batches[i % 4]
        out = tr.training_step(batch, i, sync=True)
        tr.optimizer_step()
        lh.append(float(out["loss"]))
        work = copy.deepcopy(master)                       # bf16 working copy of the fp32 master weights
        round_weights_bf16_(work)
        bq = dict(batch)
        for k in ("enc", "enc_uncond", "teacher_ehs", "teacher_neg", "teacher_pooled"):
            bq[k] = batch[k].to(torch.bfloat16).float()
        ref = training_step_ref(work, us, ut, bq, cast_hook_ref)
        ref["loss"].backward()
        for pm, pw in zip(master.parameters(), work.parameters()):
            pm.grad = pw.grad.detach().clone()
        for g in opt.param_groups:                          # optimizer step k (1-indexed) runs with the schedule's lambda(k - 1)
            g["lr"] = polynomial_lr(i, 1e-3, 0, 1000, 0.0)
        opt.step()
        lr_.append(float(ref["loss"]))
    torch.cuda.synchronize()
    wK = torch.cat([p.detach().reshape(-1) for p in master.parameters()])
    hK = ad_hip.flat_param.float().cpu()
    e_w, e_d = rel_l2(hK, wK), rel_l2(hK - w0, wK - w0)
    worst = max(abs(a - b) / abs(b) for a, b in zip(lh, lr_))
    print(f"[trajectory {K} steps] loss hip {lh[0]:.5f} -> {lh[-1]:.5f}, oracle {lr_[0]:.5f} -> {lr_[-1]:.5f}; worst per-step "
          f"loss deviation {worst:.2e}; parameters rel_l2 {e_w:.2e}, accumulated update rel_l2 {e_d:.2e} "
          f"(|update| / |w| = {float((wK - w0).norm() / w0.norm()):.2e})")
    assert worst < TRAJ_LOSS_LIM, (lh, lr_)
    assert e_w < TRAJ_W_LIM and e_d < TRAJ_D_LIM, (e_w, e_d)
    for j in range(4):                                      # the loss of each batch falls between its first and its last visit
        first, last = j, j + 4 * ((K - 1 - j) // 4)
        assert lr_[last] < lr_[first], (j, lr_)             # (what the oracle does ...)
        assert lh[last] < lh[first], (j, lh)                # (... the HIP path does as well)


def test_checkpoint_resume_is_bit_exact(gpu, tmp_path):
    """Optimizer-state checkpoint + resume (train_sdxl_zh.py:443-448 writes `proj_{step}/pytorch_model.bin`; Lightning +
    DeepSpeed keep global_step / global_samples and the Adam moments in their own checkpoint, `on_load_checkpoint` :454-458,
    utils/universal.py:24-26): 10 steps, save, 10 more steps -- against a FRESH adapter (other initial weights) + fresh trainer
    that resumes from the directory and runs the same 10 batches: every loss, the parameters and both moments bit for bit."""
    import os
    from oracle.step_ref import synthetic_batch
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    B, L = 2, 12
    cfg, us, ut, _, ad, hs, ht, _, tr = _train_pair(B, L)
    hp = dict(lr=1e-3, warmup_steps=4, total_steps=500, lr_end=1e-6)
    for k, v in hp.items():
        setattr(tr, k, v)
    batches = [synthetic_batch(cfg, B, L=L, enc_dim=128, seed=70 + j) for j in range(4)]

    def run(trainer, lo, hi):
        out = []
        for i in range(lo, hi):
            o = trainer.training_step(batches[i % 4], i)
            trainer.optimizer_step()
            out.append(o["loss"])
        torch.cuda.synchronize()
        return torch.stack(out).cpu()

    run(tr, 0, 10)
    d = tr.save_adapter(str(tmp_path))
    assert os.path.basename(d) == "proj_10" and sorted(os.listdir(d)) == ["pytorch_model.bin", "trainer_state.pt"]
    la = run(tr, 10, 20)
    wa, ma, va = ad.flat_param.clone(), tr._m.clone(), tr._v.clone()
    assert tr.global_step == 20 and tr.consumed_samples == 20 * B
    del tr
    torch.manual_seed(999)
    ad2 = PEAAdapter(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False).cuda()     # other weights: resume must replace them
    tr2 = PEATrainer(ad2, hs, ht)                                                         # default hyper-parameters: resume restores them
    assert tr2.resume(str(tmp_path)) == d
    assert tr2.global_step == 10 and tr2.consumed_samples == 10 * B and tr2.lr == hp["lr"] and tr2.warmup_steps == 4
    assert abs(tr2.current_lr() - tr2.lr_end - (hp["lr"] - hp["lr_end"]) * (1 - 6 / 496)) < 1e-12
    lb = run(tr2, 10, 20)
    assert torch.equal(la, lb), (la, lb)
    assert torch.equal(wa, ad2.flat_param) and torch.equal(ma, tr2._m) and torch.equal(va, tr2._v)
    # the presented parameters (what `proj.state_dict()` hands to torch.save) follow the restored masters
    sd = torch.load(os.path.join(d, "pytorch_model.bin"))
    tr3 = PEATrainer(PEAAdapter(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False).cuda(), hs, ht)
    os.remove(os.path.join(d, "trainer_state.pt"))                   # a reference-style directory: weights only
    tr3.resume(str(tmp_path), 10)
    assert tr3.global_step == 10 and float(tr3._m.abs().max()) == 0.0
    for k, v in tr3.adapter.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k
    with pytest.raises(Exception):
        tr3.load_state_dict(dict(tr2.state_dict(), version=99))


def test_adamw_and_checkpoint_round_trip(gpu, tmp_path):
    """fused AdamW over the flat adapter buffer against torch.optim.AdamW (the reference's
    FusedAdam adam_w_mode, utils/model_utils.py:59-67), and the checkpoint writer of train_sdxl_zh.py:443-448:
    `proj_{step}/pytorch_model.bin`, torch-loadable, reference state-dict keys, loads into the reference-shaped MLP."""
    import copy, os
    from oracle.step_ref import AdapterRef
    cfg, us, ut, ad_ref, ad_hip, hs, ht, batch, tr = _train_pair(2, 12)
    tr.training_step(batch, 0, sync=True)
    w = ad_hip.flat_param.detach().clone().cpu().requires_grad_(True)
    opt = torch.optim.AdamW([w], lr=1e-3, betas=tr.betas, eps=tr.eps, weight_decay=tr.weight_decay)
    tr.lr, tr.warmup_steps, tr.lr_end = 1e-3, 0, 1e-3
    for _ in range(3):
        w.grad = ad_hip.flat_grad.detach().clone().cpu()
        opt.step()
        tr.optimizer_step()
    assert torch.allclose(ad_hip.flat_param.detach().cpu(), w.detach(), rtol=1e-5, atol=1e-7)
    d = tr.save_adapter(str(tmp_path))
    assert os.path.basename(d) == "proj_3"
    sd = torch.load(os.path.join(d, "pytorch_model.bin"), map_location="cpu")
    assert list(sd) == list(ad_ref.state_dict())
    fresh = copy.deepcopy(ad_ref)
    fresh.load_state_dict(sd, strict=True)
    for k, v in ad_hip.state_dict().items():
        assert torch.equal(sd[k], v.detach().cpu()), k


@pytest.mark.parametrize("L,Lt", [(77, 77), (52, 77)])
def test_merged_passes_match_two_streams_and_oracle(gpu, L, Lt):
    """(L, Lt) = (52, 77) is the reference's actual default: Chinese-CLIP emits 52 tokens
    (utils/custom_dataset_sdxl.py:352-353) while the teacher's CLIP towers emit 77 -- the merged context is then 77 tokens
    long, the student rows are zero padded and a per-sample key count masks the padding in the cross-attention kernels.
    Reference default: the teacher IS the student checkpoint (train_sdxl_zh.py:138,151 load the same model_path).  The
    trainer then runs both forwards as ONE pass over 2B samples and differentiates the first B; it must agree with the
    two-stream path and with the oracle."""
    import copy
    from oracle.step_ref import AdapterRef, synthetic_batch, training_step_ref
    from oracle.unet_ref import UNet2DConditionRef, cast_hook_ref, tiny_config
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    B = 4
    cfg = tiny_config()
    torch.manual_seed(0)
    us = UNet2DConditionRef(cfg)
    round_weights_bf16_(us)
    for p in us.parameters():
        p.requires_grad_(False)
    ut = copy.deepcopy(us)
    ad_ref = AdapterRef(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False)
    ad_hip = PEAAdapter(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False)
    ad_hip.load_state_dict(ad_ref.state_dict())
    ad_hip = ad_hip.cuda()
    round_weights_bf16_(ad_ref)
    hs = HipUNet(pc.tiny_config(), B, 16, 16, L, needs_grad=True)
    hs.load_state_dict(us.state_dict())
    ht = HipUNet(pc.tiny_config(), B, 16, 16, Lt, needs_grad=False, share_weights_from=hs)
    batch = synthetic_batch(cfg, B, L=L, enc_dim=128, seed=3)
    tr = PEATrainer(ad_hip, hs, ht)
    tr.set_option("merge_passes", 0)
    two = tr.training_step(batch, 0, sync=True)
    g_two = ad_hip.flat_grad.clone()
    e_two = (tr.export("x_t").clone(), tr.export("eps_student").clone(), tr.export("eps_teacher").clone())
    tr.set_option("merge_passes", 1)
    one = tr.training_step(batch, 0, sync=True)
    g_one = ad_hip.flat_grad.clone()
    assert lib_merge_state(tr) == 1, "teacher shares the student's weights: the merged pass must be taken"
    for k in tr.LOG_KEYS:
        print(f"[merged passes] {k}: merged={float(one[k]):.6f} two-stream={float(two[k]):.6f}")
        assert abs(float(one[k]) - float(two[k])) <= 1e-3 * max(abs(float(two[k])), 1e-3), k
    e = rel_l2(g_one, g_two)
    print(f"   adapter grad merged vs two-stream rel_l2={e:.3e}")
    assert e < 5e-3
    for a, b in zip((tr.export("x_t"), tr.export("eps_student"), tr.export("eps_teacher")), e_two):
        assert rel_l2(a, b) < 5e-3
    bq = dict(batch)
    for k in ("enc", "enc_uncond", "teacher_ehs", "teacher_neg", "teacher_pooled"):
        bq[k] = batch[k].to(torch.bfloat16).float()
    out_r = training_step_ref(ad_ref, us, ut, bq, cast_hook_ref)
    out_r["loss"].backward()
    for k in tr.LOG_KEYS:
        # with identical weights the feature / logit losses are differences of nearly equal bf16 tensors: 5 % there
        tol = 5e-2 if k in ("train_loss_features", "train_loss_logits") else 1e-2
        assert abs(float(one[k]) - float(out_r[k])) <= tol * max(abs(float(out_r[k])), 1e-3), k
    for (k, p), (_, q) in zip(ad_hip.named_parameters(), ad_ref.named_parameters()):
        assert rel_l2(p.grad, q.grad) < 4e-2, k
    again = tr.training_step(batch, 0, sync=True)          # bit-reproducible in the merged form too
    assert torch.equal(g_one, ad_hip.flat_grad) and float(again["loss"]) == float(one["loss"])


def lib_merge_state(tr):
    import ctypes
    from pea_diffusion_amd._lib import lib
    return lib().pea_trainer_get_option(tr._h, b"merge_state")


def test_asymmetric_teacher_student(gpu):
    """BASELINE config 4 shape case (SSD-1B student + SDXL teacher): student and teacher UNets with different
    transformer depths share tap shapes; the KD step must match the oracle."""
    from oracle.step_ref import AdapterRef, synthetic_batch, training_step_ref
    from oracle.unet_ref import UNet2DConditionRef, cast_hook_ref, tiny_config
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    B, L = 2, 12
    cfg_t = tiny_config()
    cfg_s = tiny_config()
    cfg_s.transformer_layers_per_block = (1, 1, 1)          # pruned student
    pcs, pct = pc.tiny_config(), pc.tiny_config()
    pcs.transformer_layers_per_block = (1, 1, 1)
    torch.manual_seed(1)
    us, ut = UNet2DConditionRef(cfg_s), UNet2DConditionRef(cfg_t)
    round_weights_bf16_(us)
    round_weights_bf16_(ut)
    for p in list(us.parameters()) + list(ut.parameters()):
        p.requires_grad_(False)
    ad_ref = AdapterRef(128, cfg_s.pooled_dim, 192, cfg_s.cross_attention_dim, False)
    ad = PEAAdapter(128, cfg_s.pooled_dim, 192, cfg_s.cross_attention_dim, False)
    ad.load_state_dict(ad_ref.state_dict())
    ad = ad.cuda()
    round_weights_bf16_(ad_ref)
    hs = HipUNet(pcs, B, 16, 16, L, needs_grad=True)
    ht = HipUNet(pct, B, 16, 16, 77)
    hs.load_state_dict(us.state_dict())
    ht.load_state_dict(ut.state_dict())
    assert hs.memory()["n_ops"] < ht.memory()["n_ops"]
    batch = synthetic_batch(cfg_s, B, L=L, enc_dim=128, seed=2)
    tr = PEATrainer(ad, hs, ht)
    out = tr.training_step(batch, 0, sync=True)
    bq = dict(batch)
    for k in ("enc", "enc_uncond", "teacher_ehs", "teacher_neg", "teacher_pooled"):
        bq[k] = batch[k].to(torch.bfloat16).float()
    ref = training_step_ref(ad_ref, us, ut, bq, cast_hook_ref)
    ref["loss"].backward()
    for k in tr.LOG_KEYS:
        assert abs(float(out[k]) - float(ref[k])) <= 1e-2 * max(abs(float(ref[k])), 1e-3), k
    g_ref = torch.cat([p.grad.reshape(-1) for p in ad_ref.parameters()])
    assert rel_l2(ad.flat_grad, g_ref) < 4e-2


@pytest.mark.parametrize("model,B,hw", [("sdxl", 1, 128)])
def test_full_size_properties(gpu, model, B, hw):
    """BASELINE.json full size (SDXL, 1024x1024 = latent 128x128; the oracle cannot finish this in seconds), checked
    through size-independent properties:
      * idempotence: a teacher that shares the student's weights and receives the student's own conditioning must
        reproduce the student bit for bit -> train_loss_logits == 0, train_loss_features == 0, loss == train_loss;
      * masks: zh_or_not = 1 for every sample -> only the noise term survives and it equals mean((eps_s - eps)^2);
      * the step is bit-reproducible and every adapter gradient is finite and non-zero."""
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    cfg = pc.sdxl_config()
    L = 77
    student = HipUNet(cfg, B, hw, hw, L, needs_grad=True)
    student.init_random(3)
    teacher = HipUNet(cfg, B, hw, hw, L, share_weights_from=student)
    torch.manual_seed(0)
    ad = PEAAdapter(1024, 1280, 1024, 2048, False).cuda()
    tr = PEATrainer(ad, student, teacher)
    g = torch.Generator(device="cuda").manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g, device="cuda")
    enc = r(B, L, 1024)
    with torch.no_grad():
        pooled, tokens = ad(enc)                                     # the student's own conditioning
    base = dict(latents=r(B, 4, hw, hw), noise=r(B, 4, hw, hw), timesteps=torch.tensor([500] * B, device="cuda"),
                enc=enc, enc_uncond=r(B, L, 1024), prompt_mask=torch.zeros(B, dtype=torch.uint8, device="cuda"),
                teacher_ehs=tokens.float(), teacher_neg=r(B, L, 2048), teacher_pooled=pooled.float(),
                time_ids=torch.tensor([[1024., 1024, 0, 0, 1024, 1024]] * B, device="cuda"))
    b0 = dict(base, zh_or_not=torch.zeros(B, dtype=torch.int64, device="cuda"))
    out = tr.training_step(b0, 0, sync=True)
    assert float(out["train_loss_logits"]) == 0.0 and float(out["train_loss_features"]) == 0.0
    assert float(out["train_loss"]) == 0.0 and float(out["loss"]) == 0.0
    assert torch.equal(tr.export("eps_student"), tr.export("eps_teacher"))
    b1 = dict(base, zh_or_not=torch.ones(B, dtype=torch.int64, device="cuda"))
    out1 = tr.training_step(b1, 0, sync=True)
    eps_s = tr.export("eps_student")
    want = ((eps_s - base["noise"]) ** 2).mean().item()
    assert abs(float(out1["train_loss"]) - want) <= 1e-4 * want and float(out1["loss"]) == float(out1["train_loss"])
    g1 = ad.flat_grad.clone()
    assert torch.isfinite(g1).all() and (g1 != 0).float().mean() > 0.9
    out2 = tr.training_step(b1, 0, sync=True)
    assert torch.equal(g1, ad.flat_grad) and float(out2["loss"]) == float(out1["loss"])


def test_adapter_full_dims_backward_and_reprepare(gpu):
    """the 6M adapter at its real dimensions vs the oracle, and a trainer-independent proj(x) call with another
    batch shape in between must not leave stale bf16 weight copies behind"""
    from oracle.step_ref import AdapterRef
    from pea_diffusion_amd.adapter import PEAAdapter
    torch.manual_seed(0)
    ref = AdapterRef(1024, 1280, 1024, 2048, False)
    hip = PEAAdapter(1024, 1280, 1024, 2048, False)
    hip.load_state_dict(ref.state_dict())
    hip = hip.cuda()
    x = torch.randn(2, 77, 1024)
    with torch.no_grad():
        hip(torch.randn(1, 52, 1024).cuda())          # different (batch, L): re-prepares the C context
    pr, tk = ref(x)
    ph, th = hip(x.cuda())
    assert rel_l2(ph, pr) < 1e-2 and rel_l2(th, tk) < 1e-2
    g1, g2 = torch.randn_like(pr), torch.randn_like(tk)
    torch.autograd.backward([pr, tk], [g1, g2])
    torch.autograd.backward([ph, th], [g1.cuda(), g2.cuda()])
    for (k, p), (_, q) in zip(hip.named_parameters(), ref.named_parameters()):
        assert rel_l2(p.grad, q.grad) < 2e-2, k


def _sd15_step_check(cfg_name, B, L, hw, enc_dim, hidden, tol_fwd, tol_grad):
    """SD1.5-shaped KD step (train_sd_zh.py:184-281: adapter returns tokens only, no added conditioning, 9 taps,
    NaN/Inf guard) on the HIP path vs the CPU oracle"""
    from oracle import unet_ref as ou
    from oracle.step_ref import AdapterRef, synthetic_batch, training_step_ref
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    cfg = getattr(ou, cfg_name)()
    torch.manual_seed(4)
    us = ou.UNet2DConditionRef(cfg)
    round_weights_bf16_(us)
    for p in us.parameters():
        p.requires_grad_(False)
    import copy
    ut = copy.deepcopy(us)       # teacher = same checkpoint, but a separate module (separate forward hooks)
    ad_ref = AdapterRef(enc_dim, cfg.cross_attention_dim, hidden, None)
    ad = PEAAdapter(enc_dim, cfg.cross_attention_dim, hidden, None, False)
    ad.load_state_dict(ad_ref.state_dict())
    ad = ad.cuda()
    round_weights_bf16_(ad_ref)
    hs = HipUNet(getattr(pc, cfg_name)(), B, hw, hw, L, needs_grad=True)
    hs.load_state_dict(us.state_dict())
    ht = HipUNet(getattr(pc, cfg_name)(), B, hw, hw, 77, share_weights_from=hs)   # teacher = same checkpoint
    batch = synthetic_batch(cfg, B, L=L, enc_dim=enc_dim, seed=1, latent_hw=hw)
    tr = PEATrainer(ad, hs, ht, nan_guard=True)
    out = tr.training_step(batch, 0, sync=True)
    bq = dict(batch)
    for k in ("enc", "enc_uncond", "teacher_ehs", "teacher_neg"):
        bq[k] = batch[k].to(torch.bfloat16).float()
    store, handles = layer_grad_hooks(us)
    ref = training_step_ref(ad_ref, us, ut, bq, ou.cast_hook_ref, nan_guard=True)
    ref["loss"].backward()
    # every cross-attention K / V projection on its own (SD1.5 has no text_time conditioning: no gradient reaches the time
    # embedding, the oracle holds none for time_emb_proj and the HIP path computes none), held to the bf16-storage noise floor
    # of that layer: the deepest levels are 8 x 8 tokens (tiny15: 2 x 2), where a layer's dK / dV is a sum over few queries
    # and a fixed limit would be wrong by an order of magnitude either way (measured: 2.7e-2 on mid_block ... to_v at full size)
    assert store and all(not k.endswith("time_emb_proj.weight") for k in store)
    from oracle.bf16_store import bf16_storage
    ref32 = (ref["noise_pred"].detach().clone(), ref["noise_pred_teacher"].detach().clone(),
             torch.cat([p.grad.reshape(-1) for p in ad_ref.parameters()]).clone(), dict(store))
    hip_layers = {k: _unpad_heads(v, store[k].shape[-1]) for k, v in hip_layer_grads(tr).items() if k in store}
    assert sorted(hip_layers) == sorted(store)
    hip_eps = (tr.export("eps_student").cpu(), tr.export("eps_teacher").cpu())
    g_hip = ad.flat_grad.float().cpu().clone()
    store.clear()
    ad_ref.zero_grad()
    with bf16_storage():
        rq = training_step_ref(ad_ref, us, ut, bq, ou.cast_hook_ref, nan_guard=True)
        rq["loss"].backward()
    for h in handles:
        h.remove()
    st16 = (rq["noise_pred"].detach(), rq["noise_pred_teacher"].detach(),
            torch.cat([p.grad.reshape(-1) for p in ad_ref.parameters()]), dict(store))
    check_against_storage_floor(f"{cfg_name}", B, hip_eps, g_hip, hip_layers, ref32, st16,
                                factor=2.0 if cfg_name.startswith("tiny") else None)
    assert len(ref["taps_s"]) == 2 * len(cfg.block_out_channels) + 1 == hs.num_taps
    e = rel_l2(tr.export("eps_student"), ref["noise_pred"])
    print(f"[{cfg_name} step] eps_student rel_l2={e:.3e}")
    assert e < tol_fwd
    total = abs(float(ref["loss"]))
    for k in tr.LOG_KEYS:
        h, r = float(out[k]), float(ref[k])
        print(f"   {k}: hip={h:.6f} oracle={r:.6f}")
        # teacher == student checkpoint: the KD terms are small differences of nearly equal bf16 tensors, so they
        # carry an absolute noise floor; 2 % relative + 0.5 % of the total loss
        assert abs(h - r) <= 2e-2 * abs(r) + 5e-3 * total, k
    g_ref = ref32[2]                                # the fp32 oracle's flat adapter gradient
    eg = rel_l2(ad.flat_grad, g_ref)
    print(f"   adapter grad rel_l2={eg:.3e}")
    assert eg < tol_grad


def test_sd15_shaped_step_tiny(gpu):
    _sd15_step_check("tiny15_config", B=2, L=12, hw=16, enc_dim=128, hidden=192, tol_fwd=2e-2, tol_grad=4e-2)


def test_sd15_full_size_step_vs_oracle(gpu):
    """BASELINE configs[0]: SD1.5 512x512 (batch 2 so that both mask values occur) -- full 859.5 M-parameter UNet, the whole KD step on the MI355X
    against the fp32 CPU oracle (train_sd_zh.py path; head dims 40/80/160)"""
    _sd15_step_check("sd15_config", B=2, L=77, hw=64, enc_dim=1024, hidden=2048, tol_fwd=2e-2, tol_grad=6e-3)   # measured grad 2.1e-3


@pytest.mark.parametrize("tag", ["sdxl_hip_mixed", "sdxl_hip_all_en", "sdxl_hip_all_zh", "sdxl_hip_shared_teacher",
                                 "sd15_hip_mixed"])
def test_step_matches_reference_golden(gpu, golden_dir, tag):
    """The reference's own step COMPOSITION against the HIP path: tests/golden/step_*_hip_*.npz hold what the reference's
    `StableDiffusion.training_step` (train_sdxl_zh.py:305-449, train_sd_zh.py:184-281; run by oracle/make_golden.py at dims
    the MFMA tiles accept) produced -- the four logged scalars, eps of both UNets and the adapter's parameter gradients
    (seven for SDXL, five for SD1.5) for forced CFG-dropout masks.  What is the reference's own code in that run: the MLP
    adapter, add_noise, the CFG-dropout `where`, cast_hook's taps, the three masked MSE groups and autograd's routing back to
    the adapter.  What is NOT: the two UNets -- diffusers is absent from this pool, so the collaborators inside that run are
    `oracle.unet_ref.UNet2DConditionRef` (oracle/make_golden.py:225-226), and `noise_pred` / `noise_pred_teacher` in these
    fixtures are therefore oracle outputs (the UNet arithmetic itself stays "parity unpinned").
    PEATrainer gets the same weights (UNets regenerated from the fixture's seed, checksum-checked; MLP weights stored) and
    the same batch.  `shared_teacher`: teacher == student checkpoint as the reference loads it (:138,151) -> the merged-pass
    launch set of bench.py."""
    import os
    from oracle import unet_ref as ou      # weights only: the fixture stores the UNets as seed + checksum
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    g = np.load(os.path.join(golden_dir, f"step_{tag}.npz"))
    T = lambda a: torch.from_numpy(np.asarray(a))
    sdxl = tag.startswith("sdxl")
    ocfg, pcfg = (ou.tiny_config(), pc.tiny_config()) if sdxl else (ou.tiny15_config(), pc.tiny15_config())
    torch.manual_seed(int(g["seed_model"]))
    us, ut = ou.UNet2DConditionRef(ocfg), ou.UNet2DConditionRef(ocfg)
    shared = bool(int(g["shared_teacher"]))
    if shared:
        ut.load_state_dict(us.state_dict())
    round_weights_bf16_(us), round_weights_bf16_(ut)
    wsum = sum(float(v.double().abs().sum()) for m in (us, ut) for v in m.state_dict().values())
    assert abs(wsum - float(g["wsum_unets"])) < 1e-6 * float(g["wsum_unets"]), "seeded UNet weights drifted"
    B, L, hw = int(g["B"]), int(g["L"]), ocfg.sample_size
    hs = HipUNet(pcfg, B, hw, hw, L, needs_grad=True)
    hs.load_state_dict(us.state_dict())
    if shared:
        ht = HipUNet(pcfg, B, hw, hw, 77, share_weights_from=hs)
    else:
        ht = HipUNet(pcfg, B, hw, hw, 77)
        ht.load_state_dict(ut.state_dict())
    a = [int(v) for v in g["mlp_args"]]
    ad = PEAAdapter(a[0], a[1], a[2], a[3], bool(a[4])) if sdxl else PEAAdapter(a[0], a[1], a[2], None, False)
    ad.load_state_dict({k[2:]: T(g[k]) for k in g.files if k.startswith("w.")})     # the reference MLP's own keys
    ad = ad.cuda()
    keys = ["latents", "noise", "timesteps", "enc", "enc_uncond", "prompt_mask", "zh_or_not", "teacher_ehs", "teacher_neg"]
    if sdxl:
        keys += ["teacher_pooled", "time_ids"]
    batch = {k: T(g[k]) for k in keys}
    tr = PEATrainer(ad, hs, ht, nan_guard=not sdxl)
    out = tr.training_step(batch, 0, sync=True)
    if shared:
        assert lib_merge_state(tr) == 1
    e_s = rel_l2(tr.export("eps_student"), T(g["noise_pred"]))
    e_t = rel_l2(tr.export("eps_teacher"), T(g["noise_pred_teacher"]))
    assert e_s < 1.5e-2 and e_t < 1.5e-2, (e_s, e_t)
    total = abs(float(g["loss"]))
    for k in tr.LOG_KEYS:
        h, r = float(out[k]), float(g[k])
        # 1 % relative; the KD terms of a shared-checkpoint pair are differences of nearly equal bf16 tensors (absolute
        # noise floor: 0.5 % of the total loss)
        assert abs(h - r) <= 1e-2 * abs(r) + (5e-3 * total if shared else 1e-3 * total), (k, h, r)
    names = [n for n, _ in ad.named_parameters()]
    assert sorted("g." + n for n in names) == sorted(k for k in g.files if k.startswith("g."))
    flat = ad.flat_grad.float().cpu()
    worst = 0.0
    for p, o in zip(ad._plist(), ad._offsets):
        n = [nm for nm, q in ad.named_parameters() if q is p][0]
        want = T(g["g." + n]).float()
        got = flat[o:o + p.numel()].view_as(want)
        if float(want.abs().max()) == 0.0:              # e.g. fc.* when no sample's pooled output reaches a loss
            assert float(got.abs().max()) == 0.0, n
            continue
        e = rel_l2(got, want)
        worst = max(worst, e)
        # (shared checkpoint: the KD gradient is a difference of nearly equal bf16 tensors; measured 3.0e-2 worst / 2.6e-2 flat)
        assert e < (6e-2 if shared else 4e-2), (n, e)
    g_ref = torch.cat([T(g["g." + [nm for nm, q in ad.named_parameters() if q is p][0]]).reshape(-1) for p in ad._plist()])
    e_all = rel_l2(flat, g_ref)
    print(f"[{tag}] eps rel_l2 student {e_s:.2e} teacher {e_t:.2e}; loss hip={float(out['loss']):.6f} ref={float(g['loss']):.6f}; "
          f"adapter grad rel_l2 flat {e_all:.2e}, worst parameter {worst:.2e}")
    assert e_all < (5e-2 if shared else 3e-2)


def _fast_fill_(module, seed=0):
    """deterministic weights without torch's single-threaded default init of 2.57 B parameters: every >= 2-D tensor is
    filled from one seeded uniform buffer read at a per-tensor offset, scaled to variance 1/(3 fan_in); norm weights 1,
    biases small"""
    g = torch.Generator().manual_seed(seed)
    buf = torch.rand(1 << 24, generator=g) - 0.5
    off = 0
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() >= 2:
                fan = p[0].numel()
                flat = p.view(-1)
                for o in range(0, flat.numel(), 1 << 22):
                    k = min(1 << 22, flat.numel() - o)
                    off = (off * 31 + 7919) % (buf.numel() - (1 << 22))
                    flat[o:o + k].copy_(buf[off:off + k])
                flat.mul_(2.0 * fan ** -0.5)
            elif n.endswith("weight"):
                p.fill_(1.0)
            else:
                p.copy_(0.02 * buf[:p.numel()])


def _sdxl_full_model_step_vs_oracle(hw, tag, eps_lim, grad_lim, layer_lim=2e-2, storage_floor=False):
    import copy
    from oracle import unet_ref as ou
    from oracle.step_ref import AdapterRef, synthetic_batch, training_step_ref
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    torch.set_num_threads(min(64, len(__import__("os").sched_getaffinity(0))))
    cfg = ou.sdxl_config()
    B, L = 2, 77
    # build without the default init (torch.nn.init on 2.57 B parameters takes about a minute): allocate, then fill
    orig = torch.nn.init.kaiming_uniform_, torch.nn.init.uniform_
    torch.nn.init.kaiming_uniform_ = lambda t, *a, **k: t
    torch.nn.init.uniform_ = lambda t, *a, **k: t
    try:
        us = ou.UNet2DConditionRef(cfg)
    finally:
        torch.nn.init.kaiming_uniform_, torch.nn.init.uniform_ = orig
    _fast_fill_(us, seed=5)
    round_weights_bf16_(us)
    for p in us.parameters():
        p.requires_grad_(False)
    ut = copy.deepcopy(us)
    torch.manual_seed(6)
    ad_ref = AdapterRef(1024, cfg.pooled_dim, 1024, cfg.cross_attention_dim, False)
    ad = PEAAdapter(1024, cfg.pooled_dim, 1024, cfg.cross_attention_dim, False)
    ad.load_state_dict(ad_ref.state_dict())
    ad = ad.cuda()
    round_weights_bf16_(ad_ref)
    hs = HipUNet(pc.sdxl_config(), B, hw, hw, L, needs_grad=True)
    missing, unexpected = hs.load_state_dict(us.state_dict())
    assert not missing and not unexpected
    ht = HipUNet(pc.sdxl_config(), B, hw, hw, L, share_weights_from=hs)
    batch = synthetic_batch(cfg, B, L=L, enc_dim=1024, seed=2, latent_hw=hw)
    assert sorted(batch["zh_or_not"].tolist()) == [0, 1]          # both mask values occur
    tr = PEATrainer(ad, hs, ht)
    out = tr.training_step(batch, 0, sync=True)
    assert lib_merge_state(tr) == 1
    bq = dict(batch)
    for k in ("enc", "enc_uncond", "teacher_ehs", "teacher_neg", "teacher_pooled"):
        bq[k] = batch[k].to(torch.bfloat16).float()
    import time
    t0 = time.time()
    store, handles = layer_grad_hooks(us)
    ref = training_step_ref(ad_ref, us, ut, bq, ou.cast_hook_ref)
    ref["loss"].backward()
    e_s, e_t = rel_l2(tr.export("eps_student"), ref["noise_pred"]), rel_l2(tr.export("eps_teacher"), ref["noise_pred_teacher"])
    print(f"[sdxl full model {tag} step] eps_student rel_l2={e_s:.3e} eps_teacher rel_l2={e_t:.3e} (oracle {time.time() - t0:.0f} s)")
    assert e_s < eps_lim and e_t < eps_lim
    total = abs(float(ref["loss"]))
    for k in tr.LOG_KEYS:
        h, r = float(out[k]), float(ref[k])
        print(f"   {k}: hip={h:.6f} oracle={r:.6f}")
        assert abs(h - r) <= 2e-2 * abs(r) + 5e-3 * total, k
    g_ref = torch.cat([p.grad.reshape(-1) for p in ad_ref.parameters()])
    eg = rel_l2(ad.flat_grad, g_ref)
    print(f"   adapter grad rel_l2={eg:.3e} |ref|={g_ref.norm():.3e}")
    assert eg < grad_lim
    # all 140 cross-attention K / V projections and all 17 time_emb_proj layers, each on its own
    assert len(store) == 2 * 70 + 17, len(store)
    check_layer_grads(tr, store, B, layer_lim, f"sdxl {tag} vs fp32 oracle")
    if storage_floor:
        # adaptive tolerance (check_against_storage_floor): the same oracle once more in bf16-storage mode gives the noise floor
        from oracle.bf16_store import bf16_storage
        ref32 = (ref["noise_pred"].detach().clone(), ref["noise_pred_teacher"].detach().clone(), g_ref.clone(), dict(store))
        hip_layers = hip_layer_grads(tr)
        hip_eps = (tr.export("eps_student").cpu(), tr.export("eps_teacher").cpu())
        g_hip = ad.flat_grad.float().cpu().clone()
        store.clear()
        ad_ref.zero_grad()
        t0 = time.time()
        with bf16_storage():
            ref = training_step_ref(ad_ref, us, ut, bq, ou.cast_hook_ref)
            ref["loss"].backward()
        st16 = (ref["noise_pred"].detach(), ref["noise_pred_teacher"].detach(),
                torch.cat([p.grad.reshape(-1) for p in ad_ref.parameters()]), dict(store))
        print(f"   (bf16-storage oracle {time.time() - t0:.0f} s)")
        check_against_storage_floor(f"sdxl {tag}", B, hip_eps, g_hip, hip_layers, ref32, st16)
    for h in handles:
        h.remove()


def test_sdxl_full_model_step_vs_oracle_512(gpu):
    """The full 2.57 B-parameter SDXL UNet (BASELINE configs[1] model, 512x512 so the fp32 CPU oracle finishes in about a
    minute; batch 2 so both mask values occur): the whole KD step -- merged passes, since the teacher is the student
    checkpoint -- against the oracle: eps, the four logged scalars, the flat adapter gradient."""
    _sdxl_full_model_step_vs_oracle(64, "512x512", 1.5e-2, 2e-2, storage_floor=True)       # measured 7.0e-3 / 8.1e-3 vs fp32


def test_sdxl_full_model_step_vs_oracle_1024(gpu):
    """The same comparison at the METRIC's resolution (train_sdxl_zh.py:397-441 on 1024x1024 images, latent 128x128):
    the 4096-token x 10-head self-attention, the 128x128 convolutions (M = 32768 rows per sample) and the persistent
    256x160 tile walk meet the fp32 oracle end to end, not only through properties (tests/test_configs_gpu.py)."""
    _sdxl_full_model_step_vs_oracle(128, "1024x1024", 1.5e-2, 2e-2)
