"""Generates tests/golden/*.npz by IMPORTING THE REFERENCE's own code in the authoring
container (it needs /root/reference, which does not exist on the GPU box; the committed
.npz fixtures are what travels).  Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

What runs from the reference (nothing is copied, only executed):
  * `MLP` of train_sdxl_zh.py:43-67, train_sd_zh.py:41-56, tests/test_sdxl_zh.py:59-84
  * `cast_hook`/`getActivation` of train_sdxl_zh.py:69-84
  * `StableDiffusion.training_step` of train_sdxl_zh.py:305-449 and train_sd_zh.py:184-281,
    called unbound on a namespace `self` whose collaborators (UNets, VAE, scheduler, text
    encoders) are toys built from oracle/ (diffusers is not installed).
  * `rescale_noise_cfg` of tests/test_sdxl_zh.py:45-56
Third-party packages absent from the image are replaced by EMPTY stub modules.
Large weights are not stored: fixtures keep the torch seed and a checksum of the weights
the seed must regenerate (tests fail loudly on a checksum mismatch).
"""
import importlib
import importlib.machinery
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
import transformers  # noqa: F401  (real one must be imported before torchvision is stubbed)
from transformers import CLIPTextModel, T5EncoderModel  # noqa: F401  force lazy imports now

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


class _Any:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, k):
        return _Any()

    def __call__(self, *a, **k):
        return _Any()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    m.__getattr__ = lambda k: _Any
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    class LightningModule(torch.nn.Module):
        pass
    _stub("pytorch_lightning", LightningModule=LightningModule)
    _stub("pytorch_lightning.callbacks")
    for n in ["diffusers", "diffusers.image_processor", "diffusers.models", "diffusers.models.attention_processor",
              "diffusers.loaders", "diffusers.utils", "diffusers.utils.torch_utils", "open_clip", "cn_clip",
              "cn_clip.clip", "torchvision", "torchvision.utils", "cv2", "utils", "utils.model_utils",
              "utils.universal", "utils.custom_dataset", "utils.custom_dataset_sdxl", "PIL", "PIL.Image", "tqdm"]:
        if n not in sys.modules or n.startswith("utils") or n in ("torchvision", "torchvision.utils"):
            _stub(n)
    sys.modules["utils.custom_dataset_sdxl"].BUCKETS = [[64, 64]] * 9
    sys.modules["tqdm"].tqdm = lambda x, *a, **k: x


def load_ref(relpath, modname):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def checksum(sd):
    return float(sum(v.double().abs().sum().item() for v in sd.values()))


def np_(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


# --------------------------------------------------------------------------------------
def gen_mlp(ref_sdxl, ref_sd, ref_test):
    cases = {}
    # small-dimension instances: weights stored in the fixture
    for tag, (ctor, args) in {
        "sdxl_small": (ref_sdxl.MLP, (64, 80, 96, 128, False)),
        "sdxl_small_residual": (ref_sdxl.MLP, (64, 64, 96, 128, True)),
        "test_small": (ref_test.MLP, (64, 80, 96, 128, False)),
        "sd_small": (ref_sd.MLP, (64, 48, 96)),
    }.items():
        torch.manual_seed(1234)
        m = ctor(*args)
        x = torch.randn(2, 5, args[0], requires_grad=True)
        out = m(x)
        outs = out if isinstance(out, tuple) else (out,)
        gs = [torch.randn_like(o) for o in outs]
        torch.autograd.backward(outs, gs)
        d = {"args": np.array([a if not isinstance(a, bool) else int(a) for a in args]), "x": x, "dx": x.grad}
        for i, (o, g) in enumerate(zip(outs, gs)):
            d[f"out{i}"] = o
            d[f"gout{i}"] = g
        for k, v in m.state_dict().items():
            d["w." + k] = v
        for k, p in m.named_parameters():
            d["g." + k] = p.grad
        cases[tag] = np_(d)
    # full-size constructors used by the reference (train_sdxl_zh.py:101,107,113,124;
    # tests/test_sdxl_zh.py:92; train_sd_zh.py:96): weights by seed + checksum
    for tag, (ctor, args, L) in {
        "sdxl_6M": (ref_sdxl.MLP, (1024, 1280, 1024, 2048, False), 52),
        "sdxl_11M": (ref_sdxl.MLP, (1024, 1280, 2048, 2048, False), 77),
        "sdxl_in2048": (ref_sdxl.MLP, (2048, 1280, 2048, 2048, False), 77),
        "sdxl_in768": (ref_sdxl.MLP, (768, 1280, 2048, 2048, False), 77),
        "sd15_full": (ref_sd.MLP, (1024, 768, 2048), 77),
    }.items():
        torch.manual_seed(77)
        m = ctor(*args)
        nparam = sum(p.numel() for p in m.parameters())
        x = torch.randn(2, L, args[0])
        out = m(x)                                  # (the same forward as before; gradients recorded since round 4)
        outs = out if isinstance(out, tuple) else (out,)
        d = {"args": np.array([int(a) for a in args]), "seed": 77, "L": L, "nparam": nparam,
             "wsum": checksum(m.state_dict()), "x": x}
        for i, o in enumerate(outs):
            d[f"out{i}"] = o.detach()
        # parameter gradients of the reference MLP for seeded output gradients: norm of every gradient + a strided sample
        # (the full 6-13 M-element gradients are not stored)
        gg = torch.Generator().manual_seed(78)
        gouts = [torch.randn(o.shape, generator=gg) for o in outs]
        torch.autograd.backward(outs, gouts)
        for i, go in enumerate(gouts):
            d[f"gout{i}"] = go
        for k, p_ in m.named_parameters():
            flat = p_.grad.reshape(-1)
            stride = max(1, flat.numel() // 4096)
            d["gnorm." + k] = float(flat.double().norm())
            d["gsample." + k] = flat[::stride].clone()
            d["gstride." + k] = stride
        d["keys"] = np.array(list(m.state_dict().keys()))
        cases[tag] = np_(d)
    for tag, d in cases.items():
        np.savez_compressed(os.path.join(OUT, f"mlp_{tag}.npz"), **d)
        print("mlp", tag, {k: v.shape for k, v in d.items() if k.startswith("out")})


# --------------------------------------------------------------------------------------
class _Dist:
    def __init__(self, x):
        self.x = x

    def sample(self):
        return self.x


class ToyVAE:
    def __init__(self, lat, sf):
        self.lat, self.config = lat, types.SimpleNamespace(scaling_factor=sf)

    def to(self, **k):
        return self

    def encode(self, px):
        return types.SimpleNamespace(latent_dist=_Dist(self.lat))


class ToySched:
    config = types.SimpleNamespace(num_train_timesteps=1000)

    def add_noise(self, x0, n, t):
        from oracle.step_ref import add_noise
        return add_noise(x0.float(), n.float(), t)


class ToyEnc:
    def __init__(self, table):
        self.table = table  # id(input_ids tensor)->embedding by first element

    def encode_text(self, ids):
        e = self.table[int(ids[0, 0])]
        return e, None            # chinese_clip branch unpacks (tokens, pooled) train_sdxl_zh.py:329


def _round_bf16_(module):
    """every >= 2-D weight to a bf16-representable fp32 value (what the device path stores)"""
    with torch.no_grad():
        for p in module.parameters():
            if p.dim() >= 2:
                p.copy_(p.to(torch.bfloat16).float())


def gen_step(ref, which, zh_pattern, mask_pattern, tag, hip=False, shared_teacher=False):
    """hip=True: the SAME reference training_step on collaborators at the dims the device path's MFMA tiles accept
    (oracle tiny_config / tiny15_config: 64/128/128 channels, cross 128, MLP (128, pooled, 192, 128), L = 12, 16x16
    latents), UNet / MLP weights and the text-side inputs bf16-representable, so that `-m gpu` can compare PEATrainer
    with what the reference itself produced (tests/test_model_gpu.py::test_step_matches_reference_golden)."""
    from oracle.unet_ref import UNet2DConditionRef, UNetConfig, tiny_config, tiny15_config, sd15_config
    from oracle.step_ref import AdapterRef
    sdxl = which == "sdxl"
    if hip and sdxl:
        cfg = tiny_config()
        mlp_args = (128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False)
    elif hip:
        cfg = tiny15_config()
        mlp_args = (128, cfg.cross_attention_dim, 192)
    elif sdxl:
        cfg = UNetConfig(sample_size=8, block_out_channels=(32, 64, 64), transformer_layers_per_block=(1, 1, 1),
                         num_attention_heads=(1, 2, 2), cross_attention_dim=64, addition_time_embed_dim=16,
                         projection_class_embeddings_input_dim=48 + 96, name="toy")
        mlp_args = (32, 48, 40, 64, False)
    else:
        b = sd15_config()
        cfg = UNetConfig(sample_size=16, block_out_channels=(32, 32, 64, 64), down_block_types=b.down_block_types,
                         up_block_types=b.up_block_types, transformer_layers_per_block=(1, 1, 1, 1),
                         num_attention_heads=(2, 2, 2, 2), cross_attention_dim=48, use_linear_projection=False,
                         addition_embed_type=None, addition_time_embed_dim=0,
                         projection_class_embeddings_input_dim=0, name="toy15")
        mlp_args = (32, 48, 40)
    B, L, hw = len(zh_pattern), (12 if hip else 6), cfg.sample_size
    torch.manual_seed(2024)
    unet_s = UNet2DConditionRef(cfg)
    unet_t = UNet2DConditionRef(cfg)          # reference loads teacher from the same path (:151); here a
    if shared_teacher:                        # different init to make KD terms non-zero -- unless shared_teacher: the
        unet_t.load_state_dict(unet_s.state_dict())     # reference's own set-up (merged passes on the device path)
    proj = ref.MLP(*mlp_args)
    if hip:
        for m in (unet_s, unet_t, proj):
            _round_bf16_(m)
    wsum = checksum(unet_s.state_dict()) + checksum(unet_t.state_dict())
    g = torch.Generator().manual_seed(5)
    q = (lambda x: x.to(torch.bfloat16).float()) if hip else (lambda x: x)
    lat16 = (torch.randn(B, 4, hw, hw, generator=g)).half()
    enc = q(torch.randn(B, L, mlp_args[0], generator=g))
    enc_u = q(torch.randn(1, L, mlp_args[0], generator=g)).repeat(B, 1, 1)
    t_ehs = q(torch.randn(B, 77, cfg.cross_attention_dim, generator=g))
    t_neg = q(torch.randn(1, 77, cfg.cross_attention_dim, generator=g)).repeat(B, 1, 1)
    t_pool = q(torch.randn(B, cfg.pooled_dim if (hip and sdxl) else 48, generator=g))
    sf = 0.5
    self = types.SimpleNamespace()
    self.vae = ToyVAE(lat16.float(), sf)
    self.unet, self.unet_teacher = unet_s, unet_t
    self.noise_scheduler = ToySched()
    self.text_encoder = ToyEnc({1: enc, 2: enc_u})
    self.proj = proj
    self.KD_student, self.KD_teacher = {}, {}
    eps_cap = {}
    unet_s.register_forward_hook(lambda m, i, o: eps_cap.__setitem__("s", o[0].detach().clone()))
    unet_t.register_forward_hook(lambda m, i, o: eps_cap.__setitem__("t", o[0].detach().clone()))
    ref.cast_hook(unet_s, self.KD_student)        # the reference's own hook installer
    ref.cast_hook(unet_t, self.KD_teacher)
    logs = {}
    self.log = lambda k, v, **kw: logs.__setitem__(k, float(v))
    self.trainer = types.SimpleNamespace(optimizers=[types.SimpleNamespace(param_groups=[{"lr": 1e-5}])],
                                         global_rank=1)
    self.global_step = 0
    if sdxl:
        self.encode_prompt = lambda texts, device: (t_ehs, t_neg, t_pool)
    else:
        self.encode_prompt = lambda texts, device: (t_ehs, t_neg)
    ref.args = types.SimpleNamespace(noise_offset=0.5, text_encoder="chinese_clip", KD=True, hybrid_training=True,
                                     every_n_steps=10, default_root_dir="/tmp")
    batch = {"pixel_values": torch.zeros(B, 3, 8, 8), "input_ids": torch.full((B, 4), 1),
             "input_ids_uncond": torch.full((B, 4), 2), "texts_en": [""] * B,
             "crops_coords_top_left": torch.zeros(B, 2, dtype=torch.int64),
             "original_size": torch.full((B, 2), 64, dtype=torch.int64), "bucket_id": 0,
             "zh_or_not": torch.tensor(zh_pattern, dtype=torch.int64)}
    # the reference draws: randn(latents) [, randn(B,4,1,1)], randint(B), rand(B) from the global RNG.
    # To force a chosen prompt_mask pattern, torch.rand is wrapped for this call only.
    real_rand = torch.rand
    torch.rand = lambda *a, **k: torch.tensor([0.05 if m else 0.9 for m in mask_pattern])
    try:
        torch.manual_seed(99)
        out = ref.StableDiffusion.training_step(self, batch, 0)
    finally:
        torch.rand = real_rand
    loss = out["loss"]
    loss.backward()
    # replay the RNG stream to recover what the reference drew
    torch.manual_seed(99)
    noise = torch.randn(B, 4, hw, hw)
    if sdxl:
        noise = noise + 0.5 * torch.randn(B, 4, 1, 1)
    timesteps = torch.randint(0, 1000, (B,))
    d = {"B": B, "L": L, "seed_model": 2024, "wsum_unets": wsum, "hip_dims": int(hip), "shared_teacher": int(shared_teacher),
         "noise_pred": eps_cap["s"], "noise_pred_teacher": eps_cap["t"], "mlp_args": np.array([int(a) for a in mlp_args]),
         "cfg_boc": np.array(cfg.block_out_channels), "cfg_heads": np.array(cfg.num_attention_heads),
         "cfg_cross": cfg.cross_attention_dim, "cfg_add_dim": cfg.addition_time_embed_dim,
         "cfg_proj_in": cfg.projection_class_embeddings_input_dim, "cfg_sample": cfg.sample_size,
         "latents": lat16.float() * sf, "noise": noise, "timesteps": timesteps, "enc": enc, "enc_uncond": enc_u,
         "teacher_ehs": t_ehs, "teacher_neg": t_neg, "teacher_pooled": t_pool,
         "prompt_mask": np.array(mask_pattern, dtype=bool), "zh_or_not": np.array(zh_pattern),
         "time_ids": torch.tensor([[64, 64, 0, 0, 64, 64]] * B), "loss": loss.detach(),
         "train_loss": logs["train_loss"], "train_loss_logits": logs["train_loss_logits"],
         "train_loss_features": logs["train_loss_features"],
         "tap_keys": np.array(list(self.KD_student.keys()))}
    for k, v in proj.state_dict().items():
        d["w." + k] = v
    for k, p in proj.named_parameters():
        d["g." + k] = p.grad
    # the reference leaves the student UNet trainable: record that its
    # weights DID receive grads, as a documented quirk (the build freezes them)
    d["unet_wgrad_populated"] = int(all(p.grad is not None for p in unet_s.parameters()))
    np.savez_compressed(os.path.join(OUT, f"step_{tag}.npz"), **np_(d))
    print("step", tag, float(loss), logs)


def gen_rescale(ref_test):
    torch.manual_seed(3)
    a, b = torch.randn(2, 4, 8, 8), torch.randn(2, 4, 8, 8)
    d = {"noise_cfg": a, "noise_pred_text": b}
    for gr in (0.0, 0.3, 0.7):
        d[f"out_{gr}"] = ref_test.rescale_noise_cfg(a, b, guidance_rescale=gr)
    np.savez_compressed(os.path.join(OUT, "rescale_noise_cfg.npz"), **np_(d))


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    sys.path.insert(0, REF)
    ref_sdxl = load_ref("train_sdxl_zh.py", "ref_train_sdxl")
    ref_sd = load_ref("train_sd_zh.py", "ref_train_sd")
    ref_test = load_ref("tests/test_sdxl_zh.py", "ref_test_sdxl")
    gen_mlp(ref_sdxl, ref_sd, ref_test)
    gen_step(ref_sdxl, "sdxl", [1, 0, 0, 1], [False, False, True, False], "sdxl_mixed")
    gen_step(ref_sdxl, "sdxl", [0, 0, 0, 0], [True, False, False, True], "sdxl_all_en")
    gen_step(ref_sdxl, "sdxl", [1, 1, 1], [False, False, False], "sdxl_all_zh")
    gen_step(ref_sd, "sd15", [1, 0, 0, 1], [False, True, False, False], "sd15_mixed")
    # the same reference step at dims the device path accepts: read by `-m gpu` tests/test_model_gpu.py
    gen_step(ref_sdxl, "sdxl", [1, 0, 0, 1], [False, False, True, False], "sdxl_hip_mixed", hip=True)
    gen_step(ref_sdxl, "sdxl", [0, 0, 0, 0], [True, False, False, True], "sdxl_hip_all_en", hip=True)
    gen_step(ref_sdxl, "sdxl", [1, 1, 1], [False, False, False], "sdxl_hip_all_zh", hip=True)
    gen_step(ref_sdxl, "sdxl", [1, 0, 0, 1], [False, True, False, False], "sdxl_hip_shared_teacher", hip=True, shared_teacher=True)
    gen_step(ref_sd, "sd15", [1, 0, 0, 1], [False, True, False, False], "sd15_hip_mixed", hip=True)
    gen_rescale(ref_test)


if __name__ == "__main__":
    main()
