"""`PEATrainer`: the reference's KD `training_step` (train_sdxl_zh.py:305-449) downstream of the frozen
VAE / text encoders, as ONE call into libpea_hip.so, plus the optimizer + LR schedule of
utils/model_utils.py:45-81 (FusedAdam(adam_w_mode=True), polynomial decay with warmup) and the
data-parallel gradient all-reduce (one RCCL all-reduce on the flat adapter-grad buffer; replaces
DeepSpeed ZeRO-1, train_sdxl_zh.sh:22,87)."""
from __future__ import annotations

import ctypes
import os
from collections import OrderedDict
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import dist as pdist
from . import ops
from ._lib import PeaError, check, lib, ptr, stream_ptr
from .adapter import PEAAdapter
from .unet import HipUNet


def ddpm_alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012):
    """DDPMScheduler(beta_schedule="scaled_linear") of train_sdxl_zh.py:140"""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def polynomial_lr(step: int, base_lr: float, warmup: int, total: int, lr_end: float, power: float = 1.0) -> float:
    """transformers' polynomial-decay-with-warmup (selected by utils/model_utils.py:136-138)"""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    if step > total:
        return lr_end
    rem = 1 - (step - warmup) / max(1, total - warmup)
    return (base_lr - lr_end) * rem ** power + lr_end


class PEATrainer:
    LOG_KEYS = ("loss", "train_loss", "train_loss_logits", "train_loss_features")

    def __init__(self, adapter: PEAAdapter, student: HipUNet, teacher: HipUNet, feat_weight: float = 0.1,
                 nan_guard: bool = False, lr: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, warmup_steps: int = 100, total_steps: int = 2232142, lr_end: float = 5e-8):
        self.adapter, self.student, self.teacher = adapter, student, teacher
        adapter.prepare(2 * student.B, student.L)
        adapter._sync()
        self._h = ctypes.c_void_p()
        self._ac = ddpm_alphas_cumprod().to(student.device)
        check(lib().pea_trainer_create(adapter._h, student._h, teacher._h, feat_weight, int(nan_guard), ptr(self._ac),
                                       ctypes.byref(self._h)))
        self.losses = torch.zeros(4, device=student.device)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.warmup_steps, self.total_steps, self.lr_end = warmup_steps, total_steps, lr_end
        self.global_step = 0
        self.consumed_samples = 0                          # Lightning's `global_samples` (train_sdxl_zh.py:456): samples seen by ALL ranks
        self._m = torch.zeros_like(adapter.flat_param)
        self._v = torch.zeros_like(adapter.flat_param)
        self.comm: Optional[pdist.NativeComm] = None      # RCCL communicator + comm stream inside libpea_hip.so
        self._pending = None                               # torch.distributed work handle (gloo / torch-NCCL path)
        self._comm_inflight = False                        # an all-reduce launched on the native communicator, not joined yet
        self.local_only = False                            # True: steps skip the gradient all-reduce (bench.py: the single-GPU-equivalent step time of an N-rank run)
        # Dead-row elimination (opt-in): with teacher == student checkpoint (merged passes) the teacher row of a sample whose
        # zh_or_not is 1 only ever meets the weight (1 - zh_or_not) = 0 (train_sdxl_zh.py:402-441), so it is not computed: the
        # merged pass runs over B + n_t samples.  Same losses and gradients; needs zh_or_not on the host (a CPU tensor from the
        # dataloader costs nothing; a device tensor costs one small synchronising copy per step).
        self.skip_dead_teacher_rows = False

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().pea_trainer_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    def _dev(self, t, dtype):
        return t.detach().to(self.student.device, dtype).contiguous()

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int = 0, sync: bool = False,
                      async_allreduce: bool = False):
        """batch: the post-encoder form of the reference batch: latents, noise, timesteps, enc,
        enc_uncond, prompt_mask, zh_or_not, teacher_ehs, teacher_neg [, teacher_pooled, time_ids].
        Leaves the adapter gradients in `adapter.flat_grad`; returns {"loss": device scalar, ...}.

        Data parallel: by default the averaged gradient is complete (on the current stream) when this returns, so
        `p.grad` can be clipped, logged or fed to any optimizer.  `async_allreduce=True` only LAUNCHES the collective
        on the communicator's stream and leaves the join to `optimizer_step()` / `join_grads()`, so work enqueued
        in between (the next batch's VAE encode) overlaps the xGMI transfer; `p.grad` must not be read before that."""
        f32, dev = torch.float32, self._dev
        self.join_grads()                   # a collective still in flight reads and writes flat_grad in place
        self.adapter.prepare(2 * self.student.B, self.student.L)   # a stand-alone proj(x) call may have re-shaped it
        self.adapter._sync()
        b = {k: dev(batch[k], f32) for k in ("latents", "noise", "enc", "enc_uncond", "teacher_ehs", "teacher_neg")}
        ts = dev(batch["timesteps"], torch.int64)
        pm = dev(batch["prompt_mask"], torch.uint8)
        zh = dev(batch["zh_or_not"], torch.int64)
        tp = dev(batch["teacher_pooled"], f32) if "teacher_pooled" in batch else None
        tid = dev(batch["time_ids"], f32) if "time_ids" in batch else None
        world = self.world_size
        if self.skip_dead_teacher_rows and self.student.B <= 30:
            live = (batch["zh_or_not"].detach().reshape(-1).cpu() == 0).tolist()
            check(lib().pea_trainer_set_option(self._h, b"live_teacher_mask", sum(1 << i for i, v in enumerate(live) if v)))
        elif lib().pea_trainer_get_option(self._h, b"live_teacher_mask") != -1:
            check(lib().pea_trainer_set_option(self._h, b"live_teacher_mask", -1))
        check(lib().pea_train_step(self._h, ptr(b["latents"]), ptr(b["noise"]), ptr(ts), ptr(b["enc"]),
                                   ptr(b["enc_uncond"]), ptr(pm), ptr(zh), ptr(b["teacher_ehs"]), ptr(b["teacher_neg"]),
                                   ptr(tp), ptr(tid), 1.0, ptr(self.adapter.flat_grad), 0, ptr(self.losses),
                                   stream_ptr()))
        self._keep = (b, ts, pm, zh, tp, tid)
        self.consumed_samples += int(b["latents"].shape[0]) * (self.comm.world if self.comm is not None else world)
        if (self.comm is not None or world > 1) and not self.local_only:
            self.all_reduce_grads_async()   # launched right behind the adapter wgrad
            if not async_allreduce:
                self.join_grads()
        for p, o in zip(self.adapter._plist(), self.adapter._offsets):
            p.grad = self.adapter.flat_grad[o:o + p.numel()].view_as(p)
        snap = self.losses.clone()          # device-side snapshot: later steps overwrite self.losses
        out = {k: snap[i] for i, k in enumerate(self.LOG_KEYS)}
        if sync:
            torch.cuda.synchronize()
        return out

    def attach_frontend(self, frontend):
        """`PEAFrontEnd` (frozen VAE + teacher CLIP towers + student text tower) for training_step_from_batch"""
        self.frontend = frontend

    def training_step_from_batch(self, batch: Dict, batch_idx: int = 0, generator: Optional[torch.Generator] = None,
                                 sync: bool = False):
        """`training_step(self, batch, batch_idx)` on the dataloader's own dictionary (utils/custom_dataset_sdxl.py:397-407:
        pixel_values, input_ids, input_ids_uncond, original_size, crops_coords_top_left, bucket_id, zh_or_not, texts_en):
        VAE encode, the three text towers, add_time_ids from BUCKETS, the random draws, then the KD step
        (train_sdxl_zh.py:305-449)."""
        fe = getattr(self, "frontend", None)
        if fe is None:
            raise PeaError("training_step_from_batch: attach_frontend(PEAFrontEnd(...)) first")
        return self.training_step(fe.prepare(batch, generator=generator), batch_idx, sync=sync)

    # ---- data parallel: ONE all-reduce over the flat adapter-grad buffer (24-46 MB), averaged
    def set_option(self, name: str, value: int):
        """`two_stream` (teacher pass on a side HIP stream), `merge_passes` (teacher == student checkpoint: both forwards
        as one pass over 2B samples), `nan_guard`"""
        check(lib().pea_trainer_set_option(self._h, name.encode(), int(value)))

    @property
    def world_size(self) -> int:
        return pdist.world_size()

    def attach_comm(self, comm: "pdist.NativeComm"):
        """use the library's own RCCL communicator (dedicated comm stream) for the gradient all-reduce"""
        self.comm = comm

    def all_reduce_grads_async(self):
        """ONE all-reduce of the flat adapter gradient, off the compute stream: RCCL on the communicator's own HIP
        stream (`pea_allreduce_grads`), or -- without an attached communicator -- torch.distributed's asynchronous
        all-reduce (ProcessGroupNCCL runs it on its internal stream).  `join_grads()` must precede any read."""
        if self.comm is not None:
            self.comm.allreduce_mean_async(self.adapter.flat_grad)
            self._comm_inflight = True
        else:
            self._pending = pdist.allreduce_mean_(self.adapter.flat_grad, async_op=True)

    def join_grads(self):
        """make the current stream wait for the gradient all-reduce launched last (no-op when none is pending)"""
        if self.comm is not None:
            if self._comm_inflight:
                self.comm.join()
                self._comm_inflight = False
        elif self._pending is not None:
            self._pending.wait()
            self.adapter.flat_grad.div_(self.world_size)
            self._pending = None

    def all_reduce_grads(self):
        self.all_reduce_grads_async()
        self.join_grads()

    def export(self, which: str) -> torch.Tensor:
        idx = {"x_t": 0, "eps_student": 1, "eps_teacher": 2}[which]
        out = torch.empty(self.student.B, self.student.cfg.in_channels, self.student.H, self.student.W,
                          device=self.student.device)
        check(lib().pea_trainer_export(self._h, idx, ptr(out), stream_ptr()))
        return out

    # ---- optimizer (FusedAdam adam_w_mode + polynomial schedule; weight_decay never reaches the
    # reference's optimizer, utils/model_utils.py:64-67 + train_sdxl_zh.py:167, so the default is 0)
    def current_lr(self) -> float:
        return polynomial_lr(self.global_step, self.lr, self.warmup_steps, self.total_steps, self.lr_end)

    def optimizer_step(self):
        # Lightning + transformers' LambdaLR: optimizer step k (1-indexed) runs with lambda(k - 1), so the very first
        # update has lr = 0 (utils/model_utils.py:98-140); Adam's bias correction uses k.
        self.join_grads()
        lr = polynomial_lr(self.global_step, self.lr, self.warmup_steps, self.total_steps, self.lr_end)
        self.global_step += 1
        ops.adamw_(self.adapter.flat_param, self.adapter.flat_grad, self._m, self._v, lr, self.global_step,
                   self.betas[0], self.betas[1], self.eps, self.weight_decay)
        self.adapter.flat_param._version  # noqa: B018  (in-place op below bumps the version counter)
        self.adapter.mark_updated()

    def save_adapter(self, root: str, optimizer_state: bool = True):
        """`torch.save(self.proj.state_dict(), f"{root}/proj_{global_step}/pytorch_model.bin")`
        (train_sdxl_zh.py:443-448) -- the file the reference's `proj.load_state_dict(torch.load(...))` reads
        (tests/test_sdxl_zh.py:153) -- and, beside it, `trainer_state.pt`: what Lightning + DeepSpeed keep in THEIR
        checkpoint so that training resumes where it stopped (`on_load_checkpoint` restores `global_step` and
        `global_samples`, train_sdxl_zh.py:454-458; the engine restores the fp32 master weights and both Adam
        moments): the flat fp32 parameters (the adapter's `state_dict` is what the reference saves -- fp16 there,
        a rounded copy of the masters), m, v, the step counter, consumed samples and the schedule's constants."""
        d = os.path.join(root, f"proj_{self.global_step}")
        os.makedirs(d, exist_ok=True)
        torch.save({k: v.detach().cpu().clone() for k, v in self.adapter.state_dict().items()},
                   os.path.join(d, "pytorch_model.bin"))
        if optimizer_state:
            self.join_grads()
            torch.cuda.current_stream().synchronize()
            torch.save(self.state_dict(), os.path.join(d, "trainer_state.pt"))
        return d

    TRAINER_STATE_VERSION = 1

    def state_dict(self) -> Dict[str, object]:
        """everything `optimizer_step` reads besides the gradient (host copies): a fresh trainer that loads this continues
        the run bit for bit (tests/test_model_gpu.py::test_checkpoint_resume_is_bit_exact)"""
        return {"version": self.TRAINER_STATE_VERSION, "global_step": int(self.global_step),
                "consumed_samples": int(getattr(self, "consumed_samples", 0)),
                "flat_param": self.adapter.flat_param.detach().cpu().clone(),
                "exp_avg": self._m.detach().cpu().clone(), "exp_avg_sq": self._v.detach().cpu().clone(),
                "param_names": [n for n, _ in self.adapter.named_parameters()],
                "param_numel": [int(p.numel()) for p in self.adapter._plist()],
                "hparams": {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                            "warmup_steps": self.warmup_steps, "total_steps": self.total_steps, "lr_end": self.lr_end}}

    def load_state_dict(self, st: Dict[str, object], load_hparams: bool = True):
        if int(st.get("version", 0)) != self.TRAINER_STATE_VERSION:
            raise PeaError(f"trainer state version {st.get('version')} (this build reads {self.TRAINER_STATE_VERSION})")
        numel = [int(p.numel()) for p in self.adapter._plist()]
        if list(st["param_numel"]) != numel or st["flat_param"].numel() != self.adapter.flat_param.numel():
            raise PeaError(f"trainer state is for an adapter with parameter sizes {list(st['param_numel'])}, this one has {numel}")
        self.join_grads()
        with torch.no_grad():
            self.adapter.flat_param.copy_(st["flat_param"].to(self.adapter.flat_param.device))
            self._m.copy_(st["exp_avg"].to(self._m.device))
            self._v.copy_(st["exp_avg_sq"].to(self._v.device))
        self.adapter.mark_updated()
        self.global_step = int(st["global_step"])
        self.consumed_samples = int(st.get("consumed_samples", 0))
        if load_hparams:
            h = st["hparams"]
            self.lr, self.betas, self.eps, self.weight_decay = h["lr"], tuple(h["betas"]), h["eps"], h["weight_decay"]
            self.warmup_steps, self.total_steps, self.lr_end = h["warmup_steps"], h["total_steps"], h["lr_end"]

    def resume(self, root: str, step: Optional[int] = None) -> str:
        """load `root/proj_{step}/trainer_state.pt` (default: the highest step that has one); returns the directory.
        A directory that only holds the reference's `pytorch_model.bin` restores the weights and nothing else (the
        moments start from zero, as they do when the reference loads `proj_0_{id}`: train_sdxl_zh.py:145)."""
        if step is None:
            steps = []
            for n in os.listdir(root):
                if n.startswith("proj_") and n[5:].isdigit() and os.path.exists(os.path.join(root, n, "pytorch_model.bin")):
                    steps.append(int(n[5:]))
            if not steps:
                raise PeaError(f"resume: no proj_<step>/pytorch_model.bin under {root}")
            step = max(steps)
        d = os.path.join(root, f"proj_{step}")
        ts = os.path.join(d, "trainer_state.pt")
        if os.path.exists(ts):
            self.load_state_dict(torch.load(ts, map_location="cpu", weights_only=False))
        else:
            self.join_grads()
            self.adapter.load_state_dict(torch.load(os.path.join(d, "pytorch_model.bin"), map_location="cpu"))
            self.adapter.mark_updated()
            self.global_step = int(step)
        return d


class BucketedTrainer(PEATrainer):
    """The KD step over the reference's aspect-ratio buckets (utils/custom_dataset_sdxl.py:30,384-409: every batch comes
    from ONE of nine (height, width) buckets, 448x896 ... 896x448, and `training_step` simply runs on whatever shape
    arrives).  The HIP tape is planned per shape, so this trainer keeps one (student, teacher, trainer) context per
    latent shape -- created on first use, all of them borrowing the weights of the contexts it was built with -- behind
    the one PEATrainer surface: one adapter, one AdamW state, one LR schedule, one gradient buffer.

    HBM: a context's activation + gradient arenas are resident from its first step (17 + 12 GB per bucket for SDXL at
    B = 4, merged passes); `max_resident_gb` bounds their sum -- before a context that is not resident runs, the least
    recently used ones are released (`pea_trainer_release_activations`) until its planned size fits.  The default keeps
    every one of the nine SDXL buckets resident at B = 4 on a 288 GB MI355X."""

    def __init__(self, adapter: PEAAdapter, student: HipUNet, teacher: HipUNet, max_resident_gb: float = 240.0, **kw):
        super().__init__(adapter, student, teacher, **kw)
        self._kw = dict(feat_weight=kw.get("feat_weight", 0.1), nan_guard=kw.get("nan_guard", False))
        self._primary = (student, teacher)
        self._ctx: "OrderedDict[Tuple[int, int], tuple]" = OrderedDict()
        self._ctx[(student.H, student.W)] = (student, teacher, self._h)
        self._options: Dict[str, int] = {}
        self._known: Dict[Tuple[int, int], int] = {}
        self.max_resident_bytes = int(max_resident_gb * 2 ** 30)

    # ---- context management
    def _planned_bytes(self, h: int, w: int) -> int:
        """upper bound of what a step at this shape allocates: the merged-pass context (2B samples, with gradients)"""
        from . import config as _cfg
        s0 = self._primary[0]
        c = _cfg.to_c(s0.cfg)
        ab, gb = ctypes.c_longlong(), ctypes.c_longlong()
        check(lib().pea_unet_plan(ctypes.byref(c), 2 * s0.B, h, w, self._primary[1].L, 1, None, None, None, None,
                                  ctypes.byref(ab), ctypes.byref(gb)))
        return ab.value + gb.value

    def _resident_bytes(self, ctx) -> int:
        st, te, h = ctx
        n = 0
        for u in (st, te):
            m = u.memory()
            n += m["activation_bytes"] + m["grad_bytes"]
        return n + (lib().pea_trainer_get_option(h, b"merged_mib") << 20)

    def resident_bytes(self) -> int:
        return sum(self._resident_bytes(c) for c in self._ctx.values())

    def _select(self, h: int, w: int):
        key = (int(h), int(w))
        if key not in self._ctx:
            s0, t0 = self._primary
            st = HipUNet(s0.cfg, s0.B, key[0], key[1], s0.L, needs_grad=True, share_weights_from=s0)
            te = HipUNet(t0.cfg, t0.B, key[0], key[1], t0.L, share_weights_from=t0)
            hd = ctypes.c_void_p()
            self.adapter.prepare(2 * s0.B, s0.L)                    # a stand-alone proj(x) call may have re-shaped it
            check(lib().pea_trainer_create(self.adapter._h, st._h, te._h, self._kw["feat_weight"],
                                           int(self._kw["nan_guard"]), ptr(self._ac), ctypes.byref(hd)))
            for name, value in self._options.items():
                check(lib().pea_trainer_set_option(hd, name.encode(), int(value)))
            self._ctx[key] = (st, te, hd)
        self._ctx.move_to_end(key)
        cur = self._ctx[key]
        # make room first: what this context is about to allocate (0 when it is resident) + what is resident must fit
        have = self._resident_bytes(cur)
        if have:
            self._known[key] = have                                 # what this shape really takes, for its next admission
        need = 0 if have else self._known.get(key) or self._planned_bytes(*key)
        for k in list(self._ctx):                                   # least recently used first
            if self.resident_bytes() + need <= self.max_resident_bytes:
                break
            if k != key and self._resident_bytes(self._ctx[k]):
                check(lib().pea_trainer_release_activations(self._ctx[k][2]))
        self.student, self.teacher, self._h = cur

    @property
    def shapes(self) -> Sequence[Tuple[int, int]]:
        """latent shapes with a context, least recently used first"""
        return list(self._ctx)

    def set_option(self, name: str, value: int):
        self._options[name] = int(value)
        for _, _, h in self._ctx.values():
            check(lib().pea_trainer_set_option(h, name.encode(), int(value)))

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int = 0, sync: bool = False,
                      async_allreduce: bool = False):
        h, w = batch["latents"].shape[-2:]
        self.join_grads()
        self._select(h, w)
        return super().training_step(batch, batch_idx, sync=sync, async_allreduce=async_allreduce)

    def __del__(self):
        try:
            for _, _, h in getattr(self, "_ctx", {}).values():
                if h and h.value:
                    lib().pea_trainer_destroy(h)
            self._ctx = OrderedDict()
            self._h = ctypes.c_void_p()
        except Exception:
            pass
