"""Inference denoise loop on the HIP UNet: the scheduler object and loop body that
`StableDiffusionTest.__call__` drives (tests/test_sdxl_zh.py:350-406; ControlNet variant
tests/test_sdxl_zh_controlnet.py:437-553).  `DPMSolverMultistep` exposes the four members the reference touches --
`set_timesteps`, `timesteps`, `scale_model_input`, `step(...)[0]` -- with the configuration the reference loads
(:145, DPMSolverMultistepScheduler on the SDXL scheduler config: scaled-linear betas, epsilon prediction, "leading"
spacing with offset 1, dpmsolver++ 2M midpoint, lower_order_final).  The schedule's scalars are host float64; the
latent update, the CFG combine and `rescale_noise_cfg` are HIP kernels (csrc/sampler.hip)."""
from __future__ import annotations

import math
from typing import Callable, Optional

import numpy as np
import torch

from . import ops


class DPMSolverMultistep:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 timestep_spacing: str = "leading", steps_offset: int = 1, solver_order: int = 2,
                 lower_order_final: bool = True, final_sigma: str = "sigma_min"):
        if solver_order not in (1, 2):
            raise ValueError("solver_order 1 or 2")
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float64) ** 2
        self.alphas_cumprod = np.cumprod(1.0 - betas)
        self.num_train_timesteps = num_train_timesteps
        self.timestep_spacing, self.steps_offset = timestep_spacing, steps_offset
        self.solver_order, self.lower_order_final, self.final_sigma = solver_order, lower_order_final, final_sigma
        self.timesteps = None

    # ------------------------------------------------------------------ schedule (host)
    def set_timesteps(self, num_inference_steps: int, device=None):
        T, n = self.num_train_timesteps, num_inference_steps
        if self.timestep_spacing == "leading":
            ts = (np.arange(0, n + 1) * (T // (n + 1))).round()[::-1][:-1].copy().astype(np.int64) + self.steps_offset
        elif self.timestep_spacing == "linspace":
            ts = np.linspace(0, T - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif self.timestep_spacing == "trailing":
            ts = (np.arange(T, 0, -T / n).round() - 1).astype(np.int64)
        else:
            raise ValueError(f"timestep_spacing {self.timestep_spacing!r}")
        sig = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        sigmas = np.interp(ts, np.arange(0, len(sig)), sig)
        last = {"sigma_min": sig[0], "zero": 0.0, "repeat": sigmas[-1]}[self.final_sigma]
        self.sigmas = np.concatenate([sigmas, [last]])
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = n
        self._i = 0
        self._lower = 0
        self._x0_prev = None
        return self.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    @staticmethod
    def _alpha_sigma(sigma):
        a = 1.0 / math.sqrt(sigma * sigma + 1.0)
        return a, sigma * a

    def _coefficients(self, i: int, order: int):
        lam = lambda a, s: (math.log(a) - math.log(s)) if s > 0 else float("inf")
        a_t, s_t = self._alpha_sigma(self.sigmas[i + 1])
        a_s, s_s = self._alpha_sigma(self.sigmas[i])
        h = lam(a_t, s_t) - lam(a_s, s_s)
        em = math.expm1(-h) if math.isfinite(h) else -1.0
        c_s = s_t / s_s
        if order == 1:
            return a_s, s_s, c_s, -a_t * em, 0.0
        a_p, s_p = self._alpha_sigma(self.sigmas[i - 1])
        r0 = (lam(a_s, s_s) - lam(a_p, s_p)) / h
        return a_s, s_s, c_s, -a_t * em * (1.0 + 0.5 / r0), 0.5 * a_t * em / r0

    # ------------------------------------------------------------------ one step (device)
    def step(self, model_output, timestep, sample, return_dict: bool = False, **kwargs):
        """`latents = scheduler.step(noise_pred, t, latents, return_dict=False)[0]` (:406).  fp32 CUDA tensors;
        `sample` is updated IN PLACE and returned."""
        i, n = self._i, len(self.timesteps)
        final = (i == n - 1) and self.lower_order_final and n < 15
        order = 1 if (self.solver_order == 1 or self._lower < 1 or final) else 2
        a_s, s_s, c_s, c0, c1 = self._coefficients(i, order)
        if sample.dtype != torch.float32 or not sample.is_contiguous():
            sample = sample.float().contiguous()
        eps = model_output.float().contiguous()
        if self._x0_prev is None:
            self._x0_prev = torch.zeros_like(sample)
        ops.dpm_update_(sample, eps, self._x0_prev, a_s, s_s, c_s, c0, c1)
        if self._lower < self.solver_order:
            self._lower += 1
        self._i += 1
        return (sample,)


def denoise(unet, scheduler, latents, prompt_embeds, added_cond_kwargs, num_inference_steps: int = 30,
            guidance_scale: float = 7.5, guidance_rescale: float = 0.0,
            residual_fn: Optional[Callable] = None, callback: Optional[Callable] = None):
    """Steps 4-7 of the reference pipeline call (tests/test_sdxl_zh.py:350-406): timesteps, CFG batch doubling, UNet,
    guidance (+ rescale), scheduler step.  `unet` is a `HipUNet` built for batch 2B when guidance_scale > 1.
    `residual_fn(latent_model_input, t) -> (down_residuals, mid_residual)` is where a ControlNet plugs in
    (tests/test_sdxl_zh_controlnet.py:510-535).  Returns the final latents (fp32, the VAE decode stays outside)."""
    do_cfg = guidance_scale > 1.0
    timesteps = scheduler.set_timesteps(num_inference_steps)
    latents = (latents.float() * scheduler.init_noise_sigma).contiguous()
    for i, t in enumerate(timesteps):
        x = torch.cat([latents] * 2) if do_cfg else latents
        x = scheduler.scale_model_input(x, t)
        kw = {}
        if residual_fn is not None:
            down, mid = residual_fn(x, t)
            kw = dict(down_block_additional_residuals=down, mid_block_additional_residual=mid)
        noise_pred = unet(x, t, encoder_hidden_states=prompt_embeds, added_cond_kwargs=added_cond_kwargs,
                          return_dict=False, **kw)[0]
        if do_cfg:
            noise_pred = ops.cfg_combine(noise_pred.float(), guidance_scale, guidance_rescale)
        latents = scheduler.step(noise_pred, t, latents, return_dict=False)[0]
        if callback is not None:
            callback(i, t, latents)
    return latents
