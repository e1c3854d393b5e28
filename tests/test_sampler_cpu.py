"""CPU checks of the denoise-loop restatement (oracle/sampler_ref.py) and of the host-side schedule math the HIP
path shares with it (pea_diffusion_amd/sampler.py computes the same scalars; the tensors move on the GPU)."""
import math

import numpy as np
import pytest
import torch

from oracle.sampler_ref import DPMSolverMultistepRef, cfg_combine_ref, denoise_ref
from oracle.step_ref import ddpm_alphas_cumprod


def test_leading_timesteps_match_the_sdxl_scheduler_config():
    s = DPMSolverMultistepRef()
    ts = s.set_timesteps(30)
    assert len(ts) == 30 and ts[0] == 961 and ts[-1] == 33 and np.all(np.diff(ts) == -32)
    assert len(s.sigmas) == 31 and np.all(np.diff(s.sigmas) < 0)
    ac = ddpm_alphas_cumprod().double().numpy()
    np.testing.assert_allclose(s.alphas_cumprod, ac, rtol=1e-6)          # same DDPM schedule as the training step
    np.testing.assert_allclose(s.sigmas[0], math.sqrt((1 - ac[961]) / ac[961]), rtol=1e-5)      # ac is the fp32 table


@pytest.mark.parametrize("n,order", [(30, 2), (10, 2), (4, 1), (1, 2)])
def test_constant_data_prediction_is_integrated_exactly(n, order):
    """DPM-Solver++ is exact when the data prediction is constant: with eps = (x_t - alpha_t c) / sigma_t every update
    (first order, second order, lower-order final) must land on alpha_next c + sigma_next n."""
    s = DPMSolverMultistepRef(solver_order=order)
    s.set_timesteps(n)
    g = torch.Generator().manual_seed(0)
    c = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    nz = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    a0, s0 = s._alpha_sigma(s.sigmas[0])
    x = a0 * c + s0 * nz
    for i, t in enumerate(s.timesteps):
        a, sg = s._alpha_sigma(s.sigmas[i])
        eps = (x - a * c) / sg
        x = s.step(eps, t, x)[0]
        a1, s1 = s._alpha_sigma(s.sigmas[i + 1])
        assert torch.allclose(x, a1 * c + s1 * nz, rtol=0, atol=1e-9), i


def test_hip_host_schedule_equals_oracle():
    from pea_diffusion_amd.sampler import DPMSolverMultistep
    for spacing, off in (("leading", 1), ("linspace", 0), ("trailing", 0)):
        a, b = DPMSolverMultistepRef(timestep_spacing=spacing, steps_offset=off), DPMSolverMultistep(timestep_spacing=spacing, steps_offset=off)
        ta, tb = a.set_timesteps(20), b.set_timesteps(20)
        assert np.array_equal(ta, tb.numpy())
        np.testing.assert_allclose(a.sigmas, b.sigmas, rtol=1e-13)
        for i in range(20):
            for order in ((1,) if i == 0 else (1, 2)):
                np.testing.assert_allclose(a.coefficients(i, order), b._coefficients(i, order), rtol=1e-12)


def test_cfg_combine_and_loop_shapes():
    g = torch.Generator().manual_seed(1)
    e = torch.randn(4, 4, 8, 8, generator=g)
    c, t = cfg_combine_ref(e, 7.5)
    assert torch.equal(t, e[2:]) and torch.allclose(c, e[:2] + 7.5 * (e[2:] - e[:2]))

    class ToyUNet:                       # eps = 0.1 * x: the loop must thread CFG batches and residual kwargs through
        calls = []
        def __call__(self, x, t, encoder_hidden_states=None, added_cond_kwargs=None, return_dict=False, **kw):
            self.calls.append((tuple(x.shape), int(t), sorted(kw)))
            return (0.1 * x,)
    u = ToyUNet()
    lat = torch.randn(2, 4, 8, 8, generator=g)
    out = denoise_ref(u, DPMSolverMultistepRef(), lat, None, None, num_inference_steps=5, guidance_scale=5.0,
                      guidance_rescale=0.7, residual_fn=lambda x, t: ([x], x))
    assert out.shape == lat.shape and torch.isfinite(out).all() and len(u.calls) == 5
    assert u.calls[0] == ((4, 4, 8, 8), 831, ["down_block_additional_residuals", "mid_block_additional_residual"])
