"""CPU (no device): every attention op of every UNet graph the product builds is fed a PRESCALED Q (the producing projection's
epilogue multiplied its Q block by softmax_scale * log2 e from the fp32 accumulator -- csrc/model.hip:Tape::tag_q_prescale).
An untagged op would silently run the operator-level plain-Q path of csrc/attention.hip, which rounds the scaled operand to
bf16 once more (round-3 advisor finding): the planning pass of the C ABI reports the census, asserted here for the BASELINE
configurations; the GPU tests assert it on the live handles of the VAE / ControlNet / text-encoder graphs as well."""
import ctypes

import pytest

from pea_diffusion_amd import config as pc
from pea_diffusion_amd._lib import check, lib


@pytest.mark.parametrize("name,hw,L", [("sdxl_config", 128, 77), ("sdxl_config", 64, 52), ("sd15_config", 64, 77),
                                       ("ssd1b_config", 128, 77), ("ssd1b_uniform_config", 64, 77), ("tiny_config", 16, 12),
                                       ("tiny15_config", 16, 12)])
@pytest.mark.parametrize("flags", [0, 1, 2])          # inference, needs_grad, ControlNet residual inputs
def test_every_unet_attention_is_prescaled(name, hw, L, flags):
    cfg = getattr(pc, name)()
    c = pc.to_c(cfg)
    n, pre = ctypes.c_int(), ctypes.c_int()
    check(lib().pea_unet_plan_attention(ctypes.byref(c), 2, hw, hw, L, flags, ctypes.byref(n), ctypes.byref(pre)))
    assert n.value > 0 and n.value == pre.value, (name, flags, n.value, pre.value)
    if name == "sdxl_config":
        assert n.value == 140          # 70 transformer blocks x (self + cross)


def _census(fn, *args):
    n, pre = ctypes.c_int(), ctypes.c_int()
    check(fn(*args, ctypes.byref(n), ctypes.byref(pre)))
    return n.value, pre.value


@pytest.mark.parametrize("name,L,want", [("clip_l_config", 77, 12), ("openclip_bigg_config", 77, 32),
                                         ("cnclip_bert_large_config", 52, 24), ("xlm_roberta_large_config", 77, 24),
                                         ("mt5_xl_config", 77, 24), ("tiny_clip_config", 12, None), ("tiny_bert_config", 12, None),
                                         ("tiny_xlmr_config", 12, None), ("tiny_t5_config", 12, None)])
def test_every_text_tower_attention_is_prescaled(name, L, want):
    """(round 4: the BERT-family towers had ONE layer on the plain-Q path -- the embedding op's weight-slot index collided
    with the tensor id of layer 0's Q|K|V in tag_q_prescale's reader scan)"""
    cfg = getattr(pc, name)()
    c = pc.text_to_c(cfg)
    n, pre = _census(lib().pea_text_plan_attention, ctypes.byref(c), 2, L)
    assert n == pre and n > 0 and (want is None or n == want), (name, n, pre)


@pytest.mark.parametrize("name,hw", [("sdxl_config", 128), ("tiny_config", 16)])
def test_every_controlnet_attention_is_prescaled(name, hw):
    c = pc.to_c(getattr(pc, name)())
    n, pre = _census(lib().pea_graph_plan_attention, 2, ctypes.byref(c), 2, hw, hw, 77)
    assert n == pre and n > 0, (name, n, pre)


def test_vae_graphs_have_no_fused_attention_op():
    """the VAE's single-head 512-wide attention is materialised per image (GEMM + softmax + GEMM), not an OP_ATTN"""
    cfg = pc.sdxl_vae_config()
    assert _census(lib().pea_graph_plan_attention, 1, ctypes.byref(pc.vae_to_c(cfg)), 1, 128, 128, 0) == (0, 0)
    assert _census(lib().pea_graph_plan_attention, 3, ctypes.byref(pc.vae_decoder_to_c(cfg)), 1, 16, 16, 0) == (0, 0)
