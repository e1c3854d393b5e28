"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): torch fp32 restatement of the ControlNet call of the reference's
inference path -- `down_block_res_samples, mid_block_res_sample = self.controlnet(control_model_input, t,
encoder_hidden_states=..., controlnet_cond=image, conditioning_scale=cond_scale, guess_mode=guess_mode,
added_cond_kwargs=..., return_dict=False)` (tests/test_sdxl_zh_controlnet.py:510-519).  `ControlNetModel` is
diffusers==0.23.0 (requirements.txt:25), absent from /root/reference: module graph and state-dict keys are restated from
its published definition on top of the UNet blocks of oracle/unet_ref.py -- **parity unpinned** at that boundary.
The restated SDXL graph (controlnet-canny-sdxl-1.0 layout) totals 1 251 014 160 parameters = 2.50 GB in fp16, the
size of the published fp16 checkpoint file."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet_ref import DownBlock, MidBlock, TimestepEmbedding, UNetConfig, timestep_embedding


class ControlNetConditioningEmbedding(nn.Module):
    def __init__(self, out_channels, cond_channels=3, block_out_channels=(16, 32, 96, 256)):
        super().__init__()
        self.conv_in = nn.Conv2d(cond_channels, block_out_channels[0], 3, padding=1)
        self.blocks = nn.ModuleList()
        for i in range(len(block_out_channels) - 1):
            a, b = block_out_channels[i], block_out_channels[i + 1]
            self.blocks.append(nn.Conv2d(a, a, 3, padding=1))
            self.blocks.append(nn.Conv2d(a, b, 3, padding=1, stride=2))
        self.conv_out = nn.Conv2d(block_out_channels[-1], out_channels, 3, padding=1)   # zero-initialised upstream

    def forward(self, x):
        x = F.silu(self.conv_in(x))
        for b in self.blocks:
            x = F.silu(b(x))
        return self.conv_out(x)


class ControlNetRef(nn.Module):
    def __init__(self, cfg: UNetConfig):
        super().__init__()
        self.config = cfg
        boc = cfg.block_out_channels
        nb = len(boc)
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], cfg.time_embed_dim)
        if cfg.addition_embed_type == "text_time":
            self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, cfg.time_embed_dim)
        self.controlnet_cond_embedding = ControlNetConditioningEmbedding(boc[0])
        self.down_blocks = nn.ModuleList()
        self.controlnet_down_blocks = nn.ModuleList([nn.Conv2d(boc[0], boc[0], 1)])
        out = boc[0]
        for i, ty in enumerate(cfg.down_block_types):
            cin, out = out, boc[i]
            self.down_blocks.append(DownBlock(cfg, cin, out, cfg.transformer_layers_per_block[i],
                                              cfg.num_attention_heads[i], ty.startswith("CrossAttn"),
                                              down=(i != nb - 1)))
            for _ in range(cfg.layers_per_block + (1 if i != nb - 1 else 0)):
                self.controlnet_down_blocks.append(nn.Conv2d(out, out, 1))
        last = cfg.transformer_layers_per_block[-1]                    # same rule as UNet2DConditionRef: element [0] of a nested entry
        self.mid_block = MidBlock(cfg, boc[-1], last if isinstance(last, int) else last[0], cfg.num_attention_heads[-1])
        self.controlnet_mid_block = nn.Conv2d(boc[-1], boc[-1], 1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def embed(self, timesteps, added_cond_kwargs, B):
        cfg = self.config
        t = timesteps
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.int64)
        if t.dim() == 0:
            t = t[None]
        t = t.expand(B)
        emb = self.time_embedding(timestep_embedding(t, cfg.block_out_channels[0]).to(self.dtype))
        if cfg.addition_embed_type == "text_time":
            te = timestep_embedding(added_cond_kwargs["time_ids"].flatten(), cfg.addition_time_embed_dim).reshape(B, -1)
            add = torch.cat([added_cond_kwargs["text_embeds"], te.to(added_cond_kwargs["text_embeds"].dtype)], dim=-1)
            emb = emb + self.add_embedding(add.to(self.dtype))
        return emb

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0,
                guess_mode=False, added_cond_kwargs=None, return_dict=False):
        B = sample.shape[0]
        emb = self.embed(timestep, added_cond_kwargs, B)
        x = self.conv_in(sample) + self.controlnet_cond_embedding(controlnet_cond)
        res = (x,)
        for blk in self.down_blocks:
            x, outs = blk(x, emb, encoder_hidden_states)
            res += outs
        x = self.mid_block(x, emb, encoder_hidden_states)
        down = [conv(r) for r, conv in zip(res, self.controlnet_down_blocks)]
        mid = self.controlnet_mid_block(x)
        if guess_mode:                       # diffusers 0.23 [ext]: residual i is weighted 0.1 .. 1.0 on a log scale (:516)
            scales = torch.logspace(-1, 0, len(down) + 1) * conditioning_scale
            down = [d * s for d, s in zip(down, scales)]
            mid = mid * scales[-1]
        else:
            down = [d * conditioning_scale for d in down]
            mid = mid * conditioning_scale
        return down, mid
