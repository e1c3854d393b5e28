"""Golden vectors for the text encoders from the INSTALLED transformers release (third-party; the reference pins 4.31.0,
this container has 5.x -- both implement the same CLIP text model, BERT and T5): seeded tiny configurations, weights + ids +
outputs -> tests/golden/text_clip.npz, text_bert.npz, text_xlmr.npz (default) and text_t5.npz (`... make_golden_text.py t5`).
Run: PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_text.py [t5]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers import (BertConfig, BertModel, CLIPTextConfig, CLIPTextModelWithProjection, XLMRobertaConfig,  # noqa: E402
                          XLMRobertaModel)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    torch.manual_seed(0)
    c = CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
                       max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=64, eos_token_id=999,
                       bos_token_id=998, pad_token_id=1)
    m = CLIPTextModelWithProjection(c).eval()
    with torch.no_grad():
        for p in m.parameters():                      # HF init is N(0, 0.02): scale up so every term matters
            if p.dim() >= 2:
                p.mul_(3.0)
            else:
                p.add_(0.05 * torch.randn_like(p))
    ids = torch.randint(2, 998, (2, 77))
    ids[:, 0] = 998
    ids[0, 10:] = 1; ids[0, 10] = 999
    ids[1, 40:] = 1; ids[1, 40] = 999
    with torch.no_grad():
        o = m(ids, output_hidden_states=True)
    d = {"w." + k: v.numpy() for k, v in m.state_dict().items()}
    d.update(ids=ids.numpy(), last_hidden_state=o.last_hidden_state.numpy(), text_embeds=o.text_embeds.numpy())
    for i, h in enumerate(o.hidden_states):
        d[f"hidden_{i}"] = h.numpy()
    np.savez_compressed(os.path.join(OUT, "text_clip.npz"), **d)

    torch.manual_seed(1)
    b = BertConfig(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                   max_position_embeddings=64, type_vocab_size=2, layer_norm_eps=1e-12)
    bm = BertModel(b, add_pooling_layer=False).eval()
    with torch.no_grad():
        for p in bm.parameters():
            if p.dim() >= 2:
                p.mul_(3.0)
            else:
                p.add_(0.05 * torch.randn_like(p))
    ids = torch.randint(1, 1000, (2, 52))
    ids[0, 20:] = 0
    ids[1, 45:] = 0
    with torch.no_grad():
        ob = bm(ids, attention_mask=(ids != 0).long(), output_hidden_states=True)
    d = {"w." + k: v.numpy() for k, v in bm.state_dict().items()}
    d.update(ids=ids.numpy(), last_hidden_state=ob.last_hidden_state.numpy())
    for i, h in enumerate(ob.hidden_states):
        d[f"hidden_{i}"] = h.numpy()
    np.savez_compressed(os.path.join(OUT, "text_bert.npz"), **d)
    torch.manual_seed(2)
    x = XLMRobertaConfig(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                         max_position_embeddings=66, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5)
    xm = XLMRobertaModel(x, add_pooling_layer=False).eval()
    with torch.no_grad():
        for p in xm.parameters():
            if p.dim() >= 2:
                p.mul_(3.0)
            else:
                p.add_(0.05 * torch.randn_like(p))
    ids = torch.randint(2, 1000, (2, 64))
    ids[0, 25:] = 1
    ids[1, 50:] = 1
    with torch.no_grad():
        ox = xm(ids, attention_mask=(ids != 1).long(), output_hidden_states=True)
    d = {"w." + k: v.numpy() for k, v in xm.state_dict().items()}
    d.update(ids=ids.numpy(), last_hidden_state=ox.last_hidden_state.numpy())
    np.savez_compressed(os.path.join(OUT, "text_xlmr.npz"), **d)
    print("wrote", [(f, os.path.getsize(os.path.join(OUT, f))) for f in ("text_clip.npz", "text_bert.npz", "text_xlmr.npz")])


def t5():
    """T5EncoderModel in the mT5 (T5 v1.1) form: gated-gelu FF, d_kv 64, heads * d_kv != d_model, right padding with id 0"""
    from transformers import T5Config, T5EncoderModel
    torch.manual_seed(3)
    c = T5Config(vocab_size=1000, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=3, relative_attention_num_buckets=8,
                 relative_attention_max_distance=20, feed_forward_proj="gated-gelu", layer_norm_epsilon=1e-6, pad_token_id=0,
                 eos_token_id=1, dropout_rate=0.0, tie_word_embeddings=False)
    m = T5EncoderModel(c).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "layer_norm" in n:
                p.add_(0.1 * torch.randn_like(p))
            elif "relative_attention_bias" in n:
                p.copy_(torch.randn_like(p))
            elif "shared" in n or "embed_tokens" in n:
                p.copy_(torch.randn_like(p))
            else:                                          # unscaled scores: keep q.k of order one
                p.copy_(torch.randn_like(p) * (1.0 / p.shape[1]) ** 0.5 * (0.6 if ".q." in n or ".k." in n else 1.0))
    ids = torch.randint(2, 1000, (2, 40))
    ids[0, 12:] = 0; ids[0, 11] = 1
    ids[1, 37:] = 0; ids[1, 36] = 1
    with torch.no_grad():
        o = m.encoder(ids, attention_mask=ids.ne(0), output_hidden_states=True)
    d = {"w." + k: v.numpy() for k, v in m.state_dict().items()}
    d.update(ids=ids.numpy(), last_hidden_state=o[0].numpy())
    for i, h in enumerate(o.hidden_states):
        d[f"hidden_{i}"] = h.numpy()
    np.savez_compressed(os.path.join(OUT, "text_t5.npz"), **d)
    print("wrote text_t5.npz", os.path.getsize(os.path.join(OUT, "text_t5.npz")))


if __name__ == "__main__":
    t5() if sys.argv[1:] == ["t5"] else main()
