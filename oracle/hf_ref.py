"""ORACLE (test infrastructure only — never imported by the product path).

Pins the encoder arithmetic on HuggingFace `transformers.DebertaV2Model`, which is the exact
module `torch.onnx.export` traced to make the reference's `onnx/model.onnx`
(`/root/reference/ONNX_CONVERTING/convert_to_onnx.py:48,71-79`), and restates the GLiClass
uni-encoder head in plain torch.

PARITY STATUS
  * encoder: pinned on transformers' DebertaV2Model (third-party; `transformers` is unpinned in the
    reference, `convert_to_onnx.py:6`; version used to make tests/golden is recorded in each fixture).
  * head   : **parity unpinned** — the `gliclass` PyPI package (unpinned, `convert_to_onnx.py:5`) is
    not available offline and its golden logits live on the HF hub (`run_GLiClass.sh:34`), so the
    head below restates the published algorithm from its call site
    (`GLiClassModel(...).logits`, `convert_to_onnx.py:11-13`).
"""
import numpy as np
import torch

from gliclass.c_amd.config import GLiClassConfig, POOL_FIRST, POOL_AVG, POOL_LAST, BACKBONE_DECODER


def build_hf_decoder(cfg: GLiClassConfig, tensors):
    """Qwen2Model with the blob's tensors (decoder backbone).  HF's Qwen2 is always causal."""
    from transformers import Qwen2Config, Qwen2Model
    assert cfg.causal, "transformers' Qwen2Model has no bidirectional mode: causal = 0 is unpinned"
    hc = Qwen2Config(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                     num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_heads, intermediate_size=cfg.inter,
                     max_position_embeddings=32768, rope_theta=cfg.rope_theta, rms_norm_eps=cfg.ln_eps,
                     attention_dropout=0.0, use_sliding_window=False, tie_word_embeddings=False, pad_token_id=cfg.pad_id,
                     attn_implementation="eager")
    assert hc.hidden_size // hc.num_attention_heads == cfg.head_dim
    with torch.no_grad():
        m = Qwen2Model(hc).eval()
        sd = m.state_dict()
        for k in sd:
            if k not in tensors:
                raise KeyError(k)
            sd[k] = torch.from_numpy(np.asarray(tensors[k])).clone()
        m.load_state_dict(sd)
    return m


def build_hf_model(cfg: GLiClassConfig, tensors):
    if cfg.backbone == BACKBONE_DECODER:
        return build_hf_decoder(cfg, tensors)
    from transformers import DebertaV2Config, DebertaV2Model
    hc = DebertaV2Config(
        vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
        num_attention_heads=cfg.heads, intermediate_size=cfg.inter,
        max_position_embeddings=cfg.max_rel_pos, relative_attention=True,
        position_buckets=cfg.pos_buckets, norm_rel_ebd="layer_norm", share_att_key=True,
        pos_att_type="p2c|c2p", position_biased_input=False, layer_norm_eps=cfg.ln_eps,
        type_vocab_size=0, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
        hidden_act="gelu", pad_token_id=cfg.pad_id)
    with torch.no_grad():
        m = DebertaV2Model(hc).eval()
        sd = m.state_dict()
        for k in sd:
            if k in tensors:
                sd[k] = torch.from_numpy(np.asarray(tensors[k])).clone()
            elif "position_ids" in k:
                pass
            else:
                raise KeyError(k)
        m.load_state_dict(sd)
    return m


def features_projector(x, w1, b1, w2, b2):
    """gliclass FeaturesProjector: Linear -> GELU(erf) -> Linear (dropout = identity in eval)."""
    h = torch.nn.functional.gelu(x @ w1.T + b1)
    return h @ w2.T + b2


def gliclass_head(cfg: GLiClassConfig, tensors, hidden, ids, mask):
    """hidden [B,S,H] -> logits [B,C].  SURVEY.md §8a row a12 (uni-encoder, scorer 'simple')."""
    t = {k: torch.from_numpy(np.asarray(v)) for k, v in tensors.items() if "projector" in k}
    B, S, H = hidden.shape
    cls_mask = ids == cfg.class_token_index
    counts = cls_mask.sum(-1)
    C = int(counts.max()) if B else 0
    classes = torch.zeros(B, C, H, dtype=hidden.dtype)
    for b in range(B):
        pos = torch.nonzero(cls_mask[b]).flatten()
        if not cfg.embed_class_token:
            pos = pos + 1
        classes[b, :len(pos)] = hidden[b, pos]
    if cfg.pooling == POOL_FIRST:
        pooled = hidden[:, 0, :]
    elif cfg.pooling == POOL_AVG:           # mean over the attended positions
        mk = mask.to(hidden.dtype).unsqueeze(-1)
        pooled = (hidden * mk).sum(1) / mk.sum(1).clamp(min=1)
    elif cfg.pooling == POOL_LAST:
        last = torch.stack([torch.nonzero(mask[b]).flatten()[-1] if mask[b].any() else torch.tensor(0) for b in range(B)])
        pooled = hidden[torch.arange(B), last]
    else:
        raise NotImplementedError
    pooled = features_projector(pooled, t["text_projector.linear_1.weight"], t["text_projector.linear_1.bias"],
                                t["text_projector.linear_2.weight"], t["text_projector.linear_2.bias"])
    classes = features_projector(classes, t["classes_projector.linear_1.weight"], t["classes_projector.linear_1.bias"],
                                 t["classes_projector.linear_2.weight"], t["classes_projector.linear_2.bias"])
    if cfg.normalize_features:
        pooled = pooled / (pooled.norm(dim=-1, keepdim=True) + 1e-8)
        classes = classes / (classes.norm(dim=-1, keepdim=True) + 1e-8)
    logits = torch.einsum("bd,bcd->bc", pooled, classes)
    if cfg.normalize_features:
        logits = logits * cfg.logit_scale
    return logits


@torch.no_grad()
def forward(cfg: GLiClassConfig, tensors, ids, mask, model=None, dtype=torch.float32, want_hidden=False):
    """ids/mask: int64 numpy [B,S].  Returns logits [B,C] (numpy, f32) and optionally all hidden states."""
    m = model if model is not None else build_hf_model(cfg, tensors)
    if dtype != torch.float32:
        m = m.to(dtype)
    tid = torch.from_numpy(np.asarray(ids, np.int64))
    tm = torch.from_numpy(np.asarray(mask, np.int64))
    out = m(input_ids=tid, attention_mask=tm, output_hidden_states=True)
    hs = [h.to(dtype) for h in out.hidden_states]          # [emb, layer0, ..., layerL-1]
    tens = tensors if dtype == torch.float32 else {k: np.asarray(v, np.float64) for k, v in tensors.items()
                                                  if "projector" in k}
    logits = gliclass_head(cfg, tens, hs[-1], tid, tm)
    if want_hidden:
        return logits.float().numpy(), [h.float().numpy() for h in hs]
    return logits.float().numpy()
