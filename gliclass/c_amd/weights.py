"""Weight blob (.glcw) writer/reader + deterministic synthetic weights.

The reference loads `onnx/model.onnx` (`/root/reference/include/paths.h:5`,
`/root/reference/src/model.c:269`); this engine loads a flat blob instead:

    [glcw_header 256 B][glcw_tensor x n (160 B each)][pad to 64][tensor data, each 64-B aligned]

Tensor names follow HF `DebertaV2Model.state_dict()` (prefix-free) plus the GLiClass head
(`text_projector.linear_{1,2}`, `classes_projector.linear_{1,2}`), so a converter from a real
`model.safetensors` is a rename (see `from_state_dict`).  The C reader is
`gliclass/c_amd/host/glc_weights.c`; layouts are mirrored in `include/gliclass_hip.h`.
"""
import math
import struct
from typing import Dict, List, Tuple

import numpy as np

from .config import GLiClassConfig, BACKBONE_DECODER, SCORER_WEIGHTED_DOT, SCORER_MLP, SCORER_MLP_HIDDEN
from . import prng

MAGIC = b"GLCW\x00\x01\x00\x00"
HEADER_BYTES = 256
TENSOR_REC_BYTES = 160
# int32 slots of the header, in order (mirrors struct glcw_header in include/gliclass_hip.h)
_INT_FIELDS = ["vocab", "hidden", "layers", "heads", "head_dim", "inter", "pos_buckets", "max_rel_pos",
               "pad_id", "cls_id", "sep_id", "class_token_index", "text_token_index",
               "pooling", "scorer", "embed_class_token", "normalize_features", "backbone", "kv_heads", "causal"]
_F32_FIELDS = ["ln_eps", "logit_scale", "rope_theta"]


def tensor_specs(cfg: GLiClassConfig) -> List[Tuple[str, Tuple[int, ...], float, float]]:
    """(name, shape, uniform amplitude, mean) for every tensor, in blob order.

    Amplitudes are chosen so random models are numerically *interesting*: attention is peaky
    (q/k targets 2.0 std => score std ~ 2-4 after the 1/sqrt(3d) scale), residual branches are
    O(1), and logits land in the sigmoid's sensitive range (|logit| ~ 1-3).
    """
    H, I, L = cfg.hidden, cfg.inter, cfg.layers
    s3 = math.sqrt(3.0)

    def lin(t, fan_in):
        return s3 * t / math.sqrt(fan_in)

    def head_specs():
        t2 = math.sqrt(1.5 / math.sqrt(H)) / 0.7
        out = []
        for proj in ("text_projector", "classes_projector"):
            out += [
                (proj + ".linear_1.weight", (H, H), lin(1.0, H), 0.0),
                (proj + ".linear_1.bias", (H,), 0.1, 0.0),
                (proj + ".linear_2.weight", (H, H), lin(t2, H), 0.0),
                (proj + ".linear_2.bias", (H,), 0.02, 0.0),
            ]
        # the scorer's own tensors (include/gliclass_hip.h; mirrors head_spec in host/glc_weights.c)
        if cfg.scorer == SCORER_WEIGHTED_DOT:
            out += [
                ("scorer.proj_text.weight", (2 * H, H), lin(1.0, H), 0.0), ("scorer.proj_text.bias", (2 * H,), 0.05, 0.0),
                ("scorer.proj_label.weight", (2 * H, H), lin(1.0, H), 0.0), ("scorer.proj_label.bias", (2 * H,), 0.05, 0.0),
                ("scorer.out_mlp.0.weight", (4 * H, 3 * H), lin(1.0, 3 * H), 0.0), ("scorer.out_mlp.0.bias", (4 * H,), 0.05, 0.0),
                ("scorer.out_mlp.3.weight", (1, 4 * H), lin(2.0, 4 * H), 0.0), ("scorer.out_mlp.3.bias", (1,), 0.1, 0.0),
            ]
        elif cfg.scorer == SCORER_MLP:
            Mh = SCORER_MLP_HIDDEN
            out += [
                ("scorer.mlp.0.weight", (Mh, 2 * H), lin(1.0, 2 * H), 0.0), ("scorer.mlp.0.bias", (Mh,), 0.05, 0.0),
                ("scorer.mlp.2.weight", (Mh // 2, Mh), lin(1.5, Mh), 0.0), ("scorer.mlp.2.bias", (Mh // 2,), 0.05, 0.0),
                ("scorer.mlp.4.weight", (1, Mh // 2), lin(3.0, Mh // 2), 0.0), ("scorer.mlp.4.bias", (1,), 0.1, 0.0),
            ]
        return out

    if cfg.backbone == BACKBONE_DECODER:
        # HF Qwen2Model.state_dict() names (prefix-free).  q/k amplitudes give score std ~ 2-3 after 1/sqrt(d).
        nqd, nkvd = cfg.heads * cfg.head_dim, cfg.kv_heads * cfg.head_dim
        specs = [("embed_tokens.weight", (cfg.vocab, H), 1.0, 0.0)]
        for i in range(L):
            p = f"layers.{i}."
            specs += [
                (p + "input_layernorm.weight", (H,), 0.2, 1.0),
                (p + "self_attn.q_proj.weight", (nqd, H), lin(1.6, H), 0.0),
                (p + "self_attn.q_proj.bias", (nqd,), 0.1, 0.0),
                (p + "self_attn.k_proj.weight", (nkvd, H), lin(1.6, H), 0.0),
                (p + "self_attn.k_proj.bias", (nkvd,), 0.1, 0.0),
                (p + "self_attn.v_proj.weight", (nkvd, H), lin(1.0, H), 0.0),
                (p + "self_attn.v_proj.bias", (nkvd,), 0.1, 0.0),
                (p + "self_attn.o_proj.weight", (H, nqd), lin(0.7, nqd), 0.0),
                (p + "post_attention_layernorm.weight", (H,), 0.2, 1.0),
                (p + "mlp.gate_proj.weight", (I, H), lin(1.0, H), 0.0),
                (p + "mlp.up_proj.weight", (I, H), lin(1.0, H), 0.0),
                (p + "mlp.down_proj.weight", (H, I), lin(0.7, I), 0.0),
            ]
        specs += [("norm.weight", (H,), 0.2, 1.0)]
        return specs + head_specs()

    specs = [
        ("embeddings.word_embeddings.weight", (cfg.vocab, H), 1.0, 0.0),
        ("embeddings.LayerNorm.weight", (H,), 0.2, 1.0),
        ("embeddings.LayerNorm.bias", (H,), 0.1, 0.0),
        ("encoder.rel_embeddings.weight", (2 * cfg.att_span, H), 1.0, 0.0),
        ("encoder.LayerNorm.weight", (H,), 0.2, 1.0),
        ("encoder.LayerNorm.bias", (H,), 0.1, 0.0),
    ]
    for i in range(L):
        p = f"encoder.layer.{i}."
        specs += [
            (p + "attention.self.query_proj.weight", (H, H), lin(2.0, H), 0.0),
            (p + "attention.self.query_proj.bias", (H,), 0.1, 0.0),
            (p + "attention.self.key_proj.weight", (H, H), lin(2.0, H), 0.0),
            (p + "attention.self.key_proj.bias", (H,), 0.1, 0.0),
            (p + "attention.self.value_proj.weight", (H, H), lin(1.0, H), 0.0),
            (p + "attention.self.value_proj.bias", (H,), 0.1, 0.0),
            (p + "attention.output.dense.weight", (H, H), lin(1.0, H), 0.0),
            (p + "attention.output.dense.bias", (H,), 0.1, 0.0),
            (p + "attention.output.LayerNorm.weight", (H,), 0.2, 1.0),
            (p + "attention.output.LayerNorm.bias", (H,), 0.1, 0.0),
            (p + "intermediate.dense.weight", (I, H), lin(1.0, H), 0.0),
            (p + "intermediate.dense.bias", (I,), 0.1, 0.0),
            (p + "output.dense.weight", (H, I), lin(1.0, I), 0.0),
            (p + "output.dense.bias", (H,), 0.1, 0.0),
            (p + "output.LayerNorm.weight", (H,), 0.2, 1.0),
            (p + "output.LayerNorm.bias", (H,), 0.1, 0.0),
        ]
    return specs + head_specs()


def make_weights(cfg: GLiClassConfig, seed: int = 42) -> Dict[str, np.ndarray]:
    out = {}
    for name, shape, amp, mean in tensor_specs(cfg):
        n = int(np.prod(shape))
        out[name] = prng.uniform_f32(seed, name, n, amp, mean).reshape(shape)
    return out


def _pack_header(cfg: GLiClassConfig, n_tensors: int) -> bytes:
    d = cfg.asdict()
    b = MAGIC + struct.pack("<II", 2, n_tensors)     # version 2: + backbone, kv_heads, causal, rope_theta
    b += struct.pack("<%di" % len(_INT_FIELDS), *[int(d[k]) for k in _INT_FIELDS])
    b += struct.pack("<%df" % len(_F32_FIELDS), *[float(d[k]) for k in _F32_FIELDS])
    assert len(b) <= HEADER_BYTES
    return b + b"\x00" * (HEADER_BYTES - len(b))


def write_blob(path: str, cfg: GLiClassConfig, tensors: Dict[str, np.ndarray]) -> None:
    names = [s[0] for s in tensor_specs(cfg)]
    missing = [n for n in names if n not in tensors]
    if missing:
        raise KeyError(f"missing tensors: {missing[:4]}...")
    table_end = HEADER_BYTES + TENSOR_REC_BYTES * len(names)
    off = (table_end + 63) // 64 * 64
    recs, offs = [], []
    for n in names:
        a = np.ascontiguousarray(tensors[n], dtype=np.float32)
        shape = list(a.shape) + [0] * (4 - a.ndim)
        nm = n.encode("utf-8")
        assert len(nm) < 96
        recs.append(nm + b"\x00" * (96 - len(nm)) + struct.pack("<II4QQQ", 0, a.ndim, *shape, off, a.nbytes) + b"\x00" * 8)
        assert len(recs[-1]) == TENSOR_REC_BYTES
        offs.append(off)
        off = (off + a.nbytes + 63) // 64 * 64
    with open(path, "wb") as f:
        f.write(_pack_header(cfg, len(names)))
        for r in recs:
            f.write(r)
        for n, o in zip(names, offs):
            f.seek(o)
            f.write(np.ascontiguousarray(tensors[n], dtype=np.float32).tobytes())
        f.truncate(off)


def read_blob(path: str) -> Tuple[GLiClassConfig, Dict[str, np.ndarray]]:
    raw = np.fromfile(path, dtype=np.uint8)
    hdr = raw[:HEADER_BYTES].tobytes()
    if hdr[:8] != MAGIC:
        raise ValueError("not a GLCW blob")
    ver, n_t = struct.unpack_from("<II", hdr, 8)
    if ver != 2:
        raise ValueError(f"unsupported GLCW version {ver}")
    ints = struct.unpack_from("<%di" % len(_INT_FIELDS), hdr, 16)
    flts = struct.unpack_from("<%df" % len(_F32_FIELDS), hdr, 16 + 4 * len(_INT_FIELDS))
    kw = dict(zip(_INT_FIELDS, ints))
    kw.update(dict(zip(_F32_FIELDS, flts)))
    cfg = GLiClassConfig(name="blob", **kw)
    tensors = {}
    for i in range(n_t):
        rec = raw[HEADER_BYTES + i * TENSOR_REC_BYTES: HEADER_BYTES + (i + 1) * TENSOR_REC_BYTES].tobytes()
        name = rec[:96].split(b"\x00", 1)[0].decode()
        _, ndim, s0, s1, s2, s3, off, nb = struct.unpack_from("<II4QQQ", rec, 96)
        shape = (s0, s1, s2, s3)[:ndim]
        tensors[name] = raw[off:off + nb].view(np.float32).reshape(shape)
    return cfg, tensors


def from_state_dict(sd: Dict[str, "np.ndarray"], cfg: GLiClassConfig) -> Dict[str, np.ndarray]:
    """Rename a (GLiClass or bare DebertaV2Model) state_dict to blob names.

    Accepts the prefixes HF / gliclass checkpoints use (`deberta.`, `encoder_model.model.`, none).
    """
    out = {}
    want = [s[0] for s in tensor_specs(cfg)]
    for n in want:
        for pre in ("", "deberta.", "encoder_model.model.", "model.encoder_model.model.", "model.", "decoder_model.model."):
            if pre + n in sd:
                v = sd[pre + n]
                out[n] = v.detach().cpu().float().numpy() if hasattr(v, "detach") else np.asarray(v, np.float32)
                break
        else:
            raise KeyError(n)
    return out
