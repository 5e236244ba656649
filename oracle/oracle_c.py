"""ctypes front-end for oracle/libgliclass_oracle.so (ORACLE — test infrastructure only).

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from gliclass.c_amd.config import GLiClassConfig, BACKBONE_DECODER
from gliclass.c_amd.weights import tensor_specs

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgliclass_oracle.so")


class GloConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab", "hidden", "layers", "heads", "head_dim", "inter",
                                         "pos_buckets", "max_rel_pos", "pad_id", "class_token_index",
                                         "embed_class_token", "pooling", "normalize_features")] + \
               [("ln_eps", C.c_float), ("logit_scale", C.c_float)] + \
               [(n, C.c_int32) for n in ("backbone", "kv_heads", "causal")] + [("rope_theta", C.c_float), ("scorer", C.c_int32)]


def build(force=False):
    src = os.path.join(_HERE, "gliclass_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgliclass_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def effective_cpus():
    """CPUs this process can really use (affinity, cgroup quota): the OpenMP team of the oracle is sized to it — a 256-thread team
    spinning inside a 16-CPU quota made one 2-second forward take minutes on a GPU box."""
    from gliclass.c_amd.hostinfo import effective_cpus as f
    return f()


def lib():
    global _lib
    if _lib is None:
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")      # honoured if libgomp is first initialised by this load
        _lib = C.CDLL(build())
        _lib.glo_set_threads.argtypes = [C.c_int]
        _lib.glo_set_threads(min(effective_cpus(), 64))
        _lib.glo_forward.restype = C.c_int
        _lib.glo_forward.argtypes = [C.POINTER(GloConfig), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        _lib.glo_forward_decoder.restype = C.c_int
        _lib.glo_forward_decoder.argtypes = [C.POINTER(GloConfig), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]
        _lib.glo_delta_table.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.glo_sigmoid.restype = C.c_float
        _lib.glo_sigmoid.argtypes = [C.c_float]
        _lib.glo_num_threads.restype = C.c_int
    return _lib


def _cfg(cfg: GLiClassConfig) -> GloConfig:
    return GloConfig(cfg.vocab, cfg.hidden, cfg.layers, cfg.heads, cfg.head_dim, cfg.inter, cfg.pos_buckets,
                     cfg.max_rel_pos, cfg.pad_id, cfg.class_token_index, cfg.embed_class_token, cfg.pooling,
                     cfg.normalize_features, cfg.ln_eps, cfg.logit_scale, cfg.backbone, cfg.kv_heads, cfg.causal,
                     cfg.rope_theta, cfg.scorer)


def delta_table(S, bucket_size=256, max_position=512):
    out = np.zeros(2 * S - 1, np.int32)
    lib().glo_delta_table(S, bucket_size, max_position, out.ctypes.data)
    return out


def forward(cfg: GLiClassConfig, tensors, ids, mask, want_hidden=False, want_scores=False, c_alloc=None):
    """fp32 CPU forward.  Returns logits [B,C] (+ hidden [(L+1),B,S,H], + layer-0 head-0 scores [S,S])."""
    ids = np.ascontiguousarray(ids, np.int64)
    mask = np.ascontiguousarray(mask, np.int64)
    B, S = ids.shape
    names = [s[0] for s in tensor_specs(cfg)]
    arrs = [np.ascontiguousarray(tensors[n], np.float32) for n in names]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    c_alloc = int(c_alloc or max(1, int((ids == cfg.class_token_index).sum(1).max())))
    logits = np.zeros((B, c_alloc), np.float32)
    hidden = np.zeros((cfg.layers + 1, B, S, cfg.hidden), np.float32) if want_hidden else None
    scores = np.zeros((S, S), np.float32) if want_scores else None
    c_out = C.c_int(0)
    cc = _cfg(cfg)
    if cfg.backbone == BACKBONE_DECODER:
        if want_scores:
            raise ValueError("scores dump exists for the encoder backbone only")
        rc = lib().glo_forward_decoder(C.byref(cc), ptrs, ids.ctypes.data, mask.ctypes.data, B, S, logits.ctypes.data,
                                       c_alloc, C.byref(c_out), hidden.ctypes.data if want_hidden else None)
    else:
        rc = lib().glo_forward(C.byref(cc), ptrs, ids.ctypes.data, mask.ctypes.data, B, S, logits.ctypes.data, c_alloc,
                               C.byref(c_out), hidden.ctypes.data if want_hidden else None,
                               scores.ctypes.data if want_scores else None)
    if rc != 0:
        raise RuntimeError(f"glo_forward failed rc={rc}")
    out = [logits[:, :min(c_out.value, c_alloc)]]
    if want_hidden:
        out.append(hidden)
    if want_scores:
        out.append(scores)
    return out[0] if len(out) == 1 else tuple(out)
