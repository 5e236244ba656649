"""Thin Python handle over the HIP engine C-ABI (tests, bench.py, smoke).  All compute happens in
libgliclass_hip.so; this module only marshals numpy buffers."""
import ctypes as C

import numpy as np

from . import _lib
from .config import GLiClassConfig
from .weights import tensor_specs

DTYPES = {"f32": 0, "bf16": 1, "f16": 2}
DTYPE_NAMES = {v: k for k, v in DTYPES.items()}


def to_c_config(cfg: GLiClassConfig) -> _lib.ModelConfig:
    return _lib.ModelConfig(cfg.vocab, cfg.hidden, cfg.layers, cfg.heads, cfg.head_dim, cfg.inter, cfg.pos_buckets,
                            cfg.max_rel_pos, cfg.pad_id, cfg.cls_id, cfg.sep_id, cfg.class_token_index, cfg.text_token_index,
                            cfg.pooling, cfg.scorer, cfg.embed_class_token, cfg.normalize_features, cfg.backbone, cfg.kv_heads,
                            cfg.causal, cfg.ln_eps, cfg.logit_scale, cfg.rope_theta)


def delta_table(S, bucket_size=256, max_position=512):
    out = np.zeros(2 * S - 1, np.int32)
    _lib.hip().glc_delta_table(S, bucket_size, max_position, out.ctypes.data)
    return out


class Engine:
    def __init__(self, cfg: GLiClassConfig, tensors, dtype="f32", device=0):
        self.L = _lib.hip()
        self.cfg = cfg
        self.dtype = dtype
        names = [s[0] for s in tensor_specs(cfg)]
        arrs = [np.ascontiguousarray(tensors[n], np.float32) for n in names]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        cc = to_c_config(cfg)
        self.h = self.L.glc_engine_create(C.byref(cc), ptrs, len(arrs), device, DTYPES[dtype])
        if not self.h:
            raise RuntimeError("glc_engine_create: " + self.L.glc_last_error().decode())

    @classmethod
    def from_spec(cls, cfg: GLiClassConfig, spec: str, dtype="f32", device=0):
        """Engine from a model path / "synthetic:<config>[:seed]" through the C weight source (glc_weights_load, the same
        code create_ort_session uses) — no Python-side copy of the tensors, which matters for the 1.5 B-parameter config."""
        M = _lib.model()
        w = _lib.Weights()
        if M.glc_weights_load(spec.encode(), C.byref(w)) != 0:
            raise RuntimeError(f"glc_weights_load({spec}) failed")
        self = cls.__new__(cls)
        self.L = _lib.hip()
        self.cfg = cfg
        self.dtype = dtype
        try:
            self.h = self.L.glc_engine_create(C.byref(w.cfg), C.cast(w.tensors, C.POINTER(C.c_void_p)), w.n_tensors, device, DTYPES[dtype])
        finally:
            M.glc_weights_free(C.byref(w))
        if not self.h:
            raise RuntimeError("glc_engine_create: " + self.L.glc_last_error().decode())
        return self

    def close(self):
        if getattr(self, "h", None):
            self.L.glc_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _err(self, what):
        return RuntimeError(f"{what}: {self.L.glc_last_error().decode()}")

    def forward(self, ids, mask, c_alloc=None):
        ids = np.ascontiguousarray(ids, np.int64)
        mask = np.ascontiguousarray(mask, np.int64)
        B, S = ids.shape
        if c_alloc is None:
            c_alloc = int((ids == self.cfg.class_token_index).sum(1).max())
        logits = np.zeros((B, max(c_alloc, 1)), np.float32)
        c_out = C.c_int(0)
        if self.L.glc_engine_forward(self.h, ids.ctypes.data, mask.ctypes.data, B, S, logits.ctypes.data, c_alloc, C.byref(c_out)) != 0:
            raise self._err("glc_engine_forward")
        self.last_c = c_out.value
        return logits[:, :c_alloc]

    def hidden(self, which, B, S):
        out = np.zeros((B, S, self.cfg.hidden), np.float32)
        if self.L.glc_debug_get_hidden(self.h, which, out.ctypes.data, out.size) != 0:
            raise self._err("glc_debug_get_hidden")
        return out

    def set_prune_last_layer(self, on=True):
        self.L.glc_engine_set_prune_last_layer(self.h, int(on))

    def set_length_buckets(self, max_groups):
        if self.L.glc_engine_set_length_buckets(self.h, int(max_groups)) != 0:
            raise self._err("glc_engine_set_length_buckets")

    def set_group_split(self, mode):
        """fp32 mode: 0 = plain fp32 activations + 128-tile split GEMMs, 1 = auto, 2 = group-split pipeline whenever the shapes allow"""
        if self.L.glc_debug_set_group_split(self.h, int(mode)) != 0:
            raise self._err("glc_debug_set_group_split")

    def last_group_split(self):
        return bool(self.L.glc_debug_last_forward_group_split(self.h))

    def last_ln_folded(self):
        return bool(self.L.glc_debug_last_forward_ln_folded(self.h))

    def set_ln_fused(self, on):
        """group-split pipeline: LayerNorm folded into the GEMMs around it (default) or as kernels of its own"""
        if self.L.glc_debug_set_ln_fused(self.h, int(bool(on))) != 0:
            raise self._err("glc_debug_set_ln_fused")

    def set_precision_mask(self, mask):
        """default mode, group-split pipeline: round operand groups to f16 (bits: include/gliclass_hip.h glc_debug_set_precision_mask)"""
        if self.L.glc_debug_set_precision_mask(self.h, int(mask)) != 0:
            raise self._err("glc_debug_set_precision_mask")

    def set_mx(self, on):
        """MX cross-term pipeline on / off (engine created under GLICLASS_MX=1 or =build)"""
        if self.L.glc_debug_set_mx(self.h, int(bool(on))) != 0:
            raise self._err("glc_debug_set_mx")

    def set_mx_attention(self, on):
        """MX pipeline: attention on MX tiles (default) or on split-f16 units"""
        self.L.glc_debug_set_mx_attention(self.h, int(bool(on)))

    def last_mx(self):
        return bool(self.L.glc_debug_last_forward_mx(self.h))

    def last_mx_attention(self):
        """the last forward's attention ran on MX tiles (attention_mx.hip)"""
        return bool(self.L.glc_debug_last_forward_mx_attention(self.h))

    def fp8_range_retries(self):
        """host-buffer forwards repeated on the split-f16 kernels because an activation left the fp8 range of the MX operand images"""
        return int(self.L.glc_debug_fp8_range_retries(self.h))

    def fp8_range_sticky(self):
        """the engine has left the MX pipeline for good (outlier channels in consecutive forwards)"""
        return bool(self.L.glc_debug_fp8_range_sticky(self.h))

    def activation_exponent(self):
        """exponent of the MX pipeline's activation rows: 0, or -5 once a forward left the fp8 range (the guard's first answer)"""
        return int(self.L.glc_debug_activation_exponent(self.h))

    def set_mx2(self, on):
        """developer builds (make DEV=1): MX attention on the bucket-space kernel (csrc/dev/attention_mx2.hip) or on the band kernel; the product library refuses on=True"""
        if self.L.glc_debug_set_mx2(self.h, int(bool(on))) != 0:
            raise RuntimeError(self.L.glc_last_error().decode())

    def range_retries(self):
        """host-buffer forwards repeated with the norms unfused because the folded forward came out non-finite"""
        return int(self.L.glc_debug_range_retries(self.h))

    def keep_hidden(self, on=True):
        self.L.glc_debug_keep_hidden(self.h, int(on))

    def set_attention_impl(self, impl):
        if self.L.glc_debug_set_attention_impl(self.h, impl) != 0:
            raise self._err("glc_debug_set_attention_impl")

    # ---- device-resident path (bench) ----
    def dev_alloc(self, nbytes):
        p = self.L.glc_device_malloc(self.h, nbytes)
        if not p:
            raise self._err("glc_device_malloc")
        return p

    def dev_free(self, p):
        self.L.glc_device_free(self.h, p)

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        if self.L.glc_memcpy_h2d(self.h, dptr, arr.ctypes.data, arr.nbytes) != 0:
            raise self._err("glc_memcpy_h2d")

    def d2h(self, arr, dptr):
        if self.L.glc_memcpy_d2h(self.h, arr.ctypes.data, dptr, arr.nbytes) != 0:
            raise self._err("glc_memcpy_d2h")

    def forward_device(self, d_ids, d_mask, B, S, Cn, d_logits):
        if self.L.glc_engine_forward_device(self.h, d_ids, d_mask, B, S, Cn, d_logits) != 0:
            raise self._err("glc_engine_forward_device")

    def sync(self):
        if self.L.glc_engine_sync(self.h) != 0:
            raise self._err("glc_engine_sync")

    def timer_start(self):
        self.L.glc_timer_start(self.h)

    def timer_stop_ms(self):
        return float(self.L.glc_timer_stop_ms(self.h))

    def profile(self, on=True):
        self.L.glc_profile_enable(self.h, int(on))

    def profile_read(self):
        n = 16
        names = (C.c_char_p * n)()
        ms = (C.c_float * n)()
        cnt = (C.c_int * n)()
        k = self.L.glc_profile_read(self.h, names, ms, cnt, n)
        return {names[i].decode(): (float(ms[i]), int(cnt[i])) for i in range(k)}
