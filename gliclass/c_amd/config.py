"""Model configurations for the GLiClass hot path (encoder backbone, and the decoder backbone of BASELINE.json configs[4]).

The reference repo never states backbone dimensions (they live in the HF hub
`config.json` that `/root/reference/run_GLiClass.sh:34-36` downloads at run time);
the values below are the standard DeBERTa-v3 shapes listed in SURVEY.md §8a.
Every field is carried in the weight-blob header (`weights.py`), so a real
checkpoint overrides them — nothing here is a compile-time constant of the engine.
"""
from dataclasses import dataclass, asdict

# scorer / pooling enums shared with include/gliclass_hip.h
POOL_FIRST, POOL_AVG, POOL_LAST = 0, 1, 2
SCORER_DOT = 0
SCORER_WEIGHTED_DOT = 1
SCORER_MLP = 2
SCORER_MLP_HIDDEN = 256
SCORER_NAMES = {"simple": SCORER_DOT, "weighted-dot": SCORER_WEIGHTED_DOT, "mlp": SCORER_MLP}
BACKBONE_DEBERTA, BACKBONE_DECODER = 0, 1


@dataclass(frozen=True)
class GLiClassConfig:
    name: str
    vocab: int
    hidden: int
    layers: int
    heads: int
    inter: int
    head_dim: int = 64
    pos_buckets: int = 256          # DebertaV2Config.position_buckets
    max_rel_pos: int = 512          # max_relative_positions (= max_position_embeddings)
    ln_eps: float = 1e-7
    pad_id: int = 0
    cls_id: int = 1
    sep_id: int = 2
    class_token_index: int = -1     # id of "<<LABEL>>"   (/root/reference/src/preprocessor.c:68)
    text_token_index: int = -1      # id of "<<SEP>>"     (/root/reference/src/preprocessor.c:69)
    pooling: int = POOL_FIRST
    scorer: int = SCORER_DOT
    embed_class_token: int = 1
    normalize_features: int = 0
    logit_scale: float = 1.0
    # decoder-style backbone (Qwen2 arithmetic — RMSNorm, RoPE, grouped-query attention, SwiGLU);
    # ln_eps doubles as rms_norm_eps, `inter` is the SwiGLU width
    backbone: int = BACKBONE_DEBERTA
    kv_heads: int = 0               # 0 => heads (no grouping)
    causal: int = 1                 # BASELINE.json says causal; upstream may wrap decoders bidirectionally (unpinned) -> flag
    rope_theta: float = 1.0e6

    def __post_init__(self):
        assert self.hidden == self.heads * self.head_dim
        if self.kv_heads <= 0:
            object.__setattr__(self, "kv_heads", self.heads)
        assert self.heads % self.kv_heads == 0
        if self.class_token_index < 0:
            object.__setattr__(self, "class_token_index", self.vocab - 2)
        if self.text_token_index < 0:
            object.__setattr__(self, "text_token_index", self.vocab - 1)

    @property
    def att_span(self) -> int:
        return self.pos_buckets if self.pos_buckets > 0 else self.max_rel_pos

    def flops_per_seq(self, S: int, C: int) -> float:
        """F_seq = L*S*(8H^2 + 4HI + 4SH + 4PH) + 8H^2(1+C); decoder:
        L*S*(4H*nq*d + 4H*nkv*d + 6HI + 4S*nq*d*kappa) + head, kappa = 1/2 when causal."""
        H, I, L, P = self.hidden, self.inter, self.layers, 2 * self.att_span
        if self.backbone == BACKBONE_DECODER:
            nqd, nkvd, kappa = self.heads * self.head_dim, self.kv_heads * self.head_dim, (0.5 if self.causal else 1.0)
            return L * S * (4 * H * nqd + 4 * H * nkvd + 6 * H * I + 4 * S * nqd * kappa) + 8 * H * H * (1 + C)
        return L * S * (8 * H * H + 4 * H * I + 4 * S * H + 4 * P * H) + 8 * H * H * (1 + C)

    def asdict(self):
        return asdict(self)


CONFIGS = {
    # parity-fixture config: real head_dim (64) and real bucket geometry (256/512) so the
    # clamp / log-bucket paths are exercised, everything else tiny.
    "tiny": GLiClassConfig("tiny", vocab=515, hidden=128, layers=2, heads=2, inter=256),
    "mini": GLiClassConfig("mini", vocab=1027, hidden=256, layers=3, heads=4, inter=512),
    "small": GLiClassConfig("small", vocab=128003, hidden=768, layers=6, heads=12, inter=3072),
    "base": GLiClassConfig("base", vocab=128003, hidden=768, layers=12, heads=12, inter=3072),
    "large": GLiClassConfig("large", vocab=128003, hidden=1024, layers=24, heads=16, inter=4096),
    # decoder-style backbones: real head_dim (128) and grouping; dec-tiny/dec-mini are the parity-fixture configs,
    # qwen-1.5b is BASELINE.json configs[4] (Qwen2-1.5B shape; vocab 151 646 + <<LABEL>>, <<SEP>>)
    "dec-tiny": GLiClassConfig("dec-tiny", vocab=515, hidden=256, layers=2, heads=2, inter=512, head_dim=128, kv_heads=1,
                               ln_eps=1e-6, backbone=BACKBONE_DECODER, pooling=POOL_LAST),
    "dec-mini": GLiClassConfig("dec-mini", vocab=1027, hidden=512, layers=3, heads=4, inter=768, head_dim=128, kv_heads=2,
                               ln_eps=1e-6, backbone=BACKBONE_DECODER, pooling=POOL_LAST),
    "qwen-1.5b": GLiClassConfig("qwen-1.5b", vocab=151648, hidden=1536, layers=28, heads=12, inter=8960, head_dim=128,
                                kv_heads=2, ln_eps=1e-6, backbone=BACKBONE_DECODER, pooling=POOL_LAST),
}
