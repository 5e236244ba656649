"""Model configurations for the GLiClass uni-encoder hot path.

The reference repo never states backbone dimensions (they live in the HF hub
`config.json` that `/root/reference/run_GLiClass.sh:34-36` downloads at run time);
the values below are the standard DeBERTa-v3 shapes listed in SURVEY.md §8a.
Every field is carried in the weight-blob header (`weights.py`), so a real
checkpoint overrides them — nothing here is a compile-time constant of the engine.
"""
from dataclasses import dataclass, asdict

# scorer / pooling enums shared with include/gliclass_hip.h
POOL_FIRST, POOL_AVG = 0, 1
SCORER_DOT = 0


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

    def __post_init__(self):
        assert self.hidden == self.heads * self.head_dim
        if self.class_token_index < 0:
            object.__setattr__(self, "class_token_index", self.vocab - 2)
        if self.text_token_index < 0:
            object.__setattr__(self, "text_token_index", self.vocab - 1)

    @property
    def att_span(self) -> int:
        return self.pos_buckets if self.pos_buckets > 0 else self.max_rel_pos

    def flops_per_seq(self, S: int, C: int) -> float:
        """F_seq = L*S*(8H^2 + 4HI + 4SH + 4PH) + 8H^2(1+C)."""
        H, I, L, P = self.hidden, self.inter, self.layers, 2 * self.att_span
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
}
