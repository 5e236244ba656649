"""Developer evidence (CPU, torch): emulates the engine's operand-rounding points, one group at a time or in mixes, and
reports the per-label PROBABILITY error each choice causes against the fp32 C oracle.  Round 2: run on the headline shape
(base, S=1024) with n-bit rounding, so that mixed modes (some operand groups at 11 bits = f16, others at >= 13 bits = f16 +
low-precision correction term, or split-f16 = ~21 bits) can be priced before they are built.
    python scripts/emulate_rounding.py [config] [B] [S] [labels]      -> profiles/r02_rounding_budget_<config>.txt
Groups: W weights, X/H1 GEMM input activations (LayerNorm outputs), QK / V QKV outputs, P softmax probabilities, CTX attention
output, FF GELU output, EMB embedding table, POS relative-position operands (PQ / PK)."""
import sys, math, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, torch
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights, synth
import oracle_c
torch.set_num_threads(8)
GROUPS = ('W','X','H1','QK','V','P','CTX','FF','EMB','POS')
def rbits(n):
    if n is None or n >= 24: return lambda x: x
    if n == 11: return lambda x: x.half().float()
    def f(x):
        m, e = torch.frexp(x)
        return torch.ldexp(torch.round(m * (1 << n)) / (1 << n), e)
    return f
def run(cfg, w, ids, mask, bits):
    """bits: dict group -> mantissa bits (absent = exact fp32)"""
    rd = {g: rbits(bits.get(g)) for g in GROUPS}
    rdW, rdX, rdQK, rdV, rdP, rdC, rdH1, rdFF, rdE, rdPos = [rd[t] for t in ('W','X','QK','V','P','CTX','H1','FF','EMB','POS')]
    W = {k: torch.from_numpy(v) for k, v in w.items()}
    B, S = ids.shape; H, nh, d = cfg.hidden, cfg.heads, 64
    ids_t = torch.from_numpy(ids); m = torch.from_numpy(mask).float()
    ln = lambda x, p: torch.nn.functional.layer_norm(x, (H,), W[p+'.weight'], W[p+'.bias'], cfg.ln_eps)
    X = ln(rdE(W['embeddings.word_embeddings.weight'])[ids_t], 'embeddings.LayerNorm') * m[..., None]
    Xr = X; Xo = rdX(X)
    R = ln(W['encoder.rel_embeddings.weight'], 'encoder.LayerNorm')
    dtab = torch.from_numpy(oracle_c.delta_table(S).astype(np.int64))
    qi = torch.arange(S); idx = dtab[(qi[:, None] - qi[None, :]) + S - 1]
    scale = math.sqrt(3 * d)
    kb = (1 - m)[:, None, None, :] * -1e30
    for l in range(cfg.layers):
        p = f'encoder.layer.{l}.'
        lin = lambda x, n: x @ rdW(W[p+n+'.weight']).T + W[p+n+'.bias']
        wq = rdW(W[p+'attention.self.query_proj.weight'] / scale); bq = W[p+'attention.self.query_proj.bias'] / scale
        Q = rdQK(Xo @ wq.T + bq).view(B, S, nh, d).transpose(1, 2)
        K = rdQK(lin(Xo, 'attention.self.key_proj')).view(B, S, nh, d).transpose(1, 2)
        V = rdV(lin(Xo, 'attention.self.value_proj')).view(B, S, nh, d).transpose(1, 2)
        PQ = rdPos(R @ wq.T + bq).view(-1, nh, d).transpose(0, 1)
        PK = rdPos(R @ rdW(W[p+'attention.self.key_proj.weight']).T + W[p+'attention.self.key_proj.bias']).view(-1, nh, d).transpose(0, 1)
        out = torch.empty(B, nh, S, d)
        for b in range(B):                                   # per sequence: bounds the [nh,S,S] temporaries
            s = Q[b] @ K[b].transpose(-1, -2)
            s += torch.gather(Q[b] @ PK.transpose(-1, -2), -1, idx[None].expand(nh, S, S))
            s += torch.gather(K[b] @ PQ.transpose(-1, -2), -1, idx.T[None].expand(nh, S, S)).transpose(-1, -2)
            out[b] = rdP(torch.softmax(s + kb[b], -1)) @ V[b]
        ctx = rdC(out.transpose(1, 2).reshape(B, S, H))
        T1 = lin(ctx, 'attention.output.dense') + Xr
        H1 = ln(T1, p+'attention.output.LayerNorm'); H1o = rdH1(H1)
        FF = rdFF(torch.nn.functional.gelu(lin(H1o, 'intermediate.dense')))
        T2 = lin(FF, 'output.dense') + H1
        X = ln(T2, p+'output.LayerNorm'); Xr = X; Xo = rdX(X)
    cm = ids_t == cfg.class_token_index
    C = int(cm.sum(1).max())
    def proj(x, n): return torch.nn.functional.gelu(x @ W[n+'.linear_1.weight'].T + W[n+'.linear_1.bias']) @ W[n+'.linear_2.weight'].T + W[n+'.linear_2.bias']
    o = torch.zeros(B, C)
    for b in range(B):
        pos = torch.nonzero(cm[b]).flatten()
        o[b, :len(pos)] = proj(Xr[b, pos], 'classes_projector') @ proj(Xr[b, 0], 'text_projector')
    return o.numpy()
sig = lambda x: 1/(1+np.exp(-x.astype(np.float64)))
if __name__ == "__main__":
    cname = sys.argv[1] if len(sys.argv) > 1 else "base"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    C = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    cfg = CONFIGS[cname]; w = weights.make_weights(cfg, 42)
    ids, mask, _ = synth.make_inputs(cfg, B, S, C, seed=1234, ragged=False)
    with torch.no_grad():
        t0 = time.time(); ref = run(cfg, w, ids, mask, {}); t1 = time.time() - t0
        pref = sig(ref)
        print(f"# {cname} B={B} S={S} labels={C}: torch fp32 forward {t1:.1f} s; {B*C} probabilities per line", flush=True)
        allg = lambda n: {g: n for g in GROUPS}
        cases = [("all groups @11 (f16 everywhere)", allg(11))]
        cases += [(f"only {g} @11", {g: 11}) for g in GROUPS]
        cases += [("all @12", allg(12)), ("all @13", allg(13)), ("all @14", allg(14))]
        mix = lambda hi, n: {g: (n if g in hi else 11) for g in GROUPS}
        cases += [
            ("W@21 rest @11 (weights split)", mix(('W',), 21)),
            ("W,EMB@21 rest @11", mix(('W','EMB'), 21)),
            ("W,EMB,POS,QK@21 rest @11", mix(('W','EMB','POS','QK'), 21)),
            ("W,X,H1,FF,CTX,EMB@21 (GEMM operands) rest @11", mix(('W','X','H1','FF','CTX','EMB'), 21)),
            ("QK,POS,P,V@21 (attention operands) rest @11", mix(('QK','POS','P','V'), 21)),
            ("all@14 but FF,W(ffn)... n/a", None),
        ]
        for name, bits in cases:
            if bits is None: continue
            lg = run(cfg, w, ids, mask, bits)
            pe = np.abs(sig(lg) - pref)
            print(f"{name:52s} prob err max {pe.max():.2e} rms {np.sqrt((pe**2).mean()):.2e}   logit err max {np.abs(lg-ref).max():.2e} rms {np.sqrt(((lg-ref)**2).mean()):.2e}", flush=True)
