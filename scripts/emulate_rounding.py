"""Developer evidence (CPU, torch): emulates the engine's f16 rounding points one group at a time on the small
config and reports the logit error each group alone causes vs the fp32 C oracle.  Output committed as
profiles/r01_f16_rounding_ablation.txt; discussed in DESIGN.md §2."""
import sys, math; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, torch
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights, synth
import oracle_c
torch.set_num_threads(8)
def run(cfg, w, ids, mask, rdf, tags, resid32=True):
    rd0 = rdf
    def mk(tag):
        return (lambda x: rd0(x)) if (tags is None or tag in tags) else (lambda x: x)
    rdW, rdX, rdQK, rdV, rdP, rdC, rdH1, rdFF, rdE, rdPos = [mk(t) for t in ('W','X','QK','V','P','CTX','H1','FF','EMB','POS')]
    """torch emulation of the engine's rounding points. rd(x): round to operand dtype. resid32: keep residual stream fp32."""
    W = {k: torch.from_numpy(v) for k, v in w.items()}
    B, S = ids.shape; H, nh, d = cfg.hidden, cfg.heads, 64
    ids_t = torch.from_numpy(ids); m = torch.from_numpy(mask).float()
    ln = lambda x, p: torch.nn.functional.layer_norm(x, (H,), W[p+'.weight'], W[p+'.bias'], cfg.ln_eps)
    rs = (lambda x: x)
    X = ln(rdE(W['embeddings.word_embeddings.weight'])[ids_t], 'embeddings.LayerNorm') * m[..., None]
    Xr = rs(X); Xo = rdX(X)
    R = ln(W['encoder.rel_embeddings.weight'], 'encoder.LayerNorm')
    dtab = torch.from_numpy(oracle_c.delta_table(S).astype(np.int64))
    qi = torch.arange(S); idx = dtab[(qi[:, None] - qi[None, :]) + S - 1]        # [S,S]
    scale = math.sqrt(3 * d)
    kb = (1 - m)[:, None, None, :] * -1e30
    for l in range(cfg.layers):
        p = f'encoder.layer.{l}.'
        lin = lambda x, n: x @ rdW(W[p+n+'.weight']).T + W[p+n+'.bias']
        wq = rdW(W[p+'attention.self.query_proj.weight'] / scale); bq = W[p+'attention.self.query_proj.bias'] / scale
        Q = rdQK(Xo @ wq.T + bq).view(B, S, nh, d).transpose(1, 2)
        K = rdQK(lin(Xo, 'attention.self.key_proj')).view(B, S, nh, d).transpose(1, 2)
        V = rdV(lin(Xo, 'attention.self.value_proj')).view(B, S, nh, d).transpose(1, 2)
        Ro = rdPos(R)
        PQ = rdPos(Ro @ wq.T + bq).view(-1, nh, d).transpose(0, 1)     # [nh,P,d]
        PK = rdPos(Ro @ rdW(W[p+'attention.self.key_proj.weight']).T + W[p+'attention.self.key_proj.bias']).view(-1, nh, d).transpose(0, 1)
        s = Q @ K.transpose(-1, -2)
        c2p = torch.gather(Q @ PK.transpose(-1, -2)[None], -1, idx[None, None].expand(B, nh, S, S))
        p2c = torch.gather(K @ PQ.transpose(-1, -2)[None], -1, idx.T[None, None].expand(B, nh, S, S)).transpose(-1, -2)
        pr = rdP(torch.softmax(s + c2p + p2c + kb, -1))
        ctx = rdC((pr @ V).transpose(1, 2).reshape(B, S, H))
        T1 = rs(lin(ctx, 'attention.output.dense') + Xr)
        H1 = ln(T1, p+'attention.output.LayerNorm'); H1r = rs(H1); H1o = rdH1(H1)
        FF = rdFF(torch.nn.functional.gelu(lin(H1o, 'intermediate.dense')))
        T2 = rs(lin(FF, 'output.dense') + H1r)
        X = ln(T2, p+'output.LayerNorm'); Xr = rs(X); Xo = rdX(X)
    hid = Xr
    cm = ids_t == cfg.class_token_index
    C = int(cm.sum(1).max())
    def proj(x, n): return torch.nn.functional.gelu(x @ W[n+'.linear_1.weight'].T + W[n+'.linear_1.bias']) @ W[n+'.linear_2.weight'].T + W[n+'.linear_2.bias']
    out = torch.zeros(B, C)
    for b in range(B):
        pos = torch.nonzero(cm[b]).flatten()
        out[b, :len(pos)] = proj(hid[b, pos], 'classes_projector') @ proj(hid[b, 0], 'text_projector')
    return out.numpy()
sig = lambda x: 1/(1+np.exp(-x.astype(np.float64)))
cfg = CONFIGS["small"]; w = weights.make_weights(cfg, 42)
ids, mask, _ = synth.make_inputs(cfg, 4, 128, 4, seed=77, ragged=True)
ref = oracle_c.forward(cfg, w, ids, mask)
f16 = lambda x: x.half().float()
with torch.no_grad():
    for tags in (None, ('W',), ('X','H1'), ('QK',), ('V',), ('P',), ('CTX',), ('FF',), ('EMB',), ('POS',), ('W','X','H1','FF','CTX','V','EMB'), ('QK','POS','P')):
        lg = run(cfg, w, ids, mask, f16, tags)
        print(f"round only {str(tags):48s} max logit err {np.abs(lg-ref).max():.2e}  rms {np.sqrt(((lg-ref)**2).mean()):.2e}", flush=True)
