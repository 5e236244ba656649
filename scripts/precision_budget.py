"""Developer tool (GPU): the precision budget of the default mode, MEASURED on the engine (round 3; replaces the CPU emulation of round 2).

For each weight seed: one c3-shaped forward (base, B x S, 8 labels) in the full split-f16 mode is the reference (8.5e-6 from the CPU
oracle, bench.py); then one forward per precision mask (glc_debug_set_precision_mask: a set bit rounds one operand group of the
group-split pipeline to f16) and the error of ALL B x 8 per-label probabilities against that reference.  The f16 / bf16 opt-in modes
run on the same inputs for comparison.  Prints one table per seed and a worst-case-over-seeds summary; writes JSON to --out.
"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine

BITS = {"QKV_A": 1 << 0, "QKV_W": 1 << 1, "AO_A": 1 << 2, "AO_W": 1 << 3, "F1_A": 1 << 4, "F1_W": 1 << 5, "F2_A": 1 << 6, "F2_W": 1 << 7,
        "Q": 1 << 8, "K": 1 << 9, "V": 1 << 10, "P": 1 << 11, "PQ": 1 << 12, "PK": 1 << 13, "RESID": 1 << 14}


def m(*names):
    v = 0
    for n in names:
        v |= BITS[n]
    return v


ALL_W = m("QKV_W", "AO_W", "F1_W", "F2_W")
ALL_A = m("QKV_A", "AO_A", "F1_A", "F2_A")
ALL_ATT = m("Q", "K", "V", "P", "PQ", "PK")
CASES = [(k, v) for k, v in BITS.items()] + [
    ("all W", ALL_W), ("all A", ALL_A), ("all attention", ALL_ATT),
    ("P V", m("P", "V")), ("P V F2_A", m("P", "V", "F2_A")), ("P V F2_A AO_A", m("P", "V", "F2_A", "AO_A")),
    ("P V PQ PK", m("P", "V", "PQ", "PK")), ("P V PQ PK F2_A AO_A", m("P", "V", "PQ", "PK", "F2_A", "AO_A")),
    ("all A + P V", ALL_A | m("P", "V")), ("all A + P V PQ PK", ALL_A | m("P", "V", "PQ", "PK")),
    ("all A + all attention", ALL_A | ALL_ATT),
    ("all W + P V", ALL_W | m("P", "V")), ("all W + P V PQ PK", ALL_W | m("P", "V", "PQ", "PK")), ("all W + all attention", ALL_W | ALL_ATT),
    ("F1_A F2_A AO_A (not QKV_A)", m("F1_A", "F2_A", "AO_A")), ("F2_A AO_A P V K", m("F2_A", "AO_A", "P", "V", "K")),
    ("everything but RESID", ALL_W | ALL_A | ALL_ATT), ("everything", ALL_W | ALL_A | ALL_ATT | BITS["RESID"]),
]


def probs(x):
    return 1.0 / (1.0 + np.exp(-x.astype(np.float64)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="base")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--labels", type=int, default=8)
    ap.add_argument("--seeds", default="42,43,44")
    ap.add_argument("--masks", default="", help="extra masks, comma-separated integers")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    ids, mask, _ = synth.make_inputs(cfg, a.batch, a.seq, a.labels, seed=1234)
    cases = list(CASES) + [(f"mask {int(x)}", int(x)) for x in a.masks.split(",") if x]
    res = {}
    for seed in [int(s) for s in a.seeds.split(",")]:
        e = Engine.from_spec(cfg, f"synthetic:{a.config}:{seed}", dtype="f32")
        e.set_length_buckets(1)
        ref = e.forward(ids, mask)
        assert e.last_group_split(), "the precision switches act on the group-split pipeline"
        pref = probs(ref)
        again = e.forward(ids, mask)
        rows = {"repeat (mask 0)": (float(np.abs(probs(again) - pref).max()), 0.0, float(np.abs(again - ref).max()))}
        for name, mk in cases:
            e.set_precision_mask(mk)
            got = e.forward(ids, mask)
            d = probs(got) - pref
            rows[name] = (float(np.abs(d).max()), float(np.sqrt((d * d).mean())), float(np.abs(got - ref).max()))
        e.set_precision_mask(0)
        e.close()
        for dt in ("f16", "bf16"):
            e2 = Engine.from_spec(cfg, f"synthetic:{a.config}:{seed}", dtype=dt)
            e2.set_length_buckets(1)
            got = e2.forward(ids, mask)
            d = probs(got) - pref
            rows[f"{dt} mode"] = (float(np.abs(d).max()), float(np.sqrt((d * d).mean())), float(np.abs(got - ref).max()))
            e2.close()
        res[seed] = rows
        print(f"--- {a.config} B={a.batch} S={a.seq} C={a.labels}, weights seed {seed}: {pref.size} probabilities; logit range [{ref.min():.2f}, {ref.max():.2f}] ---")
        print(f"{'operands rounded to f16':42s} {'max |dp|':>10s} {'rms dp':>10s} {'max |dlogit|':>13s}")
        for k, (mx, rms, ml) in rows.items():
            print(f"{k:42s} {mx:10.2e} {rms:10.2e} {ml:13.2e}")
        sys.stdout.flush()
    print("--- worst case over seeds (max |dp|) and rms over seeds ---")
    names = list(next(iter(res.values())).keys())
    for k in names:
        mx = max(res[s][k][0] for s in res)
        rms = float(np.sqrt(np.mean([res[s][k][1] ** 2 for s in res])))
        print(f"{k:42s} {mx:10.2e} {rms:10.2e}")
    if a.out:
        with open(a.out, "w") as f:
            json.dump({str(s): r for s, r in res.items()}, f, indent=1)


if __name__ == "__main__":
    main()
