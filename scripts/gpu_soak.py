#!/usr/bin/env python3
"""GPU soak: random shapes (B <= 12, S <= 1400, 0-8 labels per row, ragged rows, mask holes) for both backbones and every operand
mode against the CPU oracle, for a time budget.  usage: gpu_soak.py [seconds=240] [seed=0] [max_S=1400] [models=tiny,mini,dec-tiny].  Exit code 1 on any violation of the
tests' envelopes (f32 1e-4, f16 1e-2, bf16 6e-2) or a non-finite output."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle_c  # noqa: E402
from gliclass.c_amd import synth, weights  # noqa: E402
from gliclass.c_amd.config import CONFIGS  # noqa: E402
from gliclass.c_amd.engine import Engine  # noqa: E402

TOL = {"f32": 1e-4, "f16": 1e-2, "bf16": 6e-2}


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    sig = lambda x: 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    max_s = int(sys.argv[3]) if len(sys.argv) > 3 else 1400
    names = tuple(sys.argv[4].split(",")) if len(sys.argv) > 4 else ("tiny", "mini", "dec-tiny")
    models = {}
    for cname in names:
        cfg = CONFIGS[cname]
        w = weights.make_weights(cfg, 42)
        models[cname] = (cfg, w, {dt: Engine(cfg, w, dtype=dt) for dt in TOL})
    worst = {(c, d): 0.0 for c in models for d in TOL}
    t0, cases, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        cname = names[int(rng.integers(0, len(names)))]
        cfg, w, engs = models[cname]
        B = int(rng.integers(1, 13))
        S = int(rng.integers(1, max_s + 1)) if rng.random() < 0.8 else int(rng.integers(1, 48))
        cmax = int(rng.integers(0, 9))
        lpr = [int(x) for x in rng.integers(0, cmax + 1, size=B)]
        S = max(S, 2 + 3 * max(lpr + [0]) + 2)
        ids, mask, _ = synth.make_inputs(cfg, B, S, max(cmax, 1), seed=int(rng.integers(0, 1 << 30)), ragged=bool(rng.integers(0, 2)), labels_per_row=lpr)
        if rng.random() < 0.25 and S > 20:                     # a mask hole in row 0 — in the TEXT part: a masked class token is outside the
            n0 = int(mask[0].sum())                            # contract (its hidden state is a padding-query row, which the engine does not compute)
            lo = max(n0 // 2, 2 + 3 * lpr[0] + 1)
            if n0 > 16 and lo + 3 < n0:
                mask[0, lo: lo + 3] = 0
        ref = oracle_c.forward(cfg, w, ids, mask)
        for dt, eng in engs.items():
            got = eng.forward(ids, mask)
            ok = got.shape == ref.shape and np.isfinite(got).all()
            err = float(np.abs(sig(got) - sig(ref)).max()) if (ok and ref.size) else 0.0
            worst[(cname, dt)] = max(worst[(cname, dt)], err)
            if not ok or err > TOL[dt]:
                bad += 1
                print("VIOLATION", cname, dt, "B", B, "S", S, "labels", lpr, "err", err, flush=True)
        cases += 1
        if cases % 50 == 0:
            print(f"{cases} cases, {time.time() - t0:.0f} s", flush=True)
    print("cases", cases, "violations", bad)
    for k in sorted(worst):
        print(k, f"{worst[k]:.2e}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
