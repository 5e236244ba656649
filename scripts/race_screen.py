"""Developer tool (GPU): bitwise comparison of the two main loops of the 256-tile GEMM (full-line / half-line ring stages: other DMA maps,
LDS images, ring hazards and wait counts, the SAME accumulation order) over repeated forwards — identical bits unless a synchronisation hazard fires."""
import os, sys, subprocess, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd import weights, synth
    from gliclass.c_amd.engine import Engine
    out = []
    for cname, B, S in (("mini", 16, 512), ("base", 16, 1024)):
        cfg = CONFIGS[cname]
        w = weights.make_weights(cfg, 42)
        e = Engine(cfg, w, dtype=os.environ.get("GLC_RACE_DTYPE", "f16"))
        ids, mask, _ = synth.make_inputs(cfg, B, S, 4, seed=3, ragged=True)
        hs = set()
        for it in range(int(sys.argv[2])):
            lg = e.forward(ids, mask)
            hs.add(hashlib.sha1(lg.tobytes()).hexdigest()[:12])
        out.append((cname, sorted(hs)))
        e.close()
    print(out)
else:
    for env in ({}, {"GLC_GEMM_FL": "0"}):
        r = subprocess.run([sys.executable, __file__, "child", "12"], env=dict(os.environ, **env), capture_output=True, text=True)
        print(env, r.stdout.strip()[-300:], r.stderr.strip()[-200:] if r.returncode else "")
