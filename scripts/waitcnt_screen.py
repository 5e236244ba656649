"""Developer tool (GPU): bitwise comparison of the product build against a build of every HIP translation unit with
`-mllvm -amdgpu-waitcnt-forcezero` (the compiler waits for every counter before every instruction).  Identical logits on every case =
no result of the product build depends on a wait the compiler left out (the class of bug behind round 1's "lean loop + split operands"
miscompare, docs/LOG_r01-r05.md §2).  usage: waitcnt_screen.py <path to the forcezero libgliclass_hip.so>   (scripts/build_forcezero.sh builds it)"""
import os, sys, subprocess, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
CASES = (("mini", "f32", 3, 700), ("mini", "f16", 3, 700), ("small", "f32", 8, 512), ("base", "f32", 16, 1024), ("base", "f16", 16, 1024),
         ("base", "bf16", 8, 512), ("dec-mini", "f32", 3, 600), ("dec-mini", "bf16", 3, 600), ("small", "f32", 5, 90))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd import weights, synth
    from gliclass.c_amd.engine import Engine
    for cname, dt, B, S in CASES:
        cfg = CONFIGS[cname]
        e = Engine.from_spec(cfg, f"synthetic:{cfg.name}:42", dtype=dt)
        ids, mask, _ = synth.make_inputs(cfg, B, S, 5, seed=17, ragged=True)
        hs = []
        for gs in ((1, 2) if dt == "f32" else (1,)):
            e.set_group_split(gs)
            for rep in range(2):
                out = e.forward(ids, mask)
                hs.append(hashlib.sha1(np.ascontiguousarray(out).tobytes()).hexdigest()[:16])
        print(cname, dt, B, S, " ".join(hs), flush=True)
        e.close()
    sys.exit(0)
fz = sys.argv[1]
outs = []
for lib in (None, fz):
    env = dict(os.environ)
    if lib: env["GLC_HIP_SO"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0: print(r.stdout, r.stderr); sys.exit(2)
    outs.append(r.stdout.strip().splitlines())
bad = 0
for a, b in zip(*outs):
    same = a == b
    bad += not same
    print(("same  " if same else "DIFF  ") + a + ("" if same else "   |   " + b))
print("cases", len(outs[0]), "differing", bad)
sys.exit(1 if bad else 0)
