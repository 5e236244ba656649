"""Developer tool (GPU): the MX GEMM on random, constant and all-zero operands — how much of its time is the chip's power envelope."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd import weights
    from gliclass.c_amd.engine import Engine
    e = Engine(CONFIGS["tiny"], weights.make_weights(CONFIGS["tiny"], 1), dtype="f16")
    for (name, M, N, K, ep) in (("ffn2", 65536, 768, 3072, 2), ("ffn1", 65536, 3072, 768, 1)):
        for which, tag in ((9, "GX"), (11, "GY")):
            r = [e.L.glc_debug_gemm_bench(e.h, M, N, K, ep, 20, which) for _ in range(2)]
            print(f"{sys.argv[1]:6s} {name} {tag}: {r[-1]*1e3:7.1f} us", flush=True)
    e.close()
else:
    for d in ("random", "const", "zero"):
        env = dict(os.environ); env["GLC_BENCH_DATA"] = d
        subprocess.run([sys.executable, __file__, d], env=env)
