"""The product library ships only the kernels the forward can take: no timing-only (wrong-result) build, no stamped build,
no rejected experiment.  The kernel names the host side registers (the mangled names sit in libgliclass_hip.so as strings) are checked against
the list of shipping instantiations; a developer build (make DEV=1) is skipped — it contains the diagnostic kernels on purpose."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "gliclass", "c_amd", "libgliclass_hip.so")


def _kernel_names():
    blob = open(SO, "rb").read()
    names = sorted(set(m.decode() for m in re.findall(rb"_ZN12_GLOBAL__N_1[A-Za-z0-9_]+", blob)))
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.split("\n")
    return sorted(set(o for o in out if "_kernel" in o and "__device_stub__" not in o))


def test_product_library_ships_only_shipping_kernel_instantiations():
    L = ctypes.CDLL(SO)
    if L.glc_debug_is_developer_build():
        pytest.skip("developer build (make DEV=1): the diagnostic kernels are compiled in on purpose")
    ks = _kernel_names()
    assert ks, "no kernel names found in the library"
    text = "\n".join(ks)
    # rejected experiments live in csrc/dev/ and are not linked
    for gone in ("attn_mx2_kernel", "to_gy_kernel", "gy_to_f32_kernel"):
        assert gone not in text, gone
    # band kernel on MX tiles: <NW, ABL = 0, DIAG = false, RECOMP = true, FIXQ = true, XROT = true, DIET = true> only
    mx = [k for k in ks if "attn_mx_kernel<" in k]
    assert mx and all(re.search(r"attn_mx_kernel<[48], 0, false, true, true, true, true>", k) for k in mx), mx
    # role-split kernel: <DIAG = false, XPRIO = 0> only
    assert not [k for k in ks if "attn_mxs_kernel<" in k or "attn_mxd_kernel<" in k or "attn_mx2_kernel<" in k]      # rejected attention kernels: developer builds only
    # MX GEMM: two template parameters (epilogue, transposed tile), nothing else
    gx = [k for k in ks if "gemm256x_kernel<" in k]
    assert gx and all(re.search(r"gemm256x_kernel<\d, (true|false)>\(", k) for k in gx), gx
    # ... and (round 6) its one-wave-per-SIMD tile for the large bias / GELU / residual / SwiGLU shapes: one template parameter (the epilogue), nothing else
    gw = [k for k in ks if "gemm256w_kernel<" in k]
    assert len(gw) == 4 and all(re.search(r"gemm256w_kernel<[0124]>\(", k) for k in gw), gw
    # workgroup-shared split-f16 attention: no stamped (DIAG) and no two-MFMA timing build (NMM = 2)
    for k in ks:
        m = re.search(r"attn_wg_kernel<([^>]*)>", k)
        if m:
            a = [x.strip() for x in m.group(1).split(",")]
            assert a[6] == "false" and a[8] == "3", k


def _vregs(tok):
    """registers named by one operand token: v12 -> {12}, v[4:7] -> {4..7}; anything else -> {}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def test_resident_position_block_is_untouched_between_its_request_and_the_wait(tmp_path):
    """the band kernel's resident PQ block (FIXQ) is refilled IN PLACE by inline-asm global loads the compiler does not track
    (csrc/glc_pfrag.h); until the next key tile's `s_waitcnt vmcnt(0)` the registers hold loads in flight.  A live-range split, a copy or a spill of
    them in that window would read rows that have not arrived.  Checked on the compiler's own output of the shipping kernels: from every request block
    (.Lpfskip label), along every path of the control-flow graph up to the first `s_waitcnt vmcnt(0)`, no instruction names one of the block's registers."""
    src = os.path.join(ROOT, "gliclass", "c_amd", "csrc", "attention_mx.hip")
    asm = tmp_path / "attention_mx.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", src, "-o", str(asm)], check=True, capture_output=True)
    lines = asm.read_text().split("\n")
    labels = {m.group(1): i for i, ln in enumerate(lines) if (m := re.match(r"^([.\w$]+):", ln))}
    requests = [i for i, ln in enumerate(lines) if re.match(r"^\.Lpfskip\d+:", ln)]
    assert len(requests) >= 4, "two shipping kernels (NW = 4, 8) x two band tiles per loop trip"
    checked = 0
    for r in requests:
        # the block's registers: destinations of the eight loads just above the label
        regs, i = set(), r - 1
        while i > 0 and "s_cbranch_scc1 .Lpfskip" not in lines[i]:
            m = re.match(r"\s+global_load_dwordx4 (v\[\d+:\d+\]),", lines[i])
            if m:
                regs |= _vregs(m.group(1))
            i -= 1
        assert len(regs) == 32, (r, sorted(regs))
        seen, todo = set(), [r + 1]
        while todo:
            i = todo.pop()
            while i < len(lines) and i not in seen:
                seen.add(i)
                ln = lines[i].split(";")[0].strip()
                i += 1
                if not ln or ln.endswith(":") or ln.startswith("."):
                    continue
                if re.match(r"s_waitcnt .*vmcnt\(0\)", ln):
                    break                                             # the block is usable from here on
                assert not ln.startswith("s_endpgm"), "a path leaves the kernel with the block's loads in flight (harmless) — but then this walk is wrong"
                toks = re.split(r"[\s,]+", ln)
                used = set().union(*[_vregs(t) for t in toks[1:]]) if len(toks) > 1 else set()
                assert not (used & regs), f"line {i}: `{ln}` touches the resident block's registers before the wait"
                checked += 1
                if toks[0] == "s_branch":
                    todo.append(labels[toks[1]]); break
                if toks[0].startswith("s_cbranch"):
                    todo.append(labels[toks[1]])
    assert checked > 50
