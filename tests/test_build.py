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
    for gone in ("attn_mx2_kernel", "gemm256w_kernel", "to_gy_kernel", "gy_to_f32_kernel"):
        assert gone not in text, gone
    # band kernel on MX tiles: <NW, ABL = 0, DIAG = false, RECOMP = true, FIXQ = true, XROT = true> only
    mx = [k for k in ks if "attn_mx_kernel<" in k]
    assert mx and all(re.search(r"attn_mx_kernel<[48], 0, false, true, true, true>", k) for k in mx), mx
    # role-split kernel: <DIAG = false, XPRIO = 0> only
    assert not [k for k in ks if "attn_mxs_kernel<" in k or "attn_mxd_kernel<" in k or "attn_mx2_kernel<" in k]      # rejected attention kernels: developer builds only
    # MX GEMM: two template parameters (epilogue, transposed tile), nothing else
    gx = [k for k in ks if "gemm256x_kernel<" in k]
    assert gx and all(re.search(r"gemm256x_kernel<\d, (true|false)>\(", k) for k in gx), gx
    # workgroup-shared split-f16 attention: no stamped (DIAG) and no two-MFMA timing build (NMM = 2)
    for k in ks:
        m = re.search(r"attn_wg_kernel<([^>]*)>", k)
        if m:
            a = [x.strip() for x in m.group(1).split(",")]
            assert a[6] == "false" and a[8] == "3", k
