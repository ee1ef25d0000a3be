"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/mmvae_hip.h declares, and the ctypes table in hipops.py agrees with the header's parameter lists.
No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "mmvae_hip.h")


def _declarations():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef struct \{.*?\} \w+;", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|size_t|const char\*)\s+(mmvae_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(2)] = (m.group(1), n, args)
    return decls


def test_library_exports_every_declared_symbol():
    from multimodal_vae_comparison_amd import hipops
    assert os.path.exists(hipops.LIB_PATH), "build the HIP library first (__graft_entry__.build())"
    lib = ctypes.CDLL(hipops.LIB_PATH)
    decls = _declarations()
    assert len(decls) >= 40
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in mmvae_hip.h but not exported"


def test_ctypes_table_matches_header():
    from multimodal_vae_comparison_amd import hipops
    decls = _declarations()
    assert set(decls) == set(hipops.SIGNATURES), set(decls) ^ set(hipops.SIGNATURES)
    for name, (ret, n, args) in decls.items():
        res, argtypes = hipops.SIGNATURES[name]
        assert len(argtypes) == n, f"{name}: header has {n} parameters, ctypes table {len(argtypes)}"
        # per-parameter kind check: pointers vs int/long/float
        for a, ct in zip([x.strip() for x in args.split(",")] if n else [], argtypes):
            is_ptr = "*" in a or a.startswith("mmvae_stream_t")
            if is_ptr:
                assert ct in (ctypes.c_void_p,) or isinstance(ct, type(ctypes.POINTER(ctypes.c_float))), (name, a, ct)
            elif a.startswith("long"):
                assert ct is ctypes.c_long, (name, a, ct)
            elif a.startswith("float"):
                assert ct is ctypes.c_float, (name, a, ct)
            elif a.startswith("unsigned"):
                assert ct is ctypes.c_uint, (name, a, ct)
            elif a.startswith("int"):
                assert ct is ctypes.c_int, (name, a, ct)


def test_pure_queries_run_without_gpu():
    from multimodal_vae_comparison_amd import hipops
    L = hipops.lib()
    assert L.mmvae_version() >= 1
    assert L.mmvae_arch() == b"gfx950"
    assert L.mmvae_gemm_ws_floats(64, 64, 4) == 4 * (64 * 64 + 64)
    assert L.mmvae_conv_wgrad_ws_floats(128, 32, 32, 16) > 0
    assert L.mmvae_poe_ws_floats(128, 32) >= 32


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from multimodal_vae_comparison_amd import hipops
    monkeypatch.setattr(hipops, "_lib", None)
    monkeypatch.setattr(hipops, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no fallback"):
        hipops.lib()


def test_live_knob_list_matches_the_sources():
    """tools/live_knobs.sh (the A/B scripts refuse knobs outside it) == the MMVAE_* variables the sources really read"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = set()
    pkg = os.path.join(root, "multimodal_vae_comparison_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".hip", ".inc", ".hpp")):
                continue
            src = open(os.path.join(dirpath, f), errors="replace").read()
            read |= set(re.findall(r'getenv\("(MMVAE_[A-Z0-9_]+)"\)', src))
            read |= set(re.findall(r'environ(?:\.get\(|\[)"(MMVAE_[A-Z0-9_]+)"', src))
    read |= set(re.findall(r'environ(?:\.get\(|\[)"(MMVAE_[A-Z0-9_]+)"', open(os.path.join(root, "bench.py")).read()))
    listed = re.search(r'LIVE_KNOBS="([^"]*)"', open(os.path.join(root, "tools", "live_knobs.sh")).read()).group(1).split()
    assert set(listed) == read, (sorted(set(listed) - read), sorted(read - set(listed)))
