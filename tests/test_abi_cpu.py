"""CPU: the C-ABI shared library builds, loads and exports every symbol include/dae_hip.h
declares (no compute calls without a GPU), and the product refuses to run without it."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "dae_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dae_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from spotify_recsys_challenge_2018_amd import _lib, build
    so = build.build()
    names = _declared()
    assert len(names) >= 18 and "dae_score_topk" in names
    out = subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    missing = [n for n in names if n not in exported]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == names, "ctypes binding list must match the header"
    lib = _lib.load()
    for n in names:
        assert hasattr(lib, n)
    assert lib.dae_version() >= 1000


def test_no_gpu_is_a_loud_error_not_a_fallback():
    import torch
    from spotify_recsys_challenge_2018_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.DaeError):
        _lib.Context(0)
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.dae_create(0, ctypes.byref(h)) != 0
    assert b"HIP" in lib.dae_last_error(None) or b"device" in lib.dae_last_error(None)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "spotify_recsys_challenge_2018_amd")
    for dp, _dn, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import oracle|from oracle)", src, flags=re.M), f
                assert "liboracle" not in src, f
