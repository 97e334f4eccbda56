"""The C-ABI library loads and exports every symbol include/zero_hip.h declares (no compute)."""
import os
import re
import subprocess

import pytest

from zero_amd import hip


def test_library_exports_every_declared_symbol():
    lib = hip.lib()       # raises if the .so is missing or a declared symbol is not exported
    declared = set(lib.protos)
    assert len(declared) >= 40
    out = subprocess.check_output(["nm", "-D", "--defined-only", hip.LIB_PATH]).decode()
    exported = set(re.findall(r" T (zk_\w+)", out))
    assert declared <= exported, declared - exported
    assert exported <= declared, "exported but undeclared: %s" % (exported - declared)
    assert lib.raw("zk_version")() == 100


def test_argument_errors_are_reported_not_thrown():
    lib = hip.lib()
    with pytest.raises(hip.ZeroHipError) as ei:
        # H not a multiple of 8 -> argument error before any launch (no GPU needed)
        lib.call("zk_add_ln_fwd", None, None, None, None, None, None, None, None, 4, 13, 1e-8, 0.0, None, 0, None)
    assert "multiple of 8" in str(ei.value)
    with pytest.raises(hip.ZeroHipError):
        lib.call("zk_gemm", None, None, None, 8, 8, 8, 8, 8, 8, 0, 0, 0, 1.0, None, None, 0, 7, None, 0, 1.0, 0.0,
                 None, 0, 0, None, 0, None)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(hip.ZeroHipError) as ei:
        hip._Lib()
    assert "no CPU fallback" in str(ei.value)


def test_engine_refuses_cpu_device():
    from zero_amd.func import Engine
    with pytest.raises(hip.ZeroHipError):
        Engine("cpu")


def test_header_cites_reference_lines():
    text = open(hip.HEADER_PATH).read()
    for needle in ("func.py:14-65", "func.py:218-256", "search.py:143-176", "utils/cycle.py:86-101",
                   "transformer_aan.py:92-117", "util.py:88-103"):
        assert needle in text
