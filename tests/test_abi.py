"""CPU-side checks of the C-ABI library: it builds, loads, exports every symbol the header declares,
and its layout / argument-validation entry points behave (no GPU compute is launched here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    from mobody_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    return _lib.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from mobody_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mobody_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mobody_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mobody_abi_version() == 6


def test_struct_sizes_match_header(lib):
    from mobody_amd import _lib
    assert C.sizeof(_lib.MobodyLayer) == 32
    assert C.sizeof(_lib.MobodyDynLayout) == 16 + 13 * 32 + 8
    assert C.sizeof(_lib.MobodyMlpLayout) == 6 * 4 + 15 * 8
    assert C.sizeof(_lib.MobodyTrainDims) == 8 + 4 * 8
    assert C.sizeof(_lib.MobodyHyper) == 8 * 4


@pytest.mark.parametrize("S,A", [(17, 6), (111, 8), (45, 24), (11, 3)])
def test_dyn_layout(lib, S, A):
    from mobody_amd import _lib
    L = _lib.dyn_layout(S, A)
    assert (L.S, L.A, L.E) == (S, A, 7)
    end = 0
    for i, name in enumerate(_lib.DL_NAMES):
        l = L.layer[i]
        assert l.Kp % 8 == 0 and l.Np % 16 == 0 and l.Kp >= l.in_dim and l.Np >= l.out_dim, name
        assert l.w_off >= end and l.w_off % 4 == 0
        assert l.b_off == l.w_off + 7 * l.Kp * l.Np
        end = l.b_off + 7 * l.Np
    assert L.total_floats >= end
    assert L.layer[0].in_dim == S and L.layer[10].in_dim == 2 * S + A and L.layer[9].out_dim == S


def test_mlp_layout_counts_reference_parameters(lib):
    from mobody_amd import _lib
    L = _lib.mlp_layout(23, 1, 2)          # twin-Q at S=17, A=6: 144 386 parameters
    assert (L.Kp1, L.Np3, L.Np1t) == (24, 16, 32)
    assert 2 * (23 * 256 + 256 + 256 * 256 + 256 + 256 + 1) == 144386
    assert L.member_floats == 24 * 256 + 256 + 65536 + 256 + 256 * 16 + 16 and L.total_floats == 2 * L.member_floats
    La = _lib.mlp_layout(17, 6, 1)
    assert 17 * 256 + 256 + 65536 + 256 + 256 * 6 + 6 == 71942 and La.total_floats >= 71942


def test_argument_validation_reports_errors(lib):
    from mobody_amd import _lib
    L = _lib.MobodyDynLayout()
    assert lib.mobody_dyn_layout(1000, 6, C.byref(L)) == -1
    assert b"unsupported" in lib.mobody_last_error()
    assert lib.mobody_dyn_layout(120, 40, C.byref(L)) == -1      # 2S+A > 256
    M = _lib.MobodyMlpLayout()
    assert lib.mobody_mlp_layout(300, 1, 1, C.byref(M)) == -1
    d = _lib.MobodyTrainDims(17, 6, 0, 0, 0, 0)
    assert lib.mobody_train_workspace(C.byref(d)) == -1
    d = _lib.MobodyTrainDims(17, 6, 640, 512, 640, 512)
    assert lib.mobody_train_workspace(C.byref(d)) > 640 * 256 * 8
    # empty batches are accepted without touching any pointer
    assert lib.mobody_dyn_step(None, None, 0, 17, 6, 4, None, None, 0, None, None, None, None, 0, 0, 0, None, 0.0, 1, 1, None, None,
                               None, None, None, None, None, None) == 0
    assert lib.mobody_dyn_step(None, None, 0, 17, 6, 99, None, None, 5, None, None, None, None, 0, 0, 0, None, 0.0, 1, 1, None, None,
                               None, None, None, None, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch, lib):
    from mobody_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmobody_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
