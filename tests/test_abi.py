"""CPU tier: libpbnet_hip.so loads and exports every symbol include/pbnet_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "pbnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pbn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from pbnet_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    handle = ctypes.CDLL(_native.LIB_PATH)
    names = _declared()
    assert len(names) >= 6
    for n in names:
        assert hasattr(handle, n), "symbol %s declared in include/pbnet_hip.h is not exported" % n


def test_ctypes_signatures_cover_header():
    from pbnet_amd import _native
    assert sorted(_native.SIGNATURES) == _declared()


def test_workspace_query_is_host_only():
    from pbnet_amd import _native
    lib = _native.lib()
    assert lib.pbn_version().startswith(b"pbnet_hip")
    small = lib.pbn_cluster_workspace_bytes(1000, 3, 0)
    big = lib.pbn_cluster_workspace_bytes(100000, 3, 0)
    gen = lib.pbn_cluster_workspace_bytes(100000, 3, 1)
    assert 0 < small < big < gen


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from pbnet_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeLibraryError):
        _native.lib()


def test_cpu_tensors_are_refused():
    import torch
    from pbnet_amd import pbnet_ops
    z = torch.zeros(4, 3)
    with pytest.raises(RuntimeError):
        pbnet_ops.cluster_device(z, z, torch.zeros(4, dtype=torch.int32), torch.tensor([4], dtype=torch.int32), 0.04, 31)
