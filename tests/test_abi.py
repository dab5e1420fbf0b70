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


def test_spconv_refuses_slabs_it_cannot_address():
    """k_spconv gathers with 32-bit byte offsets through a buffer resource: an input slab (or packed weight block) of
    2 GiB or more must come back as PBN_ERR_RANGE -- the argument checks run on the host before any launch, so this
    needs no GPU (the pointers below are never dereferenced)."""
    from pbnet_amd import _native as N
    lib = N.lib()
    vp = N.c_vp
    fake = 1 << 20                                    # 16-byte aligned, never touched

    def call(n_in, ld, dtype, n_out=128, n_steps=27 * 3, cout_p=96):
        return lib.pbn_spconv_forward(vp(fake), ld, n_in, vp(fake), 27, None, None, n_out, vp(fake), 12, n_steps, cout_p, None,
                                      None, None, 0, 0, vp(fake), cout_p, dtype, 0, None, 0, None)
    # 1.2 M voxels x 3 TTA copies x 384 ch fp32 = 5.5 GB: refused; the bf16 / 96-channel slabs of the same scene pass the
    # check only when they fit
    assert call(3 * 1200000, 384, 0) == N.PBN_ERR_RANGE
    assert call(-(-(1 << 31) // (96 * 2)), 96, 1) == N.PBN_ERR_RANGE      # the first row count that reaches 2 GiB
    assert call(-1, 96, 1) == N.PBN_ERR_ARG
    assert call(100, 96, 1, n_out=0) == N.PBN_OK                          # in range, nothing to do
