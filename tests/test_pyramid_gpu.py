"""GPU tier: the sorted, hash-free coordinate pipeline of pbn_coords_prepare (csrc/pyramid.hip: own radix sort, all levels in
one pass over the Z-ordered rows, kernel maps top-down) against the hash-table pipeline (csrc/coords.hip + the library sort,
reachable through pbn_coords_prepare_hash) -- EVERY output array equal, bit for bit -- and against the CPU oracle
(oracle/sparse_ref.py) for what ME.SparseTensor / the coordinate manager expose
(/root/reference/network/PBNet.py:117,240-247,265-271; network/Mink.py:293-350).

Cases: the bench scene (unique rows), local-scene-like input with many duplicates and several batch indices, negative
coordinates, a tiny input, an input whose box needs the 4th sort digit, the capacity form (device-side count over a padded
buffer), and the configs[3]-sized scene; plus the limits: a box needing more than 44 key bits is reported as an error."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import sparse_ref as R
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import _native as N
from pbnet_amd import synth
from pbnet_amd.MinkowskiEngine import conventions as CV

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _prepare(coords, which, n_dev=None, want_k5=1):
    lib = N.lib()
    n = int(coords.shape[0])
    P = N.PrepareLayout()
    nbytes = lib.pbn_coords_prepare_bytes(n, want_k5, ctypes.byref(P))
    arena = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    fn = lib.pbn_coords_prepare_hash if which == "hash" else lib.pbn_coords_prepare_dev
    if which == "sorted" and n_dev is None:
        rc = lib.pbn_coords_prepare(N.ptr(coords), n, want_k5, int(CV.X_FASTEST), N.ptr(arena), nbytes, ctypes.byref(P),
                                    N.current_stream())
    else:
        rc = fn(N.ptr(coords), None if n_dev is None else N.ptr(n_dev), n, want_k5, int(CV.X_FASTEST), N.ptr(arena), nbytes,
                ctypes.byref(P), N.current_stream())
    N.check(rc, "prepare " + which)
    torch.cuda.synchronize()
    return arena, P


def _view(arena, off, count, dtype):
    nb = count * torch.empty(0, dtype=dtype).element_size()
    return arena[off:off + nb].view(dtype).cpu().numpy()


def _arrays(arena, P, n, n_in=None):
    """Every output of the contract (include/pbnet_hip.h: pbn_prepare_layout), cut to its meaningful extent."""
    L = P.pyramid
    counts = _view(arena, L.counts, 5, torch.int32)
    out = {"counts": counts.copy(), "n_unique": _view(arena, P.n_unique, 1, torch.int32).copy()}
    if counts[0] < 0:
        return out
    n1 = int(counts[0])
    out["unique_index"] = _view(arena, P.unique_index, n, torch.int64)[:n1]
    out["inverse"] = _view(arena, P.inverse, n, torch.int64)[:n if n_in is None else n_in]      # one entry per INPUT row
    out["perm"] = _view(arena, P.perm, n, torch.int64)[:n1]
    out["inv_perm"] = _view(arena, P.inv_perm, n, torch.int64)[:n1]
    out["ucoords"] = _view(arena, P.ucoords, n * 4, torch.int32).reshape(n, 4)[:n1]
    for l in range(5):
        nl = int(counts[l])
        out["coords%d" % l] = _view(arena, L.coords[l], n * 4, torch.int32).reshape(n, 4)[:nl]
        out["k3_%d" % l] = _view(arena, L.k3[l], n * 27, torch.int32).reshape(n, 27)[:nl]
    out["k5"] = _view(arena, L.k5, n * 125, torch.int32).reshape(n, 125)[:n1]
    for l in range(4):
        nf, nc = int(counts[l]), int(counts[l + 1])
        out["parent_row%d" % l] = _view(arena, L.parent_row[l], n, torch.int32)[:nf]
        out["child_k%d" % l] = _view(arena, L.child_k[l], n, torch.int32)[:nf]
        out["nbr_down%d" % l] = _view(arena, L.nbr_down[l], n * 8, torch.int32).reshape(n, 8)[:nc]
        out["up%d" % l] = _view(arena, L.up[l], n * 8, torch.int32).reshape(n, 8)[:nf]
    return out


def _pk(c):
    """[n,4] coordinates -> one int64 per row (order-preserving per field)."""
    c = c.astype(np.int64)
    return (c[:, 0] << 48) | ((c[:, 1] + 32768) << 32) | ((c[:, 2] + 32768) << 16) | (c[:, 3] + 32768)


def _canonical(a):
    """The same contract with every row REFERENCE replaced by the coordinate key of the row it names, and the rows of every
    level put in coordinate-key order: equal for any two Z-orders of the same lineage.  (The origin of the Z-order curve is an
    implementation choice: the hash pipeline interleaves x + 32768, pyramid.hip x - box minimum rounded to 16; both keep the
    children of a voxel contiguous, and convolution results do not depend on the row order.)"""
    out = {k: a[k] for k in ("counts", "n_unique", "unique_index", "inverse", "ucoords")}
    keys = [_pk(a["coords%d" % l]) for l in range(5)]
    order = [np.argsort(k, kind="stable") for k in keys]
    ref = lambda l, idx: np.where(idx >= 0, keys[l][np.maximum(idx, 0)], -1)           # row ids of level l -> coordinate keys
    assert np.array_equal(a["inv_perm"][a["perm"]], np.arange(len(a["perm"])))
    for l in range(5):
        out["coords%d" % l] = a["coords%d" % l][order[l]]
        out["k3_%d" % l] = ref(l, a["k3_%d" % l])[order[l]]
    out["k5"] = ref(0, a["k5"])[order[0]]
    for l in range(4):
        out["parent_row%d" % l] = ref(l + 1, a["parent_row%d" % l])[order[l]]
        out["child_k%d" % l] = a["child_k%d" % l][order[l]]
        out["nbr_down%d" % l] = ref(l, a["nbr_down%d" % l])[order[l + 1]]
        out["up%d" % l] = ref(l + 1, a["up%d" % l])[order[l]]
    return out


def _is_z_ordered(a):
    """Children of a voxel are contiguous at every level and parents are numbered in the order of their first child."""
    for l in range(4):
        p = a["parent_row%d" % l]
        if len(p) and not (p[0] == 0 and np.all(np.diff(p) >= 0) and np.all(np.diff(p) <= 1)):
            return False
    return True


def _compare(coords_np, n_dev=None, exact=False):
    coords = torch.from_numpy(np.ascontiguousarray(coords_np, np.int32)).to(DEV)
    n = int(coords.shape[0])
    nd = None if n_dev is None else torch.tensor([n_dev], dtype=torch.int32, device=DEV)
    a_new = _arrays(*_prepare(coords, "sorted", nd), n, n_dev)
    a_old = _arrays(*_prepare(coords, "hash", nd), n, n_dev)
    assert set(a_new) == set(a_old)
    assert np.array_equal(a_new["ucoords"][a_new["perm"]], a_new["coords0"])
    assert _is_z_ordered(a_new) and _is_z_ordered(a_old)
    if exact:                               # box minimum in [0, 16): the two curves have the same origin -> identical arrays
        for k in sorted(a_old):
            assert a_new[k].shape == a_old[k].shape, k
            assert np.array_equal(a_new[k], a_old[k]), k
    c_new, c_old = _canonical(a_new), _canonical(a_old)
    for k in sorted(c_old):
        assert c_new[k].shape == c_old[k].shape, k
        assert np.array_equal(c_new[k], c_old[k]), k
    return a_new


def _bench_coords(copies=1):
    batch, _, _ = synth.make_val_batch(seed=2, copies=copies)
    return batch["xyz_voxel"]


def test_bench_scene_equals_hash_pipeline_and_oracle():
    coords = _bench_coords()
    a = _compare(coords, exact=True)
    assert a["counts"][0] == len(coords) == 146038
    # against the CPU oracle: level coordinates as sets, the k = 3 map of level 2 through coordinates
    ref = R.CoordinateManager(coords)
    for l, s in enumerate((1, 2, 4, 8, 16)):
        want = ref.get_coords(s)
        got = a["coords%d" % l]
        assert len(got) == len(want)
        assert set(map(tuple, got.tolist())) == set(map(tuple, want.tolist())), "level %d" % l
    c2 = a["coords2"].astype(np.int64)
    index = R.KeyIndex(c2)
    offs = R.kernel_offsets(3, 4)
    for k in range(27):
        q = c2.copy()
        q[:, 1:] += offs[k][None, :]
        assert np.array_equal(a["k3_2"][:, k], index.lookup(q)), "k3 level 2 offset %d" % k
    # rows are in Z-order: external row perm[p] sits at sorted position p, and the map of level 0 is consistent with it
    assert np.array_equal(a["ucoords"][a["perm"]], a["coords0"])


def test_three_copies_batch_index():
    _compare(_bench_coords(copies=3), exact=True)


def test_duplicates_many_batches_negative_coordinates():
    rng = np.random.default_rng(5)
    parts = []
    for b in range(37):                                      # local scenes: small boxes, many duplicated voxels
        m = int(rng.integers(200, 3000))
        c = rng.integers(-40, 60, (m, 3)) + rng.integers(-300, 300, (1, 3))
        parts.append(np.concatenate([np.full((m, 1), b), c], 1))
        parts.append(parts[-1][rng.integers(0, m, m // 2)])   # duplicates, in another order
    coords = np.concatenate(parts).astype(np.int32)
    a = _compare(coords)
    assert a["counts"][0] < len(coords)
    # first occurrence wins, survivors ascending (ME / oracle convention)
    first, inverse = R.unique_first(coords)
    assert np.array_equal(a["unique_index"], first) and np.array_equal(a["inverse"], inverse)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4097])
def test_small_inputs(n):
    rng = np.random.default_rng(n)
    coords = np.concatenate([np.zeros((n, 1), np.int64), rng.integers(0, 9, (n, 3))], 1).astype(np.int32)
    _compare(coords)


def test_wide_box_uses_the_fourth_digit():
    rng = np.random.default_rng(9)
    n = 20000
    xyz = rng.integers(-2000, 2000, (n, 3))                   # 12 bits per axis + 6 batch bits = 42 key bits
    coords = np.concatenate([rng.integers(0, 50, (n, 1)), xyz], 1).astype(np.int32)
    _compare(coords)


def test_capacity_form_device_count():
    coords = _bench_coords()[:50000].copy()
    padded = np.concatenate([coords, np.full((7000, 4), 12345, np.int32)])     # garbage beyond the count
    a = _compare(padded, n_dev=len(coords))
    assert a["counts"][0] == len(coords)


def test_box_beyond_44_key_bits_falls_back_to_the_hash_pipeline():
    n = 1000
    rng = np.random.default_rng(2)
    coords = np.concatenate([rng.integers(0, 60000, (n, 1)), rng.integers(-32000, 32000, (n, 3))], 1).astype(np.int32)
    arena, P = _prepare(torch.from_numpy(coords).to(DEV), "sorted")
    counts = _view(arena, P.pyramid.counts, 5, torch.int32)
    assert counts[0] == -1
    # the coordinate manager retries such a lineage through the hash-table pipeline (same arena, same layout): wide scenes
    # and large batch indices keep working, as they did before the sorted pipeline became the default
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(DEV), prepare="sorted")
    assert cm.num_rows(1) == len(np.unique(coords, axis=0))
    plain = ME.CoordinateManager(torch.from_numpy(coords).to(DEV), prepare="plain")
    sv = cm.sorted()
    a = sv.pyramid.coordinates(1).cpu().numpy()
    assert {tuple(r) for r in a.tolist()} == {tuple(r) for r in plain.coordinates(1).cpu().numpy().tolist()}
    for s_ in (2, 4, 8, 16):
        assert sv.pyramid.n[sv.pyramid.level_index(s_)] == plain.num_rows(s_)
    # ... and a coordinate outside the 16-bit range is still an error
    bad = coords.copy()
    bad[0, 1] = 40000
    with pytest.raises(ValueError):
        ME.CoordinateManager(torch.from_numpy(bad).to(DEV), prepare="sorted").num_rows(1)


def test_c4_sized_scene():
    batch, _, info = synth.make_val_batch(copies=1, seed=3, room=(6.4, 5.2, 2.7), n_boxes=14, pitch=0.0112, voxel=0.01)
    assert info["n_voxels"] > 1000000
    _compare(batch["xyz_voxel"])


def test_probe_of_a_full_table_ends():
    """Open-addressing probes are bounded by the table capacity (csrc/coords_dev.h): a table without a single empty slot --
    what a skipped clear or a corrupted arena leaves behind -- answers "not found" instead of spinning for ever."""
    lib = N.lib()
    cap = 1024
    keys = torch.full((cap,), 0x0123456789abcdef, dtype=torch.int64, device=DEV)        # every slot taken by a foreign key
    vals = torch.zeros(cap, dtype=torch.int32, device=DEV)
    coords = torch.tensor([[0, 1, 2, 3], [0, 5, 5, 5]], dtype=torch.int32, device=DEV)
    nbr = torch.full((2, 27), 7, dtype=torch.int32, device=DEV)
    rc = lib.pbn_kernel_map_cube(N.ptr(coords), None, 2, 3, 1, int(CV.X_FASTEST), N.ptr(keys), N.ptr(vals), cap, N.ptr(nbr),
                                 N.current_stream())
    N.check(rc, "pbn_kernel_map_cube")
    torch.cuda.synchronize()
    assert bool((nbr == -1).all())
