"""Synthetic ScanNet-like scenes for tests and bench.py (SURVEY.md section 8d "Synthetic inputs").

The reference ships no data and ScanNet is licence-gated; every measured number in this repository is on
scenes produced here.  numpy only -- usable on CPU and on the GPU box.
"""
import numpy as np


def _sample_rect(rng, origin, u, v, normal, pitch):
    """Jittered lattice on the rectangle origin + a*u + b*v, a,b in [0,1]."""
    lu = float(np.linalg.norm(u))
    lv = float(np.linalg.norm(v))
    nu = max(int(lu / pitch), 1)
    nv = max(int(lv / pitch), 1)
    a, b = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (a.reshape(-1) + 0.5 + rng.uniform(-0.3, 0.3, nu * nv)) / nu
    b = (b.reshape(-1) + 0.5 + rng.uniform(-0.3, 0.3, nu * nv)) / nv
    pts = origin[None, :] + a[:, None] * u[None, :] + b[:, None] * v[None, :]
    pts = pts + normal[None, :] * rng.normal(0.0, 0.003, (pts.shape[0], 1))
    nrm = normal[None, :] + rng.normal(0.0, 0.05, (pts.shape[0], 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return pts, nrm


def synth_room(seed=2, pitch=0.0225, room=(4.0, 3.2, 2.6), n_boxes=12):
    """Axis-aligned room (floor + 4 walls) with ``n_boxes`` boxes standing on the floor.

    Returns dict: xyz f32[N,3] (min at 0), rgb f32[N,3] in [-1,1], normal f32[N,3], sem i64[N]
    (floor 0, wall 1, box j -> 2 + j%18), ins i64[N] (box id, -100 for floor/wall), centroid f32[N,3]
    (instance centroid, own position for floor/wall).
    """
    rng = np.random.default_rng(seed)
    X, Y, Z = room
    ex, ey, ez = np.eye(3)
    parts = []

    def add(origin, u, v, normal, sem, ins):
        p, n = _sample_rect(rng, np.asarray(origin, float), np.asarray(u, float), np.asarray(v, float),
                            np.asarray(normal, float), pitch)
        parts.append((p, n, np.full(len(p), sem), np.full(len(p), ins)))

    add((0, 0, 0), X * ex, Y * ey, ez, 0, -100)
    add((0, 0, 0), X * ex, Z * ez, ey, 1, -100)
    add((0, Y, 0), X * ex, Z * ez, -ey, 1, -100)
    add((0, 0, 0), Y * ey, Z * ez, ex, 1, -100)
    add((X, 0, 0), Y * ey, Z * ez, -ex, 1, -100)
    for j in range(n_boxes):
        sx, sy, sz = rng.uniform(0.4, 1.6), rng.uniform(0.4, 0.9), rng.uniform(0.4, 1.1)
        sx, sy, sz = min(sx, 0.6 * X), min(sy, 0.6 * Y), min(sz, 0.8 * Z)
        ox, oy = rng.uniform(0.05 * X, 0.95 * X - sx), rng.uniform(0.05 * Y, 0.95 * Y - sy)
        o = np.array([ox, oy, 0.0])
        sem = 2 + (j % 18)
        add(o + sz * ez, sx * ex, sy * ey, ez, sem, j)                # top
        add(o, sx * ex, sz * ez, -ey, sem, j)                          # 4 sides
        add(o + sy * ey, sx * ex, sz * ez, ey, sem, j)
        add(o, sy * ey, sz * ez, -ex, sem, j)
        add(o + sx * ex, sy * ey, sz * ez, ex, sem, j)
    xyz = np.concatenate([p[0] for p in parts]).astype(np.float64)
    nrm = np.concatenate([p[1] for p in parts])
    sem = np.concatenate([p[2] for p in parts]).astype(np.int64)
    ins = np.concatenate([p[3] for p in parts]).astype(np.int64)
    xyz -= xyz.min(0)
    centroid = xyz.copy()
    for j in range(n_boxes):
        m = ins == j
        if m.any():
            centroid[m] = xyz[m].mean(0)
    rgb = rng.uniform(-1.0, 1.0, (len(xyz), 3))
    return dict(xyz=xyz.astype(np.float32), rgb=rgb.astype(np.float32), normal=nrm.astype(np.float32), sem=sem,
                ins=ins, centroid=centroid.astype(np.float32))


def teacher_forced_heads(scene, seed=0, flip=0.02, offset_sigma=0.02):
    """Semantic prediction = GT with ``flip`` random flips; offset = (centroid - xyz) + N(0, sigma).

    Stands in for the network heads when the grouping stage is exercised on its own."""
    rng = np.random.default_rng(1000 + seed)
    n = scene["xyz"].shape[0]
    sem = scene["sem"].copy()
    f = rng.uniform(size=n) < flip
    sem[f] = rng.integers(0, 20, int(f.sum()))
    offset = (scene["centroid"] - scene["xyz"]) + rng.normal(0.0, offset_sigma, (n, 3)).astype(np.float32)
    offset[scene["ins"] < 0] = rng.normal(0.0, offset_sigma, (int((scene["ins"] < 0).sum()), 3))
    return sem, offset.astype(np.float32)


def voxelize_numpy(xyz, voxel_size):
    """floor(xyz / voxel) -> unique (first occurrence, ascending original index) + inverse map.

    Host-side helper used to prepare bench/test inputs; the device voxeliser lives in csrc/coords.hip."""
    q = np.floor(xyz.astype(np.float64) / voxel_size).astype(np.int32)
    _, first, inverse = np.unique(q, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    return q[first[order]], first[order], rank[inverse.reshape(-1)]
