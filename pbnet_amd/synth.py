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


TTA_ANGLES = (63.0, 183.0, 303.0)   # /root/reference/datasets/scannetv2/dataset_preprocess.py:91


def make_val_batch(seed=2, copies=1, room=(4.0, 3.2, 2.6), n_boxes=12, pitch=0.0225, classes=None, voxel=0.02):
    """A val-style batch dict as dataset_preprocess.valMerge builds it (:308-385): `copies` rotated copies of one
    scene (:324,344), each voxelised at `voxel` m, plus teacher-forced head outputs for the grouping stage.

    Returns (batch, teacher, info): batch has xyz_voxel i32[V,4], feat_voxel f32[V,6], xyz_original f32[N,3],
    v2p_index i64[N], ins i64[N]; teacher has sem_score f32[N,20] and offset f32[N,3]; numpy arrays throughout."""
    sc = synth_room(seed=seed, pitch=pitch, room=room, n_boxes=n_boxes)
    if classes is not None:
        box = sc["ins"] >= 0
        sc["sem"][box] = np.asarray(classes)[sc["ins"][box] % len(classes)]
    sem_pred, offset = teacher_forced_heads(sc, seed=seed)
    xyz_l, vox_l, feat_l, v2p_l, off_l, ins_l = [], [], [], [], [], []
    nv = 0
    for b in range(copies):
        th = np.deg2rad(TTA_ANGLES[b % 3])
        rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
        xyz = sc["xyz"].astype(np.float64) @ rot.T
        xyz = (xyz - xyz.min(0)).astype(np.float32)
        q, first, inv = voxelize_numpy(xyz, voxel)
        feats = np.concatenate([sc["rgb"], (sc["normal"].astype(np.float64) @ rot.T).astype(np.float32)], 1)
        vox_l.append(np.concatenate([np.full((len(q), 1), b, np.int32), q], 1))
        feat_l.append(feats[first].astype(np.float32))
        v2p_l.append(inv + nv)
        nv += len(q)
        xyz_l.append(xyz)
        off_l.append((offset.astype(np.float64) @ rot.T).astype(np.float32))
        ins_l.append(np.where(sc["ins"] >= 0, sc["ins"] + b * n_boxes, -100))
    sem = np.tile(sem_pred, copies)
    score = np.full((len(sem), 20), -5.0, np.float32)
    score[np.arange(len(sem)), sem] = 5.0
    batch = dict(xyz_voxel=np.concatenate(vox_l).astype(np.int32), feat_voxel=np.concatenate(feat_l),
                 xyz_original=np.concatenate(xyz_l), v2p_index=np.concatenate(v2p_l).astype(np.int64),
                 ins=np.concatenate(ins_l).astype(np.int64))
    teacher = dict(sem_score=score, offset=np.concatenate(off_l))
    info = dict(n_points=int(len(sem)), n_voxels=int(nv), copies=copies, seed=seed)
    return batch, teacher, info


def make_train_batch(seed=10, copies=1, **kw):
    """A train-style batch dict as dataset_preprocess.trainMerge builds it (:178-305): make_val_batch's geometry plus
    the label tensors model_fn reads (PBNet.py:359-363): `sem` i64[N] (-100 = ignore), `inst_info` f32[N,9] (instance
    mean xyz in columns 0:3, dataset_preprocess.py:121-147), `instance_pointnum` i32[I].  Semantic labels are the
    teacher's arg-max (so the semantic loss has a non-trivial value); floor / walls carry instance -100."""
    batch, teacher, info = make_val_batch(seed=seed, copies=copies, **kw)
    n = batch["xyz_original"].shape[0]
    ins = batch["ins"]
    n_inst = int(ins.max()) + 1 if (ins >= 0).any() else 0
    inst_info = np.zeros((n, 9), np.float32)
    pointnum = np.zeros(n_inst, np.int32)
    for i in range(n_inst):
        m = ins == i
        pointnum[i] = int(m.sum())
        if pointnum[i]:
            pts = batch["xyz_original"][m]
            inst_info[m, 0:3] = pts.mean(0)
            inst_info[m, 3:6] = pts.min(0)
            inst_info[m, 6:9] = pts.max(0)
    batch = dict(batch, sem=teacher["sem_score"].argmax(1).astype(np.int64), inst_info=inst_info,
                 instance_pointnum=pointnum)
    return batch, teacher, info
