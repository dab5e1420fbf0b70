"""Generates tests/golden/cluster_G*.npz (SURVEY.md section 8c fixtures G1-G9).

The reference (CUDA-only PB_lib) cannot run here and holds no golden vectors, so these fixtures come from the C
oracle (oracle/pb_cluster_ref.c) and are cross-checked against the independent brute-force statement
(tests/bruteforce_cluster.py) before being written: PARITY UNPINNED with respect to the upstream binary.

Run from the repo root:  python tests/golden/make_cluster_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pb_cluster_ref as oracle  # noqa: E402
import bruteforce_cluster as brute  # noqa: E402
from pbnet_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def blob(rng, center, n, sigma):
    return (np.asarray(center, np.float32)[None, :] + rng.normal(0, sigma, (n, 3))).astype(np.float32)


def plate(center, nx=8, pitch=0.004):
    c = (np.arange(nx) - (nx - 1) / 2.0) * pitch
    y, z = np.meshgrid(c, c, indexing="ij")
    p = np.stack([np.zeros(nx * nx), y.reshape(-1), z.reshape(-1)], 1)
    return (p + np.asarray(center)[None, :]).astype(np.float32)


def cases():
    out = {}
    # G1: two well separated blobs + sparse noise, class 17 (threshold 47.25)
    rng = np.random.default_rng(101)
    P = np.concatenate([blob(rng, (0.3, 0.3, 0.3), 200, 0.01), blob(rng, (1.0, 0.4, 0.3), 150, 0.01),
                        rng.uniform(0, 1.5, (60, 3)).astype(np.float32)])
    perm = rng.permutation(len(P))
    P = P[perm]
    O = (P + rng.normal(0, 0.05, P.shape)).astype(np.float32)
    out["G1"] = dict(off=P, org=O, sem=np.full(len(P), 17), seg=[len(P)], radius=0.04, min_pts=31)

    # G2a: binary lattice, pair distances exactly r (r = 2^-5, pitch 2^-6); G2b: decimal lattice r=0.04 pitch 0.02
    g = np.arange(9)
    L = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    out["G2a"] = dict(off=L * np.float32(2.0 ** -6), org=L * np.float32(2.0 ** -6), sem=np.full(len(L), 17),
                      seg=[len(L)], radius=2.0 ** -5, min_pts=31)
    Lb = (L * np.float32(0.02)).astype(np.float32) + np.float32(0.37)
    out["G2b"] = dict(off=Lb, org=Lb, sem=np.full(len(L), 10), seg=[len(L)], radius=0.04, min_pts=31)
    # G2c: lattice at pitch r/2 +- 1 ulp jitter: distances r +- ulp
    rng = np.random.default_rng(102)
    Lc = (L * np.float32(0.02)).astype(np.float32)
    Lc = np.nextafter(Lc, np.where(rng.uniform(size=Lc.shape) < 0.5, np.float32(-1), np.float32(1)).astype(np.float32))
    out["G2c"] = dict(off=Lc, org=Lc, sem=np.full(len(L), 17), seg=[len(L)], radius=0.04, min_pts=31)

    # G3: border LP within r of HPs of two different components -> takes the larger id.
    A = plate((0.0, 0.0, 0.0))
    B = plate((0.0796, 0.0, 0.0))
    M = np.array([[0.0398, 0.0, 0.0]], np.float32)
    P = np.concatenate([M, B, A])  # B is seeded first (id 0), A second (id 1); M must take id 1
    out["G3"] = dict(off=P, org=P.copy(), sem=np.full(len(P), 17), seg=[len(P)], radius=0.04, min_pts=31)

    # G4: sizes 120 / 105 / 106 / 60 on class 16 (threshold exactly 106.0): 105 and 60 dropped, 106 kept.
    rng = np.random.default_rng(104)
    P = np.concatenate([blob(rng, (0.2, 0.2, 0.2), 120, 0.004), blob(rng, (0.6, 0.2, 0.2), 105, 0.004),
                        blob(rng, (1.0, 0.2, 0.2), 106, 0.004), blob(rng, (1.4, 0.2, 0.2), 60, 0.004),
                        blob(rng, (1.8, 0.2, 0.2), 110, 0.004)])
    O = (P + rng.normal(0, 0.2, P.shape)).astype(np.float32)
    out["G4"] = dict(off=P, org=O, sem=np.full(len(P), 16), seg=[len(P)], radius=0.04, min_pts=31)

    # G5: noise equidistant (ORIGINAL coords) to assigned points of two clusters -> highest index wins; density is
    # decided in SHIFTED coords.
    rng = np.random.default_rng(105)
    Pa = blob(rng, (0.0, 0.0, 0.0), 80, 0.003)
    Pb = blob(rng, (1.0, 0.0, 0.0), 80, 0.003)
    Pn = np.array([[5.0, 5.0, 5.0], [6.0, 5.0, 5.0], [7.0, 5.0, 5.0]], np.float32)
    Oa = np.tile(np.array([[0.0, 0.0, 0.0]], np.float32), (80, 1))
    Oa[:40, 0] = -1.0
    Oa[40:, 0] = 1.0                       # cluster a has original points at x=-1 and x=+1
    Ob = np.tile(np.array([[0.0, 1.0, 0.0]], np.float32), (80, 1))
    Ob[:40, 1] = 1.0
    Ob[40:, 1] = -1.0
    On = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.5], [-1.0, 0.0, 0.0]], np.float32)  # 0,1: 4-way ties; 2: unique
    P = np.concatenate([Pa[:40], Pn[:1], Pb[:40], Pn[1:2], Pa[40:], Pb[40:], Pn[2:]])
    O = np.concatenate([Oa[:40], On[:1], Ob[:40], On[1:2], Oa[40:], Ob[40:], On[2:]])
    out["G5"] = dict(off=P, org=O, sem=np.full(len(P), 17), seg=[len(P)], radius=0.04, min_pts=31)

    # G6: B=4 segments: normal, empty, zero surviving clusters, normal (global id offsets / cluster_num).
    rng = np.random.default_rng(106)
    s0 = np.concatenate([blob(rng, (0.2, 0.2, 0.2), 90, 0.005), blob(rng, (0.9, 0.2, 0.2), 70, 0.005)])
    s2 = rng.uniform(0, 3, (50, 3)).astype(np.float32)
    s3 = np.concatenate([blob(rng, (0.2, 0.9, 0.2), 64, 0.005), rng.uniform(0, 1, (10, 3)).astype(np.float32)])
    P = np.concatenate([s0, s2, s3])
    O = (P + rng.normal(0, 0.02, P.shape)).astype(np.float32)
    out["G6"] = dict(off=P, org=O, sem=np.full(len(P), 17), seg=[len(s0), 0, len(s2), len(s3)], radius=0.04,
                     min_pts=31)

    # G7: duplicate coordinates (each duplicate counts as a neighbour, distance 0).
    rng = np.random.default_rng(107)
    base = blob(rng, (0.5, 0.5, 0.5), 30, 0.01)
    P = np.concatenate([base, base, base[:10], rng.uniform(0, 1, (20, 3)).astype(np.float32)])
    P = P[rng.permutation(len(P))]
    out["G7"] = dict(off=P, org=P[::-1].copy(), sem=np.full(len(P), 17), seg=[len(P)], radius=0.04, min_pts=31)

    # G8: mixed classes in one call (general "component x class" rule, class-restricted noise, fallback to the
    # highest-index assigned point for a class with no assigned point, a (component, class) cluster below threshold).
    rng = np.random.default_rng(108)
    P = np.concatenate([blob(rng, (0.2, 0.2, 0.2), 320, 0.006), blob(rng, (0.8, 0.2, 0.2), 260, 0.006),
                        blob(rng, (1.4, 0.2, 0.2), 90, 0.006), rng.uniform(0, 1.6, (40, 3)).astype(np.float32)])
    sem = np.concatenate([rng.choice([17, 10], 320), rng.choice([17, 10, 19], 260, p=[0.5, 0.4, 0.1]),
                          np.full(90, 10), rng.choice([17, 10, 19, 4], 40)])
    perm = rng.permutation(len(P))
    P, sem = P[perm], sem[perm]
    O = (P + rng.normal(0, 0.03, P.shape)).astype(np.float32)
    out["G8"] = dict(off=P, org=O, sem=sem, seg=[len(P)], radius=0.04, min_pts=31)
    out["G8b"] = dict(off=P, org=O, sem=sem, seg=[400, len(P) - 400], radius=0.04, min_pts=31)
    out["G4n"] = dict(out["G4"], nv=False)

    # G9: synthetic room, teacher-forced heads, one class mix per seed; 3 segments (the 3-copy TTA batch shape).
    for seed in range(5):
        sc = synth.synth_room(seed=20 + seed, pitch=0.05, room=(2.4, 2.0, 1.6), n_boxes=4 + seed)
        sem_pred, offset = synth.teacher_forced_heads(sc, seed=seed)
        pick = [17, 10, 17, 4, 16][seed]
        idx = np.nonzero(sem_pred >= 2)[0]           # every box point predicted as ONE class -> several instances
        rng = np.random.default_rng(900 + seed)
        segs, offs, orgs = [], [], []
        for copy in range(3):
            th = np.deg2rad([63.0, 183.0, 303.0][copy])
            R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], np.float32)
            o = (sc["xyz"][idx] @ R.T).astype(np.float32)
            d = ((sc["xyz"][idx] + offset[idx]) @ R.T).astype(np.float32)
            keep = rng.uniform(size=len(idx)) < 0.9
            offs.append(d[keep])
            orgs.append(o[keep])
            segs.append(int(keep.sum()))
        P = np.concatenate(offs)
        O = np.concatenate(orgs)
        out["G9_%d" % seed] = dict(off=P, org=O, sem=np.full(len(P), pick), seg=segs, radius=0.04, min_pts=31)
    return out


def main():
    for name, c in cases().items():
        off = np.ascontiguousarray(c["off"], np.float32)
        org = np.ascontiguousarray(c["org"], np.float32)
        sem = np.ascontiguousarray(c["sem"], np.int32)
        seg = np.asarray(c["seg"], np.int32)
        nv = bool(c.get("nv", True))
        res = oracle.binary_cluster(off, org, sem, seg, c["radius"], c["min_pts"], nv_flag=nv)
        chk = brute.binary_cluster(off, org, sem, seg, c["radius"], c["min_pts"], nv_flag=nv)
        for k in ("cluster_id", "cluster_num", "den_queue", "clt_sem"):
            assert np.array_equal(res[k], chk[k]), (name, k)
        assert np.array_equal(res["center"].view(np.int32), chk["center"].view(np.int32)), (name, "center bits")
        np.savez_compressed(os.path.join(OUT, "cluster_%s.npz" % name), off=off, org=org, sem=sem, seg=seg,
                            radius=np.float32(c["radius"]), min_pts=np.int32(c["min_pts"]), nv_flag=np.int32(nv), **res)
        nhp = int((res["den_queue"] >= c["min_pts"]).sum())
        print("%-5s n=%5d segs=%s clusters=%s HP=%d noise_left=%d" % (
            name, len(off), seg.tolist(), res["cluster_num"].tolist(), nhp, int((res["cluster_id"] < 0).sum())))


if __name__ == "__main__":
    main()
