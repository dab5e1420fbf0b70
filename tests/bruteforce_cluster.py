"""Independent brute-force statement of the grouping specification.

Written from the specification, NOT from oracle/pb_cluster_ref.c: dense O(n^2) distance matrices, connected
components from scipy, set-style labelling.  Used only to cross-check the C oracle on small inputs.

Specification (one batch segment, points 0..n-1, shifted coords P, original coords O, class s_i):
  1. nb_i = #{j : d2(P_i,P_j) <= r*r} - 1, d2 = (dx*dx + dy*dy) + dz*dz in unfused binary32.
  2. HP_i  <=> nb_i >= min_pts.
  3. K = connected components of the graph on HPs (all classes) with edges d2 <= r*r.
  4. clusters = non-empty sets {HP in K with class s}; ordered by their smallest member index.
  5. HP -> its cluster.  LP i -> the LAST (= largest id) cluster (K, s_i) such that i is within r of some HP of K.
  6. drop clusters with float32(size) < float32(mean_count[s-2]) * float32(para_f); renumber keeping order.
  7. (nv_flag) every unlabelled point -> label of the nearest labelled point of its class in ORIGINAL coords,
     ties -> highest index; no labelled point of that class -> label of the highest-index labelled point;
     nothing labelled -> stays -1.
  8. centre of cluster c = sequential fp32 running mean M += (p - M)/N over members in index order (shifted coords).
  9. ids are global across segments (running offset of kept clusters).
"""
import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import connected_components

MEAN_COUNT = np.array([3917., 12056., 2303., 8331., 3948., 3166., 5629., 11719., 1003., 3317., 4912., 10221.,
                       3889., 4136., 2120., 945., 3967., 2589.], dtype=np.float32)


def _d2(a, b):
    a = a.astype(np.float32)
    b = b.astype(np.float32)
    dx = a[:, None, 0] - b[None, :, 0]
    dy = a[:, None, 1] - b[None, :, 1]
    dz = a[:, None, 2] - b[None, :, 2]
    return ((dx * dx).astype(np.float32) + (dy * dy).astype(np.float32)).astype(np.float32) + (dz * dz).astype(np.float32)


def _segment(P, O, sem, radius, min_pts, para_f, nv_flag, id_base):
    n = P.shape[0]
    r = np.float32(radius)
    r2 = np.float32(r * r)
    within = _d2(P, P) <= r2
    nb = within.sum(1).astype(np.int32) - 1
    hp = nb >= min_pts
    label = np.full(n, -1, dtype=np.int64)
    clusters = []  # (seed, comp, sem)
    if hp.any():
        hp_idx = np.nonzero(hp)[0]
        sub = within[np.ix_(hp_idx, hp_idx)]
        ncomp, comp_of = connected_components(csr_matrix(sub), directed=False)
        comp = np.full(n, -1, dtype=np.int64)
        comp[hp_idx] = comp_of
        seen = {}
        for i in hp_idx:
            key = (int(comp[i]), int(sem[i]))
            if key not in seen:
                seen[key] = len(clusters)
                clusters.append(key)
            label[i] = seen[key]
        lp_idx = np.nonzero(~hp)[0]
        for i in lp_idx:
            adj = hp_idx[within[i, hp_idx]]
            best = -1
            for k in set(int(c) for c in comp[adj]):
                cid = seen.get((k, int(sem[i])), -1)
                best = max(best, cid)
            label[i] = best
    # filter
    keep_map = {}
    kept_sem = []
    pf = np.float32(para_f)
    for cid, (_, s) in enumerate(clusters):
        size = int((label == cid).sum())
        thr = np.float32(MEAN_COUNT[s - 2] * pf)
        if not (np.float32(size) < thr):
            keep_map[cid] = len(kept_sem)
            kept_sem.append(s)
    new_label = np.array([keep_map.get(int(c), -1) if c >= 0 else -1 for c in label], dtype=np.int64)
    label = new_label
    # noise
    if nv_flag:
        noise = np.nonzero(label < 0)[0]
        assigned = np.nonzero(label >= 0)[0]
        if len(noise) and len(assigned):
            dd = _d2(O[noise], O[assigned])
            out = label.copy()
            for a, i in enumerate(noise):
                same = sem[assigned] == sem[i]
                if same.any():
                    cand = assigned[same]
                    dv = dd[a][same]
                    m = dv.min()
                    out[i] = label[cand[np.nonzero(dv == m)[0].max()]]
                else:
                    out[i] = label[assigned.max()]
            label = out
    # centres
    centers = np.zeros((len(kept_sem), 3), dtype=np.float32)
    for c in range(len(kept_sem)):
        M = np.zeros(3, dtype=np.float32)
        N = 0
        for i in np.nonzero(label == c)[0]:
            N += 1
            M = (M + (P[i].astype(np.float32) - M) / np.float32(N)).astype(np.float32)
        centers[c] = M
    out_label = np.where(label >= 0, label + id_base, -1).astype(np.int32)
    return out_label, nb, centers, np.array(kept_sem, dtype=np.int32)


def binary_cluster(off_xyz, org_xyz, sem, seg_len, radius, min_pts, para_f=0.05, nv_flag=True):
    off_xyz = np.asarray(off_xyz, dtype=np.float32).reshape(-1, 3)
    org_xyz = np.asarray(org_xyz, dtype=np.float32).reshape(-1, 3)
    sem = np.asarray(sem, dtype=np.int32)
    ids, dens, ctrs, sems, nums = [], [], [], [], []
    start = 0
    base = 0
    for ln in seg_len:
        ln = int(ln)
        sl = slice(start, start + ln)
        if ln == 0:
            nums.append(0)
            continue
        lab, nb, c, s = _segment(off_xyz[sl], org_xyz[sl], sem[sl], radius, min_pts, para_f, nv_flag, base)
        ids.append(lab)
        dens.append(nb)
        ctrs.append(c)
        sems.append(s)
        nums.append(len(s))
        base += len(s)
        start += ln
    cat = lambda xs, dt, shape: (np.concatenate(xs).astype(dt) if xs else np.zeros(shape, dtype=dt))
    return dict(cluster_id=cat(ids, np.int32, (0,)), cluster_num=np.array(nums, dtype=np.int32),
                den_queue=cat(dens, np.int32, (0,)), center=cat(ctrs, np.float32, (0, 3)).reshape(-1, 3),
                clt_sem=cat(sems, np.int32, (0,)))
