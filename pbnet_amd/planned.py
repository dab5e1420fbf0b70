"""Capacity-planned inference: PBNet.forward (task 'test', /root/reference/network/PBNet.py:113-280) as a FIXED launch
sequence with no host synchronisation inside.

The reference -- and the size-exact path of pbnet_amd.network.PBNet -- learn every data-dependent size on the host (class
populations, number of clusters, their centres and sizes, rows per local scene, proposal rows, the row counts of three
coordinate pyramids: six to seven device->host copies per forward).  Here every such size stays on the device:

  * buffers are allocated at CAPACITIES (Capacities, normally measured once on a scene of the same size class by
    `measure_capacities` and padded by a slack factor);
  * the decisions the reference takes on the host are three small device launches (csrc/plan.hip: class gate, local-scene
    plan with the kNN of the cluster centres, proposal offsets);
  * every kernel bounds itself by a device-side count and the launch grids are sized by the capacities;
  * an overflow of any capacity raises a flag in the device-side counts instead of writing out of bounds; `finish` (the ONE
    read-back of the forward) reports it and the caller falls back to the size-exact path.

Because the sequence is fixed it can be captured in a HIP graph (`PlannedForward.capture`) and replayed: BASELINE configs[4].
Results are bit-identical to the size-exact path whenever the same kernel configurations are chosen (capacities equal to
the true sizes); with slack the convolution tiles / pooling splits of a level may differ, which changes fp32 summation
order only (tests/test_planned_gpu.py)."""
import ctypes
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

from . import _native as N
from . import pbnet_ops, stage_ops
from .MinkowskiEngine import conventions as CV
from .MinkowskiEngine.conv import _DT, SPLITK_WORKSPACE_BYTES

CNT = SimpleNamespace(POINTS=0, CLUSTERS=1, ENTRIES=2, ROWS=3, SCENES=4, PROPOSAL_ROWS=5, PROPOSALS=6, OVERFLOW=7, WORDS=16)
OVF_NAMES = {1: "selected points", 2: "clusters", 4: "local-scene entries", 8: "local-scene rows",
             16: "clusters of one (class, batch) segment", 32: "batch index outside [0, cluster_batch)", 64: "rows of a level", 128: "a segment of more than 25 clusters (torch.cdist ranks through its matrix-multiply path: host plan)"}
MASK_THD = 0.45
LOCAL_VOXEL = 0.02
_DEBUG = os.environ.get("PBN_PLANNED_DEBUG", "0") == "1"
HINTS = os.environ.get("PBN_PLANNED_HINTS", "1") != "0"      # kernel choice by the rows expected, not by the capacities
_STOP = os.environ.get("PBN_PLANNED_STOP", "")      # debugging: end the launch sequence after the named stage


def _dbg(stage, counts=None):
    """PBN_PLANNED_DEBUG=1: synchronise after every stage and say where the forward is (hang / fault localisation)."""
    if _DEBUG:
        torch.cuda.synchronize()
        sys.stderr.write("[planned] %s%s\n" % (stage, "" if counts is None else " counts=%s" % counts.tolist()[:8]))
        sys.stderr.flush()


class Capacities(object):
    """Buffer sizes of one planned forward.  lv1 / lv2 / lv3: rows per tensor stride (1..16) of the backbone, mask-branch
    and score-branch coordinate pyramids; points: points entering the grouping; clusters; entries / rows of the local scenes."""
    FIELDS = ("n_points", "n_voxels", "lv1", "lv2", "lv3", "points", "clusters", "entries", "rows")

    def __init__(self, **kw):
        for k in self.FIELDS:
            setattr(self, k, kw[k])
        # rows EXPECTED per level (the measured ones): the kernels' family / tile choice follows these, the grids the capacities
        # (round 5: choosing by the 1.25 x capacities picked slower kernels -- pbn_unet_set_rows_hint)
        self.expect = kw.get("expect") or {"lv1": list(self.lv1), "lv2": list(self.lv2), "lv3": list(self.lv3)}

    def padded(self, slack=1.25, quantum=256):
        """The same plan with head room: every data-dependent size x slack, rounded up to a multiple of `quantum`."""
        def up(v, lo=quantum):
            return int(max(lo, -(-int(v * slack) // quantum) * quantum))
        return Capacities(n_points=self.n_points, n_voxels=self.n_voxels,
                          lv1=[self.n_voxels] + [up(v) for v in self.lv1[1:]], lv2=[up(v) for v in self.lv2],
                          lv3=[up(v) for v in self.lv3], points=min(self.n_points, up(self.points)),
                          clusters=up(self.clusters, 64), entries=up(self.entries, 64), rows=up(self.rows), expect=self.expect)

    def __repr__(self):
        return "Capacities(%s)" % ", ".join("%s=%r" % (k, getattr(self, k)) for k in self.FIELDS)


def measure_capacities(model, feat_voxel, xyz_voxel, xyz_original, v2p_index, teacher=None):
    """One size-exact forward that records every data-dependent size (PBNet._last_sizes)."""
    with torch.no_grad():
        model(feat_voxel, xyz_voxel, xyz_original, v2p_index, None, 1, "test", teacher=teacher)
    s = model._last_sizes
    return Capacities(n_points=int(xyz_original.shape[0]), n_voxels=int(feat_voxel.shape[0]), lv1=list(s["lv1"]),
                      lv2=list(s.get("lv2", [1] * 5)), lv3=list(s.get("lv3", [1] * 5)), points=int(s.get("points", 1)),
                      clusters=int(s.get("clusters", 1)), entries=int(s.get("entries", 1)), rows=int(s.get("rows", 1)))


class _Lineage(object):
    """pbn_coords_prepare(_dev) of one SparseTensor lineage into a private arena; views of what the forward needs."""

    def __init__(self, coords, n_cap, n_dev, dev):
        lib = N.lib()
        self.P = N.PrepareLayout()
        nbytes = lib.pbn_coords_prepare_bytes(int(n_cap), 1, ctypes.byref(self.P))
        self.arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if n_dev is None:
            rc = lib.pbn_coords_prepare(N.ptr(coords), int(n_cap), 1, int(CV.X_FASTEST), N.ptr(self.arena), nbytes,
                                        ctypes.byref(self.P), N.current_stream())
        else:
            rc = lib.pbn_coords_prepare_dev(N.ptr(coords), ctypes.c_void_p(n_dev), int(n_cap), 1, int(CV.X_FASTEST),
                                            N.ptr(self.arena), nbytes, ctypes.byref(self.P), N.current_stream())
        N.check(rc, "pbn_coords_prepare")
        self.n_cap = int(n_cap)
        L = self.P.pyramid
        self.counts = self.view(L.counts, 5, torch.int32)                       # rows per level (device)
        self.perm = self.view(self.P.perm, n_cap, torch.int64)                  # unique row at Z-order position p
        self.inv_perm = self.view(self.P.inv_perm, n_cap, torch.int64)
        self.unique_index = self.view(self.P.unique_index, n_cap, torch.int64)  # input row of unique row u
        self.inverse = self.view(self.P.inverse, n_cap, torch.int64)            # unique row of input row i
        self.ucoords = self.view(self.P.ucoords, n_cap * 4, torch.int32).view(n_cap, 4)

    def view(self, offset, count, dtype):
        nbytes = int(count) * torch.empty(0, dtype=dtype).element_size()
        return self.arena[offset:offset + nbytes].view(dtype)

    def tables(self):
        base, L = self.arena.data_ptr(), self.P.pyramid
        vp = ctypes.c_void_p
        return ((vp * 5)(*[base + L.k3[l] for l in range(5)]), vp(base + L.k5),
                (vp * 4)(*[base + L.nbr_down[l] for l in range(4)]), (vp * 4)(*[base + L.up[l] for l in range(4)]))


class PlannedForward(object):
    def __init__(self, model, cap, dtype=torch.bfloat16, device=None):
        self.model, self.cap, self.dtype = model, cap, dtype
        self.dev = device or next(model.parameters()).device
        thr05, thr02 = model._class_thresholds()
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.thr05 = torch.tensor(thr05, **f32)
        self.thr02 = torch.tensor(thr02, **f32)
        self.kmax = torch.tensor(model._k_max_list(), dtype=torch.int32, device=self.dev)
        self.nb = 3                                                    # PBNet.py:167-170: cluster_batch outside training
        self.graph = None
        # level capacities on the device (compared with the row counts of each pyramid; built here, not inside a capture)
        self._caps_t = {id(lv): torch.tensor([int(v) for v in lv], dtype=torch.int32, device=self.dev)
                        for lv in (cap.lv1, cap.lv2, cap.lv3)}

    # ---- helpers ----------------------------------------------------------------------------------------------------
    def _unet(self, net, lin, cap_levels, feats, row_bytes, expect=None):
        """Fused U-Net on a lineage: feats are the INPUT rows (before de-duplication); returns the output slab in Z-order
        [cap_levels[0], cout] and flags a level whose row count exceeds its capacity."""
        lib = N.lib()
        dt, dev = self.dtype, self.dev
        es = torch.empty(0, dtype=dt).element_size()
        plan = net._plan(dt)
        cin_p = plan["cin_p"]
        n0 = int(cap_levels[0])
        padded = torch.empty(n0, cin_p, dtype=dt, device=dev)
        N.check(lib.pbn_gather_pad_rows_dev(ctypes.c_void_p(feats.data_ptr()), feats.stride(0) * es, int(row_bytes),
                                            N.ptr(lin.perm), N.ptr(lin.unique_index), n0,
                                            ctypes.c_void_p(lin.counts.data_ptr()), ctypes.c_void_p(padded.data_ptr()),
                                            cin_p * es, N.current_stream()), "pbn_gather_pad_rows_dev")
        n_rows = (ctypes.c_int32 * 5)(*[int(v) for v in cap_levels])
        offs = (ctypes.c_int64 * plan["n_bufs"])()
        nbytes = lib.pbn_unet_arena_bytes(plan["bufs"], plan["n_bufs"], n_rows, _DT[dt], offs)
        arena = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        k3, k5, down, up = lin.tables()
        # split-K scratch of THIS forward: the per-stream cache of the size-exact path must not be used here -- every graph
        # capture of a process runs on torch's one shared capture stream, so a cached block would be handed from the private
        # pool of one captured graph to the next and dangle once the first graph is released
        ws = self._splitk_ws
        vp = ctypes.c_void_p
        if expect is not None and HINTS:
            lib.pbn_unet_set_rows_hint((ctypes.c_int32 * 5)(*[max(1, min(int(e), int(c))) for e, c in zip(expect, cap_levels)]))
        N.check(lib.pbn_unet_forward_dev(plan["ops"], plan["n_ops"], plan["bufs"], plan["n_bufs"], n_rows,
                                         vp(lin.counts.data_ptr()), vp(padded.data_ptr()), cin_p, k3, k5, down, up,
                                         vp(arena.data_ptr()), nbytes, _DT[dt], vp(ws.data_ptr()), ws.numel(),
                                         N.current_stream()), "pbn_unet_forward_dev")
        o, width = offs[plan["out_buf"]], plan["out_width"]
        out = arena[o:o + n0 * width * es].view(dt).view(n0, width)
        cout = net.final_sematic.kernel.shape[-1]
        self._level_overflow.append((lin.counts > self._caps_t[id(cap_levels)]).any())
        return out if width == cout else out[:, :cout]

    def _mlp(self, head, feats, idx_a, idx_b, n_cap, n_dev_ptr):
        hp = stage_ops._HEADS.setdefault(id(head), stage_ops._HeadParams()).get(head)
        out = torch.empty(int(n_cap), hp.n_out, dtype=feats.dtype, device=feats.device)
        vp = ctypes.c_void_p
        rc = N.lib().pbn_mlp_rows_dev(vp(feats.data_ptr()), feats.stride(0), int(feats.shape[0]), hp.channels, N.ptr(idx_a),
                                      N.ptr(idx_b), int(n_cap),
                                      vp(n_dev_ptr), N.ptr(hp.w1), N.ptr(hp.scale), N.ptr(hp.shift), N.ptr(hp.slope), hp.hidden,
                                      N.ptr(hp.w2), N.ptr(hp.b2), hp.n_out, int(hp.sigmoid), vp(out.data_ptr()), hp.n_out,
                                      _DT[feats.dtype], N.current_stream())
        N.check(rc, "pbn_mlp_rows_dev")
        return out

    # ---- the forward: launches only -------------------------------------------------------------------------------------
    def run(self, feat_voxel, xyz_voxel, xyz_original, v2p_index, teacher=None):
        """Returns capacity-sized device tensors + the device-side counts; nothing is read back."""
        m, cap, dev, dt, lib = self.model, self.cap, self.dev, self.dtype, N.lib()
        vp = ctypes.c_void_p
        es = torch.empty(0, dtype=dt).element_size()
        n_pts, n_vox, nb = cap.n_points, cap.n_voxels, self.nb
        assert feat_voxel.shape[0] == n_vox and xyz_original.shape[0] == n_pts and feat_voxel.dtype == dt
        self._level_overflow = []
        self._splitk_ws = torch.empty(SPLITK_WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
        counts = torch.zeros(CNT.WORDS, dtype=torch.int32, device=dev)
        cptr = counts.data_ptr()
        cnt = lambda k: cptr + 4 * k
        xyz = xyz_original.float().contiguous()
        st = N.current_stream

        # ---- backbone + heads (PBNet.py:117-136) ----
        coords1 = xyz_voxel.to(torch.int32).contiguous()
        lin1 = _Lineage(coords1, n_vox, None, dev)
        if _STOP == "prepare1":
            return {"counts": counts, "_keep": {"lin1": lin1, "coords1": coords1}}
        feats1 = feat_voxel.contiguous()
        f = self._unet(m.MEUnet, lin1, cap.lv1, feats1, feats1.shape[1] * es, cap.expect["lv1"])
        _dbg("backbone done", counts)
        if _STOP == "backbone":
            return {"counts": counts, "_keep": {k: v for k, v in locals().items() if torch.is_tensor(v) or isinstance(v, _Lineage)}}
        v2p = v2p_index.long()
        v2p_z = lin1.inv_perm[v2p]                                   # Z-order row of every point's voxel
        point_feat_p = f[v2p_z]
        sem_score = stage_ops.mlp_rows(m.linear_sem, f, v2p_z)
        offset_p = stage_ops.mlp_rows(m.linear_offset, f, v2p_z)
        batch_head = coords1[:, 0][v2p]
        if teacher is not None:
            sem_score = teacher["sem_score"].to(dev, sem_score.dtype)
            offset_p = teacher["offset"].to(dev, offset_p.dtype)
        sem_pred, sem_prob, table, block_hist = stage_ops.sem_argmax_table(sem_score, batch_head.contiguous(), nb)
        out = {"sem_pred_p": sem_pred, "sem_pred_score_p": sem_score, "offset_pred_p": offset_p, "counts": counts}

        _dbg("heads done", counts)
        if _STOP == "heads":
            return {"counts": counts, "_keep": {k: v for k, v in locals().items() if torch.is_tensor(v) or isinstance(v, _Lineage)}}
        # ---- class gate -> selection -> grouping (PBNet.py:151-179), sizes on the device ----
        n_cls = int(m.sem_num)
        n_seg = (n_cls - 2) * nb
        class_base = torch.empty(n_cls, dtype=torch.int32, device=dev)
        seg_len = torch.empty(n_seg, dtype=torch.int32, device=dev)
        N.check(lib.pbn_class_gate(N.ptr(table), N.ptr(self.thr05), n_cls, nb, int(cap.points), n_pts, N.ptr(class_base),
                                   N.ptr(seg_len), vp(cptr), st()), "pbn_class_gate")
        ins_ind, ins_orig, ins_off, ins_sem = stage_ops.select_points(sem_pred, class_base, block_hist, xyz, offset_p,
                                                                      int(cap.points))
        res = pbnet_ops.cluster_device(ins_off, ins_orig, ins_sem, seg_len, m.radius, m.min_pts, capacity=True)

        _dbg("grouping done", counts)
        if _STOP == "grouping":
            return {"counts": counts, "_keep": {k: v for k, v in locals().items() if torch.is_tensor(v) or isinstance(v, _Lineage)}}
        # ---- local scenes (PBNet.py:182-234): plan on the device, rows by one launch ----
        c_cap, e_cap, r_cap = int(cap.clusters), int(cap.entries), int(cap.rows)
        i32 = dict(dtype=torch.int32, device=dev)
        ent_row_start = torch.empty(e_cap + 1, **i32)
        ent_member_start = torch.empty(e_cap, **i32)
        ent_scene = torch.empty(e_cap, **i32)
        ent_weight = torch.empty(e_cap, dtype=torch.float32, device=dev)
        wsb = int(lib.pbn_local_plan_workspace_bytes(c_cap))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        N.check(lib.pbn_local_plan(N.ptr(res.cluster_num), n_seg, nb, N.ptr(res.member_start), N.ptr(res.centers),
                                   N.ptr(res.n_clusters), N.ptr(self.thr02), N.ptr(self.kmax), c_cap, e_cap, r_cap,
                                   N.ptr(ent_row_start), N.ptr(ent_member_start), N.ptr(ent_scene), N.ptr(ent_weight), vp(cptr),
                                   N.ptr(ws), wsb, st()), "pbn_local_plan")
        c_in = int(point_feat_p.shape[1])
        ld2 = c_in + 2
        point_idx = torch.empty(r_cap, dtype=torch.int64, device=dev)
        row_scene = torch.empty(r_cap, dtype=torch.int64, device=dev)
        coords2 = torch.empty(r_cap, 4, **i32)
        feat2 = torch.empty(r_cap, ld2, dtype=dt, device=dev)
        sem_prob2 = sem_prob.view(-1, 1)
        N.check(lib.pbn_local_scene_rows_dev(
            N.ptr(ent_row_start), N.ptr(ent_member_start), N.ptr(ent_scene), N.ptr(ent_weight), e_cap, r_cap,
            vp(cnt(CNT.ENTRIES)), vp(cnt(CNT.ROWS)), N.ptr(res.member_idx), N.ptr(ins_ind), N.ptr(xyz),
            stage_ops.reciprocal_f32(LOCAL_VOXEL), vp(point_feat_p.data_ptr()), point_feat_p.stride(0), c_in,
            vp(sem_prob2.data_ptr()), sem_prob2.stride(0), None, _DT[dt], N.ptr(point_idx), N.ptr(row_scene), N.ptr(coords2),
            vp(feat2.data_ptr()), ld2, st()), "pbn_local_scene_rows_dev")

        _dbg("local scene rows done", counts)
        if _STOP == "local scene rows":
            return {"counts": counts, "_keep": {k: v for k, v in locals().items() if torch.is_tensor(v) or isinstance(v, _Lineage)}}
        # ---- mask branch (PBNet.py:236-252) ----
        lin2 = _Lineage(coords2, r_cap, cnt(CNT.ROWS), dev)
        f2 = self._unet(m.D_Unet, lin2, cap.lv2, feat2, ld2 * es, cap.expect["lv2"])
        mask_score = self._mlp(m.linear_binary, f2, lin2.inverse, lin2.inv_perm, r_cap, cnt(CNT.ROWS))      # [r_cap, 1]

        _dbg("mask branch done", counts)
        if _STOP == "mask branch":
            return {"counts": counts, "_keep": {k: v for k, v in locals().items() if torch.is_tensor(v) or isinstance(v, _Lineage)}}
        # ---- proposals (PBNet.py:317-347) ----
        per_scene = torch.empty(c_cap, **i32)
        block_cnt = torch.empty(max(int(lib.pbn_select_blocks(r_cap)), 1), **i32)
        N.check(lib.pbn_mask_count_dev(vp(mask_score.data_ptr()), 1, float(MASK_THD), N.ptr(row_scene), r_cap, vp(cnt(CNT.ROWS)),
                                       c_cap, _DT[dt], N.ptr(per_scene), N.ptr(block_cnt), st()), "pbn_mask_count_dev")
        proposals_offset = torch.zeros(c_cap + 1, dtype=torch.int64, device=dev)
        alive_ids = torch.zeros(c_cap, dtype=torch.int64, device=dev)
        dense_of = torch.empty(c_cap, **i32)
        N.check(lib.pbn_proposal_offsets(N.ptr(per_scene), c_cap, N.ptr(proposals_offset), N.ptr(alive_ids), N.ptr(dense_of),
                                         vp(cptr), st()), "pbn_proposal_offsets")
        prop_idx = torch.empty(r_cap, 2, dtype=torch.int64, device=dev)
        prop_ms = torch.empty(r_cap, dtype=dt, device=dev)
        coords3 = torch.empty(r_cap, 4, **i32)
        feat3 = torch.empty(r_cap, c_in, dtype=dt, device=dev)
        N.check(lib.pbn_proposal_rows_dev(
            vp(mask_score.data_ptr()), 1, float(MASK_THD), N.ptr(row_scene), N.ptr(point_idx), r_cap, vp(cnt(CNT.ROWS)),
            N.ptr(dense_of), N.ptr(block_cnt), N.ptr(xyz), float(np.float32(m.scale_size)), stage_ops.reciprocal_f32(m.voxel_size),
            vp(point_feat_p.data_ptr()), point_feat_p.stride(0), c_in, _DT[dt], N.ptr(prop_idx), vp(prop_ms.data_ptr()),
            N.ptr(coords3), vp(feat3.data_ptr()), st()), "pbn_proposal_rows_dev")

        _dbg("proposals done", counts)
        if _STOP == "proposals":
            return {"counts": counts, "_keep": {k: v for k, v in locals().items() if torch.is_tensor(v) or isinstance(v, _Lineage)}}
        # ---- score branch (PBNet.py:255-279) ----
        lin3 = _Lineage(coords3, r_cap, cnt(CNT.PROPOSAL_ROWS), dev)
        f3 = self._unet(m.score_Unet, lin3, cap.lv3, feat3, c_in * es, cap.expect["lv3"])
        n3 = int(cap.lv3[0])
        iou_feat = self._mlp(m.linear_IOU_feat, f3, lin3.inv_perm, None, n3, lin3.counts.data_ptr())     # external row order
        seg_start = torch.empty(c_cap + 1, **i32)
        N.check(lib.pbn_batch_starts(N.ptr(lin3.ucoords), vp(lin3.counts.data_ptr()), n3, c_cap, N.ptr(seg_start), st()),
                "pbn_batch_starts")
        ch = int(iou_feat.shape[1])
        mx = torch.empty(c_cap, ch, dtype=torch.float32, device=dev)
        av = torch.empty(c_cap, ch, dtype=torch.float32, device=dev)
        pws = int(lib.pbn_segment_pool_workspace_bytes(c_cap, ch))
        pw = torch.empty(max(pws, 16), dtype=torch.uint8, device=dev)
        N.check(lib.pbn_segment_pool(vp(iou_feat.data_ptr()), iou_feat.stride(0), ch, _DT[dt], N.ptr(seg_start), c_cap,
                                     N.ptr(mx), N.ptr(av), vp(pw.data_ptr()), pws, st()), "pbn_segment_pool")
        pooled = (mx + av).to(dt)
        clt_scores = stage_ops.mlp_rows(m.linear_IOU, pooled).view(-1)
        _dbg("score branch done", counts)
        if _STOP == "score branch":
            return {"counts": counts, "_keep": {k: v for k, v in locals().items() if torch.is_tensor(v) or isinstance(v, _Lineage)}}
        # a level of one of the three pyramids that outgrew its capacity (rows were dropped): flag it
        ovf = torch.stack(self._level_overflow).any().to(torch.int32) * 64
        counts[CNT.OVERFLOW:CNT.OVERFLOW + 1] |= ovf
        out.update(proposals_idx=prop_idx, proposals_offset=proposals_offset, alive_ids=alive_ids, proposals_ms=prop_ms,
                   clt_scores=clt_scores, _keep=(lin1, lin2, lin3, res, feat2, feat3, point_feat_p, self._splitk_ws))
        return out

    # ---- the one read-back ------------------------------------------------------------------------------------------------
    def finish(self, out):
        """Slice the capacity-sized outputs to their true sizes (one device->host copy of 16 ints).  Returns the dict of
        PBNet.forward, or raises CapacityOverflow."""
        return self._slice(out, clone=False)

    @staticmethod
    def _slice(out, clone):
        c = out["counts"].cpu().tolist()
        if c[CNT.OVERFLOW]:
            raise CapacityOverflow(c[CNT.OVERFLOW])
        n_prop, n_rows = c[CNT.PROPOSALS], c[CNT.PROPOSAL_ROWS]
        f = (lambda t: t.clone()) if clone else (lambda t: t)
        return {"sem_pred_p": f(out["sem_pred_p"]), "sem_pred_score_p": f(out["sem_pred_score_p"]),
                "offset_pred_p": f(out["offset_pred_p"]),
                "proposals": (f(out["proposals_idx"][:n_rows]), f(out["proposals_offset"][:n_prop + 1]),
                              f(out["alive_ids"][:n_prop]), f(out["proposals_ms"][:n_rows])),
                "clt_scores": f(out["clt_scores"][:n_prop]), "counts": c}

    def __call__(self, feat_voxel, xyz_voxel, xyz_original, v2p_index, teacher=None):
        with torch.no_grad():
            return self.finish(self.run(feat_voxel, xyz_voxel, xyz_original, v2p_index, teacher))

    # ---- HIP graph ---------------------------------------------------------------------------------------------------------
    def capture(self, feat_voxel, xyz_voxel, xyz_original, v2p_index, teacher=None):
        """Capture the forward into a HIP graph over STATIC input buffers (copies of the arguments); `replay` copies new
        inputs of the same shapes in and launches the graph.  Warm-up runs first (weight packing, allocator pools)."""
        self.static_in = [feat_voxel.clone(), xyz_voxel.clone(), xyz_original.clone(), v2p_index.clone()]
        self.static_teacher = None if teacher is None else {k: v.clone() for k, v in teacher.items()}
        with torch.no_grad():
            s = torch.cuda.Stream(self.dev)
            s.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(s):
                for _ in range(2):
                    self.run(*self.static_in, teacher=self.static_teacher)
            torch.cuda.current_stream(self.dev).wait_stream(s)
            torch.cuda.synchronize(self.dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_out = self.run(*self.static_in, teacher=self.static_teacher)
        return self

    def replay(self, feat_voxel=None, xyz_voxel=None, xyz_original=None, v2p_index=None, teacher=None):
        """Copy new inputs (same shapes) into the static buffers and launch the graph on the caller's current stream.
        Replays may be interleaved with anything (eager kernels on the outputs, read-backs, synchronisations): every fill
        the launch sequence needs is a kernel node (csrc/common.hip fill_ranges) -- hipMemsetAsync nodes turned out not to
        re-execute after an explicit stream / device synchronisation between two replays on this ROCm runtime, which used to
        leave hash tables full (endless probe loops) and count tables accumulating."""
        new = (feat_voxel, xyz_voxel, xyz_original, v2p_index)
        for dst, src in zip(self.static_in, new):
            if src is not None:
                dst.copy_(src)
        if teacher is not None:
            for k, v in teacher.items():
                self.static_teacher[k].copy_(v)
        self.graph.replay()          # on the caller's current stream (a private replay stream per graph doubles the
                                     # number of HIP streams of a serving loop: 4 graphs in flight fell from 281 to 179 scenes/s)
        return self.static_out


class CapacityOverflow(RuntimeError):
    def __init__(self, flags):
        self.flags = int(flags)
        what = [name for bit, name in OVF_NAMES.items() if self.flags & bit]
        super().__init__("planned forward: capacity exceeded for " + ", ".join(what) + " -- rerun on the size-exact path")
