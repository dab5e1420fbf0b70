"""PBNet model on the MI355X path: same constructor, forward signature, returned dict keys, model_fn /
model_fn_eval as /root/reference/network/PBNet.py (:18-111 construction, :113-280 forward, :317-347 get_proposal,
:349-460 model functions), same module / parameter names (state dicts interchange).

What differs is WHERE things run, not what is computed:
  * the 18 per-class `pbnet_ops.cluster` round trips through host memory (PBNet.py:151-179) become ONE device launch
    sequence over all (class, batch) segments (`pbnet_ops.cluster_device`); ids are re-based per class on the host so
    every number matches the per-class calls of the reference;
  * local-scene construction (PBNet.py:182-234) is split into a tiny host plan over CLUSTERS (kNN of centres with
    torch.cdist/topk on the CPU exactly as the reference does, size gates, weights) and device gathers over POINTS
    driven by the ordered member lists the grouping kernel already produced;
  * get_proposal (PBNet.py:317-347) is a device compaction instead of Python loops.
One host synchronisation per stage boundary (class counts, cluster table, proposal count), none per class/cluster.
"""
import os
import threading

import numpy as np
import torch
import torch.nn as nn

from .. import MinkowskiEngine as ME
from .. import pbnet_ops
from .. import stage_ops
from ..prof import section, mark
from .Mink import Mink_unet as unet3d

COUNT_MEAN = [-1., -1., 3917., 12056., 2303., 8331., 3948., 3166., 5629., 11719., 1003., 3317., 4912., 10221., 3889.,
              4136., 2120., 945., 3967., 2589.]                       # PBNet.py:33-34 (softgroup & HAIS)
LOCAL_VOXEL = 0.02                                                    # PBNet.py:236 (hard-coded)
HEAD_CLUSTERS = 256                                                   # clusters whose centres travel with the first grouping read-back
MASK_THD = 0.45                                                       # PBNet.py:317
# training: index-only glue between the networks on the inference path's fused launches (see PBNet.forward); "0" = plain torch
TRAIN_FUSED_GLUE = os.environ.get("PBNET_TRAIN_FUSED_GLUE", "1") == "1"
# Round 5: the size-exact inference forward takes its class gate, the selection, the grouping AND the local-scene plan on the
# device (csrc/plan.hip: pbn_class_gate / pbn_local_plan, the entries the planned forward uses) over buffers bounded by the
# number of points, and reads the four sizes back ONCE where it used to read the class table, then the cluster table, and run
# the per-(class, batch) cdist / topk plan on the host (PBNet.py:151-234).  Same integers (the device plan is the one
# tests/test_planned_gpu.py pins to the host plan); more than FRONT_CLUSTER_CAP clusters -> the host path.  "0": the host path.
DEVICE_FRONT = os.environ.get("PBNET_DEVICE_FRONT", "1") == "1"
FRONT_CLUSTER_CAP = 1024
# ... and only up to this many points: the front's buffers, fills and grids are sized by the number of POINTS where the host path sizes
# them by the selected points.  (Measured after k_count stopped folding the lanes past the count onto its last point: +2.5-3.5 % in
# flight at 162 k points, level at 485 k and at 1.1 M points.)
FRONT_MAX_POINTS = int(os.environ.get("PBNET_DEVICE_FRONT_MAX_POINTS", "2000000"))      # (round 6: 2 M -- eight merged scenes, pbnet_amd/serving.py)


def _mlp(cin, mid, cout, sigmoid=False):
    layers = [ME.MinkowskiLinear(cin, mid, bias=False), ME.MinkowskiBatchNorm(mid), ME.MinkowskiPReLU(),
              ME.MinkowskiLinear(mid, cout, bias=True)]
    if sigmoid:
        layers.append(ME.MinkowskiSigmoid())
    return nn.Sequential(*layers)


class PBNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.batch_size = cfg.batch_size
        self.cluster_batch = cfg.batch_size * 1
        self.sem_num = cfg.sem_num
        self.voxel_size = cfg.voxel_size
        self.scale_size = cfg.scale_size
        self.cluster_epoch = cfg.cluster_epoch
        self.radius = cfg.radius
        self.min_pts = cfg.min_pts
        self.method = getattr(cfg, "method", "PBNet")
        self.count_mean = torch.tensor(COUNT_MEAN)
        self.K_max = torch.ones(20, dtype=torch.float32) * 6
        # three sparse U-Nets (PBNet.py:38-40)
        self.MEUnet = unet3d(in_channels=6, out_channels=32, arch="MinkUNet34C")
        self.D_Unet = unet3d(in_channels=34, out_channels=32, arch="MinkUNet14A")
        self.score_Unet = unet3d(in_channels=32, out_channels=32, arch="MinkUNet34C")
        # heads (PBNet.py:43-82)
        self.linear_sem = _mlp(32, 16, self.sem_num)
        self.linear_offset = _mlp(32, 16, 3)
        self.linear_binary = _mlp(32, 16, 1, sigmoid=True)
        self.global_max_pool = ME.MinkowskiGlobalMaxPooling()
        self.global_avg_pool = ME.MinkowskiGlobalAvgPooling()
        self.linear_IOU_feat = _mlp(32, 32, 32)
        self.linear_IOU = _mlp(32, 16, 1, sigmoid=True)
        self.soft_max = ME.MinkowskiSoftmax()
        self.weight_initialization()
        self.fix_module = []

    # per host thread: the sizes of its last forward (several scenes in flight share one model)
    @property
    def _tls(self):
        t = self.__dict__.get("_tls_obj")
        if t is None:
            t = self.__dict__["_tls_obj"] = threading.local()
        return t

    def __getstate__(self):                     # (a threading.local cannot be copied / pickled: deepcopy, torch.save of the module)
        d = self.__dict__.copy()
        d.pop("_tls_obj", None)
        return d

    @property
    def _last_sizes(self):
        return self._tls.__dict__.setdefault("last_sizes", {})

    @_last_sizes.setter
    def _last_sizes(self, v):
        self._tls.last_sizes = v

    def weight_initialization(self):
        for m in self.modules():
            if isinstance(m, ME.MinkowskiConvolution):
                ME.utils.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")
            if isinstance(m, ME.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    # =========================================================================================================
    def forward(self, feat_voxel, xyz_voxel, xyz_original, v2p_v1, ins_label, epoch, task="train", teacher=None, n_batch=None):
        """teacher: optional dict(sem_score [N,sem_num], offset [N,3]) that REPLACES the two head outputs after they
        have been computed -- a bench/test hook: randomly initialised heads cannot produce instances.
        n_batch: batch elements of a "test" forward (the batch indices are 0 .. n_batch - 1).  The reference hard-codes 3, its
        three test-time copies of one scene (PBNet.py:167-170; dataset_preprocess.py:324); a serving front that merges the
        scenes waiting on a GPU into one forward through the same batch axis (pbnet_amd/serving.py) passes their number."""
        dev = torch.device("cuda", torch.cuda.current_device())
        if feat_voxel.shape[0] == 0 or xyz_original.shape[0] == 0:
            raise ValueError("PBNet.forward: empty scene (0 voxels / 0 points)")
        fused = not torch.is_grad_enabled()           # inference: the stage glue runs as fused launches (stage_ops)
        nb = self.batch_size if task == "train" else (3 if n_batch is None else int(n_batch))            # PBNet.py:167-170
        stage1 = self.backbone_stage(feat_voxel.to(dev), xyz_voxel.to(dev), v2p_v1.to(dev), fused)
        if teacher is not None:
            stage1["sem_pred_score_p"] = teacher["sem_score"].to(dev, stage1["sem_pred_score_p"].dtype)
            stage1["offset_pred_p"] = teacher["offset"].to(dev, stage1["offset_pred_p"].dtype)
            if not fused:
                stage1["sem_pred_score_sfp"] = torch.softmax(stage1["sem_pred_score_p"].float(), 1).to(stage1["point_feat_p"].dtype)
                stage1["sem_pred_p"] = stage1["sem_pred_score_p"].max(1)[1]
        if fused:
            stage1["sem_pred_p"], stage1["sem_prob_p"], stage1["table"], stage1["block_hist"] = \
                stage_ops.sem_argmax_table(stage1["sem_pred_score_p"], stage1["batch_head_p"], nb)
        elif TRAIN_FUSED_GLUE and stage1["sem_pred_score_p"].is_cuda and epoch > self.cluster_epoch:
            # training: everything between the networks that carries NO gradient (argmax, population table, class-major
            # selection, the rows of the local scenes, proposal rows) runs as the inference path's fused launches, under
            # no_grad; what gradients flow through (features gathered at those rows, the softmax scores) stays torch.
            # The same integers either way (tests/test_train_gpu.py); ~70 small launches fewer per step, in the part of the
            # forward where the GPU waits for the host.
            with torch.no_grad():
                sc = stage1["sem_pred_score_p"].detach()
                stage1["sem_pred_p"], _, stage1["table"], stage1["block_hist"] = stage_ops.sem_argmax_table(
                    sc if sc.stride(1) == 1 else sc.contiguous(), stage1["batch_head_p"].to(torch.int32).contiguous(), nb)
        ret = {"sem_pred_p": stage1["sem_pred_p"], "sem_pred_score_p": stage1["sem_pred_score_p"],
               "offset_pred_p": stage1["offset_pred_p"]}
        if epoch > self.cluster_epoch:
            ret.update(self.cluster_stage(stage1, xyz_original.to(dev), None if ins_label is None else ins_label.to(dev),
                                          task, nb))
        return ret

    # ---- PBNet.py:117-136 -------------------------------------------------------------------------------------
    def backbone_stage(self, feat_voxel, xyz_voxel, v2p_v1, fused=False):
        with section("a3_coords"):
            inputs_v1 = ME.SparseTensor(feat_voxel, xyz_voxel)
        with section("a4_unet"):
            point_feat = self.MEUnet(inputs_v1)
        if fused:   # data-dependent sizes of this forward (pbnet_amd.planned.measure_capacities reads them)
            self._last_sizes = {"lv1": list(inputs_v1.coordinate_manager.row_counts())}
        _sec = section("a5_heads_gather"); _sec.__enter__()
        v2p = v2p_v1.long()
        if fused:
            # heads evaluated straight at the points (one launch each): the same rows, so the same numbers, as
            # evaluating them at the voxels and gathering (PBNet.py:124-134)
            f, row_index = point_feat.rows()              # the U-Net's own row order: fold it into the point index
            if row_index is not None:
                v2p = row_index[v2p]
            out = {
                "point_feat_p": f[v2p],
                "sem_pred_score_p": stage_ops.mlp_rows(self.linear_sem, f, v2p),
                "offset_pred_p": stage_ops.mlp_rows(self.linear_offset, f, v2p),
                "batch_head_p": xyz_voxel[:, 0].to(torch.int32)[v2p_v1.long()],
            }
            _sec.__exit__(None, None, None)
            return out
        sem_pred_score = self.linear_sem(point_feat)
        sem_pred_score_sf = self.soft_max(sem_pred_score)
        offsets_pred = self.linear_offset(point_feat)
        out = {
            "point_feat_p": point_feat.F[v2p],
            "sem_pred_score_p": sem_pred_score.F[v2p],
            "sem_pred_score_sfp": sem_pred_score_sf.F[v2p],
            "offset_pred_p": offsets_pred.F[v2p],
            "batch_head_p": xyz_voxel[:, 0][v2p],
        }
        out["sem_pred_p"] = out["sem_pred_score_p"].max(1)[1]
        _sec.__exit__(None, None, None)
        return out

    # ---- PBNet.py:144-279 -------------------------------------------------------------------------------------
    def cluster_stage(self, s1, xyz_original, ins_label, task, n_batch=None):
        dev = xyz_original.device
        xyz_original = xyz_original.float()
        fused = "table" in s1
        fused_glue = not torch.is_grad_enabled()      # inference: stage glue as fused launches (stage_ops)
        sem_pred_p = s1["sem_pred_p"]
        point_feat_p, offset_pred_p = s1["point_feat_p"], s1["offset_pred_p"]
        train_glue = fused and torch.is_grad_enabled()       # fused index work, differentiable features (see forward())
        sem_sfp = s1["sem_prob_p"].view(-1, 1) if (fused and not train_glue) else s1["sem_pred_score_sfp"]
        nb = n_batch if n_batch is not None else (self.batch_size if task == "train" else 3)          # PBNet.py:167-170
        self.cluster_batch = nb
        n_cls = int(self.sem_num)

        # inference: gate, selection, grouping and local-scene plan on the device, ONE read-back (see DEVICE_FRONT)
        front = None
        if (DEVICE_FRONT and fused and not train_glue and task == "test" and ins_label is None and xyz_original.is_contiguous()
                and xyz_original.shape[0] <= FRONT_MAX_POINTS):
            front = self._device_front(s1, xyz_original, nb, n_cls)
            if isinstance(front, str):
                return self._empty_stage(dev, task)
        if front is not None:
            ins_ind, res, packed, n_ent, n_rows, n_clt, m = front
            n_scenes = n_clt                                    # test mode: every cluster heads one local scene (PBNet.py:182-234)
            with section("a17_gather"), torch.no_grad():
                point_idx, row_scene, coords, feat = stage_ops.local_scene_rows(
                    packed, n_ent, n_rows, res.member_idx, ins_ind, xyz_original, LOCAL_VOXEL, point_feat_p.detach(),
                    sem_sfp.detach(), None)
            self._last_sizes.update(points=int(m), clusters=int(n_clt), entries=int(n_ent), rows=int(n_rows))
            _sec = section("a17_plan"); _sec.__enter__()         # (the host plan of the other branch: nothing to do here)
        else:
            # (a6) per-class selection, all classes at once; host learns the [class, batch] population table
            _sec = section("a6_select"); _sec.__enter__()
            if fused:
                table = s1["table"]
            else:
                batch_head_p = s1["batch_head_p"].long()
                table = torch.bincount(sem_pred_p * nb + batch_head_p, minlength=n_cls * nb)[:n_cls * nb].view(n_cls, nb)
            mark("a6:before table sync")
            tab = table.cpu().numpy()                                                 # sync 1
            mark("a6:table on host")
            assert int(tab.sum()) == sem_pred_p.shape[0], "batch index outside [0, cluster_batch)"  # PBNet.py:286
            per_class = tab.sum(1).tolist()
            thr05, thr02 = self._class_thresholds()
            classes = [c for c in range(2, n_cls) if not (float(per_class[c]) < thr05[c])]          # PBNet.py:157
            if not classes:
                return self._empty_stage(dev, task)
            m = sum(per_class[c] for c in classes)
            seg_len_h = tab[classes].reshape(-1).astype(np.int32)                    # segments = (class, batch) in order
            if fused:
                # stable class-major selection + the grouping inputs in one launch (positions from the class totals);
                # class_base and the segment lengths travel in ONE host->device copy
                class_base = np.full(n_cls, -1, dtype=np.int32)
                run = 0
                for c in classes:
                    class_base[c] = run
                    run += per_class[c]
                up = torch.from_numpy(np.concatenate([class_base, seg_len_h])).to(dev)
                seg_len = up[n_cls:]
                with torch.no_grad():
                    ins_ind, ins_orig, ins_offseted, ins_sem = stage_ops.select_points(
                        sem_pred_p, up[:n_cls], s1["block_hist"], xyz_original, offset_pred_p.detach(), m)
            else:
                keep = torch.zeros(n_cls, dtype=torch.bool)
                keep[classes] = True
                key = torch.where(keep.to(dev)[sem_pred_p], sem_pred_p, torch.full_like(sem_pred_p, n_cls))
                order = torch.sort(key, stable=True)[1]
                ins_ind = order[:m]                                   # class-major, ascending point index inside a class
                ins_orig = xyz_original[ins_ind]
                ins_offseted = ins_orig + offset_pred_p[ins_ind].float()                 # PBNet.py:165 (fp32 add)
                ins_sem = sem_pred_p[ins_ind].to(torch.int32)
                seg_len = torch.from_numpy(seg_len_h).to(dev)
            _sec.__exit__(None, None, None)

            if fused and not train_glue:
                self._last_sizes["points"] = int(m)
            mark("a7:select queued")
            with section("a7_16_grouping"):
                res = pbnet_ops.cluster_device(ins_offseted, ins_orig, ins_sem, seg_len, self.radius, self.min_pts)
                mark("a7:grouping queued")
                # one read-back for the cluster table AND the first HEAD_CLUSTERS centres / member offsets (a scene has tens
                # of clusters; a second read-back follows only when there are more)
                n_seg = int(res.cluster_num.shape[0])
                hc = min(HEAD_CLUSTERS, int(res.member_start.shape[0]) - 1)
                head = torch.cat([res.n_clusters, res.cluster_num, res.member_start[:hc + 1],
                                  res.centers[:3 * hc].view(torch.int32)]).cpu().numpy()   # sync 2
                mark("a7:grouping done")
                n_clt = int(head[0])
            if n_clt < 0:
                raise RuntimeError("grouping rejected its input (class id outside [2,19])")
            if n_clt == 0:
                return self._empty_stage(dev, task)
            _sec = section("a17_plan"); _sec.__enter__()
            cluster_num = head[1:1 + n_seg].reshape(len(classes), nb).tolist()
            if n_clt <= hc:
                member_start = head[1 + n_seg:1 + n_seg + n_clt + 1]
                centers = torch.from_numpy(head[2 + n_seg + hc:2 + n_seg + hc + 3 * n_clt].view(np.float32).copy()).view(n_clt, 3)
            else:
                packed = torch.cat([res.centers[:3 * n_clt], res.member_start[:n_clt + 1].view(torch.float32)]).cpu()  # sync 3
                centers = packed[:3 * n_clt].view(n_clt, 3)
                member_start = packed[3 * n_clt:].view(torch.int32).numpy()
            mark("a17:centres on host")
            sizes = (member_start[1:] - member_start[:-1])
            sizes_l = sizes.tolist()
            labels_h = None
            if task != "test":
                labels_h = ins_label[ins_ind[res.member_idx[:int(member_start[-1])].long()]].cpu()

            # (a17) host plan over clusters: which clusters make up each local scene, and with which weight
            ent_cluster, ent_weight, scene_len, scene_gt = [], [], [], []
            k_max = self._k_max_list()
            g = 0                                                   # running global cluster id (class-major, batch, seed)
            for ci, cls in enumerate(classes):
                for b in range(nb):
                    c_b = cluster_num[ci][b]
                    if c_b == 0:
                        continue
                    para_k = min(c_b - 1, k_max[cls])
                    if para_k > 0:
                        peak_v = [0.5 * ((para_k + 1) - p_i) / (para_k + 1) for p_i in range(para_k + 1)]
                        ctr = centers[g:g + c_b]
                        knn_idx = torch.cdist(ctr, ctr).topk(k=c_b, dim=1, largest=False)[1].tolist()
                    big = thr02[cls]
                    for c_i in range(c_b):
                        gid = g + c_i
                        gt = None
                        if task != "test":
                            gt = int(torch.mode(labels_h[int(member_start[gid]):int(member_start[gid + 1])])[0])
                            if gt == -100:
                                continue
                        ents, wts = [gid], [1.0]
                        if float(sizes_l[gid]) > big and para_k > 0:                  # PBNet.py:199
                            row = knn_idx[c_i]
                            for k_i in range(para_k):
                                ents.append(g + row[k_i + 1])
                                wts.append(peak_v[k_i])
                        ent_cluster += ents
                        ent_weight += wts
                        scene_len.append(len(ents))
                        scene_gt.append(gt)
                    g += c_b
            if not scene_len:
                return self._empty_stage(dev, task)

            _sec.__exit__(None, None, None)
            mark("a17:plan done")
            # device gathers over points: rows of every local scene, in the reference's order
            _sec = section("a17_gather"); _sec.__enter__()
            ent_np = np.asarray(ent_cluster, dtype=np.int64)
            if not torch.is_grad_enabled() or train_glue:
                # inference: ONE launch (pbn_local_scene_rows) driven by one packed host->device copy of the entry table
                n_ent = len(ent_cluster)
                row_start = np.zeros(n_ent + 1, dtype=np.int32)
                np.cumsum(sizes[ent_np], out=row_start[1:])
                n_rows = int(row_start[-1])
                ent_scene = np.repeat(np.arange(len(scene_len), dtype=np.int32), scene_len)
                packed = torch.from_numpy(np.concatenate([row_start, member_start[:-1][ent_np].astype(np.int32), ent_scene,
                                                          np.asarray(ent_weight, dtype=np.float32).view(np.int32)])).to(dev)
                with torch.no_grad():
                    point_idx, row_scene, coords, feat = stage_ops.local_scene_rows(
                        packed, n_ent, n_rows, res.member_idx, ins_ind, xyz_original, LOCAL_VOXEL, point_feat_p.detach(),
                        sem_sfp.detach(), None if (fused and not train_glue) else sem_pred_p)
                if train_glue:
                    # the same rows with their gradients: features and own-class scores gathered by torch, the entry weight (a
                    # constant) taken from the fused launch's last column                                  PBNet.py:162-163,194,230
                    row_sem_sf = sem_sfp[point_idx, sem_pred_p[point_idx]]
                    feat = torch.cat([point_feat_p[point_idx], row_sem_sf.view(-1, 1).to(point_feat_p.dtype), feat[:, -1:]], 1)
                elif fused:
                    self._last_sizes.update(clusters=int(n_clt), entries=int(n_ent), rows=int(n_rows))
            else:
                ent_cluster_t = torch.from_numpy(ent_np)
                ent_rows = torch.from_numpy(sizes.astype(np.int64))[ent_cluster_t]
                member_start = torch.from_numpy(member_start)
                ent_scene = torch.repeat_interleave(torch.arange(len(scene_len)), torch.tensor(scene_len))
                d = lambda t: t.to(dev)
                row_ent = torch.repeat_interleave(torch.arange(len(ent_cluster), device=dev), d(ent_rows))
                ent_first = d(torch.cumsum(ent_rows, 0) - ent_rows)
                pos_in_ent = torch.arange(row_ent.shape[0], device=dev) - ent_first[row_ent]
                member_pos = d(member_start[:-1][ent_cluster_t])[row_ent] + pos_in_ent
                local_idx = res.member_idx[member_pos].long()                            # index into the grouped array
                point_idx = ins_ind[local_idx]                                           # index into the scene's points
                row_scene = d(ent_scene)[row_ent]
                row_weight = d(torch.tensor(ent_weight, dtype=torch.float32))[row_ent]
                row_sem_sf = sem_sfp[point_idx, sem_pred_p[point_idx]]                    # PBNet.py:162-163: own-class score
                feat = torch.cat([point_feat_p[point_idx], row_sem_sf.view(-1, 1).to(point_feat_p.dtype),
                                  row_weight.view(-1, 1).to(point_feat_p.dtype)], 1)     # [R, 34]  PBNet.py:194,230
                coords = torch.cat([row_scene.view(-1, 1).to(torch.int32),
                                    torch.floor(xyz_original[point_idx] / LOCAL_VOXEL).to(torch.int32)], 1)
            n_scenes = len(scene_len)
        out = {}
        _sec.__exit__(None, None, None)

        mark("a17:rows queued")
        # (a18) mask branch
        with section("a18_mask_coords"):
            inputs_v2 = ME.SparseTensor(feat, coords)
        with section("a18_mask_unet"):
            if fused_glue:   # head evaluated at the rows (PBNet.py:247): same numbers as head-then-gather, one launch
                f2, row_index = self.D_Unet(inputs_v2).rows()
                if fused:
                    self._last_sizes["lv2"] = list(inputs_v2.coordinate_manager.row_counts())
                mask_score = stage_ops.mlp_rows(self.linear_binary, f2, inputs_v2.inverse_mapping, row_index)
            else:
                mask_score = self.linear_binary(self.D_Unet(inputs_v2)).F[inputs_v2.inverse_mapping]   # [R, 1]
        if task != "test":
            gt_rows = torch.tensor(scene_gt, dtype=torch.long).to(dev)[row_scene]
            lab = ins_label[point_idx]
            gt_mask = (lab == gt_rows).long()
            gt_mask[lab == -100] = -1
            out["mask_scores"] = (mask_score, gt_mask.detach())
        mark("a18:mask unet queued")
        coords3 = feat3 = None
        with section("a19_proposals"):
            if fused_glue:
                out["proposals"], coords3, feat3 = self._proposals_fused(row_scene, point_idx, mask_score, n_scenes,
                                                                         xyz_original, point_feat_p)
            elif train_glue:
                with torch.no_grad():
                    out["proposals"], coords3, _ = self._proposals_fused(row_scene, point_idx, mask_score.detach(), n_scenes,
                                                                         xyz_original, point_feat_p.detach())
                feat3 = point_feat_p[out["proposals"][0][:, 1]]
            else:
                out["proposals"] = self.get_proposal(row_scene, point_idx, mask_score, n_scenes=n_scenes)

        mark("a19:proposals queued")
        # (a20) score branch
        proposals_idx, proposals_offset, _, _ = out["proposals"]
        if proposals_offset.shape[0] > 1:
            if coords3 is None:
                pidx = proposals_idx[:, 1]
                c3 = torch.floor(xyz_original[pidx] * self.scale_size / self.voxel_size).to(torch.int32)
                coords3 = torch.cat([proposals_idx[:, 0:1].to(torch.int32), c3], 1)
                feat3 = point_feat_p[pidx]
            with section("a20_score_coords"):
                inputs_v3 = ME.SparseTensor(feat3, coords3)
            with section("a20_score_unet"):
                if fused_glue:
                    iou_feat_f = stage_ops.mlp_rows(self.linear_IOU_feat, *self.score_Unet(inputs_v3).rows())
                    if fused:
                        self._last_sizes["lv3"] = list(inputs_v3.coordinate_manager.row_counts())
                else:
                    iou_feat = self.linear_IOU_feat(self.score_Unet(inputs_v3))
            with section("a20_pool_head"):
                # global max + avg pooling per proposal (PBNet.py:274-276); rows are grouped by proposal id
                if torch.is_grad_enabled():      # training: the same segment-pool kernel with the reductions' backward rules
                    from ..MinkowskiEngine.nn import global_max_plus_avg_pool
                    global_feat = global_max_plus_avg_pool(iou_feat) if iou_feat.F.is_cuda else \
                        self.global_max_pool(iou_feat) + self.global_avg_pool(iou_feat)
                    out["clt_scores"] = self.linear_IOU(global_feat).F.view(-1)
                else:                            # inference: one deterministic segment-pool kernel
                    from ..MinkowskiEngine.nn import segment_pool, _PooledTensor
                    n_prop = int(proposals_offset.shape[0]) - 1
                    f = iou_feat_f if fused_glue else iou_feat.F
                    mx, av = segment_pool(f, inputs_v3.C[:, 0], n_prop)
                    pooled = (mx + av).to(f.dtype)
                    if fused_glue:
                        out["clt_scores"] = stage_ops.mlp_rows(self.linear_IOU, pooled).view(-1)
                    else:
                        out["clt_scores"] = self.linear_IOU(_PooledTensor(pooled)).F.view(-1)
        else:
            out["clt_scores"] = torch.zeros(0, dtype=torch.float32, device=dev)
        mark("a20:score branch queued")
        return out

    def _device_front(self, s1, xyz_original, nb, n_cls):
        """Class gate -> selection -> grouping -> local-scene plan without a host decision in between (inference).
        Returns None (fall back to the host path), "empty", or (ins_ind, res, packed, n_ent, n_rows, n_clt, m)."""
        from .. import planned as P
        from .. import _native as N
        import ctypes
        dev = xyz_original.device
        lib = N.lib()
        vp = ctypes.c_void_p
        consts = self.__dict__.setdefault("_front_consts", {})
        c = consts.get(dev)
        if c is None:
            thr05, thr02 = self._class_thresholds()
            c = consts[dev] = (torch.tensor(thr05, dtype=torch.float32, device=dev), torch.tensor(thr02, dtype=torch.float32, device=dev),
                               torch.tensor(self._k_max_list(), dtype=torch.int32, device=dev))
        thr05_d, thr02_d, kmax_d = c
        n_pts = int(xyz_original.shape[0])
        n_seg = (n_cls - 2) * nb
        i32 = dict(dtype=torch.int32, device=dev)
        counts = torch.zeros(P.CNT.WORDS, **i32)
        class_base = torch.empty(n_cls, **i32)
        seg_len = torch.empty(n_seg, **i32)
        with section("a6_select"):
            N.check(lib.pbn_class_gate(N.ptr(s1["table"]), N.ptr(thr05_d), n_cls, nb, n_pts, n_pts, N.ptr(class_base), N.ptr(seg_len),
                                       vp(counts.data_ptr()), N.current_stream()), "pbn_class_gate")
            ins_ind, ins_orig, ins_off, ins_sem = stage_ops.select_points(s1["sem_pred_p"], class_base, s1["block_hist"], xyz_original,
                                                                          s1["offset_pred_p"].detach(), n_pts)
        mark("a7:select queued")
        with section("a7_16_grouping"):
            res = pbnet_ops.cluster_device(ins_off, ins_orig, ins_sem, seg_len, self.radius, self.min_pts, capacity=True)
            c_cap = min(FRONT_CLUSTER_CAP, n_pts)
            e_cap = 7 * c_cap
            ent = torch.empty(4 * e_cap + 1, **i32)                    # row_start (e_cap + 1) | member_start | scene | weight bits
            wsb = int(lib.pbn_local_plan_workspace_bytes(c_cap))
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            base = ent.data_ptr()
            N.check(lib.pbn_local_plan(N.ptr(res.cluster_num), n_seg, nb, N.ptr(res.member_start), N.ptr(res.centers),
                                       N.ptr(res.n_clusters), N.ptr(thr02_d), N.ptr(kmax_d), c_cap, e_cap, 2 ** 31 - 1,
                                       vp(base), vp(base + 4 * (e_cap + 1)), vp(base + 4 * (2 * e_cap + 1)),
                                       vp(base + 4 * (3 * e_cap + 1)), vp(counts.data_ptr()), N.ptr(ws), wsb, N.current_stream()),
                    "pbn_local_plan")
            mark("a7:grouping queued")
            h = torch.cat([counts, res.n_clusters]).cpu().tolist()                    # the ONE read-back of the front
            mark("a7:grouping done")
        if h[P.CNT.WORDS] < 0:
            raise RuntimeError("grouping rejected its input (class id outside [2,19])")
        if h[P.CNT.OVERFLOW] & 32:
            raise AssertionError("batch index outside [0, cluster_batch)")            # PBNet.py:286
        if h[P.CNT.OVERFLOW]:
            return None                                                               # more clusters than the plan holds: host path
        m, n_clt, n_ent, n_rows = h[P.CNT.POINTS], h[P.CNT.CLUSTERS], h[P.CNT.ENTRIES], h[P.CNT.ROWS]
        if m == 0 or n_clt == 0 or n_ent == 0:
            return "empty"
        packed = torch.cat([ent[:n_ent + 1], ent[e_cap + 1:e_cap + 1 + n_ent], ent[2 * e_cap + 1:2 * e_cap + 1 + n_ent],
                            ent[3 * e_cap + 1:3 * e_cap + 1 + n_ent]])
        return ins_ind, res, packed, n_ent, n_rows, n_clt, m

    def _class_thresholds(self):
        """Per-class population gates as Python floats of the fp32 products (PBNet.py:157 `count_mean * 0.05`,
        :199 `count_mean * 0.2`): computed once, with the same fp32 multiplication the per-element form performs."""
        hit = getattr(self, "_thr_cache", None)
        if hit is None:
            hit = ((self.count_mean * 0.05).tolist(), (self.count_mean * 0.2).tolist())
            self._thr_cache = hit
        return hit

    def _k_max_list(self):
        hit = getattr(self, "_kmax_cache", None)
        if hit is None:
            hit = [int(v) for v in self.K_max.tolist()]
            self._kmax_cache = hit
        return hit

    def _empty_stage(self, dev, task):
        z = torch.zeros(0, dtype=torch.int64, device=dev)
        out = {"proposals": (torch.zeros(0, 2, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev),
                             z, torch.zeros(0, device=dev)),
               "clt_scores": torch.zeros(0, device=dev)}
        if task != "test":
            out["mask_scores"] = (torch.zeros(0, 1, device=dev), z)
        return out

    def _proposals_fused(self, row_scene, point_idx, mask_score, n_scenes, xyz_original, point_feat_p,
                         mask_score_thd=MASK_THD):
        """get_proposal (PBNet.py:317-347) plus the score-branch inputs (:240-252) as two launches around ONE host read
        (kept rows per local scene).  Same outputs as get_proposal; also returns (coords3, feat3)."""
        dev = row_scene.device
        per_scene_d, block_cnt = stage_ops.mask_count(mask_score, mask_score_thd, row_scene, n_scenes)
        mark("a19:before count sync")
        per_scene = per_scene_d.cpu().numpy().astype(np.int64)                       # sync
        mark("a19:counts on host")
        total = int(per_scene.sum())
        alive = per_scene > 0
        n_alive = int(alive.sum())
        proposals_offset = np.zeros(n_alive + 1, dtype=np.int64)
        np.cumsum(per_scene[alive], out=proposals_offset[1:])
        dense_of = (np.cumsum(alive) - 1).astype(np.int32)                          # PBNet.py:342-345
        # proposals_offset | surviving scene ids in one copy, the renumbering (int32) in another
        up64 = torch.from_numpy(np.concatenate([proposals_offset, np.nonzero(alive)[0].astype(np.int64)])).to(dev)
        dense_of = torch.from_numpy(dense_of).to(dev)
        prop_idx, prop_ms, coords3, feat3 = stage_ops.proposal_rows(
            mask_score, mask_score_thd, row_scene, point_idx, dense_of, block_cnt, total, xyz_original,
            self.scale_size, self.voxel_size, point_feat_p)
        return (prop_idx, up64[:n_alive + 1], up64[n_alive + 1:], prop_ms), coords3, feat3

    # ---- PBNet.py:317-347 as one device compaction ----------------------------------------------------------------
    def get_proposal(self, row_scene, point_idx, mask_score, mask_score_thd=MASK_THD, n_scenes=None):
        """row_scene i64[R] (local-scene id per row, ascending), point_idx i64[R], mask_score [R,1].
        One host read (rows kept per local scene); everything else stays on the device."""
        assert row_scene.shape[0] == mask_score.shape[0]
        dev = row_scene.device
        if n_scenes is None:
            n_scenes = int(row_scene.max().item()) + 1 if row_scene.numel() else 0
        keep = mask_score.view(-1).float() > mask_score_thd
        per_scene = torch.bincount(row_scene[keep], minlength=n_scenes).cpu()       # sync
        total = int(per_scene.sum())
        valid = torch.nonzero_static(keep, size=total).view(-1)
        proposals_ms = mask_score[valid].view(-1)
        alive = per_scene > 0
        cluster_id_v = torch.nonzero(alive).view(-1)                                # surviving scene ids (host)
        proposals_offset = torch.zeros(int(alive.sum()) + 1, dtype=torch.int64)
        proposals_offset[1:] = torch.cumsum(per_scene[alive], 0)
        # "remove null proposals" (PBNet.py:342-345) == dense renumbering of the surviving local scenes
        dense_of = torch.cumsum(alive.to(torch.int64), 0) - 1
        dense = dense_of.to(dev)[row_scene[valid]]
        proposals_idx = torch.stack([dense, point_idx[valid]], 1).to(torch.int64)
        return proposals_idx.detach(), proposals_offset.to(dev), cluster_id_v.to(dev), proposals_ms


def model_losses(ret, sem_label, ins_label, instance_info, instance_pointnum, xyz_original, epoch, cfg, get_iou=None):
    """The loss arithmetic of model_fn (PBNet.py:366-416) on tensors that are already on one device.  Returns
    (loss, parts, valid, mask_label_weight, gt_mask_out): `parts` holds every term as a tensor.

    Reference quirk kept on purpose (PBNet.py:398-405): `gt_mask[gt_mask == -1.] = 0.5` runs IN PLACE on a LONG
    tensor, so the ignore rows become 0 (0.5 truncates); the later `gt_mask != -1` is therefore all-true: the dice
    term covers EVERY row with the ignore rows as target 0, the BCE term masks them out through its weight, and the
    mutated gt_mask is what pred['mask_scores'] returns."""
    sem_pred_score_p, offset_pred_p = ret["sem_pred_score_p"].float(), ret["offset_pred_p"].float()
    semantic_loss = nn.CrossEntropyLoss(ignore_index=-100)(sem_pred_score_p, sem_label)
    gt_offsets = instance_info[:, 0:3] - xyz_original
    pt_dist = torch.sum(torch.abs(offset_pred_p - gt_offsets), dim=-1)
    valid = (ins_label != -100).float()
    offset_norm_loss = torch.sum(pt_dist * valid) / (torch.sum(valid) + 1e-6)
    gt_dir = gt_offsets / (torch.norm(gt_offsets, p=2, dim=1).unsqueeze(-1) + 1e-8)
    pt_dir = offset_pred_p / (torch.norm(offset_pred_p, p=2, dim=1).unsqueeze(-1) + 1e-8)
    offset_dir_loss = torch.sum(-(gt_dir * pt_dir).sum(-1) * valid) / (torch.sum(valid) + 1e-6)
    loss = semantic_loss + offset_norm_loss + offset_dir_loss
    parts = {"semantic_loss": semantic_loss, "offset_norm_loss": offset_norm_loss, "offset_dir_loss": offset_dir_loss}
    weight = gt_mask = None
    if epoch > cfg.cluster_epoch:
        pred_mask, gt_mask = ret["mask_scores"]
        pred_mask = pred_mask.float()
        weight = (gt_mask != -1).float()
        gt_mask[gt_mask == -1] = 0                                 # the reference's `= 0.5` on a long tensor
        mask_loss = nn.BCELoss(reduction="none", weight=weight)(pred_mask.view(-1), gt_mask.float()).mean()
        dice_loss = diceLoss(pred_mask.view(-1), gt_mask.view(-1))   # every row (see the docstring)
        proposals_idx, proposals_offset, _, _ = ret["proposals"]
        if get_iou is None:
            from .. import pbnet_ops as ops
            get_iou = ops.get_iou
        ious = get_iou(proposals_idx[:, 1].contiguous(), proposals_offset, ins_label, instance_pointnum)
        gt_ious, _ = ious.max(1)
        gt_scores = get_segmented_scores(gt_ious, cfg.fg_thresh, cfg.bg_thresh)
        score_loss = nn.BCELoss()(ret["clt_scores"].float().view(-1), gt_scores).mean()
        loss = loss + mask_loss + dice_loss + score_loss
        parts.update(mask_loss=mask_loss, dice_loss=dice_loss, score_loss=score_loss)
    parts["loss"] = loss
    return loss, parts, valid, weight, gt_mask


def model_fn(batch, model, epoch, cfg, task="train"):
    """PBNet.py:349-444: forward + losses.  Losses are plain torch on the device (outside the kernel scope); their
    arithmetic is restated independently in oracle/loss_ref.py and compared in tests/test_losses.py."""
    xyz_original = batch["xyz_original"].cuda()
    ins_label = batch["ins"].cuda()
    ret = model(batch["feat_voxel"], batch["xyz_voxel"], xyz_original, batch["v2p_index"], ins_label, epoch, task)
    sem_label = batch["sem"].cuda()
    instance_info = batch["inst_info"].cuda()
    instance_pointnum = batch["instance_pointnum"].cuda()
    offset_pred_p, sem_pred_p = ret["offset_pred_p"].float(), ret["sem_pred_p"]
    loss, parts, valid, weight, gt_mask = model_losses(ret, sem_label, ins_label, instance_info, instance_pointnum,
                                                       xyz_original.float(), epoch, cfg)
    with torch.no_grad():
        pred = {"sem": sem_pred_p, "offseted_xyz": xyz_original + offset_pred_p}
        # the logged terms in ONE read-back (the reference calls .item() per term: five synchronisations)
        names = ["loss", "semantic_loss", "offset_norm_loss", "offset_dir_loss"] + (["mask_loss"] if epoch > cfg.cluster_epoch else [])
        host = torch.stack([parts[k].detach().float().reshape(()) for k in names]).tolist()
        visual_dict = dict(zip(names[:4], host[:4]))
        meter_dict = {k: (v, valid.sum()) for k, v in visual_dict.items()}
        if epoch > cfg.cluster_epoch:
            visual_dict["mask_loss"] = host[4]
            meter_dict["mask_loss"] = (visual_dict["mask_loss"], weight.sum())
            pred["mask_scores"] = ret["mask_scores"]               # carries the mutated gt_mask, as upstream
            pred["proposals"] = ret["proposals"]
            pred["clt_scores"] = ret["clt_scores"]
    return loss, pred, visual_dict, meter_dict


def model_fn_eval(batch, model, epoch, cfg, task="test", teacher=None, n_batch=None):
    """PBNet.py:446-460."""
    ret = model(batch["feat_voxel"], batch["xyz_voxel"], batch["xyz_original"], batch["v2p_index"], None, epoch, task,
                teacher=teacher, n_batch=n_batch)
    pred = {"sem": ret["sem_pred_p"]}
    if epoch > cfg.cluster_epoch:
        pred["proposals"] = ret["proposals"]
        pred["clt_scores"] = ret["clt_scores"]
    return pred


def get_segmented_scores(scores, fg_thresh=1.0, bg_thresh=0.0):
    """/root/reference/tools/mIOU.py:34-49 (linear ramp between the two thresholds)."""
    fg_mask = scores > fg_thresh
    bg_mask = scores < bg_thresh
    interval_mask = (fg_mask == 0) & (bg_mask == 0)
    out = (fg_mask > 0).float()
    k = 1 / (fg_thresh - bg_thresh)
    b = bg_thresh / (bg_thresh - fg_thresh)
    out[interval_mask] = scores[interval_mask] * k + b
    return out


def diceLoss(mask_pred, mask_gt, ep=1e-8):
    """PBNet.py:463-468."""
    inter = 2 * (mask_gt * mask_pred).sum() + 1
    union = (mask_gt ** 2.0).sum() + (mask_pred ** 2.0).sum() + 1 + ep
    return 1 - inter / union
