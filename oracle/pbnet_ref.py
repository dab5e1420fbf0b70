"""oracle/pbnet_ref.py -- TEST INFRASTRUCTURE ONLY: CPU restatement of PBNet.forward
(/root/reference/network/PBNet.py:113-280 and get_proposal :317-347), loop for loop, on top of the other oracles
(sparse_ref for everything MinkowskiEngine does, pb_cluster_ref for PB_lib.binary_cluster).  Plain torch on CPU.

Takes a state dict with the reference's parameter names.  PARITY UNPINNED (inherits both oracles' status: neither
MinkowskiEngine nor PB_lib can run here and the reference holds no golden outputs for this path).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import pb_cluster_ref, sparse_ref as R

COUNT_MEAN = torch.tensor([-1., -1., 3917., 12056., 2303., 8331., 3948., 3166., 5629., 11719., 1003., 3317., 4912.,
                           10221., 3889., 4136., 2120., 945., 3967., 2589.])   # PBNet.py:33-34
K_MAX = 6                                                                     # PBNet.py:35


def _sub(sd, prefix):
    return {k[len(prefix) + 1:]: v.detach().cpu().float() for k, v in sd.items() if k.startswith(prefix + ".")}


def mlp(sd, prefix, x, training=False, sigmoid=False):
    """Sequential(MinkowskiLinear(no bias), MinkowskiBatchNorm, MinkowskiPReLU, MinkowskiLinear(bias)[, Sigmoid])
    (PBNet.py:43-82) on a feature matrix."""
    p = _sub(sd, prefix)
    h = x @ p["0.linear.weight"].t()
    h = F.batch_norm(h, p["1.bn.running_mean"], p["1.bn.running_var"], p["1.bn.weight"], p["1.bn.bias"],
                     training=training, momentum=0.0, eps=1e-5)
    h = F.prelu(h, p["2.module.weight"])
    h = h @ p["3.linear.weight"].t() + p["3.linear.bias"]
    return torch.sigmoid(h) if sigmoid else h


def backbone_stage(sd, feat_voxel, xyz_voxel, v2p_v1, training=False):
    """PBNet.py:117-136."""
    feat_voxel = feat_voxel.detach().cpu().float()
    coords = np.asarray(xyz_voxel.detach().cpu().numpy(), dtype=np.int32)
    point_feat = R.minkunet_forward(_sub(sd, "MEUnet"), "MinkUNet34C", feat_voxel, coords, training=training)
    sem_pred_score = mlp(sd, "linear_sem", point_feat, training)
    sem_pred_score_sf = torch.softmax(sem_pred_score, 1)
    offsets_pred = mlp(sd, "linear_offset", point_feat, training)
    v2p = v2p_v1.detach().cpu().long()
    out = dict(point_feat_p=point_feat[v2p], sem_pred_score_p=sem_pred_score[v2p],
               sem_pred_score_sfp=sem_pred_score_sf[v2p], offset_pred_p=offsets_pred[v2p],
               batch_head_p=torch.from_numpy(coords[:, 0].copy())[v2p])
    out["sem_pred_p"] = out["sem_pred_score_p"].max(1)[1]
    return out


def get_proposal(list_idx_proposal, mask_score, mask_score_thd=0.45):
    """PBNet.py:317-347."""
    proposals_idx = []
    for idx_i in range(len(list_idx_proposal)):
        cur = torch.ones([list_idx_proposal[idx_i].shape[0], 2])
        cur[:, 0] = cur[:, 0] * idx_i
        cur[:, 1] = list_idx_proposal[idx_i]
        proposals_idx.append(cur)
    proposals_idx = torch.cat(proposals_idx, dim=0).type(torch.int64)
    valid_index = torch.nonzero(mask_score.view(-1) > mask_score_thd).view(-1)
    proposals_idx = proposals_idx[valid_index]
    proposals_ms = mask_score[valid_index].view(-1)
    cluster_id_v, cluster_len = torch.unique(proposals_idx[:, 0], return_counts=True)
    cluster_id_v = torch.sort(cluster_id_v)[0]
    proposals_offset = torch.zeros(cluster_len.shape[0] + 1)
    for i in range(proposals_offset.shape[0]):
        if i == 0:
            continue
        proposals_offset[i] = torch.sum(cluster_len[:i])
    if proposals_offset.shape[0] == 1:
        return proposals_idx, proposals_offset.type(torch.int64), cluster_id_v, proposals_ms
    if cluster_id_v.shape[0] != torch.max(proposals_idx[:, 0]) + 1:
        for i in range(cluster_id_v.shape[0]):
            cor_idx = proposals_idx[:, 0] == cluster_id_v[i]
            proposals_idx[cor_idx, 0] = i
    return proposals_idx, proposals_offset.type(torch.int64), cluster_id_v, proposals_ms


def cluster_stage(sd, cfg, s1, xyz_original, ins_label, task, training=False):
    """PBNet.py:144-279."""
    xyz_original = xyz_original.detach().cpu().float()
    sem_pred_p, batch_head_p = s1["sem_pred_p"], s1["batch_head_p"]
    point_feat_p, sem_sfp, offset_pred_p = s1["point_feat_p"], s1["sem_pred_score_sfp"], s1["offset_pred_p"]
    cluster_batch = cfg.batch_size if task == "train" else 3
    list_xyz, list_feat, list_gt_mask, list_ins_idx = [], [], [], []
    n_clusters_per_class = {}
    for sem_id in range(2, int(cfg.sem_num)):
        ins_ind = torch.sort(torch.nonzero(sem_pred_p == sem_id).view(-1))[0]
        if ins_ind.shape[0] < COUNT_MEAN[sem_id] * 0.05:
            continue
        ins_orig = xyz_original[ins_ind]
        ins_offset = offset_pred_p[ins_ind]
        ins_feat = point_feat_p[ins_ind]
        ins_sem = sem_pred_p[ins_ind]
        ins_sem_score = sem_sfp[:, sem_id][ins_ind]
        if task != "test":
            ins_ins_label = ins_label[ins_ind]
        ins_offseted = ins_orig + ins_offset
        ins_bh = batch_head_p[ins_ind]
        ins_bp = torch.tensor([int((ins_bh == i).sum()) for i in range(cluster_batch)], dtype=torch.int32)
        assert int(ins_bp.sum()) == ins_bh.shape[0]
        ins_bp_sum = [0]
        for i in range(cluster_batch):
            ins_bp_sum.append(ins_bp_sum[-1] + int(ins_bp[i]))
        cluster_id, cluster_num, _den, clt_ctr = pb_cluster_ref.cluster(ins_offseted.numpy(), ins_orig.numpy(),
                                                                        ins_sem.numpy(), ins_bp.numpy(), cfg.radius,
                                                                        cfg.min_pts, cluster_batch)
        cluster_id = torch.from_numpy(cluster_id)
        n_clusters_per_class[sem_id] = cluster_num.copy()
        clt_ctr = torch.from_numpy(clt_ctr).view(-1, 3)
        ctr_offset = [0]
        for i in range(cluster_batch):
            ctr_offset.append(ctr_offset[-1] + int(cluster_num[i]))
        for cur_bi in range(cluster_batch):
            if cluster_num[cur_bi] == 0:
                continue
            lo, hi = ins_bp_sum[cur_bi], ins_bp_sum[cur_bi + 1]
            batch_xyz_orig = ins_orig[lo:hi]
            batch_feat = ins_feat[lo:hi]
            batch_sem_sf = ins_sem_score[lo:hi].view(-1, 1)
            batch_ins_idx = ins_ind[lo:hi]
            batch_clt_id = cluster_id[lo:hi]
            if task != "test":
                batch_ins_label = ins_ins_label[lo:hi]
            batch_ins_feat = torch.cat((batch_feat, batch_sem_sf), dim=1)
            para_k = min(int(cluster_num[cur_bi]) - 1, K_MAX)
            if para_k > 0:
                peak_v = [0.5 * ((para_k + 1) - p_i) / (para_k + 1) for p_i in range(para_k + 1)]
                clt_center = clt_ctr[ctr_offset[cur_bi]:ctr_offset[cur_bi + 1]]
                dist = torch.cdist(clt_center, clt_center)
                knn_idx = dist.topk(k=int(cluster_num[cur_bi]), dim=1, largest=False)[1]
            for c_i in range(int(cluster_num[cur_bi])):
                valid_idx = torch.nonzero(batch_clt_id == c_i + ctr_offset[cur_bi]).view(-1)
                if task != "test":
                    cur_gt_ins_label = torch.mode(batch_ins_label[valid_idx])[0]
                    if cur_gt_ins_label == -100:
                        continue
                cur_dpn = torch.ones(valid_idx.shape[0])
                if valid_idx.shape[0] > COUNT_MEAN[sem_id] * 0.2 and para_k > 0:
                    sub_valid_list, sub_dpn_list = [valid_idx], [cur_dpn]
                    for k_i in range(para_k):
                        valid_idx = torch.nonzero(batch_clt_id == knn_idx[c_i, k_i + 1] + ctr_offset[cur_bi]).view(-1)
                        sub_valid_list.append(valid_idx)
                        sub_dpn_list.append(torch.ones(valid_idx.shape[0]) * peak_v[k_i])
                    valid_idx = torch.cat(sub_valid_list, dim=0)
                    cur_dpn = torch.cat(sub_dpn_list, dim=0)
                if task != "test":
                    valid_ins_label = batch_ins_label[valid_idx]
                    cur_gt_mask = (valid_ins_label == cur_gt_ins_label).long()
                    cur_gt_mask[torch.nonzero(valid_ins_label == -100).view(-1)] = -1
                    list_gt_mask.append(cur_gt_mask)
                assert cur_dpn.min() > 0.0
                list_xyz.append(batch_xyz_orig[valid_idx])
                list_feat.append(torch.cat((batch_ins_feat[valid_idx], cur_dpn.view(-1, 1)), dim=1))
                list_ins_idx.append(batch_ins_idx[valid_idx])
    out = {"n_clusters_per_class": n_clusters_per_class, "n_local_scenes": len(list_xyz)}
    if not list_xyz:
        return out
    # ---- mask branch (PBNet.py:236-252)
    coords = R.batched_coordinates([x / 0.02 for x in list_xyz])
    sem_feat_tensor = torch.cat(list_feat, dim=0)
    f2, c2, v2p_v2 = R.sparse_tensor(sem_feat_tensor, coords)
    each_sem_feat = R.minkunet_forward(_sub(sd, "D_Unet"), "MinkUNet14A", f2, c2, training=training)
    mask_score = mlp(sd, "linear_binary", each_sem_feat, training, sigmoid=True)[v2p_v2]
    if task != "test":
        out["mask_scores"] = (mask_score, torch.cat(list_gt_mask, dim=0))
    out["proposals"] = get_proposal(list_ins_idx, mask_score)
    out["local_scene_rows"] = sem_feat_tensor.shape[0]
    # row-level view for the tests: (local scene, point index, mask score before the threshold) of every row
    out["row_scene"] = torch.cat([torch.full((len(x),), i, dtype=torch.int64) for i, x in enumerate(list_ins_idx)])
    out["row_point"] = torch.cat(list_ins_idx).long()
    out["row_mask_score"] = mask_score.view(-1)
    # ---- score branch (PBNet.py:255-279)
    proposals_idx, proposals_offset, _, _ = out["proposals"]
    if proposals_offset.shape[0] == 1:
        out["clt_scores"] = torch.zeros(0)
        return out
    clt_length = (proposals_offset[1:] - proposals_offset[:-1]).tolist()
    ins_orig_sort = xyz_original[proposals_idx[:, 1]] * cfg.scale_size / cfg.voxel_size
    ins_feat_sort = point_feat_p[proposals_idx[:, 1]]
    coords3 = R.batched_coordinates(torch.split(ins_orig_sort, clt_length, dim=0))
    f3, c3, _ = R.sparse_tensor(ins_feat_sort, coords3)
    iou_feat = R.minkunet_forward(_sub(sd, "score_Unet"), "MinkUNet34C", f3, c3, training=training)
    iou_feat = mlp(sd, "linear_IOU_feat", iou_feat, training)
    nb = len(clt_length)
    global_feat = R.global_pool(iou_feat, c3[:, 0], nb, "max") + R.global_pool(iou_feat, c3[:, 0], nb, "avg")
    out["clt_scores"] = mlp(sd, "linear_IOU", global_feat, training, sigmoid=True).view(-1)
    return out
