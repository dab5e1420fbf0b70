"""ScanNet instance-segmentation AP -- the evaluator behind eval_map.py:126-151, i.e.
/root/reference/tools/eval.py (`assign_instances_for_scan` :205-250, `evaluate_matches` :27-190, `compute_averages`
:193-210) and the ground-truth id encoding of datasets/scannetv2/get_val_gt.py:26-39.

What changes against the reference is the data model, not the numbers:

* association: the reference walks the scene once per (prediction, ground-truth instance) pair.  Here the scene's ids are
  turned into indices into their sorted unique list once, and ONE device pass per prediction fills a row of the
  [predictions, unique ids] overlap table (`pbn_instance_overlap`, csrc/post.hip).  Vertex counts, void overlap and every
  `intersection` field are sums / entries of that integer table.
* a scene's matches are a `SceneMatches` record of flat arrays instead of dicts of dicts of copies;
  `to_reference()` / `from_reference()` convert to and from the reference's layout so either evaluator accepts either.
* matching is the reference's greedy rule run on arrays; the precision/recall integration is vectorised.

Everything here except the overlap table is host bookkeeping on a few hundred integers, as in the reference."""
import numpy as np
import torch

from . import _native as N

# tools/eval.py:8-24
CLASS_LABELS = ['cabinet', 'bed', 'chair', 'sofa', 'table', 'door', 'window', 'bookshelf', 'picture', 'counter', 'desk',
                'curtain', 'refrigerator', 'shower curtain', 'toilet', 'sink', 'bathtub', 'otherfurniture']
VALID_CLASS_IDS = np.array([3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])
OVERLAPS = np.append(np.arange(0.5, 0.95, 0.05), 0.25)
MIN_REGION_SIZE = 100
_CLASS_OF_ID = {int(c): i for i, c in enumerate(VALID_CLASS_IDS)}
# datasets/scannetv2/get_val_gt.py:8 (20 training classes -> NYU40 ids; 0/1 = wall/floor)
SEMANTIC_LABEL_IDX = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])


# ---------------------------------------------------------------------------------------------- ground-truth formats
def encode_gt_ids(sem_label, ins_label):
    """get_val_gt.py:26-37: per point `NYU40 id * 1000 + instance + 1`, 0 where the point has no instance.  The class of
    an instance is the class of its lowest-index point (`instance_mask[0]`); semantic label -100 counts as class 0."""
    sem = np.asarray(sem_label).astype(np.int64)
    ins = np.asarray(ins_label).astype(np.int64)
    out = np.zeros(ins.shape, np.int32)
    has = ins >= 0
    if not has.any():
        return out
    n_inst = int(ins.max()) + 1
    first = np.full(n_inst, ins.shape[0], np.int64)
    np.minimum.at(first, ins[has], np.nonzero(has)[0])
    present = first < ins.shape[0]                       # the reference raises on an id without points; skipped here
    sem_of = np.zeros(n_inst, np.int64)
    sem_of[present] = sem[first[present]]
    sem_of[sem_of == -100] = 0
    code = SEMANTIC_LABEL_IDX[sem_of] * 1000 + np.arange(n_inst) + 1
    out[has] = code[ins[has]]
    return out


def save_gt_ids(path, ids):
    """get_val_gt.py:39: one decimal id per line."""
    np.savetxt(path, np.asarray(ids), fmt="%d")


def load_gt_ids(path):
    """tools/getins.py:7-10."""
    with open(path) as f:
        return np.array(f.read().split(), dtype=np.int64)


# ------------------------------------------------------------------------------------------------------ association
class SceneMatches:
    """One scene's association tables.  Ground-truth rows are in ascending id order, prediction rows in input order
    (the orders the reference's lists have); `inter[q, g]` is zero where the classes differ."""

    __slots__ = ("scene", "gt_class", "gt_id", "gt_vert", "pred_class", "pred_id", "pred_label_id", "pred_vert", "pred_void",
                 "pred_conf", "inter")

    def __init__(self, scene, gt_class, gt_id, gt_vert, pred_class, pred_id, pred_label_id, pred_vert, pred_void, pred_conf,
                 inter):
        self.scene = scene
        self.gt_class, self.gt_id, self.gt_vert = gt_class, gt_id, gt_vert
        self.pred_class, self.pred_id, self.pred_label_id = pred_class, pred_id, pred_label_id
        self.pred_vert, self.pred_void, self.pred_conf = pred_vert, pred_void, pred_conf
        self.inter = inter

    def to_reference(self):
        """(gt2pred, pred2gt) exactly as tools/eval.py:205-250 returns them."""
        gt2pred = {name: [] for name in CLASS_LABELS}
        pred2gt = {name: [] for name in CLASS_LABELS}

        def gt_dict(g):
            return dict(instance_id=int(self.gt_id[g]), label_id=int(self.gt_id[g] // 1000), vert_count=int(self.gt_vert[g]),
                        med_dist=-1, dist_conf=0.0)

        def pred_dict(q):
            return dict(filename="%s_%03d" % (self.scene, int(self.pred_id[q])), pred_id=int(self.pred_id[q]),
                        label_id=int(self.pred_label_id[q]), vert_count=int(self.pred_vert[q]),
                        confidence=self.pred_conf[q], void_intersection=int(self.pred_void[q]))

        for g in range(self.gt_id.shape[0]):
            d = gt_dict(g)
            d["matched_pred"] = [dict(pred_dict(q), intersection=int(self.inter[q, g]))
                                 for q in np.nonzero(self.inter[:, g])[0]]
            gt2pred[CLASS_LABELS[self.gt_class[g]]].append(d)
        for q in range(self.pred_id.shape[0]):
            d = pred_dict(q)
            d["matched_gt"] = [dict(gt_dict(g), intersection=int(self.inter[q, g]), matched_pred=[])
                               for g in np.nonzero(self.inter[q])[0]]
            pred2gt[CLASS_LABELS[self.pred_class[q]]].append(d)
        return gt2pred, pred2gt

    @classmethod
    def from_reference(cls, scene, gt2pred, pred2gt):
        """Flatten the reference's dicts (e.g. matches produced by tools/eval.py itself)."""
        gts = [(li, g) for li, name in enumerate(CLASS_LABELS) for g in gt2pred[name]]
        gts.sort(key=lambda t: t[1]["instance_id"])
        preds = [(li, p) for li, name in enumerate(CLASS_LABELS) for p in pred2gt[name]]
        preds.sort(key=lambda t: t[1]["pred_id"])
        col = {g["instance_id"]: j for j, (_, g) in enumerate(gts)}
        inter = np.zeros((len(preds), len(gts)), np.int64)
        for q, (_, p) in enumerate(preds):
            for g in p["matched_gt"]:
                inter[q, col[g["instance_id"]]] = g["intersection"]
        i64 = lambda v: np.array(v, np.int64)                                                        # noqa: E731
        return cls(scene, i64([li for li, _ in gts]), i64([g["instance_id"] for _, g in gts]),
                   i64([g["vert_count"] for _, g in gts]), i64([li for li, _ in preds]), i64([p["pred_id"] for _, p in preds]),
                   i64([p["label_id"] for _, p in preds]), i64([p["vert_count"] for _, p in preds]),
                   i64([p["void_intersection"] for _, p in preds]),
                   np.array([p["confidence"] for _, p in preds], np.float32), inter)


def overlap_table(masks, gt_ids, device=None):
    """[predictions, unique ids] overlap counts on the device.  `masks`: [P, N] tensor or array (non-zero = inside);
    `gt_ids`: int[N].  Returns (inter int64[P, U] on the host, unique ids int64[U])."""
    gt_ids = np.asarray(gt_ids.cpu() if torch.is_tensor(gt_ids) else gt_ids).astype(np.int64).reshape(-1)
    uid, index = np.unique(gt_ids, return_inverse=True)
    if torch.is_tensor(masks):
        dev = masks.device if device is None else torch.device(device)
        m = masks.to(dev)
    else:
        dev = torch.device("cuda" if device is None else device)
        m = torch.from_numpy(np.ascontiguousarray(masks)).to(dev)
    if m.dim() != 2 or m.shape[1] != gt_ids.shape[0]:
        # tools/eval.py:222-224 only logs this; counting over mismatched lengths has no meaning, so it is an error here
        raise ValueError("prediction masks are %s but the scene has %d vertices" % (tuple(m.shape), gt_ids.shape[0]))
    n_pred, n_pts, n_gt = int(m.shape[0]), int(m.shape[1]), max(1, int(uid.shape[0]))
    if n_pred == 0 or n_pts == 0:
        return np.zeros((n_pred, int(uid.shape[0])), np.int64), uid
    m = (m if m.dtype == torch.int32 else (m != 0).to(torch.int32)).contiguous()
    N.require_cuda(m)
    idx = torch.from_numpy(index.astype(np.int32).reshape(-1)).to(dev)
    inter = torch.empty(n_pred, n_gt, dtype=torch.int32, device=dev)
    N.check(N.lib().pbn_instance_overlap(N.ptr(m), n_pred, n_pts, N.ptr(idx), n_gt, N.ptr(inter), N.current_stream()),
            "pbn_instance_overlap")
    return inter.cpu().numpy().astype(np.int64), uid


def assign_instances_for_scan(scene_name, pred_info, gt, device=None):
    """tools/eval.py:205-250.  `pred_info` = {'conf' [P], 'label_id' [P], 'mask' [P, N]} (eval_map.py:128-131; the mask may
    stay a device tensor), `gt` = the scene's id vector or the path of its val_gt file.  Returns a SceneMatches."""
    gt_ids = load_gt_ids(gt) if isinstance(gt, (str, bytes)) or hasattr(gt, "__fspath__") else gt
    inter_all, uid = overlap_table(pred_info["mask"], gt_ids, device)
    label_id = np.asarray(pred_info["label_id"].cpu() if torch.is_tensor(pred_info["label_id"]) else pred_info["label_id"])
    conf = np.asarray(pred_info["conf"].cpu() if torch.is_tensor(pred_info["conf"]) else pred_info["conf"])
    # ground-truth instances: ids of benchmark classes (getins.py:59-70; 0 = unannotated never qualifies)
    uid_class = np.array([_CLASS_OF_ID.get(int(u // 1000), -1) for u in uid], np.int64)
    g_cols = np.nonzero((uid != 0) & (uid_class >= 0))[0]
    counts_all = np.bincount(np.searchsorted(uid, np.asarray(gt_ids).astype(np.int64).reshape(-1)), minlength=uid.shape[0])
    void_cols = uid_class < 0                                                     # :217 (id 0 included: 0 // 1000 = 0)
    # predictions: benchmark label and at least MIN_REGION_SIZE vertices (:219-229), numbered in input order
    vert_all = inter_all.sum(1)
    pred_class_all = np.array([_CLASS_OF_ID.get(int(l), -1) for l in label_id], np.int64)
    keep = np.nonzero((pred_class_all >= 0) & (vert_all >= MIN_REGION_SIZE))[0]
    inter = inter_all[np.ix_(keep, g_cols)]
    inter = inter * (pred_class_all[keep][:, None] == uid_class[g_cols][None, :])  # :238 same-class instances only
    return SceneMatches(scene_name, uid_class[g_cols], uid[g_cols], counts_all[g_cols].astype(np.int64), pred_class_all[keep],
                        np.arange(keep.shape[0], dtype=np.int64), label_id[keep].astype(np.int64), vert_all[keep],
                        inter_all[keep][:, void_cols].sum(1), conf[keep], inter)


# ----------------------------------------------------------------------------------------------------------- AP
def _as_records(matches):
    out = []
    for scene, m in (matches.items() if isinstance(matches, dict) else enumerate(matches)):
        out.append(m if isinstance(m, SceneMatches) else SceneMatches.from_reference(scene, m["gt"], m["pred"]))
    return out


def _match_scene_class(m, g_sel, q_sel, th, visited):
    """Greedy assignment of one class in one scene at one overlap threshold (tools/eval.py:60-125).  Returns
    (y_true, y_score, hard false negatives).  `visited` (per prediction of the scene) is updated in place."""
    inter = m.inter[np.ix_(q_sel, g_sel)]
    union = m.pred_vert[q_sel][:, None] + m.gt_vert[g_sel][None, :] - inter
    over = (inter > 0) & (inter.astype(np.float64) / np.maximum(union, 1) > th)
    ok_gt = (m.gt_id[g_sel] >= 1000) & (m.gt_vert[g_sel] >= MIN_REGION_SIZE)                  # :49-51
    conf = m.pred_conf[q_sel]
    y_true, y_score, hard_fn = [], [], 0
    for g in np.nonzero(ok_gt)[0]:
        best = None
        for q in np.nonzero(inter[:, g] > 0)[0]:              # `matched_pred`, in prediction order
            if visited[q_sel[q]] or not over[q, g]:
                continue
            if best is None:
                best = conf[q]
                visited[q_sel[q]] = True
            else:                                             # second hit on a matched instance: the lower score is a
                y_true.append(0.0)                            # false positive and the prediction stays unvisited (:78-86)
                y_score.append(min(best, conf[q]))
                best = max(best, conf[q])
        if best is None:
            hard_fn += 1
        else:
            y_true.append(1.0)
            y_score.append(best)
    # predictions without any instance above the threshold (:102-121) -- all same-class instances count, also small ones
    lonely = ~over.any(axis=1)
    ignore = m.pred_void[q_sel] + (inter * (~ok_gt)[None, :]).sum(1)
    for q in np.nonzero(lonely)[0]:
        if float(ignore[q]) / m.pred_vert[q_sel[q]] <= th:
            y_true.append(0.0)
            y_score.append(conf[q])
    return y_true, y_score, hard_fn


def _average_precision(y_true, y_score, hard_fn):
    """Area under the precision/recall curve as tools/eval.py:131-176 integrates it."""
    order = np.argsort(y_score, kind="stable")
    score, true = y_score[order], y_true[order]
    _, first = np.unique(score, return_index=True)            # first position of every distinct score
    below = np.concatenate([[0.0], np.cumsum(true)])[first]    # true matches scored strictly lower
    n_true = float(true.sum())
    tp = n_true - below
    fp = (score.shape[0] - first) - tp
    fn = below + hard_fn
    precision = np.append(tp / (tp + fp), 1.0)                # the artificial last point (:161-162)
    recall = np.append(tp / (tp + fn), 0.0)
    r = np.concatenate([recall[:1], recall, [0.0]])
    width = np.convolve(r, [-0.5, 0, 0.5], "valid")
    return np.dot(precision, width)


def evaluate_matches(matches):
    """tools/eval.py:27-190.  `matches`: {scene: SceneMatches} or the reference's {scene: {'gt':…, 'pred':…}}.
    Returns ap float32[1, classes, overlaps] (nan = class without ground truth)."""
    recs = _as_records(matches)
    ap = np.zeros((1, len(CLASS_LABELS), len(OVERLAPS)), np.float32)
    sel = [[(np.nonzero(m.gt_class == li)[0], np.nonzero(m.pred_class == li)[0]) for li in range(len(CLASS_LABELS))]
           for m in recs]
    for oi, th in enumerate(OVERLAPS):
        visited = [np.zeros(m.pred_id.shape[0], bool) for m in recs]
        for li in range(len(CLASS_LABELS)):
            y_true, y_score, hard_fn, has_gt, has_pred = [], [], 0, False, False
            for si, m in enumerate(recs):
                g_sel, q_sel = sel[si][li]
                has_gt |= bool(((m.gt_id[g_sel] >= 1000) & (m.gt_vert[g_sel] >= MIN_REGION_SIZE)).any())
                has_pred |= q_sel.shape[0] > 0
                if g_sel.shape[0] == 0 and q_sel.shape[0] == 0:
                    continue
                t, s, h = _match_scene_class(m, g_sel, q_sel, th, visited[si])
                y_true += t
                y_score += s
                hard_fn += h
            if has_gt and has_pred:
                ap[0, li, oi] = _average_precision(np.array(y_true, np.float64), np.array(y_score, np.float64), hard_fn)
            elif has_gt:
                ap[0, li, oi] = 0.0
            else:
                ap[0, li, oi] = float("nan")
    return ap


def compute_averages(aps):
    """tools/eval.py:193-210 (same keys)."""
    o50 = np.where(np.isclose(OVERLAPS, 0.5))
    o25 = np.where(np.isclose(OVERLAPS, 0.25))
    rest = np.where(np.logical_not(np.isclose(OVERLAPS, 0.25)))
    avg = {"all_ap": np.nanmean(aps[0, :, rest]), "all_ap_50%": np.nanmean(aps[0, :, o50]),
           "all_ap_25%": np.nanmean(aps[0, :, o25]), "classes": {}}
    for li, name in enumerate(CLASS_LABELS):
        avg["classes"][name] = {"ap": np.average(aps[0, li, rest]), "ap50%": np.average(aps[0, li, o50]),
                                "ap25%": np.average(aps[0, li, o25])}
    return avg


def format_results(avgs):
    """The table tools/eval.py:253-290 prints, as a list of lines."""
    lines = ["", "#" * 64, "{:<15}:{:>15}{:>15}{:>15}".format("what", "AP", "AP_50%", "AP_25%"), "#" * 64]
    for name in CLASS_LABELS:
        c = avgs["classes"][name]
        lines.append("{:<15}:{:>15.3f}{:>15.3f}{:>15.3f}".format(name, c["ap"], c["ap50%"], c["ap25%"]))
    lines += ["-" * 64, "{:<15}:{:>15.3f}{:>15.3f}{:>15.3f}".format("average", avgs["all_ap"], avgs["all_ap_50%"],
                                                                   avgs["all_ap_25%"]), ""]
    return lines


def print_results(avgs, logger=None):
    """tools/eval.py:253-326: the table through `logger.info` (or print)."""
    for line in format_results(avgs):
        (print if logger is None else logger.info)(line)
