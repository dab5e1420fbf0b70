"""On-disk scene format of the reference's ScanNet pipeline (loader side):
/root/reference/datasets/scannetv2/decode_scannet.py:194-200 writes seven `.npy` arrays per scene,
dataset_preprocess.py:222-228 reads them back, get_val_gt.py:19-39 turns two of them into the evaluator's ground-truth
ids (pbnet_amd.evaluate.encode_gt_ids / save_gt_ids).  Plain numpy I/O: host-side file handling around the hot path."""
import os

import numpy as np

# suffix -> (dtype on disk, trailing shape) exactly as decode_scannet.py produces them
SCENE_ARRAYS = {
    "xyz": (np.float32, (3,)),        # vertex coordinates, mean-centred (decode_scannet.py:92-97)
    "rgb": (np.float32, (3,)),        # colours / 127.5 - 1
    "sem_label": (np.float64, ()),    # 0..19 or -100 (np.ones(...) * -100: float64 on disk)
    "ins_label": (np.float64, ()),    # 0..I-1 or -100
    "nl": (np.float32, (3,)),         # per-vertex normals
    "face": (np.int32, (3,)),         # mesh triangles
    "sup": (np.int64, ()),            # superpoint (segment) id per vertex
}


def scene_path(npy_dir, scene, suffix):
    return os.path.join(npy_dir, "%s_%s.npy" % (scene, suffix))


def save_scene(npy_dir, scene, **arrays):
    """decode_scannet.py:194-200: `<scene>_{xyz,rgb,sem_label,ins_label,nl,face,sup}.npy`."""
    missing = set(SCENE_ARRAYS) - set(arrays)
    if missing:
        raise ValueError("save_scene: missing arrays %s" % sorted(missing))
    os.makedirs(npy_dir, exist_ok=True)
    n = None
    for suffix, (dtype, tail) in SCENE_ARRAYS.items():
        a = np.asarray(arrays[suffix], dtype=dtype)
        if a.shape[1:] != tail:
            raise ValueError("save_scene: %s must have shape [*, %s]" % (suffix, tail))
        if suffix != "face":
            if n is None:
                n = a.shape[0]
            elif a.shape[0] != n:
                raise ValueError("save_scene: %s has %d rows, expected %d" % (suffix, a.shape[0], n))
        np.save(scene_path(npy_dir, scene, suffix), a)


def load_scene(npy_dir, scene, with_mesh=True):
    """dataset_preprocess.py:222-228 (train / val read xyz, rgb, sem_label, ins_label, nl; eval_map.py adds sup)."""
    names = list(SCENE_ARRAYS) if with_mesh else ["xyz", "rgb", "sem_label", "ins_label", "nl"]
    out = {}
    for suffix in names:
        path = scene_path(npy_dir, scene, suffix)
        if not os.path.exists(path):
            raise FileNotFoundError("scene %s: %s is missing" % (scene, path))
        out[suffix] = np.load(path)
    n = out["xyz"].shape[0]
    for suffix, a in out.items():
        if suffix != "face" and a.shape[0] != n:
            raise ValueError("scene %s: %s has %d rows, xyz has %d" % (scene, suffix, a.shape[0], n))
    return out


def write_val_gt(npy_dir, gt_dir, scenes):
    """get_val_gt.py:16-39 for a list of scenes: `<gt_dir>/<scene>.txt`, one id per line."""
    from .evaluate import encode_gt_ids, save_gt_ids
    os.makedirs(gt_dir, exist_ok=True)
    for scene in scenes:
        sem = np.load(scene_path(npy_dir, scene, "sem_label"))
        ins = np.load(scene_path(npy_dir, scene, "ins_label"))
        save_gt_ids(os.path.join(gt_dir, scene + ".txt"), encode_gt_ids(sem, ins))
