"""CPU tier: the reference's on-disk scene format (decode_scannet.py:194-200, dataset_preprocess.py:222-228) and the
ground-truth id files of get_val_gt.py, pinned by files the reference's OWN get_val_gt.py wrote (tests/golden/scene_io.npz,
made by tests/golden/make_scene_io_golden.py)."""
import os

import numpy as np
import pytest

from pbnet_amd import scene_io
from pbnet_amd.evaluate import load_gt_ids


def test_val_gt_files_equal_the_reference_scripts_output(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "scene_io.npz"))
    names = [str(n) for n in g["names"]]
    npy = str(tmp_path / "npy")
    rng = np.random.default_rng(0)
    for i, name in enumerate(names):
        n = g["sem_%d" % i].shape[0]
        scene_io.save_scene(npy, name, xyz=rng.normal(size=(n, 3)), rgb=rng.normal(size=(n, 3)), sem_label=g["sem_%d" % i],
                            ins_label=g["ins_%d" % i], nl=rng.normal(size=(n, 3)), face=np.zeros((4, 3)), sup=np.zeros(n))
    scene_io.write_val_gt(npy, str(tmp_path / "val_gt"), names)
    for i, name in enumerate(names):
        got = open(str(tmp_path / "val_gt" / (name + ".txt")), "rb").read()
        assert got == g["gt_txt_%d" % i].tobytes(), name                       # byte for byte
        ids = load_gt_ids(str(tmp_path / "val_gt" / (name + ".txt")))
        assert ids.shape[0] == g["sem_%d" % i].shape[0] and ((ids == 0) == (g["ins_%d" % i] < 0)).all()


def test_scene_round_trip_dtypes_and_errors(tmp_path):
    n = 64
    rng = np.random.default_rng(1)
    arrays = dict(xyz=rng.normal(size=(n, 3)), rgb=rng.normal(size=(n, 3)), sem_label=rng.integers(0, 20, n), ins_label=np.full(n, -100),
                  nl=rng.normal(size=(n, 3)), face=rng.integers(0, n, (10, 3)), sup=rng.integers(0, 5, n))
    scene_io.save_scene(str(tmp_path), "s", **arrays)
    back = scene_io.load_scene(str(tmp_path), "s")
    for k, (dt, tail) in scene_io.SCENE_ARRAYS.items():
        assert back[k].dtype == dt and back[k].shape[1:] == tail
        assert np.array_equal(back[k], np.asarray(arrays[k], dtype=dt))
    assert set(scene_io.load_scene(str(tmp_path), "s", with_mesh=False)) == {"xyz", "rgb", "sem_label", "ins_label", "nl"}
    with pytest.raises(FileNotFoundError):
        scene_io.load_scene(str(tmp_path), "missing")
    with pytest.raises(ValueError):
        scene_io.save_scene(str(tmp_path), "bad", **dict(arrays, rgb=arrays["rgb"][:10]))
    with pytest.raises(ValueError):
        scene_io.save_scene(str(tmp_path), "bad", **{k: v for k, v in arrays.items() if k != "sup"})
