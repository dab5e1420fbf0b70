#!/usr/bin/env python3
"""Golden files for the scene-file row (loader side), produced IN THE BUILD CONTAINER by the reference's
OWN script: /root/reference/datasets/scannetv2/get_val_gt.py is executed (runpy, unmodified) in a scratch directory that
holds `.npy` scenes written by pbnet_amd.scene_io.save_scene; what it writes to val_gt/ is stored beside the inputs.

    python tests/golden/make_scene_io_golden.py        # writes tests/golden/scene_io.npz
"""
import os
import runpy
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from pbnet_amd import scene_io  # noqa: E402


def scene(seed, n, n_inst):
    rng = np.random.default_rng(seed)
    ins = rng.integers(0, n_inst, n).astype(np.float64)
    ins[rng.random(n) < 0.3] = -100
    sem_of = rng.integers(0, 20, n_inst)
    sem_of[0] = 0
    sem = np.full(n, -100.0)
    m = ins >= 0
    sem[m] = sem_of[ins[m].astype(np.int64)]
    sem[(~m) & (rng.random(n) < 0.5)] = rng.integers(0, 2)           # unannotated-instance points with a stuff class
    if n_inst > 3:                                                  # an instance whose first point is semantic -100 -> class 0
        first = np.nonzero(ins == 3)[0][0]
        sem[first] = -100
    return dict(xyz=rng.normal(size=(n, 3)), rgb=rng.uniform(-1, 1, (n, 3)), sem_label=sem, ins_label=ins,
                nl=rng.normal(size=(n, 3)), face=rng.integers(0, n, (2 * n, 3)), sup=rng.integers(0, n // 20 + 1, n))


def main():
    out = {}
    names = ["scene0000_00", "scene0001_01", "scene0002_00"]
    with tempfile.TemporaryDirectory() as tmp:
        base = os.path.join(tmp, "datasets", "scannetv2")
        os.makedirs(base)
        for i, name in enumerate(names):
            s = scene(i, 500 + 311 * i, 4 + 3 * i)
            scene_io.save_scene(os.path.join(base, "npy"), name, **s)
            out["sem_%d" % i], out["ins_%d" % i] = s["sem_label"], s["ins_label"]
        np.savetxt(os.path.join(base, "scannetv2_val.txt"), np.array(names), fmt="%s")
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            runpy.run_path("/root/reference/datasets/scannetv2/get_val_gt.py", run_name="__main__")   # reference code, executed here only
        finally:
            os.chdir(cwd)
        for i, name in enumerate(names):
            out["gt_txt_%d" % i] = np.frombuffer(open(os.path.join(base, "val_gt", name + ".txt"), "rb").read(), dtype=np.uint8)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "scene_io.npz"), **out)
    print("wrote scene_io.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
