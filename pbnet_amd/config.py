"""The cfg contract PBNet reads (/root/reference/config/config.py:10-67, config_test.py): field names and defaults
only -- the reference's argparse CLI itself is outside the hot path."""
from types import SimpleNamespace

TRAIN_DEFAULTS = dict(task="train", manual_seed=22, voxel_size=0.02, scale_size=1, sem_num=20, batch_size=4,
                      batch_size_v=1, cluster_epoch=128, min_pts=31, radius=0.04, method=0, fg_thresh=0.95,
                      bg_thresh=0.20, TEST_NMS_THRESH=0.10, TEST_SCORE_THRESH=0.07, TEST_NPOINT_THRESH=101)
TEST_OVERRIDES = dict(task="test", batch_size=1, cluster_epoch=-1)


def get_config(test=False, **overrides):
    cfg = dict(TRAIN_DEFAULTS)
    if test:
        cfg.update(TEST_OVERRIDES)
    cfg.update(overrides)
    return SimpleNamespace(**cfg)
