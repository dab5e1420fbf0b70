"""Device-side "Voxel and Batch" block of the reference's collate functions
(/root/reference/datasets/scannetv2/dataset_preprocess.py:266-296 trainMerge, :345-375 valMerge): per scene
`ME.utils.sparse_quantize(xyz, feats, quantization_size, return_index, return_inverse)`, `v2p_index = inverse + offset`,
then `ME.utils.sparse_collate`.  Here the scenes of a batch are voxelised together by ONE de-duplication on the MI355X
(the batch index is part of the key, survivors keep the input order, i.e. scene-major with first occurrence inside a
scene -- exactly what quantising scene by scene and concatenating yields).  SURVEY.md 8(f) rank 2.

Dataset I/O, augmentation and instance bookkeeping stay with the caller (out of scope)."""
import numpy as np
import torch

from .MinkowskiEngine.core import CoordinateManager


def voxelize_batch(xyz_list, feat_list, voxel_size, device="cuda"):
    """xyz_list[i] [Ni,3] float, feat_list[i] [Ni,C] -> (xyz_voxel i32[V,4], feat_voxel [V,C], v2p_index i64[sum Ni])
    on `device`: the 'xyz_voxel' / 'feat_voxel' / 'v2p_index' entries of the reference's batch dict."""
    assert len(xyz_list) == len(feat_list) and len(xyz_list) > 0
    dev = torch.device(device)
    coords, feats = [], []
    for b, (xyz, f) in enumerate(zip(xyz_list, feat_list)):
        x = torch.as_tensor(xyz).to(dev)
        q = torch.floor(x.double() / voxel_size).to(torch.int32)               # ME.utils.sparse_quantize: floor(x / size)
        coords.append(torch.cat([torch.full((q.shape[0], 1), b, dtype=torch.int32, device=dev), q], 1))
        feats.append(torch.as_tensor(f).to(dev))
    c4 = torch.cat(coords, 0)
    cm = CoordinateManager(c4, prepare="unique")
    cm.num_rows(1)
    index, inverse = cm.unique_index, cm.inverse_mapping
    return c4[index], torch.cat(feats, 0)[index], inverse
