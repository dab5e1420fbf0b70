"""ME.utils subset (/root/reference/network/PBNet.py:107,237,264; dataset_preprocess.py:269-272,296,348,375)."""
import math

import numpy as np
import torch

from .core import CoordinateManager


def batched_coordinates(coords, dtype=torch.int32, device=None):
    """List of [Ni,3] coordinates -> [sum Ni, 4] (batch, x, y, z); float input is FLOORED (ME behaviour)."""
    out = []
    for b, c in enumerate(coords):
        c = torch.as_tensor(c)
        if c.is_floating_point():
            c = torch.floor(c)
        c = c.to(dtype)
        out.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=dtype, device=c.device), c], 1))
    res = torch.cat(out, 0) if out else torch.zeros(0, 4, dtype=dtype)
    return res if device is None else res.to(device)


def sparse_collate(coords, feats, labels=None, dtype=torch.int32, device=None):
    bc = batched_coordinates(coords, dtype=dtype, device=device)
    bf = torch.cat([torch.as_tensor(f) for f in feats], 0)
    if labels is not None:
        return bc, bf, torch.cat([torch.as_tensor(l) for l in labels], 0)
    return bc, bf


def sparse_quantize(coordinates, features=None, labels=None, ignore_label=-100, return_index=False,
                    return_inverse=False, return_maps_only=False, quantization_size=None, device="cuda"):
    """Voxelise: floor(coordinates / quantization_size) -> unique (first occurrence) on the MI355X.

    Returns what the reference consumes at dataset_preprocess.py:269-272:
    (quantized_coords, features[index], index, inverse_index)."""
    is_np = isinstance(coordinates, np.ndarray)
    c = torch.as_tensor(coordinates)
    if quantization_size is not None:
        c = torch.floor(c.double() / quantization_size)
    q = c.to(torch.int32).to(device)
    c4 = torch.cat([torch.zeros(q.shape[0], 1, dtype=torch.int32, device=q.device), q], 1)
    cm = CoordinateManager(c4, prepare="unique")
    cm.num_rows(1)
    index, inverse = cm.unique_index, cm.inverse_mapping
    if return_maps_only:
        return (index, inverse) if return_inverse else index
    qc = q[index]
    outs = [qc.cpu().numpy() if is_np else qc]
    if features is not None:
        f = torch.as_tensor(features)
        fo = f.to(index.device)[index]
        outs.append(fo.cpu().numpy() if isinstance(features, np.ndarray) else fo)
    if return_index:
        outs.append(index.cpu().numpy() if is_np else index)
    if return_inverse:
        outs.append(inverse.cpu().numpy() if is_np else inverse)
    return tuple(outs) if len(outs) > 1 else outs[0]


def _calculate_fan_in_and_fan_out(tensor):
    if tensor.dim() < 2:
        raise ValueError("fan in/out need at least 2 dimensions")
    if tensor.dim() == 2:  # ME treats 2-D kernels like nn.Linear weights
        return tensor.size(1), tensor.size(0)
    receptive = tensor.size(0)
    return tensor.size(1) * receptive, tensor.size(2) * receptive


def kaiming_normal_(tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
    """ME.utils.kaiming_normal_ for kernels laid out [K, Cin, Cout] (Mink.py:68-69, PBNet.py:106-107)."""
    fan_in, fan_out = _calculate_fan_in_and_fan_out(tensor)
    fan = fan_in if mode == "fan_in" else fan_out
    gain = torch.nn.init.calculate_gain(nonlinearity, a)
    std = gain / math.sqrt(fan)
    with torch.no_grad():
        return tensor.normal_(0, std)
