"""oracle/sparse_ref.py -- TEST INFRASTRUCTURE ONLY: CPU restatement (numpy + plain torch) of the sparse-voxel
backbone the reference builds from MinkowskiEngine calls.  Never imported by the product path.

What it follows (paths relative to /root/reference/):
  network/Mink.py:218-288   MinkUNetBase.network_initialization (topology, channel plans)
  network/Mink.py:291-354   MinkUNetBase.forward (wiring, skip concatenations)
  network/Mink.py:75-107    _make_layer (1x1 conv + BN "downsample" iff the channel count changes)
  network/Mink.py:357-419   LAYERS / PLANES of MinkUNet14A, 34C, ...
  network/PBNet.py:117,240-247,265-271  SparseTensor construction (dedupe + inverse_mapping)
  datasets/scannetv2/dataset_preprocess.py:269-272  ME.utils.sparse_quantize

PARITY UNPINNED: the arithmetic itself lives in MinkowskiEngine, which is a third-party dependency that is
neither vendored nor version-pinned by the reference (README.md:15-27 installs git HEAD; the era-appropriate
release is v0.5.4) and is not installable here.  The conventions below restate ME 0.5.x's published
behaviour; they are assumptions and are kept in this one block so they can be flipped:

  C1 kernel offset index k -> offset: first spatial dimension (x) fastest.
  C2 odd kernel size K: offsets k_d - K//2; even K: offsets 0..K-1 (not centred); all scaled by the INPUT tensor stride.
  C3 convolution is cross-correlation: out[o] = sum_k in[o + delta_k] @ W[k],  W: [K^3, Cin, Cout].
  C4 strided conv output coordinates: floor(c / s_out) * s_out, de-duplicated, first-occurrence order.
  C5 transposed conv (k=2,s=2) writes onto the EXISTING coordinate map at the finer stride and uses the forward
     kernel map with in/out swapped: out[c] = in[parent(c)] @ W[k(c)].
  C6 duplicate input coordinates: the first occurrence survives (ME CPU map), survivors keep ascending original
     order, inverse_mapping sends every input row to its survivor; unique input keeps its row order.
  C7 MinkowskiBatchNorm = torch.nn.BatchNorm1d on the feature matrix (eps 1e-5, momentum 0.1).
  C8 1x1 stride-1 convolution kernel is stored [Cin, Cout] and is a plain matmul; bias is [1, Cout].
  C9 global pooling reduces per batch index, rows in ascending batch order; avg = sum / count.
"""
import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------------------------
# conventions (C1, C2)


def kernel_offsets(kernel_size, tensor_stride):
    k = int(kernel_size)
    rng = np.arange(k) - (k // 2 if k % 2 == 1 else 0)
    zz, yy, xx = np.meshgrid(rng, rng, rng, indexing="ij")  # x fastest when flattened
    off = np.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], 1).astype(np.int64)
    return off * int(tensor_stride)


# --------------------------------------------------------------------------------------------------------------
# coordinate bookkeeping (numpy, 64-bit packed keys)

_BIAS = 1 << 19
_BITS = 20


def pack(coords):
    c = np.asarray(coords, dtype=np.int64)
    assert (np.abs(c[:, 1:]) < _BIAS).all() and (c[:, 0] >= 0).all()
    return (c[:, 0] << (3 * _BITS)) | ((c[:, 1] + _BIAS) << (2 * _BITS)) | ((c[:, 2] + _BIAS) << _BITS) | (c[:, 3] + _BIAS)


def unique_first(coords):
    """C6: survivors (first occurrence) in ascending original order, and the inverse map."""
    keys = pack(coords)
    _, first, inverse = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    return first[order], rank[inverse.reshape(-1)]


class KeyIndex(object):
    """coordinate -> row lookup via sorted keys."""

    def __init__(self, coords):
        keys = pack(coords)
        self.order = np.argsort(keys, kind="stable")
        self.sorted = keys[self.order]

    def lookup(self, coords):
        keys = pack(coords)
        pos = np.searchsorted(self.sorted, keys)
        pos = np.minimum(pos, len(self.sorted) - 1)
        hit = self.sorted[pos] == keys
        return np.where(hit, self.order[pos], -1)


def stride_coords(coords, out_stride):
    """C4."""
    c = np.asarray(coords, dtype=np.int64).copy()
    c[:, 1:] = np.floor_divide(c[:, 1:], out_stride) * out_stride
    first, inverse = unique_first(c)
    return c[first].astype(np.int32), inverse


def kernel_map(in_coords, out_coords, offsets):
    """For every offset k: arrays (in_rows, out_rows) with in_coord = out_coord + offset_k (C3)."""
    index = KeyIndex(in_coords)
    oc = np.asarray(out_coords, dtype=np.int64)
    maps = []
    for k in range(len(offsets)):
        q = oc.copy()
        q[:, 1:] += offsets[k][None, :]
        rows = index.lookup(q)
        out_rows = np.nonzero(rows >= 0)[0]
        maps.append((rows[out_rows], out_rows))
    return maps


class CoordinateManager(object):
    """Caches coordinate sets per tensor stride and kernel maps per (stride_in, stride_out, kernel)."""

    def __init__(self, coords):
        self.coords = {1: np.asarray(coords, dtype=np.int32)}
        self.maps = {}

    def get_coords(self, stride):
        if stride not in self.coords:
            self.coords[stride], _ = stride_coords(self.get_coords(stride // 2), stride)
        return self.coords[stride]

    def get_map(self, in_stride, out_stride, kernel_size):
        key = (in_stride, out_stride, kernel_size)
        if key not in self.maps:
            self.maps[key] = kernel_map(self.get_coords(in_stride), self.get_coords(out_stride),
                                        kernel_offsets(kernel_size, in_stride))
        return self.maps[key]

    def n_pairs(self, in_stride, out_stride, kernel_size):
        return int(sum(len(m[0]) for m in self.get_map(in_stride, out_stride, kernel_size)))


# --------------------------------------------------------------------------------------------------------------
# arithmetic (plain torch on CPU): gather -> mm -> index_add_


def conv(feats, kernel, maps, n_out, bias=None):
    out = torch.zeros(n_out, kernel.shape[-1], dtype=feats.dtype)
    if kernel.dim() == 2:  # C8
        out = feats @ kernel
    else:
        for k, (in_rows, out_rows) in enumerate(maps):
            if len(in_rows) == 0:
                continue
            out.index_add_(0, torch.from_numpy(out_rows), feats[torch.from_numpy(in_rows)] @ kernel[k])
    if bias is not None:
        out = out + bias.reshape(1, -1)
    return out


def conv_transpose(feats, kernel, maps, n_out):
    """C5: maps are the FORWARD maps (in = fine rows, out = coarse rows); roles swapped here."""
    out = torch.zeros(n_out, kernel.shape[-1], dtype=feats.dtype)
    for k, (fine_rows, coarse_rows) in enumerate(maps):
        if len(fine_rows) == 0:
            continue
        out.index_add_(0, torch.from_numpy(fine_rows), feats[torch.from_numpy(coarse_rows)] @ kernel[k])
    return out


def batch_norm(feats, sd, prefix, training, eps=1e-5):
    """C7.  Does not update running statistics (the oracle is stateless)."""
    return F.batch_norm(feats, sd[prefix + ".bn.running_mean"].to(feats.dtype), sd[prefix + ".bn.running_var"].to(feats.dtype),
                        sd[prefix + ".bn.weight"].to(feats.dtype), sd[prefix + ".bn.bias"].to(feats.dtype),
                        training=training, momentum=0.0, eps=eps)


# --------------------------------------------------------------------------------------------------------------
# network restatement

ARCH = {  # network/Mink.py:357-419 (BasicBlock variants only; expansion = 1)
    "MinkUNet14A": dict(layers=(1, 1, 1, 1, 1, 1, 1, 1), planes=(32, 64, 128, 256, 128, 128, 96, 96)),
    "MinkUNet14B": dict(layers=(1, 1, 1, 1, 1, 1, 1, 1), planes=(32, 64, 128, 256, 128, 128, 128, 128)),
    "MinkUNet14C": dict(layers=(1, 1, 1, 1, 1, 1, 1, 1), planes=(32, 64, 128, 256, 192, 192, 128, 128)),
    "MinkUNet14D": dict(layers=(1, 1, 1, 1, 1, 1, 1, 1), planes=(32, 64, 128, 256, 384, 384, 384, 384)),
    "MinkUNet18A": dict(layers=(2, 2, 2, 2, 2, 2, 2, 2), planes=(32, 64, 128, 256, 128, 128, 96, 96)),
    "MinkUNet18B": dict(layers=(2, 2, 2, 2, 2, 2, 2, 2), planes=(32, 64, 128, 256, 128, 128, 128, 128)),
    "MinkUNet18D": dict(layers=(2, 2, 2, 2, 2, 2, 2, 2), planes=(32, 64, 128, 256, 384, 384, 384, 384)),
    "MinkUNet34A": dict(layers=(2, 3, 4, 6, 2, 2, 2, 2), planes=(32, 64, 128, 256, 256, 128, 64, 64)),
    "MinkUNet34B": dict(layers=(2, 3, 4, 6, 2, 2, 2, 2), planes=(32, 64, 128, 256, 256, 128, 64, 32)),
    "MinkUNet34C": dict(layers=(2, 3, 4, 6, 2, 2, 2, 2), planes=(32, 64, 128, 256, 256, 128, 96, 96)),
}


def basic_block(x, sd, prefix, cm, stride, training):
    """MinkowskiEngine.modules.resnet_block.BasicBlock (imported at Mink.py:11): conv3-BN-ReLU-conv3-BN, + residual
    (through `downsample` = 1x1 conv + BN when present, Mink.py:77-87), ReLU."""
    n = x.shape[0]
    m3 = cm.get_map(stride, stride, 3)
    out = conv(x, sd[prefix + ".conv1.kernel"], m3, n)
    out = torch.relu(batch_norm(out, sd, prefix + ".norm1", training))
    out = conv(out, sd[prefix + ".conv2.kernel"], m3, n)
    out = batch_norm(out, sd, prefix + ".norm2", training)
    if (prefix + ".downsample.0.kernel") in sd:
        res = conv(x, sd[prefix + ".downsample.0.kernel"], None, n)
        res = batch_norm(res, sd, prefix + ".downsample.1", training)
    else:
        res = x
    return torch.relu(out + res)


def minkunet_forward(sd, arch, feats, coords, training=False, dtype=torch.float32, cm=None, taps=None, detach=True):
    """MinkUNetBase.forward (Mink.py:291-354) on a state dict with the reference's parameter names.

    feats [V,Cin], coords int [V,4] (b,x,y,z), unique.  Returns features [V,Cout] in input row order."""
    cfg = ARCH[arch]
    if detach:
        sd = {k: v.detach().to("cpu").to(dtype) if v.is_floating_point() else v.detach().cpu() for k, v in sd.items()}
        x = feats.detach().cpu().to(dtype)
    else:  # keep the autograd graph: reference gradients for the training-path tests
        x = feats
    cm = cm or CoordinateManager(np.asarray(coords))
    n = {s: cm.get_coords(s).shape[0] for s in (1, 2, 4, 8, 16)}

    def tap(name, t):
        if taps is not None:
            taps[name] = t.clone()

    def block(x, name, nblocks, stride):
        for i in range(nblocks):
            x = basic_block(x, sd, "%s.%d" % (name, i), cm, stride, training)
        return x

    def down(x, conv_name, bn_name, s):
        out = conv(x, sd[conv_name + ".kernel"], cm.get_map(s, 2 * s, 2), n[2 * s])
        return torch.relu(batch_norm(out, sd, bn_name, training))

    def up(x, conv_name, bn_name, s):  # s = coarse stride
        out = conv_transpose(x, sd[conv_name + ".kernel"], cm.get_map(s // 2, s, 2), n[s // 2])
        return torch.relu(batch_norm(out, sd, bn_name, training))

    L = cfg["layers"]
    out = conv(x, sd["conv0p1s1.kernel"], cm.get_map(1, 1, 5), n[1])
    out_p1 = torch.relu(batch_norm(out, sd, "bn0", training))
    tap("out_p1", out_p1)
    out = down(out_p1, "conv1p1s2", "bn1", 1)
    out_b1p2 = block(out, "block1", L[0], 2)
    tap("out_b1p2", out_b1p2)
    out = down(out_b1p2, "conv2p2s2", "bn2", 2)
    out_b2p4 = block(out, "block2", L[1], 4)
    out = down(out_b2p4, "conv3p4s2", "bn3", 4)
    out_b3p8 = block(out, "block3", L[2], 8)
    out = down(out_b3p8, "conv4p8s2", "bn4", 8)
    out = block(out, "block4", L[3], 16)
    tap("block4", out)
    out = up(out, "convtr4p16s2", "bntr4", 16)
    out = block(torch.cat([out, out_b3p8], 1), "block5", L[4], 8)
    out = up(out, "convtr5p8s2", "bntr5", 8)
    out = block(torch.cat([out, out_b2p4], 1), "block6", L[5], 4)
    out = up(out, "convtr6p4s2", "bntr6", 4)
    out = block(torch.cat([out, out_b1p2], 1), "block7", L[6], 2)
    out = up(out, "convtr7p2s2", "bntr7", 2)
    out = block(torch.cat([out, out_p1], 1), "block8", L[7], 1)
    tap("block8", out)
    return conv(out, sd["final_sematic.kernel"], None, n[1], bias=sd["final_sematic.bias"])


# --------------------------------------------------------------------------------------------------------------
# ME.SparseTensor / utils restatements


def sparse_tensor(feats, coords):
    """ME.SparseTensor(features, coordinates) with the default RANDOM_SUBSAMPLE quantisation (C6).
    Returns (F_unique, C_unique, inverse_mapping)."""
    first, inverse = unique_first(np.asarray(coords))
    return feats[torch.from_numpy(first)], np.asarray(coords)[first], torch.from_numpy(inverse).long()


def batched_coordinates(list_xyz):
    """ME.utils.batched_coordinates: float coordinates are floored, batch index prepended."""
    out = []
    for b, c in enumerate(list_xyz):
        c = np.floor(np.asarray(c, dtype=np.float64) if not torch.is_tensor(c) else c.detach().cpu().double().numpy())
        out.append(np.concatenate([np.full((len(c), 1), b), c], 1).astype(np.int32))
    return np.concatenate(out, 0) if out else np.zeros((0, 4), np.int32)


def sparse_quantize(xyz, feats, quantization_size):
    """ME.utils.sparse_quantize(..., return_index=True, return_inverse=True) (dataset_preprocess.py:269-272)."""
    q = np.floor(np.asarray(xyz, dtype=np.float64) / quantization_size).astype(np.int32)
    c4 = np.concatenate([np.zeros((len(q), 1), np.int32), q], 1)
    first, inverse = unique_first(c4)
    return q[first], feats[first], first, inverse


def global_pool(feats, batch_idx, n_batch, mode):
    """C9."""
    b = torch.as_tensor(batch_idx).long()
    if mode == "avg":
        s = torch.zeros(n_batch, feats.shape[1], dtype=feats.dtype).index_add_(0, b, feats)
        cnt = torch.zeros(n_batch, dtype=feats.dtype).index_add_(0, b, torch.ones(len(b), dtype=feats.dtype))
        return s / cnt[:, None]
    out = torch.full((n_batch, feats.shape[1]), float("-inf"), dtype=feats.dtype)
    return out.scatter_reduce(0, b[:, None].expand_as(feats), feats, reduce="amax", include_self=True)
