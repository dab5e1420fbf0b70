"""tests/dense_backbone.py -- an INDEPENDENT dense statement of the sparse-voxel backbone, the backbone's counterpart of
tests/bruteforce_cluster.py.  TEST INFRASTRUCTURE ONLY.

oracle/sparse_ref.py and pbnet_amd/csrc/spconv.hip share one reading of MinkowskiEngine's conventions (offset order,
weight layout, even kernels not centred, transposed map).  This file states the same network WITHOUT any coordinate
map, kernel map or gather: the active voxels are scattered into a dense [B, C, X, Y, Z] grid and the layers are
torch.nn.functional.conv3d / conv_transpose3d / batch_norm on that grid, re-masked to the active set after every layer
(a sparse convolution on a fixed coordinate set IS the dense convolution restricted to that set, zeros elsewhere;
MinkowskiEngine is the sparse generalisation of exactly these dense operators):

  k=3 / k=5, stride 1   F.conv3d(padding=K//2)            weight[co,ci,kx,ky,kz] = W[kx + K*ky + K*K*kz][ci,co]
  k=2, stride 2 down    F.conv3d(stride=2)                active coarse sites = max_pool3d(mask, 2)
  k=2, stride 2 up      F.conv_transpose3d(stride=2)      restricted to the fine level's active set
  1x1                   F.conv3d(kernel 1)
  BatchNorm             F.batch_norm over the ACTIVE sites (train) or with running statistics (eval)

Follows /root/reference/network/Mink.py:291-354 (forward wiring), :75-107 (_make_layer), :218-288 (constructors).
Coordinates must be shifted so that every coordinate is >= 0 and the shift is a multiple of 16 (stride alignment)."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle.sparse_ref import ARCH


def _w3(kernel, K):
    """[K^3, Ci, Co] (x fastest) -> conv3d weight [Co, Ci, X, Y, Z]."""
    ci, co = kernel.shape[1], kernel.shape[2]
    return kernel.view(K, K, K, ci, co).permute(4, 3, 2, 1, 0).contiguous()      # (kz,ky,kx,ci,co) -> (co,ci,kx,ky,kz)


def _w3t(kernel, K):
    """[K^3, Ci, Co] -> conv_transpose3d weight [Ci, Co, X, Y, Z]."""
    ci, co = kernel.shape[1], kernel.shape[2]
    return kernel.view(K, K, K, ci, co).permute(3, 4, 2, 1, 0).contiguous()


class DenseScene(object):
    """Active-set masks of the five levels and scatter / sample between rows and the dense grid."""

    def __init__(self, coords, extent):
        c = np.asarray(coords, dtype=np.int64)
        lo = c[:, 1:].min(0)
        shift = (-(lo // 16) * 16).astype(np.int64)        # multiple of 16: floor(c / s) * s commutes with the shift
        self.c = c.copy()
        self.c[:, 1:] += shift[None, :]
        assert (self.c[:, 1:] >= 0).all() and (self.c[:, 1:] < extent).all() and extent % 16 == 0
        self.B = int(c[:, 0].max()) + 1
        self.E = extent
        m = torch.zeros(self.B, 1, extent, extent, extent)
        m[self.c[:, 0], 0, self.c[:, 1], self.c[:, 2], self.c[:, 3]] = 1.0
        self.mask = {1: m}
        for s in (2, 4, 8, 16):
            self.mask[s] = F.max_pool3d(self.mask[s // 2], 2)

    def scatter(self, feats):
        g = torch.zeros(self.B, feats.shape[1], self.E, self.E, self.E, dtype=feats.dtype)
        g[self.c[:, 0], :, self.c[:, 1], self.c[:, 2], self.c[:, 3]] = feats
        return g

    def sample(self, grid):
        return grid[self.c[:, 0], :, self.c[:, 1], self.c[:, 2], self.c[:, 3]]

    def rows_at(self, grid, coords, stride):
        """Rows of a level-`stride` grid at the given (un-shifted-consistent) coordinates [n,4] (b, x, y, z multiples of stride)."""
        c = np.asarray(coords, dtype=np.int64)
        return grid[c[:, 0], :, c[:, 1] // stride, c[:, 2] // stride, c[:, 3] // stride]


def bn_dense(x, mask, sd, prefix, training, eps=1e-5):
    w, b = sd[prefix + ".bn.weight"], sd[prefix + ".bn.bias"]
    if training:                                            # statistics over the active sites only
        act = mask.expand_as(x) > 0
        C = x.shape[1]
        rows = x.permute(0, 2, 3, 4, 1)[act.permute(0, 2, 3, 4, 1)].view(-1, C)
        mean, var = rows.mean(0), rows.var(0, unbiased=False)
    else:
        mean, var = sd[prefix + ".bn.running_mean"], sd[prefix + ".bn.running_var"]
    v = lambda t: t.view(1, -1, 1, 1, 1)
    return ((x - v(mean)) / torch.sqrt(v(var) + eps) * v(w) + v(b)) * mask


def conv_same(x, kernel, K):
    if kernel.dim() == 2:
        return F.conv3d(x, kernel.t().contiguous().view(kernel.shape[1], kernel.shape[0], 1, 1, 1))
    return F.conv3d(x, _w3(kernel, K), padding=K // 2)


def basic_block_dense(x, mask, sd, prefix, training):
    out = conv_same(x, sd[prefix + ".conv1.kernel"], 3) * mask
    out = torch.relu(bn_dense(out, mask, sd, prefix + ".norm1", training))
    out = conv_same(out, sd[prefix + ".conv2.kernel"], 3) * mask
    out = bn_dense(out, mask, sd, prefix + ".norm2", training)
    if (prefix + ".downsample.0.kernel") in sd:
        res = conv_same(x, sd[prefix + ".downsample.0.kernel"], 1) * mask
        res = bn_dense(res, mask, sd, prefix + ".downsample.1", training)
    else:
        res = x
    return torch.relu(out + res) * mask


def minkunet_dense(sd, arch, feats, coords, extent, training=False, taps=None):
    """The whole U-Net on dense grids; returns features [V, Cout] in input row order (coords unique)."""
    sd = {k: v.detach().cpu().float() if v.is_floating_point() else v.detach().cpu() for k, v in sd.items()}
    sc = DenseScene(coords, extent)
    L = ARCH[arch]["layers"]
    M = sc.mask

    def block(x, name, nb, s):
        for i in range(nb):
            x = basic_block_dense(x, M[s], sd, "%s.%d" % (name, i), training)
        return x

    def down(x, cname, bname, s):
        out = F.conv3d(x, _w3(sd[cname + ".kernel"], 2), stride=2) * M[2 * s]
        return torch.relu(bn_dense(out, M[2 * s], sd, bname, training))

    def up(x, cname, bname, s):
        out = F.conv_transpose3d(x, _w3t(sd[cname + ".kernel"], 2), stride=2) * M[s // 2]
        return torch.relu(bn_dense(out, M[s // 2], sd, bname, training))

    x = sc.scatter(feats.detach().cpu().float())
    out = conv_same(x, sd["conv0p1s1.kernel"], 5) * M[1]
    out_p1 = torch.relu(bn_dense(out, M[1], sd, "bn0", training))
    if taps is not None:
        taps["out_p1"] = sc.sample(out_p1)
    out_b1p2 = block(down(out_p1, "conv1p1s2", "bn1", 1), "block1", L[0], 2)
    out_b2p4 = block(down(out_b1p2, "conv2p2s2", "bn2", 2), "block2", L[1], 4)
    out_b3p8 = block(down(out_b2p4, "conv3p4s2", "bn3", 4), "block3", L[2], 8)
    out = block(down(out_b3p8, "conv4p8s2", "bn4", 8), "block4", L[3], 16)
    out = block(torch.cat([up(out, "convtr4p16s2", "bntr4", 16), out_b3p8], 1), "block5", L[4], 8)
    out = block(torch.cat([up(out, "convtr5p8s2", "bntr5", 8), out_b2p4], 1), "block6", L[5], 4)
    out = block(torch.cat([up(out, "convtr6p4s2", "bntr6", 4), out_b1p2], 1), "block7", L[6], 2)
    out = block(torch.cat([up(out, "convtr7p2s2", "bntr7", 2), out_p1], 1), "block8", L[7], 1)
    fs = sd["final_sematic.kernel"]
    out = conv_same(out, fs, 1) + sd["final_sematic.bias"].view(1, -1, 1, 1, 1)
    return sc.sample(out)
