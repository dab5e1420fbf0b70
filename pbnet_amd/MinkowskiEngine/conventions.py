"""MinkowskiEngine behavioural conventions assumed by this rebuild -- ONE switchable place.

The reference neither vendors nor pins MinkowskiEngine (/root/reference/README.md:15-27; era release v0.5.4) and it
cannot be installed here, so these are restatements of its published behaviour, not observations:

  C1 kernel index -> offset: first spatial dimension (x) fastest.
  C2 odd K centred (k - K//2); even K not centred (0..K-1); offsets scaled by the input tensor stride.
  C3 cross-correlation: out[o] = sum_k in[o + delta_k] @ W[k]; W stored [K^3, Cin, Cout].
  C4 strided output coordinates floor(c / s) * s, de-duplicated.
  C5 transposed conv writes onto the existing finer coordinate map using the forward map with in/out swapped.
  C6 duplicate coordinates: first occurrence survives, ascending survivor order, unique input keeps its order.
  C8 1x1 stride-1 kernels are stored [Cin, Cout]; bias is [1, Cout].
"""
import torch

X_FASTEST = True


def kernel_offsets(kernel_size, tensor_stride):
    """int32 [K^3, 3] offsets in voxel units of the finest grid."""
    k = int(kernel_size)
    r = torch.arange(k, dtype=torch.int32) - (k // 2 if k % 2 == 1 else 0)
    zz, yy, xx = torch.meshgrid(r, r, r, indexing="ij")
    if X_FASTEST:
        off = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], 1)
    else:
        off = torch.stack([zz.reshape(-1), yy.reshape(-1), xx.reshape(-1)], 1)
    return (off * int(tensor_stride)).contiguous()
