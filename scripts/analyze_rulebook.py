"""CPU analysis of the bench scene's rulebooks (no GPU): how much of the dense (row fragment x kernel offset) work of an
output-stationary tile is populated, per level and fragment height, in Z-order.  Guides the tile shape of k_spconv."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import sparse_ref as R
from pbnet_amd import synth


def spread3(v):
    x = v.astype(np.uint64) & np.uint64(0xffff)
    x = (x | (x << np.uint64(32))) & np.uint64(0x00ff00000000ffff)
    x = (x | (x << np.uint64(16))) & np.uint64(0x00ff0000ff0000ff)
    x = (x | (x << np.uint64(8))) & np.uint64(0xf00f00f00f00f00f)
    x = (x | (x << np.uint64(4))) & np.uint64(0x30c30c30c30c30c3)
    x = (x | (x << np.uint64(2))) & np.uint64(0x9249249249249249)
    return x


def morton_order(c):
    m = spread3(c[:, 1] + 32768) | (spread3(c[:, 2] + 32768) << np.uint64(1)) | (spread3(c[:, 3] + 32768) << np.uint64(2))
    key = (c[:, 0].astype(np.uint64) << np.uint64(48)) | (m & np.uint64(0xffffffffffff))
    return np.argsort(key, kind="stable")


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    batch, _, info = synth.make_val_batch(seed=seed, copies=1)
    c = batch["xyz_voxel"]
    c = c[morton_order(c)]
    cm = R.CoordinateManager(c)
    for s in (1, 2, 4, 8, 16):
        cc = cm.get_coords(s)
        n = len(cc)
        maps = cm.get_map(s, s, 3)
        nbr = np.full((n, 27), -1, np.int64)
        for k, (i, o) in enumerate(maps):
            nbr[o, k] = i
        pop = nbr >= 0
        pairs = int(pop.sum())
        line = "stride %2d: %7d rows, %8d pairs (%.2f/row)" % (s, n, pairs, pairs / n)
        for F in (16, 32, 64, 128):
            nf = (n + F - 1) // F
            pad = np.zeros((nf * F, 27), bool)
            pad[:n] = pop
            fr = pad.reshape(nf, F, 27)
            anyp = fr.any(1)
            dense_rows = int(anyp.sum()) * F
            line += " | F=%3d: populated %.2f, MFMA work/useful %.2f" % (F, anyp.mean(), dense_rows / max(pairs, 1))
        print(line)
        # mask-sorted inside Z-order blocks of 2048 rows: rows with the same neighbour mask adjacent
        bits = (pop * (1 << np.arange(27))).sum(1)
        B = 2048
        order = np.concatenate([np.arange(b, min(b + B, n))[np.argsort(bits[b:min(b + B, n)], kind="stable")] for b in range(0, n, B)])
        pop2 = pop[order]
        line = "      mask-sorted in %d-row blocks:" % B
        for F in (16, 32, 64):
            nf = (n + F - 1) // F
            pad = np.zeros((nf * F, 27), bool)
            pad[:n] = pop2
            anyp = pad.reshape(nf, F, 27).any(1)
            line += " F=%3d: populated %.2f, work/useful %.2f |" % (F, anyp.mean(), int(anyp.sum()) * F / max(pairs, 1))
        print(line)


if __name__ == "__main__":
    main()
