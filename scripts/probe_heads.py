"""pbn_mlp_rows on the bench scene's point count: microseconds per head, matrix-core form against the scalar kernel
(PBN_MLP_FORM=0 in a second process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import stage_ops
dev = "cuda:0"
torch.manual_seed(0)
n_vox, n_pts = 146038, 161517
for dt in (torch.bfloat16, torch.float32):
    f = torch.randn(n_vox, 32, device=dev).to(dt)
    idx = torch.randint(0, n_vox, (n_pts,), device=dev)
    for hidden, n_out, sig in ((32, 32, False), (16, 20, False), (16, 3, False), (16, 1, True)):
        layers = [ME.MinkowskiLinear(32, hidden, bias=False), ME.MinkowskiBatchNorm(hidden), ME.MinkowskiPReLU(), ME.MinkowskiLinear(hidden, n_out, bias=True)]
        if sig:
            layers.append(ME.MinkowskiSigmoid())
        head = torch.nn.Sequential(*layers).to(dev).eval()
        for _ in range(3):
            stage_ops.mlp_rows(head, f, idx)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()                     # replayed from a HIP graph: the host side of a call costs more than the kernel
        with torch.cuda.graph(g):
            for _ in range(20):
                stage_ops.mlp_rows(head, f, idx)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        print("%s 32->%d->%d%s over %d gathered rows: %.1f us (form %s, blocks/wave %s)" % (dt, hidden, n_out, " sigmoid" if sig else "", n_pts,
              e0.elapsed_time(e1) / 40 * 1e3, os.environ.get("PBN_MLP_FORM", "1"), os.environ.get("PBN_MLP_BLOCKS", "1")))
