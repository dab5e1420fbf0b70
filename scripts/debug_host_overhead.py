"""Host-side cost per call of the training path's Python wrappers (tiny tensors: the device work is negligible)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd.MinkowskiEngine.nn import bn_act
from pbnet_amd.MinkowskiEngine.modules.resnet_block import BasicBlock
dev = "cuda:0"
n = 256
i = torch.arange(n, dtype=torch.int32)
coords = torch.stack([torch.zeros_like(i), i % 16, (i // 16) % 16, torch.zeros_like(i)], 1).to(dev)
x = ME.SparseTensor(torch.randn(n, 32, device=dev).to(torch.bfloat16), coords)
conv = ME.MinkowskiConvolution(32, 32, kernel_size=3, dimension=3).to(dev).train()
bn = ME.MinkowskiBatchNorm(32).to(dev).train()
blk = BasicBlock(32, 32, dimension=3).to(dev).train()


def t(fn, reps=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / reps * 1e6


with torch.no_grad():
    print("conv forward, no grad      %.1f us" % t(lambda: conv(x)))
    print("bn_act forward, no grad    %.1f us" % t(lambda: bn_act(bn, x)))
print("conv forward, grad         %.1f us" % t(lambda: conv(x)))
print("bn_act forward, grad       %.1f us" % t(lambda: bn_act(bn, x)))
print("BasicBlock forward, grad   %.1f us" % t(lambda: blk(x)))
xg = ME.SparseTensor(x.F.clone().requires_grad_(True), coords, coordinate_manager=x.coordinate_manager) if hasattr(x, "coordinate_manager") else x


def fb():
    y = blk(x).F
    y.float().sum().backward()
print("BasicBlock fwd+bwd         %.1f us" % t(fb, 100))
