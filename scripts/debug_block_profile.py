"""cProfile of one BasicBlock forward + backward on the native training path (tiny tensors: host cost only)."""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.autograd.set_multithreading_enabled(False)
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd.MinkowskiEngine.modules.resnet_block import BasicBlock
dev = "cuda:0"
n = 256
i = torch.arange(n, dtype=torch.int32)
coords = torch.stack([torch.zeros_like(i), i % 16, (i // 16) % 16, torch.zeros_like(i)], 1).to(dev)
x = ME.SparseTensor(torch.randn(n, 32, device=dev).to(torch.bfloat16).requires_grad_(True), coords)
blk = BasicBlock(32, 32, dimension=3).to(dev).train()
def fb():
    y = blk(x).F
    y.backward(torch.ones_like(y))
for _ in range(30): fb()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): fb()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
