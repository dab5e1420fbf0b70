import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import pbnet_amd.MinkowskiEngine as ME
dev = torch.device("cuda:0")
cfg, model, b, t, info, raw = bench.build_workload(0, 1, torch.bfloat16, dev)
ret = bench.one_step(model, b, t)
pidx = ret["proposals"][0]
xyz = b["xyz_original"].float()
c3 = torch.floor(xyz[pidx[:, 1]] * 1 / 0.02).to(torch.int32)
coords3 = torch.cat([pidx[:, 0:1].to(torch.int32), c3], 1)
feats = torch.randn(coords3.shape[0], 32, device=dev).to(torch.bfloat16)
def wall(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    x = ME.SparseTensor(feats, coords3)
    print("rows", [x.coordinate_manager.num_rows(s) for s in (1, 2, 4, 8, 16)])
    print("score_Unet cached maps: %.2f ms" % wall(lambda: model.score_Unet(x)))
    print("D_Unet-like 14A on same coords (in=32 of 34): skip")
    def fresh():
        xx = ME.SparseTensor(feats, coords3)
        return model.score_Unet(xx)
    print("score_Unet fresh coords: %.2f ms" % wall(fresh))
    x1 = ME.SparseTensor(b["feat_voxel"], b["xyz_voxel"])
    print("MEUnet cached maps: %.2f ms" % wall(lambda: model.MEUnet(x1)))
    def fresh1():
        return model.MEUnet(ME.SparseTensor(b["feat_voxel"], b["xyz_voxel"]))
    print("MEUnet fresh coords: %.2f ms" % wall(fresh1))
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): fresh()
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
