"""Where the HOST spends the forward / backward of the configs[2] training step: cProfile over a few steps (scripts/train_step.py's
model, scene and optimiser), top functions by own and by cumulative time.  The waits for the GPU show up as the read-backs'
own time (`cpu`, `tolist`, `item`)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    from pbnet_amd import synth
    from pbnet_amd.config import get_config
    from pbnet_amd.network.PBNet import PBNet, model_fn
    dev = torch.device("cuda:0")
    cfg = get_config(batch_size=1, cluster_epoch=0)
    torch.manual_seed(22)
    model = PBNet(cfg).to(dev).train()
    batch_np, teacher_np, info = synth.make_train_batch(seed=10, copies=1)
    t = torch.from_numpy
    batch = {k: t(v).to(dev) for k, v in batch_np.items()}
    batch["feat_voxel"] = batch["feat_voxel"].to(torch.bfloat16)
    teacher = {k: t(v).to(dev) for k, v in teacher_np.items()}
    fwd = model.forward
    model.forward = lambda *a, **k: fwd(*a, teacher=teacher, **k)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)

    def step():
        opt.zero_grad(set_to_none=True)
        loss, pred, visual, meter = model_fn(batch, model, 1, cfg, "train")
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    for key in ("tottime", "cumulative"):
        st = pstats.Stats(pr, stream=sys.stdout)
        st.sort_stats(key)
        print("==== by %s (totals over %d steps)" % (key, steps))
        st.print_stats(45)


if __name__ == "__main__":
    main()
