"""Manual stress (not collected by pytest): N random scenes of different sizes, each evaluated once alone, then ROUNDS times
by WORKERS host threads on their own HIP streams in shuffled order (bf16 slabs, the bench configuration).  Every in-flight
result must equal the stand-alone result bit for bit.  usage: fuzz_inflight.py [scenes=8] [workers=4] [rounds=6]"""
import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from pbnet_amd import synth
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import PBNet

NS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
WORKERS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda", 0)
cfg = get_config(test=True)
torch.manual_seed(22)
model = PBNet(cfg).to(dev).eval()
for p in model.parameters():
    p.data = p.data.to(torch.float32)
scenes = []
for seed in range(1, NS + 1):
    rng = np.random.default_rng(100 + seed)
    room = (float(rng.uniform(1.2, 4.0)), float(rng.uniform(1.0, 3.2)), float(rng.uniform(0.9, 2.6)))
    nb = int(rng.integers(2, 12))
    classes = tuple(int(c) for c in rng.choice(np.arange(2, 20), size=min(5, nb), replace=False))
    batch, teacher, info = synth.make_val_batch(seed=seed, copies=int(rng.integers(1, 4)), room=room, n_boxes=nb,
                                                pitch=float(rng.choice([0.0225, 0.03])), classes=classes)
    b = {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}
    b["feat_voxel"] = b["feat_voxel"].to(torch.bfloat16)
    t = {k: torch.from_numpy(v).to(dev) for k, v in teacher.items()}
    scenes.append((b, t, info["n_points"]))


def run(i):
    b, t, _ = scenes[i]
    with torch.no_grad():
        r = model(b["feat_voxel"], b["xyz_voxel"], b["xyz_original"], b["v2p_index"], None, 1, "test", teacher=t)
    return [r["proposals"][0].cpu(), r["proposals"][1].cpu(), r["proposals"][3].float().cpu(), r["clt_scores"].float().cpu(),
            r["sem_pred_p"].cpu()]


want = [run(i) for i in range(NS)]
torch.cuda.synchronize()


def same(a, b):
    if a.shape != b.shape:
        return False
    return torch.equal(a, b)


print("scenes:", ", ".join("%dk pts/%d proposals" % (s[2] // 1000, w[1].shape[0] - 1) for s, w in zip(scenes, want)))
bad, errors = [], []


def worker(w):
    try:
        torch.cuda.set_device(dev)
        rng = np.random.default_rng(w)
        with torch.cuda.stream(torch.cuda.Stream(dev)):
            for r in range(ROUNDS):
                for i in rng.permutation(NS):
                    got = run(int(i))
                    if not all(same(a, b) for a, b in zip(got, want[int(i)])):
                        bad.append((w, r, int(i)))
    except BaseException as e:
        errors.append(e)


ths = [threading.Thread(target=worker, args=(w,)) for w in range(WORKERS)]
for th in ths: th.start()
for th in ths: th.join()
print("forwards in flight: %d, mismatches: %d, errors: %d" % (WORKERS * ROUNDS * NS, len(bad), len(errors)), bad[:5], errors[:2])
sys.exit(1 if bad or errors else 0)
