"""CPU: checkpoint format against files written by the reference's own tools.log.checkpoint_save and
the names of the reference's module tree (tests/golden/make_ckpt_golden.py)."""
import json
import os
import shutil

import torch

from pbnet_amd import checkpoint as C
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import PBNet

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
EXPECT = json.load(open(os.path.join(GOLD, "ckpt_expect.json")))
NAMES = json.load(open(os.path.join(GOLD, "ckpt_names.json")))


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(3, 2)
        self.b = torch.nn.BatchNorm1d(2)

    def forward(self, x):
        return self.b(self.a(x))


def test_restore_reads_reference_files(tmp_path):
    for sub in ("ckpt_ref", "ckpt_mod"):
        d = str(tmp_path / sub) + "/"
        shutil.copytree(os.path.join(GOLD, sub), d)
        raw = torch.load(os.path.join(d, EXPECT[sub]["file"]), map_location="cpu")
        assert sorted(raw) == ["model", "optimizer"] and sorted(raw["model"]) == EXPECT[sub]["keys"]
        model = Tiny()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        start, picked = C.checkpoint_restore(model, opt, d)
        assert start == EXPECT[sub]["start_epoch"] and os.path.basename(picked) == EXPECT[sub]["picked"]
        want = {k[len("module."):] if k.startswith("module.") else k: v for k, v in raw["model"].items()}
        got = model.state_dict()
        assert set(got) == set(want) and all(torch.equal(got[k], want[k]) for k in want)
        st = opt.state_dict()["state"]
        assert len(st) == len(raw["optimizer"]["state"]) > 0
        assert all(torch.equal(st[i]["exp_avg"], raw["optimizer"]["state"][i]["exp_avg"]) for i in st)
        # explicit epoch and explicit file
        assert C.checkpoint_restore(Tiny(), None, d, epoch=start - 1)[0] == start
        assert C.checkpoint_restore(Tiny(), None, "/nonexistent/", pretrain_file=picked) == (start, picked)
    assert C.checkpoint_restore(Tiny(), None, str(tmp_path / "empty") + "/") == (1, "")


def test_save_layout_and_pruning(tmp_path):
    d = str(tmp_path / "exp" / "PBNet") + "/"
    model = Tiny()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    model(torch.randn(4, 3)).sum().backward()
    opt.step()
    files = [C.checkpoint_save(model, opt, d, e, save_freq=4) for e in range(1, 10)]
    assert files[-1] == d + "000000009.pth"
    assert sorted(os.listdir(d)) == ["000000004.pth", "000000008.pth", "000000009.pth"]       # tools/log.py:112-115
    raw = torch.load(files[-1], map_location="cpu")
    ref = torch.load(os.path.join(GOLD, "ckpt_ref", EXPECT["ckpt_ref"]["file"]), map_location="cpu")
    assert sorted(raw) == sorted(ref) and sorted(raw["model"]) == sorted(ref["model"])
    assert sorted(raw["optimizer"]) == sorted(ref["optimizer"])
    fresh = Tiny()
    assert C.checkpoint_restore(fresh, None, d) == (10, files[-1])                              # newest by name
    assert all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), model.state_dict().values()))


def _leaves(kind):
    bn = ["bn.weight", "bn.bias", "bn.running_mean", "bn.running_var", "bn.num_batches_tracked"]
    return {"MinkowskiConvolution": ["kernel", "bias"], "MinkowskiConvolutionTranspose": ["kernel", "bias"],
            "MinkowskiBatchNorm": bn}[kind]


def test_state_dict_keys_follow_the_reference_module_tree():
    """A released checkpoint is loaded by key name (strict=False drops what does not match, silently): every key of
    this model must be spelled from the reference's attribute names and MinkowskiEngine's leaf names."""
    keys = list(PBNet(get_config(test=True)).state_dict().keys())
    unet, top = NAMES["MinkUNetBase"], NAMES["PBNet"]
    nets = [n for n, k in top.items() if k == "unet3d"]
    assert sorted(nets) == ["D_Unet", "MEUnet", "score_Unet"]
    bn = _leaves("MinkowskiBatchNorm")
    seen_attr = set()
    for key in keys:
        head, rest = key.split(".", 1)
        assert head in top, key
        if head in nets:
            attr, leaf = rest.split(".", 1)
            assert attr in unet, key
            seen_attr.add((head, attr))
            kind = unet[attr]
            if kind == "_make_layer":                        # BasicBlock: conv1/norm1/conv2/norm2/downsample(conv, norm)
                idx, part, leaf2 = leaf.split(".", 2)
                assert idx.isdigit() and part in ("conv1", "conv2", "norm1", "norm2", "downsample"), key
                if part.startswith("conv"):
                    assert leaf2 == "kernel", key
                elif part.startswith("norm"):
                    assert leaf2 in bn, key
                else:
                    assert leaf2 == "0.kernel" or (leaf2.startswith("1.") and leaf2[2:] in bn), key
            else:
                assert leaf in _leaves(kind), key
        elif top[head] == "Sequential":                      # heads: MinkowskiLinear / BatchNorm / PReLU / Linear
            idx, leaf = rest.split(".", 1)
            assert idx.isdigit() and leaf in ["linear.weight", "linear.bias", "module.weight"] + bn, key
        else:
            raise AssertionError("unexpected parameter under %s: %s" % (head, key))
    # and nothing of the reference's parameterised tree is missing
    for net in nets:
        for attr, kind in unet.items():
            if kind != "MinkowskiReLU":
                assert (net, attr) in seen_attr, (net, attr)
    for head, kind in top.items():
        if kind == "Sequential":
            assert any(k.startswith(head + ".") for k in keys), head
