#!/usr/bin/env python3
"""Golden data for the checkpoint row, produced IN THE BUILD CONTAINER from the reference:

  ckpt_ref/000000007.pth, ckpt_mod/000000003.pth -- files written by the reference's OWN `tools.log.checkpoint_save` (imported from
      /root/reference) for a tiny model + Adam optimizer; the second one through a wrapper whose keys carry the
      `module.` prefix DistributedDataParallel adds (train.py:345).  `ckpt_expect.json` holds what the reference's own
      `checkpoint_restore` returns for them (start epoch, file picked).
  ckpt_names.json -- the attribute names of the reference's module tree that end up in state-dict keys, read from the
      reference's source text with a regular expression (network/Mink.py `MinkUNetBase` -- the base of the 34C and 14A
      nets PBNet builds, PBNet.py:38-40 --,
      network/PBNet.py `PBNet.__init__`): name -> constructor.  The real model cannot be instantiated here
      (MinkowskiEngine is not installed); the leaf names under each constructor (`kernel`, `bn.weight`,
      `linear.weight`, `module.weight`) are MinkowskiEngine's, as listed in SURVEY.md 8b.

    python tests/golden/make_ckpt_golden.py"""
import json
import os
import re
import sys

import torch

sys.path.insert(0, "/root/reference")
import tools.log as ref_log          # noqa: E402  (reference code, executed here only)

HERE = os.path.dirname(os.path.abspath(__file__))


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(3, 2)
        self.b = torch.nn.BatchNorm1d(2)

    def forward(self, x):
        return self.b(self.a(x))


class Wrapped(torch.nn.Module):       # state-dict keys 'module.a.weight', ... like DistributedDataParallel
    def __init__(self, m):
        super().__init__()
        self.module = m


def class_body(src, name):
    m = re.search(r"^class %s\b.*?(?=^class |\Z)" % name, src, re.S | re.M)
    return m.group(0)


def attrs(body):
    out = {}
    for name, ctor in re.findall(r"self\.(\w+)\s*=\s*\(?\s*([\w\.]+)\(", body):
        out.setdefault(name, ctor.split(".")[-1])
    return out


if __name__ == "__main__":
    torch.manual_seed(3)
    expect = {}
    for sub, epoch, wrap in (("ckpt_ref", 7, False), ("ckpt_mod", 3, True)):
        d = os.path.join(HERE, sub) + "/"
        os.makedirs(d, exist_ok=True)
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        model = Tiny()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        model(torch.randn(5, 3)).sum().backward()
        opt.step()                                                  # optimizer state tensors in the file
        path = ref_log.checkpoint_save(Wrapped(model) if wrap else model, opt, d, epoch)
        fresh = Tiny()
        start, picked = ref_log.checkpoint_restore(fresh, None, d)   # optimizer=None: the reference calls .cuda() on its state
        assert all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), model.state_dict().values()))
        expect[sub] = dict(file=os.path.basename(path), start_epoch=start, picked=os.path.basename(picked),
                           keys=sorted(torch.load(path)["model"].keys()))
    json.dump(expect, open(os.path.join(HERE, "ckpt_expect.json"), "w"), indent=1, sort_keys=True)
    mink = open("/root/reference/network/Mink.py").read()
    pb = open("/root/reference/network/PBNet.py").read()
    names = dict(MinkUNetBase=attrs(class_body(mink, "MinkUNetBase")), PBNet=attrs(class_body(pb, "PBNet")))
    json.dump(names, open(os.path.join(HERE, "ckpt_names.json"), "w"), indent=1, sort_keys=True)
    print({k: len(v) for k, v in names.items()}, expect)
