"""Checkpoint files of the reference's training loop (/root/reference/tools/log.py:64-116).

Layout: `<logpath><epoch:09d>.pth` (logpath is a string PREFIX, normally ending in '/'), a `torch.save` of
`{'model': state_dict, 'optimizer': state_dict}`.  Keys saved from a DistributedDataParallel wrapper carry a `module.`
prefix, stripped on load; loading is non-strict.  State-dict key names of pbnet_amd.network.PBNet follow the reference's
module tree and MinkowskiEngine's leaf names (tests/test_checkpoint.py), so a released PBNet checkpoint loads by name."""
import glob
import os

import torch


def checkpoint_restore(model, optimizer, logpath, epoch=0, dist=False, pretrain_file='', gpu=0):
    """tools/log.py:64-103.  Returns (epoch to start from, file that was loaded or '')."""
    if not pretrain_file:
        if epoch > 0:
            pretrain_file = logpath + '%09d' % epoch + '.pth'
            assert os.path.isfile(pretrain_file), pretrain_file
        else:                                    # newest file by name = highest epoch (zero-padded numbers)
            found = sorted(glob.glob(logpath + '*.pth'))
            pretrain_file = found[-1] if found else ''
    if not pretrain_file:
        return epoch + 1, ''
    target = None
    for p in model.parameters():
        target = p.device
        break
    # the reference remaps 'cuda:0' -> 'cuda:<gpu>' (:76); here tensors are loaded on the host and land on the devices
    # the model / optimizer state already live on
    checkpoint = torch.load(pretrain_file, map_location='cpu')
    model_dict = checkpoint['model']
    epoch = int(os.path.basename(pretrain_file).split('.')[0])
    if any(k.startswith('module.') for k in list(model_dict)[:1]):
        model_dict = {k[len('module.'):]: v for k, v in model_dict.items()}
    (model.module if dist else model).load_state_dict(model_dict, strict=False)
    if optimizer is not None:
        optimizer.load_state_dict(checkpoint['optimizer'])
        for state in optimizer.state.values():   # :93-97 moves every state tensor with .cuda(); follow the parameters instead
            if state is None:
                continue
            for k, v in state.items():
                if torch.is_tensor(v) and target is not None:
                    state[k] = v.to(target)
    if dist:
        torch.distributed.barrier()
    return epoch + 1, pretrain_file


def checkpoint_save(model, optimizer, logpath, epoch, save_freq=1):
    """tools/log.py:106-116: write this epoch's file, then delete the previous epoch's unless it is a multiple of
    `save_freq`."""
    pretrain_file = logpath + '%09d' % epoch + '.pth'
    directory = os.path.dirname(pretrain_file)
    if directory:
        os.makedirs(directory, exist_ok=True)
    torch.save({'model': model.state_dict(), 'optimizer': optimizer.state_dict()}, pretrain_file)
    previous = logpath + '%09d' % (epoch - 1) + '.pth'
    if os.path.isfile(previous) and (epoch - 1) % save_freq != 0:
        os.remove(previous)
    return pretrain_file
