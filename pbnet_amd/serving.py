"""Serving front of the inference path (round 6): the scenes WAITING on a GPU are merged into one `PBNet.forward` through the
reference's own batch axis, and the results are handed back per scene.

Why: the coarse levels of the three U-Nets (stride 4 / 8 / 16: 93 of the 138 convolution launches of a forward) are launches of a
few hundred to a few thousand rows -- latency chains that one scene cannot fill; B scenes in one forward give
every launch B x the rows for the same number of launches.  The reference already batches this way: `dataset_preprocess.py:296`
collates with a batch index, `network/PBNet.py:167-176` groups per (class, batch element), its evaluation feeds the three
test-time copies of a scene as a batch of 3 (`eval_map.py:48-50`, `dataset_preprocess.py:324`).  `tests/test_batched_gpu.py` pins
the property this module rests on: a merged forward returns, for every scene, the proposals and scores of that scene's own forward.

    server = SceneServer(model, max_batch=4, forwards_in_flight=2)
    fut = server.submit(scene)           # scene: dict(xyz_voxel [V,4] int32, feat_voxel [V,C], xyz_original [N,3], v2p_index [N]) on the GPU
    res = fut.result()                   # dict(sem_pred_p [N], proposals (idx [M,2], offset [P+1]), clt_scores [P]) of THAT scene
    server.close()

`merge_scenes` / `split_results` are the two pure functions; `SceneServer` is the small scheduler around them: F worker threads,
each with its own HIP stream (create the server FIRST in a process: the runtime maps a process's first streams to distinct
hardware queues), each taking up to `max_batch` waiting scenes per forward -- it never waits for a batch to fill:
a lone scene is served alone.
"""
import queue
import threading
from concurrent.futures import Future

import torch


def merge_scenes(scenes, teachers=None):
    """B scenes -> one batch: scene j becomes batch element j (the batch column of its voxel coordinates is overwritten), voxel
    and point arrays are concatenated in scene order, `v2p_index` is shifted by the voxels in front.  Returns (batch, teacher or
    None, point_starts [B + 1] -- the rows of xyz_original / sem_pred_p that belong to scene j are point_starts[j] : [j + 1])."""
    if len(scenes) == 1:                                  # a lone scene: nothing to concatenate
        s = scenes[0]
        xv = s["xyz_voxel"].clone()
        xv[:, 0] = 0
        return ({"xyz_voxel": xv, "feat_voxel": s["feat_voxel"], "xyz_original": s["xyz_original"], "v2p_index": s["v2p_index"]},
                None if teachers is None else teachers[0], [0, int(s["xyz_original"].shape[0])])
    vox, feat, xyz, v2p, starts = [], [], [], [], [0]
    nv = 0
    for j, s in enumerate(scenes):
        xv = s["xyz_voxel"].clone()
        xv[:, 0] = j
        vox.append(xv)
        feat.append(s["feat_voxel"])
        xyz.append(s["xyz_original"])
        v2p.append(s["v2p_index"] + nv)
        nv += int(xv.shape[0])
        starts.append(starts[-1] + int(s["xyz_original"].shape[0]))
    batch = {"xyz_voxel": torch.cat(vox), "feat_voxel": torch.cat(feat), "xyz_original": torch.cat(xyz), "v2p_index": torch.cat(v2p)}
    teacher = None
    if teachers is not None and teachers[0] is not None:
        teacher = {k: torch.cat([t[k] for t in teachers]) for k in teachers[0]}
    return batch, teacher, starts


def split_results(ret, point_starts):
    """The merged forward's results -> one result per scene, in the reference's own output form (`proposals_idx` rows are
    (proposal, point) with the proposal numbered from 0 and the point index local to the scene; `proposals_offset` starts at 0).
    A proposal belongs to the scene its points lie in (a local scene never crosses batch elements: PBNet.py:167-176).
    One small read-back (proposals and rows per scene); everything else stays on the device."""
    n_scenes = len(point_starts) - 1
    if n_scenes == 1:
        out = {"sem_pred_p": ret["sem_pred_p"]}
        if "proposals" in ret:
            out["proposals"], out["clt_scores"] = (ret["proposals"][0], ret["proposals"][1]), ret["clt_scores"]
        return [out]
    dev = ret["sem_pred_p"].device
    sem = ret["sem_pred_p"]
    out = [{"sem_pred_p": sem[point_starts[j]:point_starts[j + 1]]} for j in range(n_scenes)]
    if "proposals" not in ret:
        return out
    idx, off = ret["proposals"][0], ret["proposals"][1]
    scores = ret["clt_scores"]
    n_prop = int(off.shape[0]) - 1
    if n_prop <= 0:
        for o in out:
            o["proposals"] = (idx[:0], off[:1].clone())
            o["clt_scores"] = scores[:0]
        return out
    starts_d = torch.tensor(point_starts, dtype=idx.dtype, device=dev)
    off = off.to(torch.int64)
    sizes = off[1:] - off[:-1]
    first_pt = idx[off[:-1], 1]
    scene = torch.searchsorted(starts_d, first_pt, right=True) - 1                  # [P]
    order = torch.sort(scene, stable=True)[1]                                       # proposals grouped by scene, their order kept
    scene_s, sizes_s = scene[order], sizes[order]
    new_off = torch.zeros(n_prop + 1, dtype=torch.int64, device=dev)
    new_off[1:] = torch.cumsum(sizes_s, 0)
    # rows of proposal order[q] move to new_off[q] .. : source row of every destination row
    rep = torch.repeat_interleave(torch.arange(n_prop, device=dev), sizes_s, output_size=int(idx.shape[0]))   # destination row -> its (new) proposal
    src = off[order][rep] + (torch.arange(rep.shape[0], device=dev) - new_off[:-1][rep])
    rows = idx[src]
    pts_local = rows[:, 1] - starts_d[scene_s][rep]
    props = torch.bincount(scene_s, minlength=n_scenes)                             # proposals per scene
    nrows = torch.bincount(scene_s, weights=sizes_s.double(), minlength=n_scenes).long()
    counts = torch.stack([props, nrows]).cpu().tolist()                             # the one read-back
    scores_s = scores[order]
    p0 = r0 = 0
    for j in range(n_scenes):
        p1, r1 = p0 + int(counts[0][j]), r0 + int(counts[1][j])
        pid = rep[r0:r1] - p0
        out[j]["proposals"] = (torch.stack([pid.to(idx.dtype), pts_local[r0:r1]], 1), (new_off[p0:p1 + 1] - r0).to(ret["proposals"][1].dtype))
        out[j]["clt_scores"] = scores_s[p0:p1]
        p0, r0 = p1, r1
    return out


class SceneServer(object):
    """F forwards in flight x up to B scenes per forward.  `submit` returns a Future; `close` drains the queue."""

    def __init__(self, model, max_batch=4, forwards_in_flight=2, device=None, epoch=1, split=True, streams=None):
        """streams: the HIP streams of the workers (default: `forwards_in_flight` new ones; a process that already owns its
        in-flight streams passes them: streams created later can share a hardware queue -- DESIGN.md section 5, round 5 item 6b)."""
        self.model, self.max_batch, self.epoch, self.split = model, int(max_batch), epoch, split
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._q = queue.Queue()
        self._closed = False
        self.forwards = 0              # merged forwards run so far
        self.scenes = 0                # scenes served so far
        self._lock = threading.Lock()
        self._streams = list(streams) if streams is not None else [torch.cuda.Stream(self.device) for _ in range(int(forwards_in_flight))]
        self._threads = [threading.Thread(target=self._worker, args=(st,), daemon=True) for st in self._streams]
        for t in self._threads:
            t.start()

    def submit(self, scene, teacher=None):
        if self._closed:
            raise RuntimeError("SceneServer is closed")
        f = Future()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))        # the scene's tensors are ready when the submitting stream gets here
        self._q.put((scene, teacher, f, ev))
        return f

    def close(self):
        self._closed = True
        for _ in self._threads:
            self._q.put(None)
        for t in self._threads:
            t.join()

    def _take(self):
        """Block for one scene, then take what else is waiting, up to max_batch; None = shut down."""
        first = self._q.get()
        if first is None:
            return None
        items = [first]
        while len(items) < self.max_batch:
            try:
                nxt = self._q.get_nowait()
            except queue.Empty:
                break
            if nxt is None:                 # a shutdown token meant for some worker: put it back behind this batch
                self._q.put(None)
                break
            items.append(nxt)
        return items

    def _worker(self, stream):
        torch.cuda.set_device(self.device)
        with torch.cuda.stream(stream):
            while True:
                items = self._take()
                if items is None:
                    return
                futs = [it[2] for it in items]
                try:
                    for it in items:
                        stream.wait_event(it[3])
                    batch, teacher, starts = merge_scenes([it[0] for it in items], [it[1] for it in items])
                    with torch.no_grad():
                        ret = self.model(batch["feat_voxel"], batch["xyz_voxel"], batch["xyz_original"], batch["v2p_index"], None,
                                         self.epoch, "test", teacher=teacher, n_batch=len(items))
                    res = split_results(ret, starts) if self.split else [dict(ret, point_starts=starts, scene=j) for j in range(len(items))]
                    stream.synchronize()
                    with self._lock:
                        self.forwards += 1
                        self.scenes += len(items)
                    for f, r in zip(futs, res):
                        f.set_result(r)
                except BaseException as e:      # noqa: BLE001 -- the exception belongs to the callers that wait on the futures
                    for f in futs:
                        if not f.done():
                            f.set_exception(e)
