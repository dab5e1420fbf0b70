"""Train-mode MinkUNet body on the native executor (csrc/train_exec.hip: pbn_unet_train_forward / _backward).

The body = every layer of /root/reference/network/Mink.py:291-350 but the final 1x1 convolution: stem, four k2s2 down
convolutions, the residual stages, four transposed convolutions with their skip concatenations.  The module path issues it as
~330 native calls per network and direction from Python (MinkowskiEngine/fused_train.py) and the configs[2] step was
host-bound on exactly that; here the whole body is ONE autograd node whose forward and backward are one C call each over a
static plan (the train-mode twin of MinkUNet._build_plan):

  * plan   : ops (convolution -> batch norm with its tail) over symbolic buffers; skip concatenations are written in place
             (encoder output -> right-hand columns of the decoder's slab, transposed convolution -> left-hand columns);
             which input gradients are accumulated in the convolution epilogue (`dx_accumulate`) is decided here, statically,
             by walking the ops backwards;
  * arenas : activations (kept for the backward) and their gradients share one layout (pbn_unet_arena_bytes);
  * grads  : one flat fp32 buffer for all kernel / gamma / beta gradients, handed to autograd as views.

Same kernels, same order and the same arithmetic as the module path (tests/test_train_engine_gpu.py compares the two)."""
import ctypes
import weakref
import os

import torch

from .. import _native as N
from ..MinkowskiEngine import conv as C
from ..MinkowskiEngine.nn import _bn_workspace
from ..MinkowskiEngine.core import SparseTensor

ENABLED = os.environ.get("PBN_TRAIN_ENGINE", "1") == "1"     # "0": the module path (one autograd node per block)
# PBN_TRAIN_SORTED=1: the body runs on the lineage in Z-order (as the fused inference path does: a 128-row tile is a compact
# piece of space, the lineage comes from the one-call pbn_coords_prepare); rows are permuted on the way in (with the padding
# pass) and out.  Off by default: on the bench scene, whose points arrive spatially coherent as ScanNet's mesh vertices do,
# the step time is the same within noise (33.96 vs 34.36 ms), and in the caller's row order the arithmetic is, sum for sum,
# the module path's -- which is what the exactness tests pin.
SORTED = os.environ.get("PBN_TRAIN_SORTED", "0") == "1"
# a list while a caller collects what one step's bodies compute (scripts/train_step.py: the step's roofline figure); None otherwise
ACCOUNTING = None
_DOWN = ("conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2")
_DOWN_BN = ("bn1", "bn2", "bn3", "bn4")
_UP = ("convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2")
_UP_BN = ("bntr4", "bntr5", "bntr6", "bntr7")
INIT_DIM = 32
_K_OF = {0: 1, 1: 27, 2: 125, 3: 8, 4: 8}


def _pair_slot(map_kind, level_in, level_out):
    """include/pbnet_hip.h: 0..4 k3 of levels 0..4, 5 k5, 6..9 down of fine levels, 10..13 up of fine levels."""
    return {1: level_out, 2: 5, 3: 6 + level_in, 4: 10 + level_out}[map_kind]


class _Covered(object):
    """Which column ranges of which gradient buffers already hold a contribution (the backward walk of the planner)."""

    def __init__(self):
        self.ranges = {}

    def state(self, buf, c0, c1):
        """'full' / 'none' for a view; anything in between is not something the executor can express."""
        hit = [(a, b) for a, b in self.ranges.get(buf, ()) if a < c1 and b > c0]
        if not hit:
            return "none"
        if sum(min(b, c1) - max(a, c0) for a, b in hit) == c1 - c0:      # marked ranges never overlap
            return "full"
        raise NotImplementedError("partially initialised gradient view (buffer %d, columns %d:%d)" % (buf, c0, c1))

    def mark(self, buf, c0, c1):
        if self.state(buf, c0, c1) == "none":
            self.ranges.setdefault(buf, []).append((c0, c1))


class TrainPlan(object):
    """The static part of one (network, dtype, input-gradient wanted) combination."""

    def __init__(self, net, dtype, want_input_grad):
        self.dtype, self.want_input_grad = dtype, want_input_grad
        P = net.PLANES
        bufs = [(0, 0)]
        recs = []                   # (conv, norm, map_kind, lin, lout, in view, pre, res view, out view, relu)
        skip_c = (INIT_DIM, P[0], P[1], P[2])
        up_c = (P[7], P[6], P[5], P[4])

        def new_buf(level, width):
            bufs.append((level, width))
            return len(bufs) - 1

        def add(conv, norm, src, map_kind, lin, lout, relu=True, res=None, out=None):
            cout = int(conv.kernel.shape[-1])
            pre = new_buf(lout, cout)
            if out is None:
                out = (new_buf(lout, cout), 0)
            recs.append((conv, norm, map_kind, lin, lout, src, pre, res, out, relu))
            return out

        def stage(blocks, cur, l, out=None):
            for bi, blk in enumerate(blocks):
                h = add(blk.conv1, blk.norm1, cur, 1, l, l)
                res = cur
                if blk.downsample is not None:
                    res = add(blk.downsample[0], blk.downsample[1], cur, 0, l, l, relu=False)
                cur = add(blk.conv2, blk.norm2, h, 1, l, l, relu=True, res=res, out=out if bi == len(blocks) - 1 else None)
            return cur

        slab = [new_buf(l, up_c[l] + skip_c[l]) for l in range(4)]
        cur = add(net.conv0p1s1, net.bn0, (0, 0), 2, 0, 0, out=(slab[0], up_c[0]))
        l = 0
        for i in range(4):
            cur = add(getattr(net, _DOWN[i]), getattr(net, _DOWN_BN[i]), cur, 3, l, l + 1)
            l += 1
            cur = stage(getattr(net, "block%d" % (i + 1)), cur, l, out=(slab[l], up_c[l]) if l < 4 else None)
        for i in range(4):
            add(getattr(net, _UP[i]), getattr(net, _UP_BN[i]), cur, 4, l, l - 1, out=(slab[l - 1], 0))
            l -= 1
            cur = stage(getattr(net, "block%d" % (i + 5)), (slab[l], 0), l)
        self.out_view = cur
        self.out_channels = int(recs[-1][0].kernel.shape[-1])
        e = C._ELEMS[dtype]
        cin0 = int(net.conv0p1s1.kernel.shape[-2])
        self.cin = cin0
        self.cin_p = C._vpo(cin0, dtype) * e
        self.dinput_width = (cin0 + 15) // 16 * 16
        bufs[0] = (0, self.cin_p)
        self.bufs = bufs
        self.recs = recs
        # the backward walk: who writes a gradient view first, who adds to it
        cov = _Covered()
        cov.mark(self.out_view[0], self.out_view[1], self.out_view[1] + self.out_channels)
        flags = [None] * len(recs)
        for i in range(len(recs) - 1, -1, -1):
            conv, norm, mk, lin, lout, src, pre, res, out, relu = recs[i]
            cin, cout = int(conv.kernel.shape[-2]), int(conv.kernel.shape[-1])
            if cov.state(out[0], out[1], out[1] + cout) != "full":
                raise NotImplementedError("op %d: its output has no consumer" % i)
            if res is not None:
                if not relu or cov.state(res[0], res[1], res[1] + cout) != "none":
                    raise NotImplementedError("op %d: the residual gradient must be the first contribution of its buffer" % i)
                cov.mark(res[0], res[1], res[1] + cout)
            want_dx = src[0] != 0 or want_input_grad
            width = cin if src[0] != 0 else self.dinput_width
            acc = False
            if want_dx:
                acc = cov.state(src[0], src[1], src[1] + width) == "full"
                cov.mark(src[0], src[1], src[1] + width)
            flags[i] = (want_dx, acc)
        # parameter-gradient and statistics layouts
        self.params = []
        off_g = off_s = 0
        self.layout = []
        for conv, norm, *_ in recs:
            k3 = conv.kernel if conv.kernel.dim() == 3 else conv.kernel.unsqueeze(0)
            nk, cin, cout = int(k3.shape[0]), int(k3.shape[1]), int(k3.shape[2])
            dw, dg, db = off_g, off_g + nk * cin * cout, off_g + nk * cin * cout + cout
            off_g = db + cout
            self.layout.append((dw, dg, db, off_s))
            off_s += 2 * cout
            self.params += [conv.kernel, norm.bn.weight, norm.bn.bias]
        self.grad_floats, self.stat_floats = off_g, off_s
        self.split_sizes = []
        for conv, norm, *_ in recs:
            self.split_sizes += [conv.kernel.numel(), norm.bn.weight.numel(), norm.bn.bias.numel()]
        self.norms = [r[1] for r in recs]
        self.max_channels = max(int(r[0].kernel.shape[-1]) for r in recs)
        self.wgrad_ws_bytes = max(int(N.lib().pbn_spconv_wgrad_workspace_bytes(_K_OF[r[2]], int(r[0].kernel.shape[-2]),
                                                                                int(r[0].kernel.shape[-1]))) for r in recs)
        self.pair_slots = sorted({_pair_slot(r[2], r[3], r[4]) for r in recs if r[2] != 0})
        # the ctypes image: everything but the packed-weight pointers is fixed
        ops = (N.TrainOp * len(recs))()
        for i, (conv, norm, mk, lin, lout, src, pre, res, out, relu) in enumerate(recs):
            o, bn = ops[i], norm.bn
            o.map_kind, o.level_in, o.level_out = mk, lin, lout
            o.in_buf, o.in_col = src
            o.pre_buf = pre
            o.res_buf, o.res_col = res if res is not None else (-1, 0)
            o.out_buf, o.out_col = out
            o.relu = int(relu)
            o.cin, o.cout = int(conv.kernel.shape[-2]), int(conv.kernel.shape[-1])
            o.want_dx, o.dx_accumulate = int(flags[i][0]), int(flags[i][1])
            o.gamma, o.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
            if bn.track_running_stats:
                o.running_mean, o.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            o.eps, o.momentum = float(bn.eps), float(bn.momentum)
            o.dw_off, o.dgamma_off, o.dbeta_off, o.stat_off = self.layout[i]
        self.ops = ops
        self.bufs_arr = (N.UnetBuf * len(bufs))(*[N.UnetBuf(lv, w) for lv, w in bufs])
        self.packed = [None] * (2 * len(recs))        # the packed tensors behind ops[i].w / w_d (identity = still current)
        self.pointer_key = tuple(p.data_ptr() for p in self.params)

    def refresh_weights(self):
        """Packed forward / input-gradient weights of this step: the first stale layer repacks every layer in one launch
        (conv._BatchPacker); the pointers only move when a buffer had to be re-allocated."""
        dt, ops = self.dtype, self.ops
        for i, rec in enumerate(self.recs):
            conv = rec[0]
            cache, kernel = conv._cache, conv.kernel
            ver = kernel._version
            hit = cache.store.get(("f", dt))
            pk = cache._note_stream(hit) if (hit is not None and hit[0][1] == ver) else cache.get(kernel, dt)
            if pk[0] is not self.packed[2 * i]:
                self.packed[2 * i] = pk[0]
                ops[i].w, ops[i].vpo, ops[i].n_steps, ops[i].cout_p = pk[0].data_ptr(), pk[1], pk[2], pk[3]
            if ops[i].want_dx:
                hit = cache.store.get(("d", dt))
                pk = cache._note_stream(hit) if (hit is not None and hit[0][1] == ver) else cache.get_dgrad(kernel, dt, conv._dgrad_flip)
                if pk[0] is not self.packed[2 * i + 1]:
                    self.packed[2 * i + 1] = pk[0]
                    ops[i].w_d, ops[i].vpo_d, ops[i].n_steps_d, ops[i].cout_p_d = pk[0].data_ptr(), pk[1], pk[2], pk[3]


def usable(net, x):
    f = x.F
    if not (ENABLED and net.training and torch.is_grad_enabled() and f.is_cuda and f.dtype in C._DT and x.tensor_stride == 1):
        return False
    ok = net.__dict__.get("_train_engine_ok")
    if ok is None:
        from ..MinkowskiEngine import MinkowskiBatchNorm, MinkowskiConvolution, MinkowskiConvolutionTranspose
        ok = True
        for name, m in net.named_modules():
            if isinstance(m, (MinkowskiConvolution, MinkowskiConvolutionTranspose)) and name != "final_sematic":
                ok = ok and m.bias is None and m.kernel.dtype == torch.float32 and int(m.kernel.shape[-1]) % 16 == 0
            if isinstance(m, MinkowskiBatchNorm):
                bn = m.bn
                ok = ok and m.NATIVE_TRAIN and m.FUSE_ACT and bn.momentum is not None and bn.weight is not None \
                    and bn.weight.dtype == torch.float32
        net.__dict__["_train_engine_ok"] = bool(ok)
    if not ok:
        return False
    norms = net.__dict__.get("_train_engine_norms")
    if norms is None:
        from ..MinkowskiEngine import MinkowskiBatchNorm
        norms = [m.bn for name, m in net.named_modules() if isinstance(m, MinkowskiBatchNorm)]
        net.__dict__["_train_engine_norms"] = norms
    return all(bn.training for bn in norms)                     # a frozen (eval-mode) batch norm anywhere: the module path


def _plan(net, dtype, want_input_grad):
    plans = net.__dict__.setdefault("_train_plans", {})
    key = (dtype, bool(want_input_grad))
    plan = plans.get(key)
    if plan is not None and plan.pointer_key != tuple(p.data_ptr() for p in plan.params):
        plan = None                                               # parameters were re-allocated (.to(), load_state_dict(assign=True))
    if plan is None:
        plan = TrainPlan(net, dtype, want_input_grad)
        plans[key] = plan
    return plan


def _pair_array(pyr, plan):
    """pbn_pair_lists[14] of a pyramid (the lists are cached on the map tensors: every layer and every network shares them)."""
    cache = pyr.__dict__.setdefault("_pair_arrays", {})
    key = tuple(plan.pair_slots)
    hit = cache.get(key)
    if hit is None:
        arr = (N.PairLists * 14)()
        keep = []
        maps = {}
        for slot in plan.pair_slots:
            if slot < 5:
                maps[slot] = pyr.kernel_map(1 << slot, 3)
            elif slot == 5:
                maps[slot] = pyr.kernel_map(1, 5)
            elif slot < 10:
                maps[slot] = pyr.down_map(1 << (slot - 6))
            else:
                maps[slot] = pyr.up_map(2 << (slot - 10))
        C.rulebook_pairs_dev_multi([maps[sl] for sl in plan.pair_slots])       # all lists of the lineage in three launches
        for slot in plan.pair_slots:
            nbr = maps[slot]
            in_idx, out_idx, seg_begin, counts = C.rulebook_pairs_dev(nbr)
            k = int(nbr.shape[1])
            p = arr[slot]
            p.in_idx, p.out_idx, p.seg_begin, p.counts = in_idx.data_ptr(), out_idx.data_ptr(), seg_begin.data_ptr(), counts.data_ptr()
            p.segment = C.WGRAD_PAIR_SEGMENT
            p.n_pairs_estimate = max(C.WGRAD_PAIR_SEGMENT, (int(nbr.shape[0]) * k) // (2 if k >= 27 else 4))
            keep.append((in_idx, out_idx, seg_begin, counts))
        hit = (arr, keep)
        cache[key] = hit
    return hit[0]


class _State(object):
    """What the backward needs from the forward (attributes of the autograd context)."""
    __slots__ = ("plan", "pyr", "padded", "arena", "stats", "offs", "nbytes", "n_rows", "tables", "perm", "inv_perm",
                 "param_versions", "out_ref", "out_version")


class _BodyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, net, pyr, perm, inv_perm, *params):
        want_dx = bool(feats.requires_grad)
        dt, dev = feats.dtype, feats.device
        plan = _plan(net, dt, want_dx)
        plan.refresh_weights()
        lib = N.lib()
        es = feats.element_size()
        n = int(feats.shape[0])
        cin_p = plan.cin_p
        if perm is None and feats.shape[1] == cin_p and feats.stride(1) == 1 and (feats.stride(0) * es) % 16 == 0 \
                and feats.data_ptr() % 16 == 0:
            padded = feats
        elif feats.stride(1) == 1 and (feats.shape[1] * es) % 4 == 0 and (feats.stride(0) * es) % 4 == 0 and feats.data_ptr() % 4 == 0:
            padded = torch.empty(n, cin_p, dtype=dt, device=dev)
            N.check(lib.pbn_gather_pad_rows(ctypes.c_void_p(feats.data_ptr()), feats.stride(0) * es, feats.shape[1] * es,
                                            None if perm is None else ctypes.c_void_p(perm.data_ptr()), n,
                                            ctypes.c_void_p(padded.data_ptr()), cin_p * es, N.current_stream()), "pbn_gather_pad_rows")
        else:
            padded = torch.zeros(n, cin_p, dtype=dt, device=dev)
            padded[:, :feats.shape[1]] = feats if perm is None else feats[perm]
        rows = list(pyr.n)
        n_rows = (ctypes.c_int32 * 5)(*rows)
        offs = (ctypes.c_int64 * len(plan.bufs))()
        nbytes = lib.pbn_unet_arena_bytes(plan.bufs_arr, len(plan.bufs), n_rows, C._DT[dt], offs)
        arena = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        stats = torch.empty(plan.stat_floats, dtype=torch.float32, device=dev)
        k3, k5, down, up = pyr.native_tables()
        vp = ctypes.c_void_p
        tables = ((vp * 5)(*k3), vp(k5), (vp * 4)(*down), (vp * 4)(*up))
        ws, bws = C._workspace(dev), _bn_workspace(dev, plan.max_channels)
        for nm in plan.norms:
            nm._tick()
        N.check(lib.pbn_unet_train_forward(plan.ops, len(plan.recs), plan.bufs_arr, len(plan.bufs), n_rows, vp(padded.data_ptr()),
                                           padded.stride(0), tables[0], tables[1], tables[2], tables[3], vp(arena.data_ptr()),
                                           nbytes, vp(stats.data_ptr()), C._DT[dt], vp(ws.data_ptr()), ws.numel(),
                                           vp(bws.data_ptr()), bws.numel(), N.current_stream()), "pbn_unet_train_forward")
        if ACCOUNTING is not None:
            ACCOUNTING.append((plan, pyr, es))
        st = _State()
        st.plan, st.pyr, st.padded, st.arena, st.stats, st.offs, st.nbytes, st.n_rows, st.tables = \
            plan, pyr, padded, arena, stats, offs, nbytes, n_rows, tables
        st.perm, st.inv_perm = perm, inv_perm
        # the backward re-reads the activation arena (ReLU masks, batch-norm inputs) and the kernels behind the packed weights:
        # autograd cannot see either (the arena is not a saved tensor), so their versions are checked by hand in backward()
        st.param_versions = [int(p._version) for p in plan.params]
        ctx.state = st
        ctx.in_shape, ctx.in_dtype = (n, int(feats.shape[1])), feats.dtype
        ob, oc = plan.out_view
        width = plan.bufs[ob][1]
        out = arena[offs[ob]:offs[ob] + rows[0] * width * es].view(dt).view(rows[0], width)
        out = out[:, oc:oc + plan.out_channels] if (oc or width != plan.out_channels) else out
        out = out if inv_perm is None else out.index_select(0, inv_perm)       # external row i = stored row inv_perm[i]
        st.out_ref, st.out_version = weakref.ref(out), int(out._version)       # an in-place op on the output would corrupt the arena
        return out

    @staticmethod
    def backward(ctx, dout):
        st = ctx.state
        if st is None:
            raise RuntimeError("native training executor: backward ran twice through the same forward (retain_graph=True is not "
                               "supported: the activation arena is released after the first backward; PBN_TRAIN_ENGINE=0 runs the modules)")
        o_ = st.out_ref()
        if o_ is not None and int(o_._version) != st.out_version:
            raise RuntimeError("native training executor: the body's output was modified in place between forward and backward "
                               "(it aliases the activation arena the backward reads)")
        plan, pyr = st.plan, st.pyr
        if [int(p._version) for p in plan.params] != st.param_versions:
            raise RuntimeError("native training executor: a parameter changed between forward and backward (an optimizer step or an "
                               "in-place update): the packed weights of the forward no longer match")
        dt, dev = plan.dtype, dout.device
        lib = N.lib()
        es = torch.empty(0, dtype=dt).element_size()
        rows = list(pyr.n)
        garena = torch.empty(max(st.nbytes, 16), dtype=torch.uint8, device=dev)
        ob, oc = plan.out_view
        width = plan.bufs[ob][1]
        gout = garena[st.offs[ob]:st.offs[ob] + rows[0] * width * es].view(dt).view(rows[0], width)
        gout[:, oc:oc + plan.out_channels].copy_(dout if st.perm is None else dout.index_select(0, st.perm))
        pgrads = torch.empty(plan.grad_floats, dtype=torch.float32, device=dev)
        dinput = torch.empty(rows[0], plan.dinput_width, dtype=dt, device=dev) if plan.want_input_grad else None
        pairs = _pair_array(pyr, plan)
        vp = ctypes.c_void_p
        ws, bws = C._workspace(dev), _bn_workspace(dev, plan.max_channels)
        wws = C._WGRAD_WS.get(dev, plan.wgrad_ws_bytes)
        t = st.tables
        N.check(lib.pbn_unet_train_backward(
            plan.ops, len(plan.recs), plan.bufs_arr, len(plan.bufs), st.n_rows, vp(st.padded.data_ptr()), st.padded.stride(0),
            t[0], t[1], t[2], t[3], pairs, vp(st.arena.data_ptr()), vp(garena.data_ptr()), st.nbytes, vp(st.stats.data_ptr()),
            vp(pgrads.data_ptr()), None if dinput is None else vp(dinput.data_ptr()), plan.dinput_width, C._DT[dt],
            vp(ws.data_ptr()), ws.numel(), vp(bws.data_ptr()), bws.numel(), vp(wws.data_ptr()), wws.numel(), N.current_stream()),
            "pbn_unet_train_backward")
        ctx.state = None
        grads = [g.view_as(p) for g, p in zip(pgrads.split(plan.split_sizes), plan.params)]
        dfeats = None
        if dinput is not None:
            dfeats = dinput[:, :ctx.in_shape[1]]
            if st.inv_perm is not None:
                dfeats = dfeats.index_select(0, st.inv_perm)
            if dfeats.dtype != ctx.in_dtype:
                dfeats = dfeats.to(ctx.in_dtype)
        return (dfeats, None, None, None, None) + tuple(grads)


def forward_body(net, x):
    """The body's output (a SparseTensor over the input's coordinates) on the native executor; None when it does not apply
    (the caller then runs the modules)."""
    if not usable(net, x):
        return None
    cm = x.coordinate_manager
    if SORTED:
        sv = cm.sorted()
        pyr, perm, inv_perm = sv.pyramid, sv.perm, sv.inv_perm
    else:
        pyr, perm, inv_perm = cm.plain(), None, None
    if min(pyr.n) <= 0:
        return None
    feats = x.F
    plan = _plan(net, feats.dtype, bool(feats.requires_grad))
    out = _BodyFn.apply(feats, net, pyr, perm, inv_perm, *plan.params)
    return SparseTensor(out, coordinate_manager=cm, tensor_stride=1)


def step_accounting(records):
    """Algorithmic flops and bytes of the bodies recorded in ACCOUNTING (one entry per network and forward): per op the forward
    convolution, its input gradient (where the plan computes one) and its weight gradient -- 2 x pairs x C_in x C_out each, bytes as
    SURVEY 8d counts them (input rows + output rows + kernel + 8 bytes per rule pair; the weight gradient writes its kernel in
    fp32) -- and the batch norm's passes over the op's output slab (forward: statistics, apply [+ residual]; backward: two passes
    over x, dy, y and the write of dx [+ the residual gradient]).  Reads the pair counts back: call it outside the timed region."""
    flops = {"forward": 0, "dgrad": 0, "wgrad": 0}
    nbytes = {"forward": 0, "dgrad": 0, "wgrad": 0, "batch_norm": 0}
    n_ops = 0
    for plan, pyr, es in records:
        rows = list(pyr.n)
        pair_cache = {}
        for i, (conv, norm, mk, lin, lout, src, pre, res, out, relu) in enumerate(plan.recs):
            cin, cout = int(conv.kernel.shape[-2]), int(conv.kernel.shape[-1])
            k = _K_OF[mk]
            v_in, v_out = rows[lin], rows[lout]
            if mk == 0:
                pairs = v_out
            else:
                slot = _pair_slot(mk, lin, lout)
                if slot not in pair_cache:
                    if slot < 5:
                        nbr = pyr.kernel_map(1 << slot, 3)
                    elif slot == 5:
                        nbr = pyr.kernel_map(1, 5)
                    elif slot < 10:
                        nbr = pyr.down_map(1 << (slot - 6))
                    else:
                        nbr = pyr.up_map(2 << (slot - 10))
                    pair_cache[slot] = int(C.rulebook_pairs_dev(nbr)[3].sum().item())
                pairs = pair_cache[slot]
            f = 2 * pairs * cin * cout
            io = (v_in * cin + v_out * cout) * es
            maps = 8 * pairs if mk else 0
            flops["forward"] += f
            nbytes["forward"] += io + k * cin * cout * es + maps
            if plan.ops[i].want_dx:
                flops["dgrad"] += f
                nbytes["dgrad"] += io + k * cin * cout * es + maps
            flops["wgrad"] += f
            nbytes["wgrad"] += io + maps + k * cin * cout * 4
            slab = v_out * cout * es
            nbytes["batch_norm"] += (3 + (1 if res is not None else 0)) * slab + (7 + (1 if res is not None else 0)) * slab
            n_ops += 1
    return {"ops": n_ops, "flops": flops, "bytes": nbytes, "flops_total": sum(flops.values()), "bytes_total": sum(nbytes.values())}
