// train_exec.hip -- the train-mode MinkUNet body forward and backward behind ONE C call each (include/pbnet_hip.h:
// pbn_unet_train_forward / _backward).  Only sequences the kernels of spconv.hip / spconv_wave.hip (convolutions and
// their input gradients), bnorm.hip (batch norm with the block tail) and wgrad.hip; no new arithmetic lives here.
// Mirrors /root/reference/network/Mink.py:291-350 in training mode (MinkowskiEngine's BasicBlock.forward for the stages).
#include "pbn_common.h"

using namespace pbn;

namespace {

inline int esize(int dtype) { return dtype == PBN_F32 ? 4 : 2; }

struct Tables {
    const int32_t* const* k3; const int32_t* k5; const int32_t* const* down; const int32_t* const* up;
};

// forward table of an op (rows = output level) and the table of its input gradient (rows = input level)
inline bool op_maps(const pbn_train_op& o, const Tables& t, const int32_t*& fwd, int& K, const int32_t*& bwd, int& pair_slot) {
    fwd = bwd = nullptr; K = 1; pair_slot = -1;
    switch (o.map_kind) {
        case 0: return true;
        case 1: fwd = bwd = t.k3[o.level_out]; K = 27; pair_slot = o.level_out; break;      // centred cube: the mirrored offsets of the same table
        case 2: fwd = bwd = t.k5; K = 125; pair_slot = 5; break;
        case 3: fwd = t.down[o.level_in]; bwd = t.up[o.level_in]; K = 8; pair_slot = 6 + o.level_in; break;     // k2s2: level_in = fine level
        case 4: fwd = t.up[o.level_out]; bwd = t.down[o.level_out]; K = 8; pair_slot = 10 + o.level_out; break; // transposed: level_out = fine
        default: return false;
    }
    return fwd != nullptr && bwd != nullptr;
}

inline bool op_ok(const pbn_train_op& o, int n_bufs) {
    return o.in_buf >= 0 && o.in_buf < n_bufs && o.pre_buf >= 1 && o.pre_buf < n_bufs && o.out_buf >= 1 && o.out_buf < n_bufs &&
           o.res_buf < n_bufs && o.level_in >= 0 && o.level_in <= 4 && o.level_out >= 0 && o.level_out <= 4 && o.gamma && o.beta &&
           o.cout == o.cout_p && o.w;
}

}  // namespace

extern "C" int pbn_unet_train_forward(const pbn_train_op* ops, int n_ops, const pbn_unet_buf* bufs, int n_bufs,
                                      const int32_t* n_rows, const void* input, int ld_input, const int32_t* const* k3,
                                      const int32_t* k5, const int32_t* const* down, const int32_t* const* up, void* act_arena,
                                      size_t arena_bytes, float* stats, int dtype, void* splitk_ws, size_t splitk_bytes,
                                      void* bn_ws, size_t bn_ws_bytes, pbn_stream_t stream) {
    if (!ops || !bufs || !n_rows || !input || !act_arena || !stats || n_ops < 1 || n_bufs < 2 || n_bufs > 512) return PBN_ERR_ARG;
    int64_t offs[512];
    if (pbn_unet_arena_bytes(bufs, n_bufs, n_rows, dtype, offs) > arena_bytes) return PBN_ERR_WORKSPACE;
    const int es = esize(dtype);
    char* A = (char*)act_arena;
    auto base = [&](int b) -> char* { return b == 0 ? (char*)input : A + offs[b]; };
    auto ld = [&](int b) -> int { return b == 0 ? ld_input : bufs[b].width; };
    const Tables T{k3, k5, down, up};
    for (int i = 0; i < n_ops; ++i) {
        const pbn_train_op& o = ops[i];
        if (!op_ok(o, n_bufs)) return PBN_ERR_ARG;
        const int32_t *fwd, *bwd;
        int K, slot;
        if (!op_maps(o, T, fwd, K, bwd, slot)) return PBN_ERR_ARG;
        const int n_in = n_rows[o.level_in], n_out = n_rows[o.level_out];
        void* pre = base(o.pre_buf);
        int rc = pbn_spconv_forward(base(o.in_buf) + (size_t)o.in_col * es, ld(o.in_buf), n_in, fwd, K, nullptr, nullptr, n_out,
                                    o.w, o.vpo, o.n_steps, o.cout_p, nullptr, nullptr, nullptr, 0, 0, pre, ld(o.pre_buf), dtype,
                                    0, splitk_ws, splitk_bytes, stream);
        if (rc != PBN_OK) return rc;
        const void* res = o.res_buf >= 0 ? base(o.res_buf) + (size_t)o.res_col * es : nullptr;
        rc = pbn_bn_act_train_forward(pre, ld(o.pre_buf), n_out, o.cout, dtype, o.gamma, o.beta, o.eps, o.momentum,
                                      o.running_mean, o.running_var, res, o.res_buf >= 0 ? ld(o.res_buf) : 0, o.relu,
                                      base(o.out_buf) + (size_t)o.out_col * es, ld(o.out_buf), stats + o.stat_off,
                                      stats + o.stat_off + o.cout, bn_ws, bn_ws_bytes, stream);
        if (rc != PBN_OK) return rc;
    }
    return PBN_OK;
}

extern "C" int pbn_unet_train_backward(const pbn_train_op* ops, int n_ops, const pbn_unet_buf* bufs, int n_bufs,
                                       const int32_t* n_rows, const void* input, int ld_input, const int32_t* const* k3,
                                       const int32_t* k5, const int32_t* const* down, const int32_t* const* up,
                                       const pbn_pair_lists* pairs, const void* act_arena, void* grad_arena, size_t arena_bytes,
                                       const float* stats, float* param_grads, void* dinput, int ld_dinput, int dtype,
                                       void* splitk_ws, size_t splitk_bytes, void* bn_ws, size_t bn_ws_bytes, void* wgrad_ws,
                                       size_t wgrad_ws_bytes, pbn_stream_t stream) {
    if (!ops || !bufs || !n_rows || !input || !pairs || !act_arena || !grad_arena || !stats || !param_grads || n_ops < 1 ||
        n_bufs < 2 || n_bufs > 512)
        return PBN_ERR_ARG;
    int64_t offs[512];
    if (pbn_unet_arena_bytes(bufs, n_bufs, n_rows, dtype, offs) > arena_bytes) return PBN_ERR_WORKSPACE;
    const int es = esize(dtype);
    const char* A = (const char*)act_arena;
    char* G = (char*)grad_arena;
    auto act = [&](int b) -> const char* { return b == 0 ? (const char*)input : A + offs[b]; };
    auto grad = [&](int b) -> char* { return b == 0 ? (char*)dinput : G + offs[b]; };
    auto ld = [&](int b) -> int { return b == 0 ? ld_input : bufs[b].width; };
    auto ldg = [&](int b) -> int { return b == 0 ? ld_dinput : bufs[b].width; };
    const Tables T{k3, k5, down, up};
    for (int i = n_ops - 1; i >= 0; --i) {
        const pbn_train_op& o = ops[i];
        if (!op_ok(o, n_bufs)) return PBN_ERR_ARG;
        const int32_t *fwd, *bwd;
        int K, slot;
        if (!op_maps(o, T, fwd, K, bwd, slot)) return PBN_ERR_ARG;
        const int n_in = n_rows[o.level_in], n_out = n_rows[o.level_out];
        // 1. batch norm with its tail: g = d(pre), the masked gradient to the residual branch
        char* gpre = grad(o.pre_buf);
        const char* y = act(o.out_buf) + (size_t)o.out_col * es;
        if (o.res_buf >= 0 && !o.relu) return PBN_ERR_UNSUPPORTED;      // the residual gradient is then dy itself: not on the path
        int rc = pbn_bn_act_train_backward(act(o.pre_buf), ld(o.pre_buf), grad(o.out_buf) + (size_t)o.out_col * es, ldg(o.out_buf),
                                           o.relu ? y : nullptr, o.relu ? ld(o.out_buf) : 0, n_out, o.cout, dtype, o.gamma,
                                           stats + o.stat_off, stats + o.stat_off + o.cout, gpre, ldg(o.pre_buf),
                                           o.res_buf >= 0 ? grad(o.res_buf) + (size_t)o.res_col * es : nullptr,
                                           o.res_buf >= 0 ? ldg(o.res_buf) : 0, param_grads + o.dgamma_off,
                                           param_grads + o.dbeta_off, bn_ws, bn_ws_bytes, stream);
        if (rc != PBN_OK) return rc;
        // 2. input gradient on the convolution kernel (the mirrored / up / down table), accumulated in the epilogue
        if (o.want_dx) {
            if (!o.w_d || (o.in_buf == 0 && !dinput)) return PBN_ERR_ARG;
            char* dx = grad(o.in_buf) + (size_t)o.in_col * es;
            rc = pbn_spconv_forward(gpre, ldg(o.pre_buf), n_out, bwd, K, nullptr, nullptr, n_in, o.w_d, o.vpo_d, o.n_steps_d,
                                    o.cout_p_d, nullptr, nullptr, o.dx_accumulate ? dx : nullptr,
                                    o.dx_accumulate ? ldg(o.in_buf) : 0, 0, dx, ldg(o.in_buf), dtype, 0, splitk_ws, splitk_bytes,
                                    stream);
            if (rc != PBN_OK) return rc;
        }
        // 3. weight gradient over the rule pairs of the forward map
        const void* x = act(o.in_buf) + (size_t)o.in_col * es;
        if (o.map_kind == 0) {
            rc = pbn_spconv_wgrad_checked(x, ld(o.in_buf), n_in, gpre, ldg(o.pre_buf), n_out, dtype, nullptr, nullptr, nullptr,
                                          nullptr, 0, 0, n_out, 1, o.cin, o.cout, param_grads + o.dw_off, wgrad_ws, wgrad_ws_bytes,
                                          stream);
        } else {
            const pbn_pair_lists& P = pairs[slot];
            if (!P.in_idx || !P.out_idx || !P.seg_begin) return PBN_ERR_ARG;
            rc = pbn_spconv_wgrad_checked(x, ld(o.in_buf), n_in, gpre, ldg(o.pre_buf), n_out, dtype, P.in_idx, P.out_idx,
                                          P.seg_begin, P.counts, 0, P.segment, P.n_pairs_estimate, K, o.cin, o.cout,
                                          param_grads + o.dw_off, wgrad_ws, wgrad_ws_bytes, stream);
        }
        if (rc != PBN_OK) return rc;
    }
    return PBN_OK;
}
