// post.hip -- what follows PBNet.forward at evaluation time (/root/reference/eval_map.py:55-123), on the device:
// TTA fold of the proposal member lists, proposal sizes, pairwise mask IoU, per-point label of the picked clusters,
// superpoint vote (tools/getins.py:72-98) and the rebuilt cluster masks.  The reference builds dense [P, N] int masks
// and a [P, P] product with torch.mm; here a proposal is a BITSET over the N/3 folded points (32 points per word), so
// the IoU matrix is popcounts of ANDs and every step is integer work -- bit-exact by construction.
// The greedy NMS itself (tools/mIOU.py:77-87: a few hundred scalars) stays on the host, exactly as the reference runs it.
#include "pbn_common.h"

namespace pbn {
namespace {

constexpr int TPB = 256;

// bit (point % n_fold) of row proposal: eval_map.py:67-70 (the three rotated copies fold onto one index range)
__global__ __launch_bounds__(TPB) void k_set_bits(const long long* __restrict__ proposals_idx, int n_entries, int n_fold,
                                                 int n_prop, int words, unsigned* __restrict__ masks) {
    const int e = blockIdx.x * TPB + threadIdx.x;
    if (e >= n_entries) return;
    const long long p = proposals_idx[2 * (size_t)e + 0];
    const long long pt = proposals_idx[2 * (size_t)e + 1] % n_fold;
    if (p < 0 || p >= n_prop || pt < 0) return;
    atomicOr(&masks[(size_t)p * words + (pt >> 5)], 1u << (pt & 31));
}

// counts[p] = popcount of row p (one wave per row)
__global__ __launch_bounds__(64) void k_row_popcount(const unsigned* __restrict__ masks, int words, int* __restrict__ counts) {
    const unsigned* row = masks + (size_t)blockIdx.x * words;
    int c = 0;
    for (int w = threadIdx.x; w < words; w += 64) c += __popc(row[w]);
    c = wave_reduce_add(c);
    if (threadIdx.x == 0) counts[blockIdx.x] = c;
}

// iou[i][j] = |Mi & Mj| / (|Mi| + |Mj| - |Mi & Mj|) in fp32, the arithmetic of eval_map.py:90-96 on exact integers
__global__ __launch_bounds__(64) void k_mask_iou(const unsigned* __restrict__ masks, const int* __restrict__ rows, int n_rows,
                                                int words, const int* __restrict__ counts, float* __restrict__ iou) {
    const int i = blockIdx.x, j = blockIdx.y;
    const int ri = rows ? rows[i] : i, rj = rows ? rows[j] : j;
    const unsigned* a = masks + (size_t)ri * words;
    const unsigned* b = masks + (size_t)rj * words;
    int c = 0;
    for (int w = threadIdx.x; w < words; w += 64) c += __popc(a[w] & b[w]);
    c = wave_reduce_add(c);
    if (threadIdx.x == 0) {
        const float inter = (float)c;
        iou[(size_t)i * n_rows + j] = inter / (((float)counts[ri] + (float)counts[rj]) - inter);
    }
}

// seg[pt] = the LAST picked cluster that contains the point (eval_map.py:104-107 overwrites in order), else -100
__global__ __launch_bounds__(TPB) void k_point_labels(const unsigned* __restrict__ masks, const int* __restrict__ pick,
                                                     int n_pick, int words, int n_fold, long long* __restrict__ seg) {
    const int pt = blockIdx.x * TPB + threadIdx.x;
    if (pt >= n_fold) return;
    long long lab = -100;
    for (int c = n_pick - 1; c >= 0; --c)
        if ((masks[(size_t)pick[c] * words + (pt >> 5)] >> (pt & 31)) & 1u) { lab = c; break; }
    seg[pt] = lab;
}

// histogram of point labels per superpoint (tools/getins.py:88-92: negative labels go to bucket n_label)
__global__ __launch_bounds__(TPB) void k_sp_hist(const long long* __restrict__ seg, const long long* __restrict__ superpoint,
                                                int n, int n_sp, int n_label, int* __restrict__ hist) {
    const int pt = blockIdx.x * TPB + threadIdx.x;
    if (pt >= n) return;
    const long long sp = superpoint[pt];
    if (sp < 0 || sp >= n_sp) return;
    long long l = seg[pt];
    if (l < 0 || l > n_label) l = n_label;
    atomicAdd(&hist[(size_t)sp * (n_label + 1) + l], 1);
}

// sp_label = first arg-max bucket (np.argmax), bucket n_label -> -100 (tools/getins.py:93-94)
__global__ __launch_bounds__(TPB) void k_sp_argmax(const int* __restrict__ hist, int n_sp, int n_label,
                                                  long long* __restrict__ sp_label) {
    const int sp = blockIdx.x * TPB + threadIdx.x;
    if (sp >= n_sp) return;
    const int* h = hist + (size_t)sp * (n_label + 1);
    int best = h[0], arg = 0;
    for (int l = 1; l <= n_label; ++l)
        if (h[l] > best) { best = h[l]; arg = l; }
    sp_label[sp] = arg == n_label ? -100 : arg;
}

// seg2 = sp_label[superpoint]; cluster bitsets rebuilt from it (eval_map.py:109-116)
__global__ __launch_bounds__(TPB) void k_relabel_points(const long long* __restrict__ sp_label,
                                                       const long long* __restrict__ superpoint, int n, int n_sp,
                                                       int n_label, int words, long long* __restrict__ seg2,
                                                       unsigned* __restrict__ masks_out) {
    const int pt = blockIdx.x * TPB + threadIdx.x;
    if (pt >= n) return;
    const long long sp = superpoint[pt];
    const long long l = (sp >= 0 && sp < n_sp) ? sp_label[sp] : -100;
    seg2[pt] = l;
    if (l >= 0 && l < n_label) atomicOr(&masks_out[(size_t)l * words + (pt >> 5)], 1u << (pt & 31));
}

// dense int32 [rows, n_fold] view of selected bitset rows (the reference's `clusters` tensor)
__global__ __launch_bounds__(TPB) void k_bits_to_dense(const unsigned* __restrict__ masks, const int* __restrict__ rows,
                                                      int n_rows, int words, int n_fold, int* __restrict__ dense) {
    const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
    if (e >= (long long)n_rows * n_fold) return;
    const int r = (int)(e / n_fold), pt = (int)(e - (long long)r * n_fold);
    const int src = rows ? rows[r] : r;
    dense[e] = (masks[(size_t)src * words + (pt >> 5)] >> (pt & 31)) & 1u;
}

// inter[p][gt_index[i]] += 1 for every point i inside prediction p (mask value != 0): the association counts of
// tools/eval.py:230-245 (`count_nonzero(logical_and(gt_ids == id, pred_mask))` per (prediction, instance) pair, one pass
// over the scene per pair in the reference) as ONE pass per prediction.  A block owns OVERLAP_CHUNK consecutive points of
// one prediction; instances are runs of neighbouring vertices, so a wave first merges lanes that hit the same bin
// (leader adds the lane count) before touching the LDS histogram.
constexpr int OVERLAP_CHUNK = 4096;
constexpr int OVERLAP_LDS_BINS = 8192;

__global__ __launch_bounds__(TPB) void k_instance_overlap(const int* __restrict__ masks, int n_pts,
                                                         const int* __restrict__ gt_index, int n_gt, int use_lds,
                                                         int* __restrict__ inter) {
    __shared__ int s_hist[OVERLAP_LDS_BINS];
    const int p = blockIdx.y;
    const int lo = blockIdx.x * OVERLAP_CHUNK;
    const int hi = min(n_pts, lo + OVERLAP_CHUNK);
    int* out = inter + (size_t)p * n_gt;
    if (use_lds) {
        for (int b = threadIdx.x; b < n_gt; b += TPB) s_hist[b] = 0;
        __syncthreads();
    }
    const int* row = masks + (size_t)p * n_pts;
    for (int base = lo; base < hi; base += TPB) {         // uniform trip count: the ballots below need whole waves
        const int i = base + threadIdx.x;
        int bin = -1;
        if (i < hi && row[i] != 0) {
            const int g = gt_index[i];
            if (g >= 0 && g < n_gt) bin = g;
        }
        unsigned long long todo = __ballot(bin >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int lb = __shfl(bin, leader);
            const unsigned long long same = __ballot(bin == lb) & todo;
            if ((int)(threadIdx.x & 63) == leader) {
                const int c = __popcll(same);
                if (use_lds) atomicAdd(&s_hist[lb], c);
                else atomicAdd(&out[lb], c);
            }
            todo &= ~same;
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int b = threadIdx.x; b < n_gt; b += TPB) {
            const int c = s_hist[b];
            if (c) atomicAdd(&out[b], c);
        }
    }
}

}  // namespace
}  // namespace pbn

using namespace pbn;

extern "C" int pbn_post_words(int n_fold) { return n_fold > 0 ? (n_fold + 31) / 32 : 0; }

extern "C" int pbn_proposal_bitmask(const int64_t* proposals_idx, int n_entries, int n_fold, int n_prop, uint32_t* masks,
                                    int32_t* counts, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_entries < 0 || n_fold < 1 || n_prop < 0) return PBN_ERR_ARG;
    if (n_prop == 0) return PBN_OK;
    if (!masks || !counts || (n_entries > 0 && !proposals_idx)) return PBN_ERR_ARG;
    const int words = pbn_post_words(n_fold);
    { const int frc_ = fill_bytes(masks, 0, sizeof(uint32_t) * (size_t)n_prop * words, stream); if (frc_ != PBN_OK) return frc_; }
    if (n_entries > 0)
        hipLaunchKernelGGL(k_set_bits, dim3(cdiv(n_entries, TPB)), dim3(TPB), 0, stream, (const long long*)proposals_idx,
                           n_entries, n_fold, n_prop, words, masks);
    hipLaunchKernelGGL(k_row_popcount, dim3(n_prop), dim3(64), 0, stream, masks, words, counts);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_mask_iou(const uint32_t* masks, const int32_t* rows, int n_rows, int n_fold, const int32_t* counts,
                            float* iou, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || n_fold < 1) return PBN_ERR_ARG;
    if (n_rows == 0) return PBN_OK;
    if (!masks || !counts || !iou) return PBN_ERR_ARG;
    hipLaunchKernelGGL(k_mask_iou, dim3(n_rows, n_rows), dim3(64), 0, stream, masks, rows, n_rows, pbn_post_words(n_fold),
                       counts, iou);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_superpoint_refine(const uint32_t* masks, const int32_t* pick, int n_pick, int n_fold,
                                     const int64_t* superpoint, int n_sp, int64_t* seg, int32_t* hist, int64_t* sp_label,
                                     int64_t* seg_refined, uint32_t* masks_out, int32_t* counts_out, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_pick < 0 || n_fold < 1 || n_sp < 1) return PBN_ERR_ARG;
    if (!masks || !superpoint || !seg || !hist || !sp_label || !seg_refined || (n_pick > 0 && (!pick || !masks_out || !counts_out)))
        return PBN_ERR_ARG;
    const int words = pbn_post_words(n_fold);
    const int nb = cdiv(n_fold, TPB);
    { const int frc_ = fill_bytes(hist, 0, sizeof(int) * (size_t)n_sp * (n_pick + 1), stream); if (frc_ != PBN_OK) return frc_; }
    if (n_pick > 0) { const int frc_ = fill_bytes(masks_out, 0, sizeof(uint32_t) * (size_t)n_pick * words, stream); if (frc_ != PBN_OK) return frc_; }
    hipLaunchKernelGGL(k_point_labels, dim3(nb), dim3(TPB), 0, stream, masks, pick, n_pick, words, n_fold, (long long*)seg);
    hipLaunchKernelGGL(k_sp_hist, dim3(nb), dim3(TPB), 0, stream, (const long long*)seg, (const long long*)superpoint, n_fold,
                       n_sp, n_pick, hist);
    hipLaunchKernelGGL(k_sp_argmax, dim3(cdiv(n_sp, TPB)), dim3(TPB), 0, stream, hist, n_sp, n_pick, (long long*)sp_label);
    hipLaunchKernelGGL(k_relabel_points, dim3(nb), dim3(TPB), 0, stream, (const long long*)sp_label,
                       (const long long*)superpoint, n_fold, n_sp, n_pick, words, (long long*)seg_refined, masks_out);
    if (n_pick > 0) hipLaunchKernelGGL(k_row_popcount, dim3(n_pick), dim3(64), 0, stream, masks_out, words, counts_out);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_bitmask_to_dense(const uint32_t* masks, const int32_t* rows, int n_rows, int n_fold, int32_t* dense,
                                    pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || n_fold < 1) return PBN_ERR_ARG;
    if (n_rows == 0) return PBN_OK;
    if (!masks || !dense) return PBN_ERR_ARG;
    hipLaunchKernelGGL(k_bits_to_dense, dim3(cdiv((long long)n_rows * n_fold, TPB)), dim3(TPB), 0, stream, masks, rows, n_rows,
                       pbn_post_words(n_fold), n_fold, dense);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_instance_overlap(const int32_t* masks, int n_pred, int n_pts, const int32_t* gt_index, int n_gt,
                                    int32_t* inter, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_pred < 0 || n_pts < 0 || n_gt < 1) return PBN_ERR_ARG;
    if (n_pred == 0) return PBN_OK;
    if (!inter) return PBN_ERR_ARG;
    { const int frc_ = fill_bytes(inter, 0, sizeof(int32_t) * (size_t)n_pred * n_gt, stream); if (frc_ != PBN_OK) return frc_; }
    if (n_pts == 0) return PBN_OK;
    if (!masks || !gt_index) return PBN_ERR_ARG;
    hipLaunchKernelGGL(k_instance_overlap, dim3(cdiv(n_pts, OVERLAP_CHUNK), n_pred), dim3(TPB), 0, stream, masks, n_pts,
                       gt_index, n_gt, n_gt <= OVERLAP_LDS_BINS ? 1 : 0, inter);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}
