// stages.hip -- the glue between the big kernels of PBNet.forward (/root/reference/network/PBNet.py:113-280), fused:
// every entry point replaces a run of small tensor ops (index / cat / where / floor / arange ...) of the reference by ONE
// launch with the same integer results and the same fp32 arithmetic.  Nothing here is arithmetic-heavy; the point is
// launch count (host and device) on the inference path, where the stages between U-Nets are latency bound.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include "pbn_common.h"

namespace pbn {
namespace {

constexpr int TPB = 256;

// largest e in [0, n) with start[e] <= r  (start ascending, start[0] = 0, start[n] = total > r)
__device__ __forceinline__ int upper_entry(const int* __restrict__ start, int n, int r) {
    int lo = 0, hi = n;  // invariant: start[lo] <= r < start[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (start[mid] <= r) lo = mid; else hi = mid;
    }
    return lo;
}

// ---- local-scene rows (PBNet.py:182-247) ---------------------------------------------------------------------------
// One thread per (row, 32-bit word of the output feature row).  Word w of a row of ESZ-byte elements:
//   elements [0, C)      : point_feat[p, :]
//   element  C           : sem_score[p, sem_pred[p]]      (PBNet.py:162-163: softmax score of the point's own class)
//   element  C + 1       : entry weight                   (PBNet.py:194,230)
//   elements [C+2, ld)   : 0
template <int ESZ>
__global__ __launch_bounds__(TPB) void k_local_scene_rows(
    const int* __restrict__ ent_row_start, const int* __restrict__ ent_member_start, const int* __restrict__ ent_scene,
    const float* __restrict__ ent_weight, int n_ent, int n_rows, const int* __restrict__ member_idx,
    const long long* __restrict__ ins_ind, const float* __restrict__ xyz, float inv_voxel,
    const unsigned char* __restrict__ point_feat, int ld_feat, int channels, const unsigned char* __restrict__ sem_score,
    int ld_sem, const long long* __restrict__ sem_pred, int dtype, long long* __restrict__ point_idx,
    long long* __restrict__ row_scene, int* __restrict__ coords, unsigned* __restrict__ feat_out, int words_per_row) {
    const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
    if (e >= (long long)n_rows * words_per_row) return;
    const int r = (int)(e / words_per_row), w = (int)(e - (long long)r * words_per_row);
    const int ent = upper_entry(ent_row_start, n_ent, r);
    const int local = member_idx[ent_member_start[ent] + (r - ent_row_start[ent])];
    const long long p = ins_ind[local];
    if (w == 0) {
        point_idx[r] = p;
        row_scene[r] = ent_scene[ent];
        // the reference divides a device tensor by a host scalar: that is a multiplication by the fp32 reciprocal
        const float x = xyz[3 * p + 0] * inv_voxel, y = xyz[3 * p + 1] * inv_voxel, z = xyz[3 * p + 2] * inv_voxel;
        reinterpret_cast<int4*>(coords)[r] = make_int4(ent_scene[ent], (int)floorf(x), (int)floorf(y), (int)floorf(z));
    }
    constexpr int EPW = 4 / ESZ;  // elements per word
    unsigned out = 0;
#pragma unroll
    for (int k = 0; k < EPW; ++k) {
        const int c = w * EPW + k;
        unsigned bits = 0;
        if (c < channels) {
            if (ESZ == 4) bits = *reinterpret_cast<const unsigned*>(point_feat + ((size_t)p * ld_feat + c) * 4);
            else bits = *reinterpret_cast<const unsigned short*>(point_feat + ((size_t)p * ld_feat + c) * 2);
        } else if (c == channels) {
            const size_t o = (size_t)p * ld_sem + (size_t)sem_pred[p];
            if (ESZ == 4) bits = *reinterpret_cast<const unsigned*>(sem_score + o * 4);
            else bits = *reinterpret_cast<const unsigned short*>(sem_score + o * 2);
        } else if (c == channels + 1) {
            const float wt = ent_weight[ent];
            if (ESZ == 4) bits = __float_as_uint(wt);
            else if (dtype == PBN_BF16) bits = __builtin_bit_cast(unsigned short, __float2bfloat16(wt));
            else bits = __builtin_bit_cast(unsigned short, __float2half(wt));
        }
        out |= bits << (8 * ESZ * k);
    }
    feat_out[e] = out;
}

// ---- out[i, :C] = in[idx[i], :C], out[i, C:ld_out] = 0  (32-bit words) -----------------------------------------------
__global__ __launch_bounds__(TPB) void k_gather_pad_rows(const unsigned* __restrict__ in, int ld_in_w, int row_w,
                                                        const long long* __restrict__ idx, int n,
                                                        unsigned* __restrict__ out, int ld_out_w) {
    const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
    if (e >= (long long)n * ld_out_w) return;
    const int i = (int)(e / ld_out_w), w = (int)(e - (long long)i * ld_out_w);
    unsigned v = 0;
    if (w < row_w) v = in[(size_t)(idx ? idx[i] : i) * ld_in_w + w];
    out[e] = v;
}

}  // namespace
}  // namespace pbn

using namespace pbn;

extern "C" int pbn_local_scene_rows(const int32_t* ent_row_start, const int32_t* ent_member_start,
                                    const int32_t* ent_scene, const float* ent_weight, int n_ent, int n_rows,
                                    const int32_t* member_idx, const int64_t* ins_ind, const float* xyz, float inv_voxel,
                                    const void* point_feat, int ld_feat, int channels, const void* sem_score, int ld_sem,
                                    const int64_t* sem_pred, int dtype, int64_t* point_idx, int64_t* row_scene,
                                    int32_t* coords, void* feat_out, int ld_out, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_ent < 0 || n_rows < 0 || channels < 0 || ld_out < channels + 2 || ld_feat < channels) return PBN_ERR_ARG;
    if (n_rows == 0) return PBN_OK;
    if (n_ent == 0 || !ent_row_start || !ent_member_start || !ent_scene || !ent_weight || !member_idx || !ins_ind || !xyz ||
        !point_feat || !sem_score || !sem_pred || !point_idx || !row_scene || !coords || !feat_out)
        return PBN_ERR_ARG;
    const int esz = dtype == PBN_F32 ? 4 : 2;
    if ((ld_out * esz) % 4 || ((uintptr_t)feat_out & 3) || ((uintptr_t)coords & 15)) return PBN_ERR_ARG;
    const int wpr = ld_out * esz / 4;
    const long long total = (long long)n_rows * wpr;
    const dim3 grid(cdiv(total, TPB));
#define PBN_ARGS                                                                                                        \
    ent_row_start, ent_member_start, ent_scene, ent_weight, n_ent, n_rows, member_idx, (const long long*)ins_ind, xyz,   \
        inv_voxel, (const unsigned char*)point_feat, ld_feat, channels, (const unsigned char*)sem_score, ld_sem,         \
        (const long long*)sem_pred, dtype, (long long*)point_idx, (long long*)row_scene, coords, (unsigned*)feat_out, wpr
    if (esz == 4) hipLaunchKernelGGL(k_local_scene_rows<4>, grid, dim3(TPB), 0, stream, PBN_ARGS);
    else hipLaunchKernelGGL(k_local_scene_rows<2>, grid, dim3(TPB), 0, stream, PBN_ARGS);
#undef PBN_ARGS
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_gather_pad_rows(const void* in, int ld_in_bytes, int row_bytes, const int64_t* idx, int n, void* out,
                                   int ld_out_bytes, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || row_bytes <= 0 || (row_bytes & 3) || (ld_in_bytes & 3) || (ld_out_bytes & 3) || ld_out_bytes < row_bytes ||
        ld_in_bytes < row_bytes)
        return PBN_ERR_ARG;
    if (n == 0) return PBN_OK;
    if (!in || !out || (((uintptr_t)in | (uintptr_t)out) & 3)) return PBN_ERR_ARG;
    const long long total = (long long)n * (ld_out_bytes / 4);
    hipLaunchKernelGGL(k_gather_pad_rows, dim3(cdiv(total, TPB)), dim3(TPB), 0, stream, (const unsigned*)in, ld_in_bytes / 4,
                       row_bytes / 4, (const long long*)idx, n, (unsigned*)out, ld_out_bytes / 4);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}
