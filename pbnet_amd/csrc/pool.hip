// pool.hip -- per-batch global max / average pooling over a feature slab whose rows are grouped by batch index.
// Replaces MinkowskiGlobalMaxPooling / MinkowskiGlobalAvgPooling as used by the score branch
// (/root/reference/network/PBNet.py:67-68,274-276).  One workgroup per (batch, 32-channel chunk); rows are reduced in a
// fixed order (8 interleaved partial streams, then a fixed tree), so the result is deterministic.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include "pbn_common.h"

namespace pbn {
namespace {

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <> __device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }

template <typename T>
__global__ __launch_bounds__(256) void k_segment_pool(const T* __restrict__ feats, int ld, int channels,
                                                     const int* __restrict__ seg_start, float* __restrict__ out_max,
                                                     float* __restrict__ out_avg) {
    __shared__ float s_max[8][32], s_sum[8][32];
    const int b = blockIdx.x, c = blockIdx.y * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
    const int beg = seg_start[b], end = seg_start[b + 1];
    float mx = -__builtin_inff(), sm = 0.f;
    if (c < channels)
        for (int r = beg + rg; r < end; r += 8) {
            const float v = to_f32<T>(feats[(size_t)r * ld + c]);
            mx = fmaxf(mx, v);
            sm += v;
        }
    s_max[rg][threadIdx.x & 31] = mx;
    s_sum[rg][threadIdx.x & 31] = sm;
    __syncthreads();
    if (rg == 0 && c < channels) {
#pragma unroll
        for (int k = 1; k < 8; ++k) { mx = fmaxf(mx, s_max[k][threadIdx.x]); sm += s_sum[k][threadIdx.x]; }
        if (out_max) out_max[(size_t)b * channels + c] = mx;
        if (out_avg) out_avg[(size_t)b * channels + c] = sm / (float)(end - beg);
    }
}

}  // namespace
}  // namespace pbn

using namespace pbn;

extern "C" int pbn_segment_pool(const void* feats, int ld, int channels, int dtype, const int32_t* seg_start, int n_seg,
                                float* out_max, float* out_avg, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_seg < 0 || channels <= 0 || ld < channels) return PBN_ERR_ARG;
    if (n_seg == 0) return PBN_OK;
    if (!feats || !seg_start || (!out_max && !out_avg)) return PBN_ERR_ARG;
    const dim3 grid(n_seg, cdiv(channels, 32));
    switch (dtype) {
        case PBN_F32:
            hipLaunchKernelGGL(k_segment_pool<float>, grid, dim3(256), 0, stream, (const float*)feats, ld, channels,
                               seg_start, out_max, out_avg);
            break;
        case PBN_BF16:
            hipLaunchKernelGGL(k_segment_pool<__hip_bfloat16>, grid, dim3(256), 0, stream, (const __hip_bfloat16*)feats,
                               ld, channels, seg_start, out_max, out_avg);
            break;
        case PBN_F16:
            hipLaunchKernelGGL(k_segment_pool<__half>, grid, dim3(256), 0, stream, (const __half*)feats, ld, channels,
                               seg_start, out_max, out_avg);
            break;
        default: return PBN_ERR_ARG;
    }
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}
