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

// ---- two-pass form for long segments -------------------------------------------------------------------------------
// pass 1: workgroup (segment b, slice z of SPLIT) reduces its contiguous slice of rows with 16-byte row-vector loads
//         (a row of C channels = C*sizeof(T)/16 vectors; 256/vpr rows per sweep, 4 sweeps in flight);
// pass 2: one thread per (segment, channel) folds the SPLIT partials in slice order.  Fixed order => deterministic.
constexpr int POOL_SPLIT_MAX = 64;

template <typename T> struct PoolVec;
template <> struct PoolVec<float> {
    static constexpr int E = 4;
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
};
template <> struct PoolVec<__hip_bfloat16> {
    static constexpr int E = 8;
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
};
template <> struct PoolVec<__half> {
    static constexpr int E = 8;
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 t = __half22float2(__builtin_bit_cast(__half2, w[i]));
            f[2 * i] = t.x; f[2 * i + 1] = t.y;
        }
    }
};

template <typename T>
__global__ __launch_bounds__(256) void k_pool_partial(const uint4* __restrict__ feats, int ld_vec, int vpr,
                                                     const int* __restrict__ seg_start, int split,
                                                     float* __restrict__ part_max, float* __restrict__ part_sum) {
    constexpr int E = PoolVec<T>::E;
    extern __shared__ float s_red[];                       // [2][rows_per_sweep][vpr * E]
    const int b = blockIdx.x, z = blockIdx.y;
    const int beg = seg_start[b], end = seg_start[b + 1];
    const long long len = end - beg;
    const int lo = beg + (int)(len * z / split), hi = beg + (int)(len * (z + 1) / split);
    const int rps = 256 / vpr;                             // rows per sweep
    const int q = threadIdx.x % vpr, rl = threadIdx.x / vpr;
    float mx[E], sm[E];
#pragma unroll
    for (int i = 0; i < E; ++i) { mx[i] = -__builtin_inff(); sm[i] = 0.f; }
    if (rl < rps) {
        int r = lo + rl;
        for (; r + 3 * rps < hi; r += 4 * rps) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = feats[(size_t)(r + u * rps) * ld_vec + q];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[E];
                PoolVec<T>::unpack(v[u], f);
#pragma unroll
                for (int i = 0; i < E; ++i) { mx[i] = fmaxf(mx[i], f[i]); sm[i] += f[i]; }
            }
        }
        for (; r < hi; r += rps) {
            float f[E];
            PoolVec<T>::unpack(feats[(size_t)r * ld_vec + q], f);
#pragma unroll
            for (int i = 0; i < E; ++i) { mx[i] = fmaxf(mx[i], f[i]); sm[i] += f[i]; }
        }
    }
    const int C = vpr * E;
    float* s_mx = s_red;
    float* s_sm = s_red + rps * C;
    if (rl < rps) {
#pragma unroll
        for (int i = 0; i < E; ++i) { s_mx[rl * C + q * E + i] = mx[i]; s_sm[rl * C + q * E + i] = sm[i]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float m = -__builtin_inff(), t = 0.f;
        for (int k = 0; k < rps; ++k) { m = fmaxf(m, s_mx[k * C + c]); t += s_sm[k * C + c]; }
        part_max[((size_t)b * split + z) * C + c] = m;
        part_sum[((size_t)b * split + z) * C + c] = t;
    }
}

__global__ __launch_bounds__(256) void k_pool_final(const float* __restrict__ part_max, const float* __restrict__ part_sum,
                                                   const int* __restrict__ seg_start, int n_seg, int C, int split,
                                                   float* __restrict__ out_max, float* __restrict__ out_avg) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_seg * C) return;
    const int b = e / C, c = e - b * C;
    float m = -__builtin_inff(), t = 0.f;
    for (int z = 0; z < split; ++z) {
        m = fmaxf(m, part_max[((size_t)b * split + z) * C + c]);
        t += part_sum[((size_t)b * split + z) * C + c];
    }
    if (out_max) out_max[e] = m;
    if (out_avg) out_avg[e] = t / (float)(seg_start[b + 1] - seg_start[b]);
}

int pool_split(int n_seg) {
    int s = 1024 / (n_seg > 0 ? n_seg : 1);
    return s < 1 ? 1 : (s > POOL_SPLIT_MAX ? POOL_SPLIT_MAX : s);
}

}  // namespace
}  // namespace pbn

using namespace pbn;

extern "C" size_t pbn_segment_pool_workspace_bytes(int n_seg, int channels) {
    if (n_seg <= 0 || channels <= 0) return 0;
    return (size_t)2 * n_seg * pool_split(n_seg) * channels * sizeof(float);
}

extern "C" int pbn_segment_pool(const void* feats, int ld, int channels, int dtype, const int32_t* seg_start, int n_seg,
                                float* out_max, float* out_avg, void* workspace, size_t workspace_bytes,
                                pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_seg < 0 || channels <= 0 || ld < channels) return PBN_ERR_ARG;
    if (n_seg == 0) return PBN_OK;
    if (!feats || !seg_start || (!out_max && !out_avg)) return PBN_ERR_ARG;
    const int esz = dtype == PBN_F32 ? 4 : 2;
    const int split = pool_split(n_seg);
    if (workspace && workspace_bytes >= pbn_segment_pool_workspace_bytes(n_seg, channels) && split > 1 &&
        (channels * esz) % 16 == 0 && (ld * esz) % 16 == 0 && ((uintptr_t)feats & 15) == 0 &&
        ((uintptr_t)workspace & 15) == 0 && channels * esz / 16 <= 256) {
        const int vpr = channels * esz / 16, rps = 256 / vpr;
        float* pm = reinterpret_cast<float*>(workspace);
        float* ps = pm + (size_t)n_seg * split * channels;
        const size_t lds = (size_t)2 * rps * channels * sizeof(float);
        const dim3 grid(n_seg, split);
        if (dtype == PBN_F32)
            hipLaunchKernelGGL(k_pool_partial<float>, grid, dim3(256), lds, stream, (const uint4*)feats, ld * esz / 16, vpr,
                               seg_start, split, pm, ps);
        else if (dtype == PBN_BF16)
            hipLaunchKernelGGL(k_pool_partial<__hip_bfloat16>, grid, dim3(256), lds, stream, (const uint4*)feats,
                               ld * esz / 16, vpr, seg_start, split, pm, ps);
        else if (dtype == PBN_F16)
            hipLaunchKernelGGL(k_pool_partial<__half>, grid, dim3(256), lds, stream, (const uint4*)feats, ld * esz / 16,
                               vpr, seg_start, split, pm, ps);
        else
            return PBN_ERR_ARG;
        hipLaunchKernelGGL(k_pool_final, dim3(cdiv((long long)n_seg * channels, 256)), dim3(256), 0, stream, pm, ps, seg_start,
                           n_seg, channels, split, out_max, out_avg);
        PBN_LAUNCH_CHECK();
        return PBN_OK;
    }
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
