// spconv.hip -- output-stationary implicit-GEMM sparse convolution on gfx950 matrix cores (wave64, MFMA 16x16).
//
// Re-creates the arithmetic the reference delegates to MinkowskiEngine's convolution kernels
// (/root/reference/network/Mink.py:221-288 constructors, :293-350 call sites; PBNet.py:43-82 linear heads):
//     out[o, :] = epilogue( sum_k  in[nbr[o,k], :] @ W[k] )        nbr from coords.hip, -1 = no neighbour
// One kernel serves k=5 / k=3 / k=2,s=2 down / k=2,s=2 transposed / 1x1 / linear: they differ only in the table.
//
// Mapping (MI355X-first, not a translation of gather-GEMM-scatter):
//   * workgroup = 4 waves = TM output rows (TM = 64 or 128) x NT*16 output channels; accumulators stay in registers
//     for the whole K*Cin reduction, so there is no scatter, no atomics and a fixed summation order (deterministic);
//   * the reduction axis is the flattened (kernel offset, input channel) axis cut into STEPS of 4 x 16-byte vectors;
//     each lane gathers its MFMA operand (16 B of one neighbour row) straight from the feature slab in HBM/L2 --
//     rows are contiguous, a 16-lane group reads whole 64-B row segments;
//   * the rulebook tile nbr[TM][K] is staged once in LDS; steps whose offsets have no neighbour in the tile are
//     dropped from the step list (sparse scenes: ~7 of 27 offsets populated);
//   * weights are pre-packed in MFMA-fragment order, double-buffered through LDS and shared by the 4 waves;
//   * the products are computed transposed (D = W^T-tile x X^T-tile) so that each lane ends up with 4 CONSECUTIVE
//     output channels of one row: the epilogue (folded BN scale/shift, bias, residual, ReLU, down-cast) is applied
//     in registers and written with 8/16-byte stores;
//   * blockIdx -> tile mapping is XCD-aware (contiguous tile ranges per XCD) so neighbouring tiles, which gather
//     overlapping rows, share an L2.
//
// Numerics: fp32 accumulate always.  f32 slabs use v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain) -- this is the
// parity configuration (1e-4 vs the oracle); bf16/f16 slabs use v_mfma_f32_16x16x32_{bf16,f16}.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include "pbn_common.h"

namespace pbn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct ConvArgs {
    const void* in;      // feature slab [*, ld_in] elements of T (already offset to the first input column)
    const int* nbr;      // [n_out, K] input rows, -1 = none; nullptr => identity (K must be 1)
    const int* row_perm; // optional processing order: tile position p handles output row row_perm[p]
    const int* n_out_dev;
    const void* w;       // packed weights [n_steps][ntiles_total][64][16 B]
    const float* scale;  // [cout_p] or nullptr
    const float* shift;  // [cout_p] or nullptr (bias / folded BN shift)
    const void* residual;
    void* out;
    int ld_in, ld_res, ld_out;
    int K, vpo, n_steps, ntiles_total;
    int n_out, relu;
};

template <typename T> struct Tr;
template <> struct Tr<float> { static constexpr int ELEMS = 4; };
template <> struct Tr<__hip_bfloat16> { static constexpr int ELEMS = 8; };
template <> struct Tr<__half> { static constexpr int ELEMS = 8; };

template <typename T>
__device__ __forceinline__ void mfma_step(const uint4& w, const uint4& x, f32x4& acc);

template <>
__device__ __forceinline__ void mfma_step<float>(const uint4& w, const uint4& x, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(x.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mfma_step<__hip_bfloat16>(const uint4& w, const uint4& x, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mfma_step<__half>(const uint4& w, const uint4& x, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
}

// 4 consecutive channels: load as float4 / store from float4
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    return f32x4{v.x, v.y, v.z, v.w};
}
template <> __device__ __forceinline__ f32x4 load4<__hip_bfloat16>(const __hip_bfloat16* p) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    return f32x4{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                 __uint_as_float(v.y & 0xffff0000u)};
}
template <> __device__ __forceinline__ f32x4 load4<__half>(const __half* p) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    const __half2 a = __builtin_bit_cast(__half2, v.x), b = __builtin_bit_cast(__half2, v.y);
    const float2 fa = __half22float2(a), fb = __half22float2(b);
    return f32x4{fa.x, fa.y, fb.x, fb.y};
}
__device__ __forceinline__ unsigned bf16_rne(float f) {  // round-to-nearest-even, NaN preserved
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
template <typename T> __device__ __forceinline__ void store4(T* p, const f32x4& v);
template <> __device__ __forceinline__ void store4<float>(float* p, const f32x4& v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<__hip_bfloat16>(__hip_bfloat16* p, const f32x4& v) {
    uint2 o;
    o.x = bf16_rne(v[0]) | (bf16_rne(v[1]) << 16);
    o.y = bf16_rne(v[2]) | (bf16_rne(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = o;
}
template <> __device__ __forceinline__ void store4<__half>(__half* p, const f32x4& v) {
    const __half2 a = __floats2half2_rn(v[0], v[1]), b = __floats2half2_rn(v[2], v[3]);
    uint2 o;
    o.x = __builtin_bit_cast(unsigned, a);
    o.y = __builtin_bit_cast(unsigned, b);
    *reinterpret_cast<uint2*>(p) = o;
}

__device__ __forceinline__ int xcd_tile(int b, int nt) {  // contiguous tile range per XCD (bijective for any nt)
    const int q = nt >> 3, r = nt & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

constexpr int CONV_TPB = 256;

template <typename T, int RW, int NT>
__global__ __launch_bounds__(CONV_TPB) void k_spconv(const ConvArgs a) {
    constexpr int ELEMS = Tr<T>::ELEMS;
    constexpr int TM = 4 * RW;
    constexpr int NF = RW / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = a.K;
    const int KS = K | 1;  // odd row pitch: conflict-free column reads of the rulebook tile
    uint4* s_w = reinterpret_cast<uint4*>(smem);                                   // 2 * NT * 64 uint4
    int* s_nbr = reinterpret_cast<int*>(smem + 2 * NT * 1024);                      // TM * KS
    int* s_valid = s_nbr + TM * KS;                                                 // K
    int* s_steps = s_valid + ((K + 3) & ~3);                                        // n_steps + 1 (last = count)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n_out_dev ? min(*a.n_out_dev, a.n_out) : a.n_out;
    const int row0 = xcd_tile(blockIdx.x, gridDim.x) * TM;
    if (row0 >= n) return;
    const int tile0 = blockIdx.y * NT;

    for (int k = tid; k < K; k += CONV_TPB) s_valid[k] = 0;
    __syncthreads();
    for (int e = tid; e < TM * K; e += CONV_TPB) {
        const int r = e / K, k = e - r * K;
        const int p = row0 + r;
        int v = -1;
        if (p < n) {
            const int row = a.row_perm ? a.row_perm[p] : p;
            v = a.nbr ? a.nbr[(size_t)row * K + k] : row;
        }
        s_nbr[r * KS + k] = v;
        if (v >= 0) s_valid[k] = 1;
    }
    __syncthreads();
    // list of steps that touch at least one populated offset (wave 0, ordered compaction)
    const int vpo = a.vpo;
    if (wave == 0) {
        int base = 0;
        for (int s0 = 0; s0 < a.n_steps; s0 += 64) {
            const int s = s0 + lane;
            bool ok = false;
            if (s < a.n_steps) {
                if ((vpo & 3) == 0) {
                    ok = s_valid[s / (vpo >> 2)] != 0;
                } else {
                    for (int g = 0; g < 4; ++g) {
                        const int ko = (s * 4 + g) / vpo;
                        ok = ok || (ko < K && s_valid[ko] != 0);
                    }
                }
            }
            const unsigned long long m = __ballot(ok);
            if (ok) s_steps[base + __popcll(m & ((1ULL << lane) - 1ULL))] = s;
            base += __popcll(m);
        }
        if (lane == 0) s_steps[a.n_steps] = base;
    }
    __syncthreads();
    const int ns = s_steps[a.n_steps];

    f32x4 acc[NF][NT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const T* in = reinterpret_cast<const T*>(a.in);
    const uint4* wp = reinterpret_cast<const uint4*>(a.w);
    const int g = lane >> 4, rl = lane & 15;
    constexpr int WREGS = (NT * 64 + CONV_TPB - 1) / CONV_TPB;

    auto load_x = [&](int s, uint4 (&x)[NF]) {
        const int v = s * 4 + g;
        int ko, cv;
        if ((vpo & 3) == 0) { const int q = vpo >> 2; ko = s / q; cv = (s - ko * q) * 4 + g; }
        else { ko = v / vpo; cv = v - ko * vpo; }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int r = wave * RW + f * 16 + rl;
            const int src = (ko < K) ? s_nbr[r * KS + ko] : -1;
            x[f] = make_uint4(0u, 0u, 0u, 0u);
            if (src >= 0) x[f] = *reinterpret_cast<const uint4*>(in + (size_t)src * a.ld_in + cv * ELEMS);
        }
    };
    auto load_w = [&](int s, uint4 (&wr)[WREGS]) {
#pragma unroll
        for (int i = 0; i < WREGS; ++i) {
            const int q = tid + i * CONV_TPB;
            if (q < NT * 64) wr[i] = wp[((size_t)s * a.ntiles_total + tile0) * 64 + q];
        }
    };
    auto store_w = [&](int buf, const uint4 (&wr)[WREGS]) {
#pragma unroll
        for (int i = 0; i < WREGS; ++i) {
            const int q = tid + i * CONV_TPB;
            if (q < NT * 64) s_w[buf * NT * 64 + q] = wr[i];
        }
    };

    if (ns > 0) {
        uint4 xcur[NF], xnext[NF], wr[WREGS];
        load_w(s_steps[0], wr);
        load_x(s_steps[0], xcur);
        store_w(0, wr);
        __syncthreads();
        for (int si = 0; si < ns; ++si) {
            const int cur = si & 1;
            const bool more = si + 1 < ns;
            if (more) {
                const int sn = s_steps[si + 1];
                load_w(sn, wr);
                load_x(sn, xnext);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint4 wf = s_w[cur * NT * 64 + t * 64 + lane];
#pragma unroll
                for (int f = 0; f < NF; ++f) mfma_step<T>(wf, xcur[f], acc[f][t]);
            }
            if (more) store_w(cur ^ 1, wr);
            __syncthreads();
#pragma unroll
            for (int f = 0; f < NF; ++f) xcur[f] = xnext[f];
        }
    }

    // epilogue: lane holds channels c0..c0+3 of output row (wave*RW + f*16 + rl)
    T* out = reinterpret_cast<T*>(a.out);
    const T* res = reinterpret_cast<const T*>(a.residual);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int p = row0 + wave * RW + f * 16 + rl;
        if (p >= n) continue;
        const int orow = a.row_perm ? a.row_perm[p] : p;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c0 = (tile0 + t) * 16 + g * 4;
            f32x4 v = acc[f][t];
            if (a.scale) {
                const float4 sc = *reinterpret_cast<const float4*>(a.scale + c0);
                v[0] *= sc.x; v[1] *= sc.y; v[2] *= sc.z; v[3] *= sc.w;
            }
            if (a.shift) {
                const float4 sh = *reinterpret_cast<const float4*>(a.shift + c0);
                v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w;
            }
            if (res) {
                const f32x4 rv = load4<T>(res + (size_t)orow * a.ld_res + c0);
                v += rv;
            }
            if (a.relu) {
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            }
            store4<T>(out + (size_t)orow * a.ld_out + c0, v);
        }
    }
}

template <typename T, int RW, int NT>
int launch_one(const ConvArgs& a, int ngroups, hipStream_t stream) {
    constexpr int TM = 4 * RW;
    const int KS = a.K | 1;
    const size_t lds = 2 * NT * 1024 + sizeof(int) * ((size_t)TM * KS + ((a.K + 3) & ~3) + a.n_steps + 1);
    if (lds > 160 * 1024) return PBN_ERR_UNSUPPORTED;
    auto kern = k_spconv<T, RW, NT>;
    if (lds > 64 * 1024)
        PBN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tiles = cdiv(a.n_out, TM);
    hipLaunchKernelGGL(kern, dim3(tiles, ngroups), dim3(CONV_TPB), lds, stream, a);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

template <typename T, int RW>
int launch_nt(const ConvArgs& a, hipStream_t stream) {
    const int ntt = a.ntiles_total;
    if (ntt % 8 == 0) return launch_one<T, RW, 8>(a, ntt / 8, stream);
    if (ntt % 6 == 0) return launch_one<T, RW, 6>(a, ntt / 6, stream);
    if (ntt % 4 == 0) return launch_one<T, RW, 4>(a, ntt / 4, stream);
    if (ntt % 2 == 0) return launch_one<T, RW, 2>(a, ntt / 2, stream);
    return launch_one<T, RW, 1>(a, ntt, stream);
}

template <typename T>
int launch_t(const ConvArgs& a, int rows_per_wave, hipStream_t stream) {
    if (rows_per_wave == 32) return launch_nt<T, 32>(a, stream);
    return launch_nt<T, 16>(a, stream);
}

// ---- row gather: out[i, :] = in[idx[i], :]  (voxel -> point, PBNet.py:130-134) ----------------------------------
__global__ __launch_bounds__(256) void k_gather_rows(const uint4* __restrict__ in, const long long* __restrict__ idx,
                                                    int n, int vec_per_row, int ld_in_vec, int ld_out_vec,
                                                    uint4* __restrict__ out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)n * vec_per_row) return;
    const int i = (int)(e / vec_per_row), v = (int)(e % vec_per_row);
    out[(size_t)i * ld_out_vec + v] = in[(size_t)idx[i] * ld_in_vec + v];
}

}  // namespace
}  // namespace pbn

using namespace pbn;

extern "C" int pbn_spconv_forward(const void* in_feat, int ld_in, const int32_t* nbr, int n_offsets,
                                  const int32_t* row_perm, const int32_t* n_out_dev, int n_out, const void* w_packed,
                                  int vecs_per_offset, int n_steps, int cout_padded, const float* scale,
                                  const float* shift, const void* residual, int ld_res, int relu, void* out_feat,
                                  int ld_out, int dtype, int rows_per_wave, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_out < 0 || n_offsets < 1 || vecs_per_offset < 1 || n_steps < 1 || cout_padded < 16 || (cout_padded & 15))
        return PBN_ERR_ARG;
    if (!(vecs_per_offset == 1 || vecs_per_offset == 2 || (vecs_per_offset & 3) == 0)) return PBN_ERR_ARG;
    if (n_steps != (n_offsets * vecs_per_offset + 3) / 4) return PBN_ERR_ARG;
    if (!nbr && n_offsets != 1) return PBN_ERR_ARG;
    if (n_out == 0) return PBN_OK;
    if (!in_feat || !w_packed || !out_feat) return PBN_ERR_ARG;
    const int esz = dtype == PBN_F32 ? 4 : 2;
    if ((ld_in * esz) % 16 || (ld_out * esz) % 8 || (residual && (ld_res * esz) % 8)) return PBN_ERR_ARG;
    if (((uintptr_t)in_feat | (uintptr_t)w_packed) & 15) return PBN_ERR_ARG;
    ConvArgs a;
    a.in = in_feat; a.nbr = nbr; a.row_perm = row_perm; a.n_out_dev = n_out_dev; a.w = w_packed; a.scale = scale;
    a.shift = shift; a.residual = residual; a.out = out_feat; a.ld_in = ld_in; a.ld_res = ld_res; a.ld_out = ld_out;
    a.K = n_offsets; a.vpo = vecs_per_offset; a.n_steps = n_steps; a.ntiles_total = cout_padded / 16; a.n_out = n_out;
    a.relu = relu;
    if (rows_per_wave != 16 && rows_per_wave != 32) rows_per_wave = (n_out >= 64 * 1024) ? 32 : 16;
    switch (dtype) {
        case PBN_F32: return launch_t<float>(a, rows_per_wave, stream);
        case PBN_BF16: return launch_t<__hip_bfloat16>(a, rows_per_wave, stream);
        case PBN_F16: return launch_t<__half>(a, rows_per_wave, stream);
        default: return PBN_ERR_ARG;
    }
}

extern "C" int pbn_gather_rows(const void* in, int ld_in_bytes, const int64_t* idx, int n, int row_bytes, void* out,
                               int ld_out_bytes, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || row_bytes <= 0 || (row_bytes & 15) || (ld_in_bytes & 15) || (ld_out_bytes & 15)) return PBN_ERR_ARG;
    if (n == 0) return PBN_OK;
    if (!in || !idx || !out) return PBN_ERR_ARG;
    const int vpr = row_bytes / 16;
    const long long total = (long long)n * vpr;
    hipLaunchKernelGGL(k_gather_rows, dim3(cdiv(total, 256)), dim3(256), 0, stream, (const uint4*)in,
                       (const long long*)idx, n, vpr, ld_in_bytes / 16, ld_out_bytes / 16, (uint4*)out);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}
