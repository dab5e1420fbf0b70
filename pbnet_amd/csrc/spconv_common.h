// spconv_common.h -- types and per-lane helpers shared by the two convolution kernel families (spconv.hip: LDS weight
// ring, workgroup tiles; spconv_wave.hip: wave-autonomous tiles, K split over the waves of a workgroup).
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include "pbn_common.h"

namespace pbn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
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
    unsigned in_bytes, w_bytes;   // extents of the two buffer resources (< 2 GiB, checked by pbn_spconv_forward)
    int K, vpo, n_steps, ntiles_total;
    int n_out, relu;
    int n_sel;           // rows the kernel CHOICE is made for (family, tile shape): n_out, or the caller's expectation when n_out is a capacity
    int ksplit;          // >1: this launch writes fp32 partial sums, k_spconv_reduce applies the epilogue
    float* partial;      // [ksplit][n_out_pad][ntiles_total*16]
    int n_out_pad;
    int cg;              // steps per barrier group (1..4)
    int wmajor;          // wave family: weight-major block order (spconv_wave.hip map_block)
    int dbg;             // ablation switches for scripts/probe_conv_ablate.py (PBN_CONV_DBG); 0 in production
    // second source (round 4: a BasicBlock's 1x1 shortcut folded into its second convolution, Mink.py:77-87): reduction steps
    // n_main .. n_steps-1 read row o of `in2` (identity map), vpo2 vectors per row; null = none
    const void* in2;
    int ld_in2, vpo2, n_main;
    unsigned in2_bytes;
    // next op's packed weights (round 4): every workgroup of THIS launch touches a slice of them so that they are in the L2
    // of the XCD that will read them (weight-major launches: an XCD owns channel-tile groups) or at least in the memory-side
    // cache when the next launch starts.  pf_groups = channel-tile groups of the next launch (0 = no ownership: plain slices)
    const void* pf_w;
    int pf_steps, pf_ntt, pf_nt, pf_groups;
};

// what the executor knows about the launch an op will turn into (spconv_wave.hip: describe_launch)
struct LaunchDesc { int wave_family, nt, groups, wmajor; };
struct NextWeights { const void* w; int steps, ntt, nt, groups; };
extern thread_local NextWeights g_next_weights;     // set by the executor around pbn_spconv_forward (spconv.hip)
// round 5: expected rows of the NEXT convolution call's output level (capacity-planned forwards: n_out is a capacity 1.25 x larger,
// and choosing families / tile shapes by it picks slower kernels); 0 = none.  Set by the executor around a call, cleared behind it
extern thread_local int g_rows_hint;

namespace {

template <typename T> struct Tr;
template <> struct Tr<float> { static constexpr int ELEMS = 4; };
template <> struct Tr<__hip_bfloat16> { static constexpr int ELEMS = 8; };
template <> struct Tr<__half> { static constexpr int ELEMS = 8; };

template <typename T>
__device__ __forceinline__ void mfma_step(const u32x4& w, const u32x4& x, f32x4& acc);

template <>
__device__ __forceinline__ void mfma_step<float>(const u32x4& w, const u32x4& x, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w[0]), __uint_as_float(x[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w[1]), __uint_as_float(x[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w[2]), __uint_as_float(x[2]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w[3]), __uint_as_float(x[3]), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mfma_step<__hip_bfloat16>(const u32x4& w, const u32x4& x, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mfma_step<__half>(const u32x4& w, const u32x4& x, f32x4& acc) {
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

// Touch this workgroup's share of the next op's weights (1 KiB pieces [step][channel tile][64 lanes][16 B]): loads into a
// register nobody reads, never waited for (they retire while the main loop runs).  With ownership, the blocks of XCD x
// (blockIdx % 8) share the pieces of the groups that XCD will read; without, all blocks share all pieces.
// The destination register is returned and must stay allocated until the loads have landed: the caller names it in
// prefetch_drain() behind its own first loads (the rulebook tile), whose wait covers these older loads anyway -- a load that
// lands in a register the compiler has meanwhile given to an address is a memory fault (seen).
__device__ __forceinline__ void prefetch_drain(u32x4& sink) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink)); }
__device__ __forceinline__ u32x4 prefetch_next_weights(const ConvArgs& a, int block, int n_blocks, int tid, int tpb) {
    u32x4 sink = {0u, 0u, 0u, 0u};
    if (!a.pf_w) return sink;
    const int lane = tid & 63, wave = tid >> 6, waves = tpb >> 6;
    int owners = 1, xcd = 0, n_owned = a.pf_ntt / (a.pf_nt > 0 ? a.pf_nt : 1), g0 = 0, gstride = 1;
    if (a.pf_groups > 0) {
        xcd = block & 7;
        owners = 8;
        if (a.pf_groups >= 8) { g0 = xcd; gstride = 8; n_owned = (a.pf_groups - xcd + 7) >> 3; }
        else { g0 = xcd % a.pf_groups; gstride = a.pf_groups; n_owned = 1; }
    } else {
        n_owned = 1;                                  // one "group" spanning all tiles
    }
    const int nt = a.pf_groups > 0 ? a.pf_nt : a.pf_ntt;
    const int rank = a.pf_groups > 0 ? (block >> 3) : block;
    const int nrank = a.pf_groups > 0 ? (n_blocks + 7 - xcd) >> 3 : n_blocks;
    const long long pieces = (long long)a.pf_steps * n_owned * nt;
    const long long per = (pieces + (nrank > 0 ? nrank : 1) - 1) / (nrank > 0 ? nrank : 1);
    const long long lo = (long long)rank * per, hi = lo + per < pieces ? lo + per : pieces;
    (void)owners;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(a.pf_w);
    for (long long q = lo + wave; q < hi; q += waves) {
        const int s = (int)(q / (n_owned * nt));
        const int rem = (int)(q - (long long)s * (n_owned * nt));
        const int gi = rem / nt, t = rem - gi * nt;
        const int tile = (g0 + gi * gstride) * nt + t;
        const u32x4* src = reinterpret_cast<const u32x4*>(base + ((size_t)s * a.pf_ntt + tile) * 1024) + lane;
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink) : "v"(src) : "memory");     // same register: loads land in order
    }
    return sink;
}

__device__ __forceinline__ int xcd_tile(int b, int nt) {  // contiguous tile range per XCD (bijective for any nt)
    const int q = nt >> 3, r = nt & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace

// spconv_wave.hip: wave-autonomous kernel family.  Returns PBN_ERR_UNSUPPORTED when no instantiation fits.
int launch_wave(const ConvArgs& a, int dtype, int force_cfg, hipStream_t stream);
// spconv_wave.hip: which levels take the wave-autonomous family (rows of the output level, shapes)
bool wave_family_wanted(const ConvArgs& a, int dtype);
// spconv_wave.hip: channel tiles per workgroup / groups / block order of the launch `a` will become (automatic configuration)
void describe_launch(const ConvArgs& a, int dtype, LaunchDesc* d);

// spconv_rs.hip (round 5): row-stationary big-tile family for the wide levels.  cfg 0 = automatic tile height, 1..5 = fragments per wave
bool rs_family_wanted(const ConvArgs& a, int dtype);
int launch_rs(const ConvArgs& a, int dtype, int cfg, hipStream_t stream);


#ifdef PBN_EXPERIMENTS
// experiments/spconv_pc.hip (round 6, `make experiments` only): pair-compacted family for the wide levels (fp32 output tile in
// LDS, fragments of 16 real rule pairs).  rows: 0 = automatic tile height, otherwise the tile height
int launch_pc(const ConvArgs& a, int dtype, int rows, hipStream_t stream);
bool pc_family_wanted(const ConvArgs& a, int dtype);
#endif

}  // namespace pbn
