// wgrad.hip -- weight gradient of the sparse convolution (training, BASELINE configs[2]) on the matrix cores:
//     dW[k, ci, co] = sum over the rule pairs (i, o) of offset k of  x[i, ci] * g[o, co]
// (MinkowskiEngine's convolution backward w.r.t. the kernel, reached from /root/reference/train.py:57 `loss.backward()`
// through network/Mink.py:293-350).  Replaces the gather + library GEMM of round 1 (no rocBLAS in the step any more).
//
// Mapping: the contraction runs over PAIRS, so both MFMA operands are read straight from the row-major slabs with no
// transpose: v_mfma_f32_16x16x4_f32 takes A[i][k] = x[pair k][ci0 + i] and B[k][j] = g[pair k][co0 + j] -- for a fixed
// pair the 16 lanes of a lane group read 16 CONSECUTIVE channels of one row (one 32/64-byte segment).  bf16 / f16 slabs
// are widened on load (exact), products are exact in fp32, accumulation is fp32 in pair order: the result does not
// depend on the slab precision beyond the operands' own rounding.
//   * workgroup = 4 waves; a wave owns one 16-channel input tile x NTW 16-channel output tiles of ONE offset and a
//     contiguous quarter of the workgroup's pair range; the four partial tiles are summed through LDS in wave order;
//   * grid = (tile strips, offsets, pair splits); split partials go to a workspace and are summed in split order by
//     k_wgrad_reduce: fixed summation order everywhere -> bit-identical run to run (no atomics).
// Rate: the f32-input MFMA runs at 1/16 of the bf16 rate (MI355X_MICROARCH.md).  16-bit slabs with 16-byte aligned rows
// therefore take k_wgrad16 below: rows are gathered with 16-byte loads into LDS and reach v_mfma_f32_16x16x32_{bf16,f16}
// through the hardware transpose read (ds_read_b64_tr_b16); k_wgrad serves fp32 slabs and unaligned 16-bit rows.
#include "spconv_common.h"

namespace pbn {
namespace {

constexpr int NTW = 4;   // output-channel tiles per wave

template <typename T> __device__ __forceinline__ float widen(const T* p);
template <> __device__ __forceinline__ float widen<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float widen<__hip_bfloat16>(const __hip_bfloat16* p) {
    return __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(p) << 16);
}
template <> __device__ __forceinline__ float widen<__half>(const __half* p) { return __half2float(*p); }

// Flat grid -> (strip, offset, pair split), XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs, so split s
// is pinned to XCD s % 8 and that XCD's dispatch order walks all strips and offsets of one split before the next one --
// the workgroups that read the SAME output-gradient rows (and neighbouring input rows) share one L2 while they run.
struct WgradTile { int strip, k, split; bool live; };
__device__ __forceinline__ WgradTile wgrad_tile(int strips, int K, int splits) {
    WgradTile t;
    if (splits < 8) {
        // few splits (the coarse levels, whose rows fit one L2 many times over): no padding, consecutive workgroups --
        // i.e. all 8 XCDs and their 8 x 16 L2 channels -- share the tiles of one split
        const int i = blockIdx.x, j = i / splits;
        t.split = i - j * splits;
        t.k = j / strips;
        t.strip = j - t.k * strips;
        t.live = true;
        return t;
    }
    const int i = blockIdx.x, xcd = i & 7, j = i >> 3;
    const int inner = strips * K;
    const int sgrp = j / inner, rem = j - sgrp * inner;
    t.split = xcd + 8 * sgrp;
    t.k = rem / strips;
    t.strip = rem - t.k * strips;
    t.live = t.split < splits;
    return t;
}
__host__ inline long long wgrad_grid(int strips, int K, int splits) {
    return (long long)strips * K * (splits < 8 ? splits : ((splits + 7) & ~7));
}

struct WgradArgs {
    const void* x; const void* g;
    const int* in_idx; const int* out_idx;               // pair lists (nullptr: identity, pair p = row p)
    const int* seg_begin;                                 // [K+1] first segment of every offset (nullptr: one offset, n_pairs pairs)
    const int* counts;                                    // [K] pairs of every offset (nullptr: whole segments, -1 padded)
    float* out;                                           // dW [K, cin, cout] (splits == 1) or partial [splits, K, cin, cout]
    int ld_x, ld_g, cin, cout, K, segment, n_pairs, splits, co_groups, strips, dbg;
};

template <typename T>
__global__ __launch_bounds__(256) void k_wgrad(const WgradArgs a) {
    __shared__ float s_red[3][NTW][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kb = lane >> 4;
    const WgradTile wt = wgrad_tile(a.strips, a.K, a.splits);
    if (!wt.live) return;
    const int k = wt.k;
    const int ci0 = (wt.strip / a.co_groups) * 16, co0 = (wt.strip % a.co_groups) * (NTW * 16);
    // pair range of this offset, of this workgroup's split, of this wave
    long long p_lo, p_hi;
    if (a.seg_begin) { p_lo = (long long)a.seg_begin[k] * a.segment; p_hi = a.counts ? p_lo + a.counts[k] : (long long)a.seg_begin[k + 1] * a.segment; }
    else { p_lo = 0; p_hi = a.n_pairs; }
    const long long len = p_hi - p_lo;
    const long long per_split = ((len + a.splits - 1) / a.splits + 15) & ~15LL;
    const long long s_lo = min(p_lo + per_split * wt.split, p_hi), s_hi = min(s_lo + per_split, p_hi);
    const long long per_wave = (((s_hi - s_lo) + 3) / 4 + 3) & ~3LL;
    const long long w_lo = min(s_lo + per_wave * wave, s_hi), w_hi = min(w_lo + per_wave, s_hi);

    // both slabs through buffer resources: 32-bit byte offsets (row * pitch + channel), one multiply + add per row instead of
    // 64-bit pointer arithmetic per element; an absent pair / a channel past the tensor uses an out-of-range offset and
    // reads zero.  (Slabs stay far below 4 GiB: pbn_spconv_forward already refuses 2 GiB.)
    constexpr unsigned ESZ = (unsigned)sizeof(T);
    constexpr unsigned NOREC = 0xfffffff0u, OOB = 0xfffffff8u;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)NOREC, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.g), 0, (int)NOREC, 0x00020000);
    const unsigned ldx_b = (unsigned)a.ld_x * ESZ, ldg_b = (unsigned)a.ld_g * ESZ;
    const bool ci_ok = ci0 + j < a.cin;
    const unsigned cx_b = (unsigned)(ci0 + j) * ESZ;
    unsigned cg_b[NTW];
    bool co_ok[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) { co_ok[t] = co0 + t * 16 + j < a.cout; cg_b[t] = (unsigned)(co0 + t * 16 + j) * ESZ; }
    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto ld = [&](__amdgpu_buffer_rsrc_t rs, unsigned off) -> float {
        if constexpr (sizeof(T) == 4) return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
        else {
            const unsigned short h = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, 0, 0);
            if constexpr (__is_same(T, __hip_bfloat16)) return __uint_as_float((unsigned)h << 16);
            else return __half2float(__builtin_bit_cast(__half, h));
        }
    };
    // Blocks of 64 pairs: ONE coalesced index load per lane and block (lane l holds pair p0 + l; the next block's indices are
    // fetched a block ahead), the 16 MFMA steps of the block take their row indices by lane shuffles and issue all their
    // operand loads up front (16 x (1 + NTW) independent loads per lane), then the 64 MFMAs run behind them: two dependent
    // memory latencies per 64 MFMAs.
    auto load_idx = [&](long long p, int& ri, int& ro) {
        ri = -1; ro = -1;
        if (p < w_hi) {
            if (a.in_idx) { ri = a.in_idx[p]; ro = a.out_idx[p]; }
            else { ri = (int)p; ro = (int)p; }
        }
    };
    int ri_n, ro_n;
    load_idx(w_lo + lane, ri_n, ro_n);
    for (long long p0 = w_lo; p0 < w_hi; p0 += 64) {
        const int ri = ri_n, ro = ro_n;
        load_idx(p0 + 64 + lane, ri_n, ro_n);
        float av[16], bv[16][NTW];
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int src = st * 4 + kb;
            const int r_i = __shfl(ri, src, 64), r_o = __shfl(ro, src, 64);
            const bool ok = r_i >= 0;
            av[st] = ld(rs_x, (ok && ci_ok) ? (unsigned)r_i * ldx_b + cx_b : OOB);
            const unsigned gbase = (unsigned)r_o * ldg_b;
#pragma unroll
            for (int t = 0; t < NTW; ++t) bv[st][t] = ld(rs_g, (ok && co_ok[t]) ? gbase + cg_b[t] : OOB);
        }
#pragma unroll
        for (int st = 0; st < 16; ++st)
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st], bv[st][t], acc[t], 0, 0, 0);
    }
    // waves 1..3 -> LDS, wave 0 adds them in wave order and stores.  D layout: column = lane & 15 (co), row = kb * 4 + r (ci)
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_red[wave - 1][t][r * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (wave != 0) return;
    float* out = a.out + ((size_t)wt.split * a.K + k) * (size_t)a.cin * a.cout;
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[t][r];
            v += s_red[0][t][r * 64 + lane];
            v += s_red[1][t][r * 64 + lane];
            v += s_red[2][t][r * 64 + lane];
            const int ci = ci0 + kb * 4 + r, co = co0 + t * 16 + j;
            if (ci < a.cin && co < a.cout) out[(size_t)ci * a.cout + co] = v;
        }
}

// ---- 16-bit slabs: pairs are the MFMA's K axis, so both operands must be read TRANSPOSED (8 consecutive pairs of one
// channel per lane).  A step = 32 pairs: the workgroup's 256 threads gather the pairs' rows of both slabs with 16-byte
// loads (whole 8-channel chunks, coalesced along the row) one step ahead, store them row-major in LDS, and every wave
// builds its operands with ds_read_b64_tr_b16 -- a 16-lane group reads a [4 pairs][16 channels] block and each lane
// receives 4 pairs of one channel; two reads make the 8-pair operand of v_mfma_f32_16x16x32.
//   * workgroup = 2 x 2 waves; a wave owns WA x WB 16-channel tiles of dW[k]: the workgroup covers 32 WA x 32 WB channels;
//   * the pair -> MFMA k mapping is k = 8 kg + j  <->  LDS row 4 kg + (j & 3) + 16 (j >> 2) (the same for both operands, so
//     the contraction is unchanged): with a row pitch of 16 x odd elements the 8 rows a 32-lane half reads per LDS cycle
//     fall into 8 distinct 8-bank ranges -- conflict-free (MI355X_MICROARCH.md, LDS table);
//   * one barrier per step (two LDS images), index lists fetched two steps ahead, gathers one step ahead.
typedef short i16x4 __attribute__((ext_vector_type(4)));
constexpr int W16_SP = 32;   // pairs per step
__host__ __device__ constexpr int w16_pitch(int channels) { return ((channels / 16) | 1) * 16; }

template <typename T, int WA, int WB>
__global__ __launch_bounds__(256) void k_wgrad16(const WgradArgs a) {
    constexpr int CIT = 2 * WA, COT = 2 * WB;               // 16-channel tiles per workgroup
    constexpr int PX = w16_pitch(CIT * 16), PG = w16_pitch(COT * 16);
    constexpr int CXC = CIT * 2, CGC = COT * 2;             // 16-byte chunks per row
    constexpr int NCH = W16_SP * (CXC + CGC);
    constexpr int NPT = (NCH + 255) / 256;                  // chunks per thread and step
    __shared__ __attribute__((aligned(16))) unsigned short s_x[2][W16_SP * PX];
    __shared__ __attribute__((aligned(16))) unsigned short s_g[2][W16_SP * PG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kg = lane >> 4;
    const WgradTile wt = wgrad_tile(a.strips, a.K, a.splits);
    if (!wt.live) return;
    const int k = wt.k;
    const int cib = wt.strip / a.co_groups, cob = wt.strip % a.co_groups;
    const int ci_base = cib * (CIT * 16), co_base = cob * (COT * 16);
    const int wa = wave >> 1, wb = wave & 1;

    long long p_lo, p_hi;
    if (a.seg_begin) { p_lo = (long long)a.seg_begin[k] * a.segment; p_hi = a.counts ? p_lo + a.counts[k] : (long long)a.seg_begin[k + 1] * a.segment; }
    else { p_lo = 0; p_hi = a.n_pairs; }
    const long long len = p_hi - p_lo;
    const long long per_split = ((len + a.splits - 1) / a.splits + W16_SP - 1) & ~(long long)(W16_SP - 1);
    const long long s_lo = min(p_lo + per_split * wt.split, p_hi), s_hi = min(s_lo + per_split, p_hi);
    const int steps = (int)((s_hi - s_lo + W16_SP - 1) / W16_SP);

    constexpr unsigned NOREC = 0xfffffff0u, OOB = 0xfffffff8u;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)NOREC, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.g), 0, (int)NOREC, 0x00020000);
    const unsigned ldx_b = (unsigned)a.ld_x * 2u, ldg_b = (unsigned)a.ld_g * 2u;
    const int cin8 = (a.cin + 7) & ~7, cout8 = (a.cout + 7) & ~7;   // whole chunks inside the row (ld >= these: checked by the host)

    // this thread's chunks: slab, row of the step, byte offset inside the slab row (or "no such channel"), LDS element
    int c_row[NPT], c_lds[NPT];
    unsigned c_col[NPT];
    bool c_isx[NPT], c_ok[NPT];
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
        const int q = tid + 256 * e;
        c_isx[e] = q < W16_SP * CXC;
        const int qq = c_isx[e] ? q : q - W16_SP * CXC;
        const int per = c_isx[e] ? CXC : CGC;
        const int row = qq / per, cc = qq - row * per;
        c_row[e] = row;
        const int ch = (c_isx[e] ? ci_base : co_base) + cc * 8;
        c_ok[e] = q < NCH && ch < (c_isx[e] ? cin8 : cout8);
        c_col[e] = (unsigned)ch * 2u;
        c_lds[e] = row * (c_isx[e] ? PX : PG) + cc * 8;
        if (q >= NCH) c_lds[e] = -1;
    }
    auto load_idx = [&](int step, int (&idx)[NPT]) {
#pragma unroll
        for (int e = 0; e < NPT; ++e) {
            const long long p = s_lo + (long long)step * W16_SP + c_row[e];
            idx[e] = -1;
            if (p < s_hi && c_ok[e]) idx[e] = a.in_idx ? (c_isx[e] ? a.in_idx[p] : a.out_idx[p]) : (int)p;
        }
    };
    auto gather = [&](const int (&idx)[NPT], u32x4 (&v)[NPT]) {
#pragma unroll
        for (int e = 0; e < NPT; ++e) {
            const unsigned off = idx[e] >= 0 ? (unsigned)idx[e] * (c_isx[e] ? ldx_b : ldg_b) + c_col[e] : OOB;
            v[e] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(c_isx[e] ? rs_x : rs_g, (int)off, 0, 0));
        }
    };
    auto stage = [&](int buf, const u32x4 (&v)[NPT]) {
#pragma unroll
        for (int e = 0; e < NPT; ++e) {
            if (c_lds[e] < 0) continue;
            unsigned short* dst = (c_isx[e] ? s_x[buf] : s_g[buf]) + c_lds[e];
            *reinterpret_cast<u32x4*>(dst) = v[e];
        }
    };

    f32x4 acc[WA][WB];
#pragma unroll
    for (int ta = 0; ta < WA; ++ta)
#pragma unroll
        for (int tb = 0; tb < WB; ++tb) acc[ta][tb] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (steps > 0) {
        int idx_a[NPT], idx_b[NPT];
        u32x4 v[NPT];
        load_idx(0, idx_a);
        load_idx(1, idx_b);
        gather(idx_a, v);
        stage(0, v);
        __syncthreads();
        // per-lane LDS element offsets of the two transpose reads of tile 0 (see the row mapping above)
        const int rd_x = (kg * 4 + (i >> 2)) * PX + (wa * WA) * 16 + (i & 3) * 4;
        const int rd_g = (kg * 4 + (i >> 2)) * PG + (wb * WB) * 16 + (i & 3) * 4;
        for (int st = 0; st < steps; ++st) {
            const int buf = st & 1;
            if (st + 1 < steps) gather(idx_b, v);          // in flight behind this step's MFMAs
            load_idx(st + 2, idx_b);
            u32x4 fa[WA], fb[WB];
#pragma unroll
            for (int ta = 0; ta < WA; ++ta) {
                const unsigned short* src = s_x[buf] + rd_x + ta * 16;
                const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(src));
                const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(src + 16 * PX));
                const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                fa[ta] = u32x4{l2.x, l2.y, h2.x, h2.y};
            }
#pragma unroll
            for (int tb = 0; tb < WB; ++tb) {
                const unsigned short* src = s_g[buf] + rd_g + tb * 16;
                const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(src));
                const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(src + 16 * PG));
                const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                fb[tb] = u32x4{l2.x, l2.y, h2.x, h2.y};
            }
#pragma unroll
            for (int ta = 0; ta < WA; ++ta)
#pragma unroll
                for (int tb = 0; tb < WB; ++tb) mfma_step<T>(fa[ta], fb[tb], acc[ta][tb]);
            if (st + 1 < steps) stage(buf ^ 1, v);
            __syncthreads();
        }
    }
    // D layout: column = lane & 15 (co), row = kg * 4 + r (ci)
    float* out = a.out + ((size_t)wt.split * a.K + k) * (size_t)a.cin * a.cout;
#pragma unroll
    for (int ta = 0; ta < WA; ++ta)
#pragma unroll
        for (int tb = 0; tb < WB; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci_base + (wa * WA + ta) * 16 + kg * 4 + r, co = co_base + (wb * WB + tb) * 16 + i;
                if (ci < a.cin && co < a.cout) out[(size_t)ci * a.cout + co] = acc[ta][tb][r];
            }
}


// debug build only (-DPBN_WGRAD_TIMING, make -C pbnet_amd/csrc timing; scripts/wgrad_stamps.py): s_memtime of wave 0 of workgroup 0
// at 6 points of its first 32 steps
#ifdef PBN_WGRAD_TIMING
__device__ unsigned long long g_wgrad_stamp[32 * 6];
#define PBN_WSTAMP(ST, I) { if (blockIdx.x == 0 && wave == 0 && (ST) < 32) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_wgrad_stamp[(ST) * 6 + (I)] = t_; } }
#else
#define PBN_WSTAMP(ST, I)
#endif

// ---- k_wgrad_ring (round 3): k_wgrad16's tiling and LDS images behind a register ring of WR_DEPTH steps --------------
// k_wgrad16 keeps ONE step in flight: every step is two dependent memory latencies (pair indices, then rows) behind a
// barrier.  (Tried first and measured out: the gathers on the LDS-DMA path, `buffer_load_dwordx4 ... lds` with per-lane
// addresses into a swizzled dense image.  Its loads never stalled -- cycle stamps -- but one such instruction takes ~700
// cycles to ISSUE with 64 active lanes, ~3000 cycles per 16 KiB step; the register ring below spends 1500.)
// Workgroups of one split walk pair lists that name the SAME rows in the same ascending order: started together they pull
// the same few cache lines through one L2 channel at any moment (measured with the DMA form on the stride-16 level: 14 us
// per step), so every workgroup starts at its own point of its range and wraps around (`rot`): a fixed function of the
// workgroup, hence a fixed summation order.
// A thread owns row tid / 8 of the step and up to four 16-byte chunks of it (chunks c8, c8 + 8 of either image), so it
// needs two pair indices per step.  Issue order per step t: indices of step t + 2 D - 1, rows of step t + D (addresses from
// the indices of step t + D, loaded at step t + 1 - D); at the top of step s the rows of step s + 1 are waited for with
// vmcnt((D - 2) x loads per step) -- loads complete in issue order, so the same wait covers the indices of step s + D --
// and written to the LDS image that step s - 1 has just released.  All vector-memory loads of the loop are inline asm with
// "+v" destinations and hand-counted waits (see spconv_wave.hip for why hipcc's own counts drain the queue).
constexpr int WR_DEPTH = 4;
template <int S> struct StageTag { static constexpr int value = S; };

template <typename T, int WA, int WB, bool IDENT>
__global__ __launch_bounds__(256) void k_wgrad_ring(const WgradArgs a) {
    constexpr int CIT = 2 * WA, COT = 2 * WB;
    constexpr int PX = w16_pitch(CIT * 16), PG = w16_pitch(COT * 16);
    constexpr int CXC = CIT * 2, CGC = COT * 2;             // 16-byte chunks per row
    constexpr int NX = (CXC + 7) / 8, NG = (CGC + 7) / 8, NPT = NX + NG;
    constexpr int NID = IDENT ? 0 : 2, NLD = NID + NPT, D = WR_DEPTH;
    __shared__ __attribute__((aligned(16))) unsigned short s_x[2][W16_SP * PX];
    __shared__ __attribute__((aligned(16))) unsigned short s_g[2][W16_SP * PG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kg = lane >> 4;
    const WgradTile wt = wgrad_tile(a.strips, a.K, a.splits);
    if (!wt.live) return;
    const int k = wt.k;
    const int cib = wt.strip / a.co_groups, cob = wt.strip % a.co_groups;
    const int ci_base = cib * (CIT * 16), co_base = cob * (COT * 16);
    const int wa = wave >> 1, wb = wave & 1;

    long long p_lo, p_hi;
    if (a.seg_begin) { p_lo = (long long)a.seg_begin[k] * a.segment; p_hi = a.counts ? p_lo + a.counts[k] : (long long)a.seg_begin[k + 1] * a.segment; }
    else { p_lo = 0; p_hi = a.n_pairs; }
    const long long len = p_hi - p_lo;
    const long long per_split = ((len + a.splits - 1) / a.splits + W16_SP - 1) & ~(long long)(W16_SP - 1);
    const long long s_lo = min(p_lo + per_split * wt.split, p_hi), s_hi = min(s_lo + per_split, p_hi);
    const int rem = (int)(s_hi - s_lo);
    const int steps = (a.dbg & 1) ? 0 : (rem + W16_SP - 1) / W16_SP;
    const int steps_pad = (steps + D - 1) / D * D;          // the loop is unrolled over the register stages
    // staggered start (see above): issue step t works on block (t + rot) % steps
    const int rot = (a.dbg & 4) ? 0 : (int)(((long long)(k * a.strips + wt.strip) * steps) / (a.K * a.strips));
    auto block_of = [&](int t) { const int b = t + rot; return t >= steps ? -1 : (b >= steps ? b - steps : b); };

    constexpr unsigned NOREC = 0xfffffff0u, OOB = 0xfffffff8u;
    const auto rsrc = [](const void* p) {
        const unsigned long long v = (unsigned long long)(uintptr_t)p;
        return i32x4{(int)(unsigned)v, (int)(unsigned)(v >> 32), (int)NOREC, 0x00020000};
    };
    const i32x4 rs_x = rsrc(a.x), rs_g = rsrc(a.g), rs_ii = rsrc(a.in_idx), rs_io = rsrc(a.out_idx);
    const unsigned ldx_b = (unsigned)a.ld_x * 2u, ldg_b = (unsigned)a.ld_g * 2u;
    const int cin8 = (a.cin + 7) & ~7, cout8 = (a.cout + 7) & ~7;
    const int row = tid >> 3, c8 = tid & 7;
    unsigned c_col[NPT];
    int c_lds[NPT];
    bool c_ok[NPT];
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
        const bool isx = e < NX;
        const int cc = c8 + 8 * (isx ? e : e - NX);
        const int ch = (isx ? ci_base : co_base) + cc * 8;
        c_ok[e] = cc < (isx ? CXC : CGC) && ch < (isx ? cin8 : cout8);
        c_col[e] = (unsigned)ch * 2u;
        c_lds[e] = cc < (isx ? CXC : CGC) ? row * (isx ? PX : PG) + cc * 8 : -1;
    }
    u32x4 stg[D][NPT];
    unsigned idr[D][2];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        idr[d][0] = idr[d][1] = 0u;
#pragma unroll
        for (int e = 0; e < NPT; ++e) stg[d][e] = u32x4{0u, 0u, 0u, 0u};
    }
    // pair indices of step t -> idr[t % D]
    auto issue_idx = [&](int t, unsigned (&dst)[2]) {
        if constexpr (!IDENT) {
            const int blk = block_of(t);
            const int pr = blk * W16_SP + row;
            const unsigned off = (blk >= 0 && pr < rem) ? (unsigned)(s_lo + pr) * 4u : OOB;
            asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(dst[0]) : "v"(off), "s"(rs_ii));
            asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(dst[1]) : "v"(off), "s"(rs_io));
        }
    };
    auto issue_rows = [&](int t, const unsigned (&id)[2], u32x4 (&dst)[NPT]) {
        const int blk = block_of(t);
        const int pr = blk * W16_SP + row;
        const bool live = blk >= 0 && pr < rem && !(a.dbg & 8);
        int ri, ro;
        if constexpr (IDENT) { ri = ro = (int)s_lo + pr; }
        else { ri = (int)id[0]; ro = (int)id[1]; }
#pragma unroll
        for (int e = 0; e < NPT; ++e) {
            const bool isx = e < NX;
            const int r = isx ? ri : ro;
            const unsigned off = (live && c_ok[e] && r >= 0) ? (unsigned)r * (isx ? ldx_b : ldg_b) + c_col[e] : OOB;
            if (isx) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(dst[e]) : "v"(off), "s"(rs_x));
            else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(dst[e]) : "v"(off), "s"(rs_g));
        }
    };
    auto to_lds = [&](int buf, const u32x4 (&v)[NPT]) {
#pragma unroll
        for (int e = 0; e < NPT; ++e) {
            if (c_lds[e] < 0) continue;
            unsigned short* dst = (e < NX ? s_x[buf] : s_g[buf]) + c_lds[e];
            *reinterpret_cast<u32x4*>(dst) = v[e];
        }
    };

    f32x4 acc[WA][WB];
#pragma unroll
    for (int ta = 0; ta < WA; ++ta)
#pragma unroll
        for (int tb = 0; tb < WB; ++tb) acc[ta][tb] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (steps > 0) {
        // prologue (drained: the loop's first waits are then trivially met): indices of steps 0 .. D - 1, rows of the same
        // steps + indices of steps D .. 2 D - 2, image of step 0
#pragma unroll
        for (int d = 0; d < D; ++d) issue_idx(d, idr[d]);
#pragma unroll
        for (int d = 0; d < D; ++d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(idr[d][0]), "+v"(idr[d][1]));
#pragma unroll
        for (int d = 0; d < D; ++d) issue_rows(d, idr[d], stg[d]);
#pragma unroll
        for (int d = 0; d < D - 1; ++d) issue_idx(D + d, idr[d]);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(idr[d][0]), "+v"(idr[d][1]));
#pragma unroll
            for (int e = 0; e < NPT; ++e) asm volatile("" : "+v"(stg[d][e]));
        }
        to_lds(0, stg[0]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");

        const int rd_x = (kg * 4 + (i >> 2)) * PX + (wa * WA) * 16 + (i & 3) * 4;
        const int rd_g = (kg * 4 + (i >> 2)) * PG + (wb * WB) * 16 + (i & 3) * 4;
        auto body = [&](auto tag, int st) {
            constexpr int S = decltype(tag)::value;              // = st % D
            constexpr int S1 = (S + 1) % D, SP = (S + D - 1) % D;
            const int buf = S & 1;                               // D is even: st & 1 == S & 1
            PBN_WSTAMP(st, 0);
            // rows of step st + 1 (and the indices of step st + D behind them) have landed
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(idr[S][0]), "+v"(idr[S][1]) : "n"((D - 2) * NLD));
#pragma unroll
            for (int e = 0; e < NPT; ++e) asm volatile("" : "+v"(stg[S1][e]));
            PBN_WSTAMP(st, 1);
            to_lds(buf ^ 1, stg[S1]);
            PBN_WSTAMP(st, 2);
            issue_idx(st + 2 * D - 1, idr[SP]);
            issue_rows(st + D, idr[S], stg[S]);
            PBN_WSTAMP(st, 3);
            u32x4 fa[WA], fb[WB];
#pragma unroll
            for (int ta = 0; ta < WA; ++ta) {
                const unsigned short* src = s_x[buf] + rd_x + ta * 16;
                const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(src));
                const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(src + 16 * PX));
                const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                fa[ta] = u32x4{l2.x, l2.y, h2.x, h2.y};
            }
#pragma unroll
            for (int tb = 0; tb < WB; ++tb) {
                const unsigned short* src = s_g[buf] + rd_g + tb * 16;
                const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(src));
                const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(src + 16 * PG));
                const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                fb[tb] = u32x4{l2.x, l2.y, h2.x, h2.y};
            }
            PBN_WSTAMP(st, 4);
            if (!(a.dbg & 16)) {
#pragma unroll
                for (int ta = 0; ta < WA; ++ta)
#pragma unroll
                    for (int tb = 0; tb < WB; ++tb) mfma_step<T>(fa[ta], fb[tb], acc[ta][tb]);
            }
            PBN_WSTAMP(st, 5);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
        };
        static_assert(D == 4, "the loop below is unrolled over four register stages");
        for (int st = 0; st < steps_pad; st += D) {
            body(StageTag<0>{}, st);
            body(StageTag<1>{}, st + 1);
            body(StageTag<2>{}, st + 2);
            body(StageTag<3>{}, st + 3);
        }
        // the loads issued for the steps past the end still target these registers
#pragma unroll
        for (int d = 0; d < D; ++d) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(idr[d][0]), "+v"(idr[d][1]));
#pragma unroll
            for (int e = 0; e < NPT; ++e) asm volatile("" : "+v"(stg[d][e]));
        }
    }
    if (a.dbg & 2) return;
    float* out = a.out + ((size_t)wt.split * a.K + k) * (size_t)a.cin * a.cout;
#pragma unroll
    for (int ta = 0; ta < WA; ++ta)
#pragma unroll
        for (int tb = 0; tb < WB; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci_base + (wa * WA + ta) * 16 + kg * 4 + r, co = co_base + (wb * WB + tb) * 16 + i;
                if (ci < a.cin && co < a.cout) out[(size_t)ci * a.cout + co] = acc[ta][tb][r];
            }
}

template <typename T, int WA, int WB>
void launch_ring_c(const WgradArgs& a, dim3 grid, hipStream_t stream) {
    if (a.in_idx) hipLaunchKernelGGL((k_wgrad_ring<T, WA, WB, false>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((k_wgrad_ring<T, WA, WB, true>), grid, dim3(256), 0, stream, a);
}
template <typename T, int WA>
void launch_ring_b(const WgradArgs& a, int wb, dim3 grid, hipStream_t stream) {
    switch (wb) {
        case 1: launch_ring_c<T, WA, 1>(a, grid, stream); break;
        case 2: launch_ring_c<T, WA, 2>(a, grid, stream); break;
        case 3: launch_ring_c<T, WA, 3>(a, grid, stream); break;
        default: launch_ring_c<T, WA, 4>(a, grid, stream); break;
    }
}
template <typename T>
void launch_ring(const WgradArgs& a, int wa, int wb, dim3 grid, hipStream_t stream) {
    switch (wa) {
        case 1: launch_ring_b<T, 1>(a, wb, grid, stream); break;
        case 2: launch_ring_b<T, 2>(a, wb, grid, stream); break;
        case 3: launch_ring_b<T, 3>(a, wb, grid, stream); break;
        default: launch_ring_b<T, 4>(a, wb, grid, stream); break;
    }
}

template <typename T, int WA>
void launch16_b(const WgradArgs& a, int wb, dim3 grid, hipStream_t stream) {
    switch (wb) {
        case 1: hipLaunchKernelGGL((k_wgrad16<T, WA, 1>), grid, dim3(256), 0, stream, a); break;
        case 2: hipLaunchKernelGGL((k_wgrad16<T, WA, 2>), grid, dim3(256), 0, stream, a); break;
        case 3: hipLaunchKernelGGL((k_wgrad16<T, WA, 3>), grid, dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL((k_wgrad16<T, WA, 4>), grid, dim3(256), 0, stream, a); break;
    }
}
template <typename T>
void launch16(const WgradArgs& a, int wa, int wb, dim3 grid, hipStream_t stream) {
    switch (wa) {
        case 1: launch16_b<T, 1>(a, wb, grid, stream); break;
        case 2: launch16_b<T, 2>(a, wb, grid, stream); break;
        case 3: launch16_b<T, 3>(a, wb, grid, stream); break;
        default: launch16_b<T, 4>(a, wb, grid, stream); break;
    }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ partial, int splits, long long n,
                                                     float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += partial[(size_t)s * n + i];
    out[i] = v;
}

}  // namespace
}  // namespace pbn

using namespace pbn;

#ifdef PBN_WGRAD_TIMING
extern "C" int pbn_wgrad_timing_read(unsigned long long* out) {
    PBN_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(pbn::g_wgrad_stamp), sizeof(unsigned long long) * 32 * 6));
    return PBN_OK;
}
#endif

extern "C" size_t pbn_spconv_wgrad_workspace_bytes(int n_offsets, int cin, int cout) {
    if (n_offsets < 1 || cin < 1 || cout < 1) return 0;
    return (size_t)64 * n_offsets * cin * cout * sizeof(float);   // up to 64 pair splits
}

extern "C" int pbn_spconv_wgrad(const void* x, int ld_x, const void* g, int ld_g, int dtype, const int32_t* in_idx,
                                const int32_t* out_idx, const int32_t* seg_begin, const int32_t* pair_counts, int segment, int n_pairs_total,
                                int n_offsets, int cin, int cout, float* dw, void* workspace, size_t workspace_bytes,
                                pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_offsets < 1 || cin < 1 || cout < 1 || n_pairs_total < 0 || !dw) return PBN_ERR_ARG;
    if ((in_idx == nullptr) != (out_idx == nullptr)) return PBN_ERR_ARG;
    if (seg_begin && (segment < 1 || !in_idx)) return PBN_ERR_ARG;
    if (!seg_begin && n_offsets != 1) return PBN_ERR_ARG;
    const long long n_out = (long long)n_offsets * cin * cout;
    if (n_pairs_total == 0) {
        { const int frc_ = fill_bytes(dw, 0, sizeof(float) * (size_t)n_out, stream); if (frc_ != PBN_OK) return frc_; }
        return PBN_OK;
    }
    if (!x || !g) return PBN_ERR_ARG;
    WgradArgs a;
    a.x = x; a.g = g; a.in_idx = in_idx; a.out_idx = out_idx; a.seg_begin = seg_begin;
    a.counts = seg_begin ? pair_counts : nullptr;
    a.dbg = getenv("PBN_WGRAD_DBG") ? atoi(getenv("PBN_WGRAD_DBG")) : 0;   // measurement only: 1 = no main loop, 2 = no stores
    a.ld_x = ld_x; a.ld_g = ld_g; a.cin = cin; a.cout = cout; a.K = n_offsets; a.segment = segment; a.n_pairs = n_pairs_total;
    // 16-bit slabs whose rows can be read in 16-byte chunks take the bf16/f16 matrix-core forms: k_wgrad_ring (default) or
    // k_wgrad16 (PBN_WGRAD_FORM=16, the round-2 kernel, kept as the cross-check); PBN_WGRAD_FORM=32: always the f32-MFMA form
    static const int form_env = getenv("PBN_WGRAD_FORM") ? atoi(getenv("PBN_WGRAD_FORM")) : 0;
    const bool form16 = form_env != 32 && dtype != PBN_F32 && (ld_x % 8) == 0 && (ld_g % 8) == 0 &&
                        ld_x >= ((cin + 7) & ~7) && ld_g >= ((cout + 7) & ~7) && (((uintptr_t)x | (uintptr_t)g) & 15) == 0;
    const bool ring = form16 && form_env != 16;
    int wa = 0, wb = 0, strips;
    bool small_level = false;
    if (form16) {
        const int cit = cdiv(cin, 16), cot = cdiv(cout, 16);
        wa = cit >= 7 ? 4 : (cit + 1) / 2;            // waves are 2 x 2: a workgroup covers 2 wa x 2 wb tiles
        wb = cot >= 7 ? 4 : (cot + 1) / 2;
        // few pairs per offset (the stride-8/16 levels): quarter tiles instead of pair splits -- 4x the workgroups with no
        // partial slabs and no reduce launch (measured, stride-16 256->256: 23 -> 13 us; stride-8: 35 us either way)
        small_level = ring && n_pairs_total / n_offsets < 3000 && cdiv(cit, 2 * wa) * cdiv(cot, 2 * wb) * n_offsets < 256;
        const int maxt = getenv("PBN_WGRAD_MAXT") ? atoi(getenv("PBN_WGRAD_MAXT")) : (small_level ? 2 : 4);
        if (wa > maxt) wa = maxt;
        if (wb > maxt) wb = maxt;
        a.co_groups = cdiv(cot, 2 * wb);
        strips = cdiv(cit, 2 * wa) * a.co_groups;
    } else {
        a.co_groups = cdiv(cout, NTW * 16);
        strips = cdiv(cin, 16) * a.co_groups;
    }
    // pair splits: enough workgroups for the chip, enough pairs per workgroup to amortise its prologue, bounded by the
    // workspace.  k_wgrad_ring: ~1024 workgroups of >= 256 pairs (8 steps); the one-step-in-flight kernels: ~2048 of >= 512
    const int want_wgs = getenv("PBN_WGRAD_WGS") ? atoi(getenv("PBN_WGRAD_WGS")) : 0;
    const int min_pairs_env = getenv("PBN_WGRAD_MIN_PAIRS") ? atoi(getenv("PBN_WGRAD_MIN_PAIRS")) : 0;
    const long long target = want_wgs > 0 ? want_wgs : (ring ? 1024 : 2048);
    const long long min_pairs = min_pairs_env > 0 ? min_pairs_env : (ring ? 256 : 512);
    const long long pairs_per_offset = n_pairs_total / n_offsets + 1;
    long long splits = target / ((long long)strips * n_offsets) + 1;
    if (splits > pairs_per_offset / min_pairs + 1) splits = pairs_per_offset / min_pairs + 1;
    // every split writes and re-reads a dW-sized partial: at most ~32 MB of partials (256->256 cubes: 4 splits)
    const long long by_traffic = (32LL << 20) / (long long)(sizeof(float) * (size_t)n_out) + 1;
    if (ring && splits > by_traffic) splits = by_traffic;
    if (small_level && want_wgs <= 0) splits = 1;
    const long long by_ws = workspace ? (long long)(workspace_bytes / (sizeof(float) * (size_t)n_out)) : 1;
    if (splits > by_ws) splits = by_ws;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
    a.splits = (int)splits;
    a.out = splits > 1 ? (float*)workspace : dw;
    a.strips = strips;
    const dim3 grid((unsigned)wgrad_grid(strips, n_offsets, (int)splits));   // wgrad_tile()
    if (ring) {
        if (dtype == PBN_BF16) launch_ring<__hip_bfloat16>(a, wa, wb, grid, stream);
        else launch_ring<__half>(a, wa, wb, grid, stream);
    } else if (form16) {
        if (dtype == PBN_BF16) launch16<__hip_bfloat16>(a, wa, wb, grid, stream);
        else launch16<__half>(a, wa, wb, grid, stream);
    } else {
        switch (dtype) {
            case PBN_F32: hipLaunchKernelGGL(k_wgrad<float>, grid, dim3(256), 0, stream, a); break;
            case PBN_BF16: hipLaunchKernelGGL(k_wgrad<__hip_bfloat16>, grid, dim3(256), 0, stream, a); break;
            case PBN_F16: hipLaunchKernelGGL(k_wgrad<__half>, grid, dim3(256), 0, stream, a); break;
            default: return PBN_ERR_ARG;
        }
    }
    if (splits > 1)
        hipLaunchKernelGGL(k_wgrad_reduce, dim3(cdiv(n_out, 256)), dim3(256), 0, stream, (const float*)workspace, (int)splits,
                           n_out, dw);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

// The same launch behind the checks its 32-bit buffer offsets rely on (the raw entry above takes no row counts): a slab of
// 4 GiB or more would wrap `row * ld * esize`, pair lists of 2^30 entries `index * 4`; device-built lists (pbn_rulebook_pair_fill_dev /
// pairs_multi) are NOT -1 padded, so walking them without their pair counts reads whatever lies behind the last pair.
extern "C" int pbn_spconv_wgrad_checked(const void* x, int ld_x, long long n_x_rows, const void* g, int ld_g, long long n_g_rows,
                                        int dtype, const int32_t* in_idx, const int32_t* out_idx, const int32_t* seg_begin,
                                        const int32_t* pair_counts, int lists_padded, int segment, int n_pairs_total, int n_offsets,
                                        int cin, int cout, float* dw, void* workspace, size_t workspace_bytes, pbn_stream_t stream) {
    if (n_x_rows < 0 || n_g_rows < 0 || ld_x < 1 || ld_g < 1) return PBN_ERR_ARG;
    const long long es = dtype == PBN_F32 ? 4 : 2;
    if (n_x_rows * (long long)ld_x * es >= (1LL << 32) || n_g_rows * (long long)ld_g * es >= (1LL << 32)) return PBN_ERR_RANGE;
    if ((long long)n_pairs_total >= (1LL << 30)) return PBN_ERR_RANGE;
    if (seg_begin && !pair_counts && !lists_padded) return PBN_ERR_ARG;
    if (!in_idx && n_pairs_total > (n_x_rows < n_g_rows ? n_x_rows : n_g_rows)) return PBN_ERR_ARG;   // identity pairs: one per row
    return pbn_spconv_wgrad(x, ld_x, g, ld_g, dtype, in_idx, out_idx, seg_begin, pair_counts, segment, n_pairs_total, n_offsets, cin,
                            cout, dw, workspace, workspace_bytes, stream);
}
