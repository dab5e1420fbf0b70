// spconv_wave.hip -- wave-autonomous implicit-GEMM sparse convolution for gfx950 (wave64, MFMA 16x16).
//
// Same arithmetic and the same operand layouts as spconv.hip (MinkowskiConvolution / ConvolutionTranspose / Linear
// forward, /root/reference/network/Mink.py:221-288,293-350):  out[o,:] = epilogue( sum_k in[nbr[o,k],:] @ W[k] ),
// but a different machine mapping, built for the levels spconv.hip serves badly:
//
//   * every WAVE runs its own software-pipelined loop over "units" (one 4-vector step of the flattened (offset, channel)
//     axis): NT weight fragments and NF gathered row fragments go global/L2 -> registers through buffer loads, one unit
//     ahead of the MFMAs that consume them.  There is NO barrier, NO LDS traffic and NO M0-addressed DMA in the main
//     loop; hipcc's own s_waitcnt counting is exact here because the loop body is straight-line code;
//   * K-SPLIT mode (coarse levels: a few hundred to a few thousand rows, 27 x 128..384 deep reductions): the KW waves of
//     a workgroup share ONE row tile and take the populated units round-robin; their fp32 partial tiles are summed
//     through LDS in a fixed order and the epilogue runs once.  Parallelism over the reduction axis is bought INSIDE the
//     workgroup: no fp32 partial slabs in HBM/L2, no second (reduce) launch, deterministic;
//   * ROW-SPLIT mode: each wave owns NF*16 rows of the workgroup's tile and walks all of ITS populated units;
//   * units whose offsets have no neighbour among the wave's rows are never visited (offset population by ballot).
//
// Bound: per-CU vector-memory return path (64 B/clk) -- each wave-unit moves (NF + NT) KiB for NF*NT MFMAs.
#include <cstdlib>
#include <cstring>
#include "spconv_common.h"

namespace pbn {
namespace {

constexpr unsigned OOB = 0x80000000u;   // >= num_records of either resource: the load returns zeros
constexpr unsigned UNIT_NONE = 0xffffu;        // a step past the end (n_steps <= 0xfffe): the unit loads nothing and adds zeros

constexpr int MAX_DEPTH = 8;

// Cycle stamps (debug build only: make timing -> libpbnet_hip_timing.so, scripts/wave_timing.py): lane 0 of every wave of
// the first WT_BLOCKS workgroups records s_memtime at 8 points, stamp 7 = s_memrealtime (100 MHz, comparable across CUs).
#ifdef PBN_CONV_TIMING
constexpr int WT_BLOCKS = 1024;
__device__ unsigned long long g_wave_timing[WT_BLOCKS * 8 * 8 + 8];
#define PBN_WSTAMP(I)                                                                                                 \
    if (lane == 0 && wt_blk < WT_BLOCKS && wave < 8)                                                                  \
        g_wave_timing[(wt_blk * 8 + wave) * 8 + (I)] = (I) == 7 ? wall_clock64() : __builtin_readcyclecounter();
#else
#define PBN_WSTAMP(I)
#endif
// units in flight per wave.  Measured on the bench scene (scripts/probe_wave.py): 2 beats the deeper pipelines on every
// level -- occupancy (registers) and the issue cost of the extra in-flight loads outweigh the latency they would hide
#ifndef PBN_WAVE_DEPTH
#define PBN_WAVE_DEPTH 2
#endif
// Row gathers in a QUAD-COALESCED lane order: lane 4 r + c fetches 16-byte chunk c of row r of the fragment, so the four
// lanes of a quad read one contiguous 64-byte piece of one row (16 cache-line accesses per wave-load); the MFMA operand
// order (lane 16 g + r holds chunk g of row r: consecutive lanes = consecutive ROWS, 64 line accesses per wave-load) is
// restored on the way in by four ds_bpermute_b32 per fragment (LDS crossbar, no LDS memory).  scripts/micro/
// gather_layout.hip: an L2-resident gather costs 62 cycles of the CU's vector-memory path per wave-load in operand order, 32
// quad-coalesced (a contiguous 1 KiB load: 30).
#ifndef PBN_WAVE_XCOAL
#define PBN_WAVE_XCOAL 1
#endif
constexpr int pipe_depth(int nf, int nt) {
    (void)nf; (void)nt;
    return PBN_WAVE_DEPTH;
}

template <int NF, int NT>
struct Stage {
    u32x4 w[NT];
    u32x4 x[NF];
};

// Every vector-memory load of the main loop is issued from inline asm and waited for with a hand-counted s_waitcnt: loads
// complete in issue order, so "unit i has landed" is a fixed count of the loads issued behind it.  (hipcc's own wait-count
// insertion merges the scoreboard states of the loop entry and the back edge and drains the whole queue at the top of every
// iteration -- measured: 940 cycles per unit with the builtin loads, i.e. no overlap at all.)
__device__ __forceinline__ void buf_load_asm(u32x4& dst, const i32x4& rs, unsigned voff, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");   // "+v": refilled in place, never through a copy
}

template <typename T>
__device__ __forceinline__ void epilogue_store(const ConvArgs& a, f32x4 v, int orow, int c0) {
    if (a.scale) {
        const float4 sc = *reinterpret_cast<const float4*>(a.scale + c0);
        v[0] *= sc.x; v[1] *= sc.y; v[2] *= sc.z; v[3] *= sc.w;
    }
    if (a.shift) {
        const float4 sh = *reinterpret_cast<const float4*>(a.shift + c0);
        v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w;
    }
    if (a.residual) v += load4<T>(reinterpret_cast<const T*>(a.residual) + (size_t)orow * a.ld_res + c0);
    if (a.relu) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    store4<T>(reinterpret_cast<T*>(a.out) + (size_t)orow * a.ld_out + c0, v);
}

// channel tiles per LDS reduction round of a K-split workgroup: all of them when the KW partial tiles fit in 64 KiB
#ifndef PBN_WAVE_RED_KB
#define PBN_WAVE_RED_KB 64
#endif
constexpr int ksplit_round_tiles(int kw, int tm, int nt) {
    return (kw * tm * nt * 64 <= PBN_WAVE_RED_KB * 1024) ? nt : (nt < 2 ? nt : 2);
}
// row pitch of the reduction buffer in floats: + 4, so that the 16 rows a ds_write_b128 lane group stores in one LDS cycle
// start 36 dwords apart = in 16 different 4-bank groups (at the natural pitch of 32 dwords they alternate between two)
constexpr int ksplit_red_pitch(int ntb) { return ntb * 16 + 4; }

// blockIdx -> (row tile, channel-tile group).  Row-major launches keep contiguous row-tile ranges per XCD (neighbouring
// tiles gather overlapping rows); WEIGHT-major launches (a.wmajor: the packed weights outweigh the input slab, i.e. the
// coarse levels) give every XCD its own channel-tile groups instead, so that a weight byte enters ONE XCD's L2 instead of
// all eight.  Both are bijections of a 1-D grid onto the (row tile, group) pairs they cover; blocks past the end exit.
struct TileMap { int row_tile, group; bool valid; };
__device__ __forceinline__ TileMap map_block(const ConvArgs& a, int n_row_tiles, int n_groups) {
    TileMap m;
    const int b = blockIdx.x;
    if (!a.wmajor) {
        m.group = b / n_row_tiles;
        m.row_tile = xcd_tile(b - m.group * n_row_tiles, n_row_tiles);
        m.valid = m.group < n_groups;
        return m;
    }
    const int xcd = b & 7, idx = b >> 3;
    if (n_groups >= 8) {                 // n_groups % 8 == 0 (checked at launch): XCD x owns groups x, x + 8, ...
        const int gl = idx / n_row_tiles;
        m.group = xcd + 8 * gl;
        m.row_tile = idx - gl * n_row_tiles;
        m.valid = m.group < n_groups;
    } else {                             // n_groups in {1, 2, 4}: 8 / n_groups XCDs share a group and split its row tiles
        const int share = 8 / n_groups;
        m.group = xcd % n_groups;
        m.row_tile = (xcd / n_groups) + share * idx;
        m.valid = m.row_tile < n_row_tiles;
    }
    return m;
}

// KLIST (K-split only, round 4): the workgroup's waves take the POPULATED steps of the tile round-robin (every wave builds the same
// list; wave w consumes entries w, w + KW, ...) instead of every step -- for the k = 5 stems, where a 32-row tile has a
// neighbour at 30-40 % of the 125 offsets (the coarse levels, 93-98 % populated, keep the computed steps: no list to build).
template <typename T, int NF, int NT, int KW, bool KSPLIT, bool KLIST = false>
__global__ __launch_bounds__(KW * 64) void k_spconv_wave(const ConvArgs a) {
    static_assert(!KLIST || KSPLIT, "the shared step list is a K-split feature");
    static_assert(Tr<T>::ELEMS * sizeof(T) == 16, "one gather vector is 16 bytes");
    constexpr int RW = NF * 16;                          // rows per wave
    constexpr int TM = KSPLIT ? RW : KW * RW;            // rows per workgroup
    constexpr int NTB = ksplit_round_tiles(KW, TM, NT);  // channel tiles per LDS reduction round (K-split)
    constexpr int TPB = KW * 64;
    constexpr int B = pipe_depth(NF, NT);                // units in flight per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = a.K;
    const int KS = K | 1;
    int* s_nbr = reinterpret_cast<int*>(smem);                                        // TM * KS
    // row-split: per-wave unit lists; K-split: units are computed (every step is visited), the space holds the epilogue
    // constants (scale | shift of this workgroup's NT*16 channels) and the reduction buffer
    unsigned* s_units = reinterpret_cast<unsigned*>(s_nbr + ((TM * KS + 3) & ~3));    // row-split: KW * (n_steps + 2 * MAX_DEPTH)
    const int units_pitch = (KSPLIT && !KLIST) ? 0 : a.n_steps + 2 * MAX_DEPTH + (KLIST ? 64 : 0);
    int* s_none = reinterpret_cast<int*>(s_units + ((KW * units_pitch + 3) & ~3));    // one word, -1: "no neighbour" for units past the end
    float* s_ss = reinterpret_cast<float*>(s_none + 4);                               // K-split: 2 * NT*16 floats
    float* s_red = s_ss + 2 * NT * 16;                                                // K-split: KW * TM * NTB*16 floats

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PBN_CONV_TIMING
    const int wt_blk = blockIdx.x;
    if (tid == 0 && wt_blk == 0) { g_wave_timing[WT_BLOCKS * 64] = gridDim.x; g_wave_timing[WT_BLOCKS * 64 + 1] = gridDim.y; }
#endif
    PBN_WSTAMP(7);
    PBN_WSTAMP(0);
    const int n_groups = a.ntiles_total / NT;
    u32x4 pf_sink = prefetch_next_weights(a, blockIdx.x, gridDim.x, tid, TPB);
    // row tiles that HAVE rows: with a device-side count the grid is sized by a capacity, and mapping the blocks over the grid's
    // tiles would hand the last XCDs only empty ones (round 5)
    const int n_rows_now = a.n_out_dev ? min(*a.n_out_dev, a.n_out) : a.n_out;
    if (n_rows_now <= 0) { prefetch_drain(pf_sink); return; }
    const TileMap tm = map_block(a, (n_rows_now + TM - 1) / TM, n_groups);
    if (!tm.valid) { prefetch_drain(pf_sink); return; }
    const int row0 = tm.row_tile * TM;
    const int tile0 = tm.group * NT;
    const int g = lane >> 4, rl = lane & 15;

    const unsigned long long in_addr = (unsigned long long)a.in, w_addr = (unsigned long long)a.w;
    const i32x4 rs_in = {(int)(unsigned)in_addr, (int)(unsigned)(in_addr >> 32), (int)a.in_bytes, 0x00020000};
    const i32x4 rs_w = {(int)(unsigned)w_addr, (int)(unsigned)(w_addr >> 32), (int)a.w_bytes, 0x00020000};
    const unsigned ld_bytes = (unsigned)a.ld_in * (unsigned)sizeof(T);
    // second source (a BasicBlock's 1x1 shortcut folded into its second convolution): steps n_main.. read row o of in2
    const unsigned long long in2_addr = (unsigned long long)a.in2;
    const i32x4 rs_in2 = {(int)(unsigned)in2_addr, (int)(unsigned)(in2_addr >> 32), (int)a.in2_bytes, 0x00020000};
    const unsigned ld2_bytes = (unsigned)a.ld_in2 * (unsigned)sizeof(T);
    const int n_main = a.in2 ? a.n_main : a.n_steps;
    const unsigned w_lane = (unsigned)lane * 16u;
    const unsigned w_tile0 = (unsigned)tile0 * 1024u;
    const unsigned w_step_bytes = (unsigned)a.ntiles_total * 1024u;
    const int vpo = a.vpo;
    const bool wide = (vpo & 3) == 0;
    const int spo = wide ? (vpo >> 2) : 1;
    const int vshift = (vpo == 2) ? 1 : 0;
    const float inv_vpo = 1.0f / (float)vpo;
    const int n_steps = a.n_steps;
    unsigned* my_units = s_units + wave * units_pitch;

    // step of this wave's unit i (>= n_steps: past the end, the unit adds zeros).  K-split: every step is visited, wave w
    // takes steps w, w + KW, ...; row-split: the wave's list of populated steps in LDS
    auto step_of = [&](int i) -> int {
        if constexpr (KLIST) return __builtin_amdgcn_readfirstlane((int)my_units[i * KW + wave]);
        else if constexpr (KSPLIT) return i * KW + wave;
        else return __builtin_amdgcn_readfirstlane((int)my_units[i]);
    };
    // the loads of one unit, branch-free (a branch in the loop body makes hipcc's wait-count insertion drain the queue at
    // every join): NT weight fragments (wave-uniform address + lane * 16) ...
    auto issue_w = [&](Stage<NF, NT>& st, int step) {
        const unsigned w_soff = (unsigned)(step < n_steps ? step : 0) * w_step_bytes + w_tile0;
#ifdef PBN_CONV_TIMING
        if (a.dbg & 2) return;                                   // ablation: no weight loads
        const unsigned w_lane = (a.dbg & 16) ? OOB : (unsigned)lane * 16u;   // ablation: weight loads out of range
#endif
#pragma unroll
        for (int t = 0; t < NT; ++t) buf_load_asm(st.w[t], rs_w, w_lane, w_soff + (unsigned)t * 1024u);
    };

    Stage<NF, NT> st[B];
#pragma unroll
    for (int s = 0; s < B; ++s) {
#pragma unroll
        for (int t = 0; t < NT; ++t) st[s].w[t] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int f = 0; f < NF; ++f) st[s].x[f] = u32x4{0u, 0u, 0u, 0u};
    }
    if constexpr (KSPLIT) {
        // epilogue constants of this workgroup's channels -> LDS (read back after the reduction)
        if (tid < NT * 32) {
            const int c = tile0 * 16 + (tid < NT * 16 ? tid : tid - NT * 16);
            const float* src = tid < NT * 16 ? a.scale : a.shift;
            s_ss[tid] = src ? src[c] : (tid < NT * 16 ? 1.0f : 0.0f);
        }
    }
    const int n = a.n_out_dev ? min(*a.n_out_dev, a.n_out) : a.n_out;
    if (row0 >= n) { prefetch_drain(pf_sink); return; }
    if (tid == 0) *s_none = -1;

    // ---- rulebook tile -> LDS ----
    if (a.nbr && !a.row_perm && row0 + TM <= n && KS == K && ((TM * K) & 3) == 0) {
        const int4* src = reinterpret_cast<const int4*>(a.nbr + (size_t)row0 * K);
        int4* dst = reinterpret_cast<int4*>(s_nbr);
        for (int e = tid; e < (TM * K) >> 2; e += TPB) dst[e] = src[e];
    } else {
        const float inv_k = 1.0f / (float)K;
        for (int e = tid; e < TM * K; e += TPB) {
            const int r = (int)(((float)e + 0.5f) * inv_k), k = e - r * K;
            const int p = row0 + r;
            int v = -1;
            if (p < n) {
                const int row = a.row_perm ? a.row_perm[p] : p;
                v = a.nbr ? a.nbr[(size_t)row * K + k] : row;
            }
            s_nbr[r * KS + k] = v;
        }
    }
    prefetch_drain(pf_sink);          // behind the rulebook loads: nothing new to wait for
    __syncthreads();
    PBN_WSTAMP(1);

    const int wrow0 = KSPLIT ? 0 : wave * RW;
    int n_units;
    if constexpr (KSPLIT && !KLIST) {
        // coarse levels: 93-98 % of the (64-row tile, offset) pairs are populated (scripts/analyze_rulebook.py) -- visiting
        // every step costs a few per cent of MFMA work and takes the population scan and the unit list out of the prologue
        // of a launch whose whole main loop is a few microseconds
        n_units = a.n_steps > wave ? (a.n_steps - wave + KW - 1) / KW : 0;
    } else {
        // ---- which offsets are populated among this wave's rows / the tile's rows (K <= 128: two ballots) ----
        constexpr int SCAN = KSPLIT ? TM : RW;
        bool any0 = false, any1 = false;
        if (lane < K)
            for (int r = 0; r < SCAN; ++r) any0 |= s_nbr[(wrow0 + r) * KS + lane] >= 0;
        if (64 + lane < K)
            for (int r = 0; r < SCAN; ++r) any1 |= s_nbr[(wrow0 + r) * KS + 64 + lane] >= 0;
        const unsigned long long pop0 = __ballot(any0), pop1 = __ballot(any1);
        auto populated = [&](int k) -> bool { return ((k < 64 ? pop0 >> k : pop1 >> (k - 64)) & 1ull) != 0ull; };
        // ---- this wave's unit list: populated steps, in order ----
        int base = 0;
        for (int s0 = 0; s0 < a.n_steps; s0 += 64) {
            const int s = s0 + lane;
            bool ok = false;
            int ko = 0;
            if (s < a.n_steps) {
                if (s >= n_main) {
                    ok = true;                   // the second source: every row has itself
                } else if (wide) {
                    ko = s / spo;
                    ok = populated(ko);
                } else {                      // a step spans 4 / vpo offsets; lane group g reads offset (4 s + g) >> vshift
                    ko = (s * 4) >> vshift;
                    const int span = 4 >> vshift;
                    for (int q = 0; q < span; ++q)
                        if (ko + q < K) ok |= populated(ko + q);
                }
            }
            const unsigned long long m = __ballot(ok);
            if (ok) my_units[base + __popcll(m & ((1ULL << lane) - 1ULL))] = (unsigned)s;
            base += __popcll(m);
        }
        // sentinels: the pipeline below always has B units in flight and needs no tail handling
        constexpr int NSENT = (2 * B + 2) * (KLIST ? KW : 1);
        static_assert(NSENT <= 64, "one store per lane writes the sentinels");
        if (lane < NSENT) my_units[base + lane] = UNIT_NONE;   // steps past the end
        if constexpr (KLIST) n_units = base > wave ? (base - wave + KW - 1) / KW : 0;   // this wave's share of the tile's list
        else n_units = base;
        // the list is wave-private and a wave's LDS operations execute in order: the reads below follow the writes above
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    PBN_WSTAMP(2);

    f32x4 acc[NF][NT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int my_row = wrow0 + (PBN_WAVE_XCOAL ? (lane >> 2) : rl);     // the row this lane gathers for (per fragment)
    const int my_chunk = PBN_WAVE_XCOAL ? (lane & 3) : g;                 // ... and its 16-byte chunk of the step
    const int xpose_addr = (4 * (lane & 15) + (lane >> 4)) * 4;          // operand-order lane 16 g + r pulls from lane 4 r + g
    // ... and NF gathered row fragments (one 16-byte vector of one neighbour row per lane; no neighbour / past the last
    // offset -> out-of-range offset -> zeros)
    // The rulebook entries of a unit are read from LDS one unit AHEAD of its gathers (Idx travels in registers across one
    // iteration): the LDS round trips are off the issue path of the loads.
    struct Idx { int src[NF]; unsigned cvb; bool second; };
    auto fetch_idx = [&](int step) -> Idx {
        // lane group g reads vector v = 4 step + g of the flattened (offset, channel) axis: offset v / vpo, channel vector
        // v % vpo (float reciprocal: exact here); past the last offset or the last step -> the "-1" word -> zeros
        const int v = step * 4 + my_chunk;
        Idx ix;
        ix.second = __builtin_amdgcn_readfirstlane((int)(step >= n_main && step < n_steps)) != 0;   // wave-uniform
        if (ix.second) {
            const int cv = v - n_main * 4;
            ix.cvb = (unsigned)cv * 16u;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int p = row0 + my_row + f * 16;
                ix.src[f] = (p < n && cv < a.vpo2) ? p : -1;
            }
            return ix;
        }
        const int ko = (int)(((float)v + 0.5f) * inv_vpo);
        const bool live = step < n_steps && ko < K;
        ix.cvb = (unsigned)(v - ko * vpo) * 16u;
#pragma unroll
        for (int f = 0; f < NF; ++f) ix.src[f] = *(live ? s_nbr + (my_row + f * 16) * KS + ko : s_none);
        return ix;
    };
    auto issue_x = [&](Stage<NF, NT>& st, const Idx& ix) {
#ifdef PBN_CONV_TIMING
        if (a.dbg & 1) return;                                   // ablation: no row gathers
#endif
        const i32x4 rs = {__builtin_amdgcn_readfirstlane(ix.second ? rs_in2[0] : rs_in[0]),      // wave-uniform select
                          __builtin_amdgcn_readfirstlane(ix.second ? rs_in2[1] : rs_in[1]),
                          __builtin_amdgcn_readfirstlane(ix.second ? rs_in2[2] : rs_in[2]), 0x00020000};
        const unsigned ldb = ix.second ? ld2_bytes : ld_bytes;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const unsigned voff = ix.src[f] >= 0 ? (unsigned)ix.src[f] * ldb + ix.cvb : OOB;
            buf_load_asm(st.x[f], rs, voff, 0u);
        }
    };
    auto compute = [&](const Stage<NF, NT>& st) {
#ifdef PBN_CONV_TIMING
        if (a.dbg & 4) return;                                   // ablation: no MFMAs
#endif
#if PBN_WAVE_XCOAL
        u32x4 xo[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int d = 0; d < 4; ++d) xo[f][d] = (unsigned)__builtin_amdgcn_ds_bpermute(xpose_addr, (int)st.x[f][d]);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int f = 0; f < NF; ++f) mfma_step<T>(st.w[t], xo[f], acc[f][t]);
#else
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int f = 0; f < NF; ++f) mfma_step<T>(st.w[t], st.x[f], acc[f][t]);
#endif
    };

    {
        // B units in flight: a unit's registers are refilled with the unit B positions ahead right behind its own MFMAs.
        // Issue order = unit order (weights, then rows), so when unit i is consumed the loads behind it are exactly those of
        // units i+1 .. i+B-1: wait for vmcnt((B-1) * (NT+NF)).  The wait names the unit's registers so that their readers
        // are ordered behind it.
        constexpr int BEHIND = (B - 1) * (NT + NF);
        auto wait_unit = [&](Stage<NF, NT>& u) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(u.w[0]) : "n"(BEHIND));
#pragma unroll
            for (int t = 1; t < NT; ++t) asm volatile("" : "+v"(u.w[t]));
#pragma unroll
            for (int f = 0; f < NF; ++f) asm volatile("" : "+v"(u.x[f]));
        };
        // issue slot k: the loads of unit k (its step and rulebook entries were fetched during slot k - 1), then the
        // fetches for slot k + 1
        int step = step_of(0);
        Idx ix = fetch_idx(step);
        int raw_next = 0;
        if constexpr (KLIST) raw_next = (int)my_units[KW + wave];
        else if constexpr (!KSPLIT) raw_next = (int)my_units[1];
        auto slot = [&](Stage<NF, NT>& u, int k) {
            issue_w(u, step);
            issue_x(u, ix);
            if constexpr (KLIST) {
                step = __builtin_amdgcn_readfirstlane(raw_next);
                raw_next = (int)my_units[(k + 2) * KW + wave];
            } else if constexpr (KSPLIT) {
                step = step_of(k + 1);
            } else {
                step = __builtin_amdgcn_readfirstlane(raw_next);
                raw_next = (int)my_units[k + 2];
            }
            ix = fetch_idx(step);
        };
#pragma unroll
        for (int s = 0; s < B; ++s) slot(st[s], s);
        PBN_WSTAMP(3);
        for (int i = 0; i < n_units; i += B) {
#pragma unroll
            for (int s = 0; s < B; ++s) {
                wait_unit(st[s]);
                compute(st[s]);
                __builtin_amdgcn_sched_barrier(0);     // the refill stays behind the unit's own MFMAs
                slot(st[s], i + B + s);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the trailing issues were units past the end; drain them.  The wait NAMES every stage register: they stay allocated
        // up to here (otherwise the epilogue's address arithmetic is scheduled into them above the wait and a landing
        // load overwrites it -- seen as run-to-run different rows)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < B; ++s) {
#pragma unroll
            for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(st[s].w[t]));
#pragma unroll
            for (int f = 0; f < NF; ++f) asm volatile("" : "+v"(st[s].x[f]));
        }
    }
    PBN_WSTAMP(4);

    if constexpr (!KSPLIT) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int p = row0 + wave * RW + f * 16 + rl;
            if (p >= n) continue;
            const int orow = a.row_perm ? a.row_perm[p] : p;
#pragma unroll
            for (int t = 0; t < NT; ++t) epilogue_store<T>(a, acc[f][t], orow, (tile0 + t) * 16 + g * 4);
        }
    } else {
        // fixed-order sum of the KW partial tiles through LDS, NTB channel tiles per round, then the epilogue
        constexpr int RP = NTB * 16;   // floats per row per round
        constexpr int RPP = ksplit_red_pitch(NTB);   // ... and the buffer's row pitch
        T* out = reinterpret_cast<T*>(a.out);
        const T* res = reinterpret_cast<const T*>(a.residual);
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += NTB) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int tt = 0; tt < NTB; ++tt) {
                    const f32x4 v = acc[f][t0 + tt];
                    *reinterpret_cast<float4*>(s_red + ((size_t)(wave * TM + f * 16 + rl) * RPP + tt * 16 + g * 4)) =
                        make_float4(v[0], v[1], v[2], v[3]);
                }
            __syncthreads();
            if (t0 == 0) { PBN_WSTAMP(5); }
            for (int e = tid; e < TM * (RP / 4); e += TPB) {
                const int r = e / (RP / 4), q = e - r * (RP / 4);
                const int p = row0 + r;
                if (p >= n) continue;
                const int orow = a.row_perm ? a.row_perm[p] : p;
                const int cl = t0 * 16 + q * 4;          // channel within this workgroup's NT*16
                f32x4 rv = f32x4{0.f, 0.f, 0.f, 0.f};
                if (res) rv = load4<T>(res + (size_t)orow * a.ld_res + tile0 * 16 + cl);   // in flight under the LDS sum
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < KW; ++w) {
                    const float4 s = *reinterpret_cast<const float4*>(s_red + ((size_t)(w * TM + r) * RPP + q * 4));
                    v[0] += s.x; v[1] += s.y; v[2] += s.z; v[3] += s.w;
                }
                if (a.scale) {
                    const float4 sc = *reinterpret_cast<const float4*>(s_ss + cl);
                    v[0] *= sc.x; v[1] *= sc.y; v[2] *= sc.z; v[3] *= sc.w;
                }
                if (a.shift) {
                    const float4 sh = *reinterpret_cast<const float4*>(s_ss + NT * 16 + cl);
                    v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w;
                }
                v += rv;
                if (a.relu) {
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                store4<T>(out + (size_t)orow * a.ld_out + tile0 * 16 + cl, v);
            }
            if (t0 + NTB < NT) __syncthreads();
        }
    }
    PBN_WSTAMP(6);
}

template <typename T, int NF, int NT, int KW, bool KSPLIT, bool KLIST = false>
int launch_cfg(ConvArgs a, hipStream_t stream) {
    constexpr int RW = NF * 16;
    constexpr int TM = KSPLIT ? RW : KW * RW;
    constexpr int NTB = ksplit_round_tiles(KW, TM, NT);
    if (a.ntiles_total % NT) return PBN_ERR_UNSUPPORTED;
    const int KS = a.K | 1;
    size_t lds = sizeof(int) * (size_t)((TM * KS + 3) & ~3);
    if (KSPLIT) lds += 16 + sizeof(float) * ((size_t)2 * NT * 16 + (size_t)KW * TM * ksplit_red_pitch(NTB));
    if (!KSPLIT || KLIST) lds += (KSPLIT ? 0 : 16) + sizeof(unsigned) * (size_t)((KW * (a.n_steps + 2 * MAX_DEPTH + (KLIST ? 64 : 0)) + 3) & ~3);
    if (lds > 160 * 1024 || a.K > 128 || a.n_steps > 0xfffe) return PBN_ERR_UNSUPPORTED;
    auto kern = k_spconv_wave<T, NF, NT, KW, KSPLIT, KLIST>;
    if (lds > 64 * 1024)
        PBN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int row_tiles = cdiv(a.n_out, TM), groups = a.ntiles_total / NT;
    // weight-major block order (see map_block) where the packed weights outweigh the input slab: coarse levels
    static const int wmajor_env = getenv("PBN_WAVE_WMAJOR") ? atoi(getenv("PBN_WAVE_WMAJOR")) : 2;   // 0 off, 1 always, 2 by size
    const bool shape_ok = groups >= 8 ? (groups % 8 == 0) : (groups == 1 || groups == 2 || groups == 4);
    a.wmajor = (wmajor_env == 1 || (wmajor_env == 2 && a.w_bytes > a.in_bytes)) && shape_ok && groups > 1 ? 1 : 0;
    int blocks = row_tiles * groups;
    if (a.wmajor && groups < 8) blocks = 8 * cdiv(row_tiles, 8 / groups);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(KW * 64), lds, stream, a);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

// cfg = 1000 * ksplit (2000: K-split with the populated-step list) + 100 * NF + NT
template <typename T>
int launch_by_cfg(const ConvArgs& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 401: return launch_cfg<T, 4, 1, 4, false>(a, stream);
        case 402: return launch_cfg<T, 4, 2, 4, false>(a, stream);
        case 404: return launch_cfg<T, 4, 4, 4, false>(a, stream);
        case 406: return launch_cfg<T, 4, 6, 4, false>(a, stream);
        case 408: return launch_cfg<T, 4, 8, 4, false>(a, stream);
        case 204: return launch_cfg<T, 2, 4, 4, false>(a, stream);
        case 206: return launch_cfg<T, 2, 6, 4, false>(a, stream);
        case 208: return launch_cfg<T, 2, 8, 4, false>(a, stream);
        case 1401: return launch_cfg<T, 4, 1, 8, true>(a, stream);
        case 1402: return launch_cfg<T, 4, 2, 8, true>(a, stream);
        case 1404: return launch_cfg<T, 4, 4, 8, true>(a, stream);
        case 1201: return launch_cfg<T, 2, 1, 8, true>(a, stream);
        case 1202: return launch_cfg<T, 2, 2, 8, true>(a, stream);
        case 1204: return launch_cfg<T, 2, 4, 8, true>(a, stream);
        case 1208: return launch_cfg<T, 2, 8, 8, true>(a, stream);
        case 1408: return launch_cfg<T, 4, 8, 8, true>(a, stream);
        case 2202: return launch_cfg<T, 2, 2, 8, true, true>(a, stream);      // K-split over the tile's populated steps
        case 2201: return launch_cfg<T, 2, 1, 8, true, true>(a, stream);
        default: return PBN_ERR_UNSUPPORTED;
    }
}

// Pick (mode, NF, NT).  The wave-level cost is load bytes through the CU's 64 B/clk return path: a workgroup moves
// (all populated K) x (NT KiB of weights + NF KiB of rows) per wave-slice; the chip wants >= 256 workgroups.  K-split
// spreads ONE row tile over 8 waves, so it is the mode of choice whenever row-split tiles alone cannot fill the CUs.
int pick_cfg(const ConvArgs& a) {
    const int ntt = a.ntiles_total;
    const long long rows = a.n_sel;
    // stem-like layers (a 16/32-byte input row, K = 125): 32 rows x 32 channels per workgroup measured best at every size
    // (146 k rows: 44.8 us against 68.1 for 64 rows and 72.6 for the workgroup-tile kernel)
    if (a.vpo <= 2 && a.K >= 64 && ntt % 2 == 0) return 1202;
    // the k = 5 stems of the local-scene networks (32 / 64-byte... 64 / 128-byte input rows, 125 offsets, 58 k rows): the same
    // shape wins there (round 4: 146 k rows 32->32 152 -> 90 us, 64->32 179 -> 156 against the
    // workgroup-tile kernel, whose 125-column rulebook tile leaves it two workgroups per CU)
    // round 4: their K-split over the tile's POPULATED steps (KLIST, 2202): 146 k-row probe 40->32 170 -> 154 us; level on 32->32
    // (94 / 95) and a loss on the 16-byte-row stem above, where a step spans four offsets and nearly every step is populated
    // (41 -> 65 us: the list costs more than the few steps it drops) -- so only rows of 4+ vectors take it
    // In the pipeline: the 14A stem (34 -> 32 channels) 78.8 -> 66.7 us with one scene alone, and four scenes in flight measured
    // 310 / 313 scenes/s with it against 318 / 328 without (run-to-run spread +-5 %): 12 us of 5 400 alone, nothing shown in
    // flight -- OFF by default (PBN_WAVE_KLIST=1), kept with its tests (cfg 2201 / 2202).
    static const int klist = getenv("PBN_WAVE_KLIST") ? atoi(getenv("PBN_WAVE_KLIST")) : 0;
    if (a.vpo <= 8 && a.K >= 64 && ntt % 2 == 0) return (klist && a.vpo >= 8) ? 2202 : 1202;
    int best = 0;
    long long best_wgs = -1;
    for (int cfg : {1404, 1204, 1402, 1202, 1401, 1201}) {
        const int nf = (cfg / 100) % 10, nt = cfg % 100;
        if (ntt % nt) continue;
        const long long wgs = ((rows + nf * 16 - 1) / (nf * 16)) * (ntt / nt);
        if (wgs >= 180) return cfg;
        if (wgs > best_wgs) { best_wgs = wgs; best = cfg; }
    }
    return best ? best : 1201;
}

template <typename T>
int launch_t(const ConvArgs& a, int force_cfg, hipStream_t stream) {
    int cfg = force_cfg > 0 ? force_cfg : pick_cfg(a);
    int rc = launch_by_cfg<T>(a, cfg, stream);
    if (rc == PBN_ERR_UNSUPPORTED && force_cfg <= 0) {   // e.g. K = 125: the rulebook tile + reduction buffer exceed LDS
        for (int alt : {1202, 1201, 204}) {
            rc = launch_by_cfg<T>(a, alt, stream);
            if (rc != PBN_ERR_UNSUPPORTED) break;
        }
    }
    return rc;
}

}  // namespace

#ifdef PBN_CONV_TIMING
}  // namespace pbn
// debug build only: the stamps of the last k_spconv_wave launch -> host
extern "C" int pbn_wave_timing_read(unsigned long long* host) {
    PBN_HIP_CHECK(hipDeviceSynchronize());
    PBN_HIP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(pbn::g_wave_timing), sizeof(unsigned long long) * (pbn::WT_BLOCKS * 64 + 8)));
    return PBN_OK;
}
namespace pbn {
#endif

int launch_wave(const ConvArgs& a, int dtype, int force_cfg, hipStream_t stream) {
    switch (dtype) {
        case PBN_F32: return launch_t<float>(a, force_cfg, stream);
        case PBN_BF16: return launch_t<__hip_bfloat16>(a, force_cfg, stream);
        case PBN_F16: return launch_t<__half>(a, force_cfg, stream);
        default: return PBN_ERR_ARG;
    }
}

// PBN_CONV_FAMILY: 0 = workgroup-tile kernels only (round 1), 1 = wave kernels only, 2 (default) = by size of the op.
// Several scenes in flight (scripts/probe_wave.py with PBN_PROBE_STREAMS=4) would pick wider channel tiles on the same
// layers (1408: L3 256->256 19.6 us against 26.3 for 1404; 43.6 against 33.6 alone) -- in the real pipeline, where the
// other streams run OTHER layers, a process-wide switch to those tiles measured 2-3 % slower (274-278 against 283-285
// scenes/s), so the choice below stays the one-scene one.
// Measured crossover (scripts/probe_wave.py, HIP-graph replay, round 3 after the hand-counted main loop and the
// quad-coalesced gathers): the K-split wave kernel is ahead of the workgroup-tile kernel + its split-K reduce launch on every
// level below 20 k rows up to ~1e10 dense MACs (rows x K x C_in x C_out; L3 384->256 30.3 against 40.2 us, L2 128->128 28.8
// against 33.8), and behind it on the wide levels (L1 96->96 48.5 against 38.0, L0 96->96 116 against 74.5).
void describe_launch(const ConvArgs& a, int dtype, LaunchDesc* d) {
    d->wave_family = 0; d->nt = 0; d->groups = 0; d->wmajor = 0;
    if (!wave_family_wanted(a, dtype)) return;
    static const int force_cfg = getenv("PBN_WAVE_CFG") ? atoi(getenv("PBN_WAVE_CFG")) : 0;
    const int cfg = force_cfg > 0 ? force_cfg : pick_cfg(a);
    const int nt = cfg % 100;
    if (nt <= 0 || a.ntiles_total % nt) return;
    d->wave_family = 1;
    d->nt = nt;
    d->groups = a.ntiles_total / nt;
    const bool shape_ok = d->groups >= 8 ? (d->groups % 8 == 0) : (d->groups == 1 || d->groups == 2 || d->groups == 4);
    static const int wmajor_env = getenv("PBN_WAVE_WMAJOR") ? atoi(getenv("PBN_WAVE_WMAJOR")) : 2;
    d->wmajor = (cfg >= 1000 && (wmajor_env == 1 || (wmajor_env == 2 && a.w_bytes > a.in_bytes)) && shape_ok && d->groups > 1) ? 1 : 0;
}

bool wave_family_wanted(const ConvArgs& a, int dtype) {
    (void)dtype;
    static const int fam = getenv("PBN_CONV_FAMILY") ? atoi(getenv("PBN_CONV_FAMILY")) : 2;
    static const int max_rows = getenv("PBN_WAVE_MAX_ROWS") ? atoi(getenv("PBN_WAVE_MAX_ROWS")) : 30000;   // 30 k instead of 20 k: level with the tile family on 26 k-row levels and no split-K reduce launch left in a forward
    static const double max_macs = getenv("PBN_WAVE_MAX_GMACS") ? atof(getenv("PBN_WAVE_MAX_GMACS")) * 1e9 : 1.0e10;
    if (a.K > 128) return false;
    if (fam == 0) return false;
    if (fam == 1) return true;
    if (a.vpo <= 8 && a.K >= 64 && !a.row_perm) return true;   // the k = 5 stems: rulebook-bound, no weight reuse to speak of
    const double elems_per_step = 4.0 * (dtype == PBN_F32 ? 4.0 : 8.0);
    const double dense = (double)a.n_sel * a.n_steps * elems_per_step * a.ntiles_total * 16.0;
    return a.n_sel < max_rows && dense <= max_macs;
}

}  // namespace pbn
