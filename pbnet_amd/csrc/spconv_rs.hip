// spconv_rs.hip -- ROW-STATIONARY implicit-GEMM sparse convolution for the wide levels (stride 1 / 2: tens of thousands of
// rows, 32..128 channels) on gfx950.  Round 5.
//
// Same arithmetic, operand layouts, packed weights and fused epilogue as spconv.hip (MinkowskiConvolution /
// ConvolutionTranspose forward, /root/reference/network/Mink.py:221-288,293-350):
//     out[o, :] = epilogue( sum_k in[nbr[o,k], :] @ W[k] )
// and the same summation order per output element (offsets ascending, channels ascending), so its results are bit-identical
// to k_spconv's.  What differs is the machine mapping, which follows from what rounds 1-4 measured on that kernel
// (profiles/README.md): its main loop sits on the CU's vector-memory path, where the row gathers and the
// weight tiles cost about the same -- every 128-row workgroup streams ALL K * Cin * Cout weights through its LDS ring
// (0.57 GB per 96->96 launch at 146 k rows against 0.21 GB of useful gathers) -- and 1 141 tiles on 768 resident workgroups run
// as two rounds.  Here:
//   * ONE workgroup per CU and ONE round: the launch cuts the level into ~(number of CUs) tiles of equal height (a multiple
//     of 16 rows, up to 640: 8 waves x NF <= 5 fragments of 16 rows), so every CU streams the weights ONCE per launch
//     (256 x 0.5 MB instead of 1 141 x 0.5 MB) and nobody waits for a second round;
//   * a tile's accumulators (tile_rows x Cout fp32 = up to 240 KB of the CU's 512 KB register file) stay in registers for
//     the whole K * Cin reduction -- the rows are stationary, the weights stream past them through the LDS-DMA ring;
//   * per (16-row fragment, offset) skipping: a fragment without a neighbour at an offset issues no MFMAs (k_spconv skips per
//     32-row wave) -- populated share 0.77 instead of 0.85 at stride 1 (scripts/analyze_rulebook.py);
//   * fragments are dealt to the waves round-robin (fragment f * 8 + wave), so a tile height that is not a multiple of 128
//     leaves every wave the same number of fragments +- 1.
// The main loop keeps k_spconv's proven structure: one barrier per reduction group (<= 4 steps of one offset), weights one
// group ahead through buffer_load ... lds, in-place refill of the gather registers, every vector-memory wait hand-counted.
#include <cstdlib>
#include <type_traits>
#include "spconv_common.h"

namespace pbn {
namespace {

#define PBN_RS_LDS_ADDR(p) ((unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)(p)))

constexpr int RS_NW = 8;              // waves per workgroup
constexpr int RS_TPB = RS_NW * 64;
constexpr int RS_NF_MAX = 5;          // fragments per wave: 8 x 5 x 16 = 640 rows per tile at most
constexpr unsigned RS_OOB = 0x80000000u;

template <typename T, int NF, int NT, int CG, int RING>
__global__ __launch_bounds__(RS_TPB) void k_spconv_rs(const ConvArgs a, const int tile_rows_launch, const int n_tiles, const int list_cap) {
    static_assert(Tr<T>::ELEMS * sizeof(T) == 16, "one gather vector is 16 bytes");
    constexpr int DEPTH = RING - 1;                         // weight tiles in flight ahead
    constexpr int PW = (CG * NT + RS_NW - 1) / RS_NW;        // weight pieces (1 KiB) per wave and group
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = a.K, KS = K | 1;
    u32x4* s_w = reinterpret_cast<u32x4*>(smem);                                        // RING slots x CG * NT KiB
    int* s_nbr = reinterpret_cast<int*>(smem + (size_t)RING * CG * NT * 1024);          // tile_rows * KS
    int* s_gko = s_nbr + ((tile_rows_launch * KS + 3) & ~3);                             // list_cap: offset | sub-group << 16
    unsigned* s_act = reinterpret_cast<unsigned*>(s_gko + list_cap);                     // [0..1] active offsets, [2] groups
    float* s_ss = reinterpret_cast<float*>(s_act + 4);                                   // scale | shift

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, rl = lane & 15;
    const int n = a.n_out_dev ? min(*a.n_out_dev, a.n_out) : a.n_out;
    // capacity form (n_out is a capacity, the rows that exist are counted on the device) WITHOUT a rows hint: the launch's tiles share
    // the rows that exist -- a tile height sized by the capacity would leave the last workgroups without work and the others with too
    // much.  With a hint (n_sel < n_out) the tiles were cut for the rows expected and those behind the count simply exit
    if (n <= 0) return;                   // (a device-side count of 0: no tile height to cut, nothing to write)
    const int tile_rows = (a.n_out_dev && a.n_sel == a.n_out) ? min(tile_rows_launch, (((n + n_tiles - 1) / n_tiles) + 15) & ~15) : tile_rows_launch;
    // tiles that have rows (the launch may hold more: capacities); the XCD-aware map runs over THOSE, so that every XCD gets the
    // same share of working tiles (mapped over the launch's tiles, the last XCDs would hold only empty ones)
    const int n_work = (n + tile_rows - 1) / tile_rows;
    if ((int)blockIdx.x >= n_work) return;
    const int tile = xcd_tile(blockIdx.x, n_work);
    const int row0 = tile * tile_rows;
    const int live = min(tile_rows, n - row0);

    if (tid < 4) s_act[tid] = 0u;
    if (tid < NT * 32) {
        const int c = tid < NT * 16 ? tid : tid - NT * 16;
        const float* src = tid < NT * 16 ? a.scale : a.shift;
        s_ss[tid] = src ? src[c] : (tid < NT * 16 ? 1.0f : 0.0f);
    }
    // ---- rulebook tile -> LDS (rows past the end of the level: -1) ----
    if (a.nbr && live == tile_rows && KS == K && ((tile_rows * K) & 3) == 0 && (((size_t)row0 * K) & 3) == 0) {
        const int4* src = reinterpret_cast<const int4*>(a.nbr + (size_t)row0 * K);
        int4* dst = reinterpret_cast<int4*>(s_nbr);
        const int nv = (tile_rows * K) >> 2;
#pragma unroll 4
        for (int e = tid; e < nv; e += RS_TPB) dst[e] = src[e];
    } else {
        const float inv_k = 1.0f / (float)K;
#pragma unroll 2
        for (int e = tid; e < tile_rows * K; e += RS_TPB) {
            const int r = (int)(((float)e + 0.5f) * inv_k), k = e - r * K;
            int v = -1;
            if (r < live) v = a.nbr ? a.nbr[(size_t)(row0 + r) * K + k] : row0 + r;
            s_nbr[r * KS + k] = v;
        }
    }
    __syncthreads();
    // ---- lane k of every wave: which of the wave's NF fragments have a neighbour at offset k (lane K: the second source) ----
    unsigned mreg = 0u;
    if (lane < K) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int r0 = (f * RS_NW + wave) * 16;
            if (r0 < live) {
                int any = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) any |= (s_nbr[(r0 + i) * KS + lane] >= 0) ? 1 : 0;
                mreg |= (unsigned)any << f;
            }
        }
    } else if (lane == K && a.in2) {
#pragma unroll
        for (int f = 0; f < NF; ++f)
            if ((f * RS_NW + wave) * 16 < live) mreg |= 1u << f;
    }
    if (a.dbg & 1) mreg = (lane <= K) ? ((1u << NF) - 1u) : 0u;
    {
        const unsigned long long pop = __ballot(mreg != 0u);
        if (lane == 0) {
            atomicOr(&s_act[0], (unsigned)pop);
            atomicOr(&s_act[1], (unsigned)(pop >> 32));
        }
    }
    __syncthreads();
    // ---- the tile's ordered list of reduction groups (wave 0): every populated offset x its sub-groups, then the second source ----
    const int gpo = (a.vpo >> 2) / CG;                         // groups per offset
    const int n_main_groups = (a.in2 ? a.n_main : a.n_steps) / CG;
    if (wave == 0) {
        const unsigned long long act = (unsigned long long)s_act[0] | ((unsigned long long)s_act[1] << 32);
        int cnt = 0;
        if (lane == 0) {
            for (int k = 0; k < K; ++k)
                if ((act >> k) & 1ull)
                    for (int s = 0; s < gpo; ++s) s_gko[cnt++] = k | (s << 16);
            if (a.in2 && ((act >> K) & 1ull))
                for (int s = 0; s < a.n_steps / CG - n_main_groups; ++s) s_gko[cnt++] = K | (s << 16);
            s_act[2] = (unsigned)cnt;
            for (int e = 0; e < 4; ++e) s_gko[cnt + e] = 0;   // entries read past the end (never used: `more` is false)
        }
    }
    __syncthreads();
    const int ng = (a.dbg & 16) ? 0 : __builtin_amdgcn_readfirstlane((int)s_act[2]);

    f32x4 acc[NF][NT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Both operands come in through buffer resources (as in k_spconv): a gather is one instruction with a 32-bit per-lane
    // byte offset, a fragment without a neighbour uses an out-of-range offset and reads zeros.
    const unsigned long long in_addr = (unsigned long long)a.in, w_addr = (unsigned long long)a.w;
    const i32x4 rs_in = {(int)(unsigned)in_addr, (int)(unsigned)(in_addr >> 32), (int)a.in_bytes, 0x00020000};
    const i32x4 rs_w = {(int)(unsigned)w_addr, (int)(unsigned)(w_addr >> 32), (int)a.w_bytes, 0x00020000};
    const unsigned ld_bytes = (unsigned)a.ld_in * (unsigned)sizeof(T);
    const unsigned long long in2_addr = (unsigned long long)a.in2;
    const i32x4 rs_in2 = {(int)(unsigned)in2_addr, (int)(unsigned)(in2_addr >> 32), (int)a.in2_bytes, 0x00020000};
    const unsigned ld2_bytes = (unsigned)a.ld_in2 * (unsigned)sizeof(T);
    const unsigned w_step_bytes = (unsigned)NT * 1024u;     // one step, all channel tiles (ntiles_total == NT)
    const unsigned w_lane = (unsigned)lane * 16u;
    constexpr unsigned slot_bytes = (unsigned)CG * NT * 1024u;
    const unsigned lds_w = PBN_RS_LDS_ADDR(s_w);

    if (ng > 0) {
        u32x4 x[CG][NF];
#pragma unroll
        for (int c = 0; c < CG; ++c)
#pragma unroll
            for (int f = 0; f < NF; ++f) x[c][f] = u32x4{0u, 0u, 0u, 0u};

        bool nx2 = false;     // the group whose rows are being fetched reads the second source ...
        int nxv = CG;         // ... and this many of its chunks exist
        // gather byte offsets of list entry `pk` (all out of range when `more` is false)
        auto group_rows = [&](int pk, bool more, unsigned (&voff)[NF]) {
            const int ko = pk & 0xffff, sub = pk >> 16;
            nx2 = a.in2 != nullptr && more && ko == K;
            nxv = nx2 ? (a.vpo2 >> 2) - sub * CG : CG;
            const unsigned ldb = nx2 ? ld2_bytes : ld_bytes;
            const unsigned cvb = (unsigned)(sub * CG * 4 + g) * 16u;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int r = (f * RS_NW + wave) * 16 + rl;
                int src = -1;
                if (more && r < live) src = nx2 ? row0 + r : (ko < K ? s_nbr[r * KS + ko] : -1);
                voff[f] = (src >= 0 && !(a.dbg & 4)) ? (unsigned)src * ldb + cvb : RS_OOB;
            }
        };
        auto pick_rs = [&](bool second) -> i32x4 {      // wave-uniform choice, made provably scalar for the "s" constraint
            const bool s2 = __builtin_amdgcn_readfirstlane((int)second) != 0;
            return i32x4{__builtin_amdgcn_readfirstlane(s2 ? rs_in2[0] : rs_in[0]), __builtin_amdgcn_readfirstlane(s2 ? rs_in2[1] : rs_in[1]),
                         __builtin_amdgcn_readfirstlane(s2 ? rs_in2[2] : rs_in[2]), 0x00020000};
        };
        // weight tile of list entry `pk` -> ring slot at LDS byte address `slot_addr`: piece p = wave + 8 i is step p / NT, channel
        // tile p % NT; every wave issues PW pieces (the tail is clamped onto the last piece: a benign duplicate copy).  `more`
        // false (past the last group): the same instructions with an out-of-range offset: no memory traffic.
        auto dma_w = [&](int pk, bool more, unsigned slot_addr) {
            const int ko = pk & 0xffff, sub = pk >> 16;
            const int gi = ko == K ? n_main_groups + sub : ko * gpo + sub;
            const unsigned gbase = (unsigned)gi * (unsigned)(CG * NT) * 1024u;
            const unsigned wv = (more && !(a.dbg & 8)) ? w_lane : RS_OOB;
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                const int p = min(wave + RS_NW * i, CG * NT - 1);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                             "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "s"(__builtin_amdgcn_readfirstlane(slot_addr + (unsigned)p * 1024u)), "v"(wv), "s"(rs_w),
                               "s"(__builtin_amdgcn_readfirstlane(gbase + (unsigned)p * 1024u))
                             : "memory");
            }
        };
        (void)w_step_bytes;

        // ---- prologue: the issue pattern of DEPTH loop bodies, so that the loop's wait counts hold from the first group on ----
        unsigned fcur = 0u;
        {
            const int pk0 = __builtin_amdgcn_readfirstlane(s_gko[0]);
            fcur = (unsigned)__builtin_amdgcn_readlane((int)mreg, pk0 & 0xffff);
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                const int pkj = __builtin_amdgcn_readfirstlane(s_gko[j]);
                dma_w(pkj, j < ng, lds_w + (unsigned)j * slot_bytes);
                unsigned v0[NF];
                group_rows(pk0, j == DEPTH - 1, v0);
                const i32x4 rs0 = pick_rs(nx2);
#pragma unroll
                for (int c = 0; c < CG; ++c)
#pragma unroll
                    for (int f = 0; f < NF; ++f)
                        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen"
                                     : "+v"(x[c][f]) : "v"(c < nxv ? v0[f] : RS_OOB), "s"(rs0), "s"(c * 64));
            }
        }
        unsigned slot = 0;
        // ---- main loop.  Issue order per group body j: DMA(j + DEPTH), then x[0..CG-1](j + 1); hence at the top of group p:
        //   barrier : loads issued behind DMA(p) = CG*NF + (DEPTH-1) * (PW + CG*NF): wait for exactly that count = this wave's pieces
        //             of the group's weight tile have landed; behind the barrier everybody's have, and the slot of group p-1 is free
        //   chunk c : loads behind x[c](p) = x[c+1..](p), DMA(p + DEPTH), x[..c-1](p + 1) = (CG-1)*NF + PW
        for (int pos = 0; pos < ng; ++pos) {
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(CG * NF + (DEPTH - 1) * (PW + CG * NF)) : "memory");
            asm volatile("s_barrier" : : : "memory");
            const bool more = pos + 1 < ng;
            const u32x4* cur = s_w + slot * (CG * NT * 64);
            const unsigned free_slot = slot == 0 ? RING - 1 : slot - 1;   // slot of group pos-1 = slot of group pos+DEPTH
            slot = slot == RING - 1 ? 0 : slot + 1;
            unsigned vnext[NF];
            unsigned fnext;
            {
                const int pkd = __builtin_amdgcn_readfirstlane(s_gko[pos + DEPTH]);
                dma_w(pkd, pos + DEPTH < ng, lds_w + free_slot * slot_bytes);
                const int pkn = __builtin_amdgcn_readfirstlane(s_gko[pos + 1]);
                fnext = more ? (unsigned)__builtin_amdgcn_readlane((int)mreg, pkn & 0xffff) : 0u;
                group_rows(pkn, more, vnext);
            }
            const i32x4 rsn = pick_rs(nx2);
            const bool active = fcur != 0u && !(a.dbg & 2);
            // the asm statements that define x[][] stay on the straight-line path (k_spconv: inside a branch the compiler would
            // merge them through register copies, i.e. read registers whose loads are still in flight)
            u32x4 wf[2][NT];
            if (active) {
#pragma unroll
                for (int t = 0; t < NT; ++t) wf[0][t] = cur[t * 64 + lane];
            }
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x[c][0]) : "n"((CG - 1) * NF + PW));
#pragma unroll
                for (int f = 1; f < NF; ++f) asm volatile("" : "+v"(x[c][f]));
                if (active) {
                    if (c + 1 < CG) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) wf[(c + 1) & 1][t] = cur[((c + 1) * NT + t) * 64 + lane];
                    }
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        if ((fcur >> f) & 1u) {
#pragma unroll
                            for (int t = 0; t < NT; ++t) mfma_step<T>(wf[c & 1][t], x[c][f], acc[f][t]);
                        }
                    }
                }
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen"
                                 : "+v"(x[c][f]) : "v"(c < nxv ? vnext[f] : RS_OOB), "s"(rsn), "s"(c * 64));
            }
            fcur = fnext;
        }
        // drain: the loads issued for the (non-existent) group past the end still target these registers
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[c][0]));
#pragma unroll
            for (int f = 1; f < NF; ++f) asm volatile("" : "+v"(x[c][f]));
        }
    }

    // ---- epilogue: lane holds channels g*4 .. g*4+3 of tile t of output row (fragment base + rl) ----
    if (a.dbg & 32) return;
    T* out = reinterpret_cast<T*>(a.out);
    const T* res = reinterpret_cast<const T*>(a.residual);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int r = (f * RS_NW + wave) * 16 + rl;
        if (r >= live) continue;
        const int orow = row0 + r;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c0 = t * 16 + g * 4;
            f32x4 v = acc[f][t];
            if (a.scale) {
                const float4 sc = *reinterpret_cast<const float4*>(s_ss + t * 16 + g * 4);
                v[0] *= sc.x; v[1] *= sc.y; v[2] *= sc.z; v[3] *= sc.w;
            }
            if (a.shift) {
                const float4 sh = *reinterpret_cast<const float4*>(s_ss + NT * 16 + t * 16 + g * 4);
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

// =====================================================================================================================
// STAGED form (k_spconv_rsh): the tile's distinct input rows staged in LDS, MFMA row operands read from LDS.
//
// What round 5 measured on k_spconv / k_spconv_rs (profiles/r05_ablate_*.json, scripts/micro/gather_layout.hip with 16 loads in
// flight per wave): a gather instruction costs the CU's vector-memory path ~31-39 cycles whether 20 % or 100 % of its lanes
// return data (12.7 when none does, 25.6 for a contiguous KiB), and the loop with its gathers, weight copies and MFMAs ALL
// switched off still takes 70 % of the full time -- the path is bound by the NUMBER of vector-memory instructions, and an
// output-stationary tile issues 27 x 3 of them per fragment to fetch rows of which it names only ~1.35 distinct ones per output
// row.  Here every distinct input row of the tile enters the CU ONCE per 64-byte piece:
//   * prologue (rsh_build_tile; or once per map: k_rs_table_build): a thread's rulebook entries go global -> registers with all
//     loads in flight; entries inside the tile's WINDOW of consecutive input rows (cube maps: the tile's own rows; strided /
//     transposed maps: the rows from the tile's smallest entry on) become slot = row - window start, the others are de-duplicated
//     through an LDS hash set (atomicCAS, linear probing), numbered behind the window and renumbered by row value (which rows share
//     a stage segment must not depend on timing); the tile's rows are counting-sorted by their 8 corner-offset bits (rows of a
//     fragment then share most of their neighbour pattern); the map is rewritten, by POSITION in that order, as the 16-bit stage
//     position of each neighbour's slot, and per wave and offset the mask of fragments that have a neighbour there is collected;
//   * the reduction runs PIECE-major: for each 64-byte piece c of the input rows (32 bf16 channels) the pieces of all staged
//     rows are copied global -> LDS by the DMA path (quad-coalesced: four lanes fetch one row's piece; chunk g of slot s sits at
//     position (g + 2 (s / 4)) & 3 of its 64 bytes: with that rotation the four lane groups of a ds_read_b128 -- {0-3, 12-15,
//     20-27}, ... -- hit 16 different 4-bank groups when a fragment's 16 neighbours are consecutive slots), then all
//     offsets are walked with the piece's weights (NT KiB per offset) arriving in batches of BO offsets through a two-slot ring:
//     ONE barrier per batch, no vector-memory instruction in the inner loop at all;
//   * an operand is ds_read_b128(stage + slot * 64 + swizzled chunk); a missing neighbour reads a zeroed slot;
//   * accumulators stay in registers across pieces and offsets (row-stationary, as k_spconv_rs).
// Summation order per output element: pieces outer, offsets inner -- NOT k_spconv's (offsets outer); results agree with it to
// fp32 rounding (deterministic, run-to-run identical).
// Robustness: a tile whose staged rows exceed the LDS stage walks them in SEGMENTS (rows outside the resident segment read the
// zero slot: exact zeros, every product is still formed once); a tile whose hash set overflows (> RSH_EMAX distinct rows
// outside the window: no real map does) fetches its operands from global memory lane by lane.
// Cycle accounting (timing build only: make timing -> libpbnet_hip_timing.so, scripts/rsh_timing.py): every wave of the first
// RSH_TBLOCKS workgroups adds the cycles between consecutive stamps to one of 16 phase counters.
#ifdef PBN_CONV_TIMING
constexpr int RSH_TBLOCKS = 256;
__device__ unsigned long long g_rsh_timing[RSH_TBLOCKS * 8 * 16];
#define RSH_T0 unsigned long long t_last_ = __builtin_readcyclecounter(); unsigned long long t_acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define RSH_STAMP(I) { const unsigned long long t_now_ = __builtin_readcyclecounter(); t_acc_[I] += t_now_ - t_last_; t_last_ = t_now_; }
#define RSH_TWRITE if (blockIdx.x < RSH_TBLOCKS && lane == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) g_rsh_timing[(blockIdx.x * 8 + wave) * 16 + i_] = t_acc_[i_]; }
#else
#define RSH_T0
#define RSH_STAMP(I)
#define RSH_TWRITE
#endif
constexpr int RSH_BUCKETS = 2048;
constexpr int RSH_EMAX = 1536;
constexpr unsigned RSH_NONE = 0xffffu;

struct RshLds { unsigned slots, stage, wring, extra, misc, ss, perm, total; };
__host__ __device__ inline RshLds rsh_layout(int tile_rows, int K, int s_cap, int nt, int bo) {
    RshLds L;
    const unsigned KP = (unsigned)((K + bo - 1) / bo * bo);   // slot-table row pitch in halfwords: whole batches of bo columns (bo % 4 == 0)
    unsigned o = 0;
    L.slots = o; o += ((unsigned)tile_rows * KP * 2u + 16u + 15u) & ~15u;      // (+ a dummy entry behind the table)
    L.stage = o; o += ((unsigned)s_cap + 1u) * 64u;
    L.wring = o; o += 2u * (unsigned)bo * (unsigned)nt * 1024u;
    L.extra = o; o += ((unsigned)RSH_EMAX + 4u) * 4u;        // (+ a dummy word for the branch-free prologue)
    L.misc = o; o += (8u + 8u * 32u) * 4u;
    L.ss = o; o += 2u * (unsigned)nt * 16u * 4u;
    L.perm = o; o += ((unsigned)tile_rows * 2u + 15u) & ~15u;
    L.total = o;
    return L;
}

template <typename T, int NF, int NT>
__device__ __forceinline__ void rs_epilogue(const ConvArgs& a, f32x4 (&acc)[NF][NT], const float* s_ss, int row0, int live, int wave, int g, int rl,
                                            const unsigned short* s_perm) {
    T* out = reinterpret_cast<T*>(a.out);
    const T* res = reinterpret_cast<const T*>(a.residual);
    const bool has_scale = a.scale != nullptr, has_shift = a.shift != nullptr, relu = a.relu != 0;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int r = (f * RS_NW + wave) * 16 + rl;
        const bool ok = r < live;
        const size_t orow = (size_t)(row0 + (ok ? (int)s_perm[r] : 0));       // position -> row of the tile
        // the fragment's NT residual vectors are requested together, then combined and stored
        f32x4 rv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) rv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (res && ok) {
#pragma unroll
            for (int t = 0; t < NT; ++t) rv[t] = load4<T>(res + orow * a.ld_res + t * 16 + g * 4);
        }
        if (ok) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 v = acc[f][t];
                if (has_scale) {
                    const float4 s4 = *reinterpret_cast<const float4*>(s_ss + t * 16 + g * 4);
                    v[0] *= s4.x; v[1] *= s4.y; v[2] *= s4.z; v[3] *= s4.w;
                }
                if (has_shift) {
                    const float4 h4 = *reinterpret_cast<const float4*>(s_ss + NT * 16 + t * 16 + g * 4);
                    v[0] += h4.x; v[1] += h4.y; v[2] += h4.z; v[3] += h4.w;
                }
                v += rv[t];
                if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                store4<T>(out + orow * a.ld_out + t * 16 + g * 4, v);
            }
        }
    }
}

// Per-map tables in global memory (pbn_rs_table_build: once per cube map and lineage, reused by every layer of the level): one
// RECORD per tile = what k_spconv_rsh's prologue would build: header + fragment masks (8 + RS_NW * 32 ints: [0] rows outside the
// window), row order, the rows outside the window, the slot table at pitch kpg = K rounded up to 4 halfwords.
struct RsRecord { unsigned masks, perm, extra, slots, kpg, total; };
__host__ __device__ inline RsRecord rs_record_layout(int tile_rows, int K) {
    RsRecord R;
    unsigned o = 0;
    R.masks = o; o += (8u + (unsigned)RS_NW * 32u) * 4u;
    R.perm = o; o += ((unsigned)tile_rows * 2u + 15u) & ~15u;
    R.extra = o; o += (unsigned)RSH_EMAX * 4u;
    R.kpg = (unsigned)(K + 3) & ~3u;
    R.slots = o; o += ((unsigned)tile_rows * R.kpg * 2u + 15u) & ~15u;
    R.total = (o + 255u) & ~255u;
    return R;
}
__host__ __device__ inline size_t rs_table_record_bytes(int tile_rows, int K) { return rs_record_layout(tile_rows, K).total; }
constexpr int RS_TABLE_MAGIC = 0x52533035;      // the table starts with {magic, tile_rows, n_tiles, K}; records from byte 256 on
constexpr int RS_TABLE_HEAD = 256;

// The tile's tables, built in LDS by the whole workgroup (RS_TPB threads): slot table (indexed by position, pitch KP halfwords),
// row order (s_perm), rows outside the window (s_extra), fragment masks (s_misc[8 + wave * 32 + offset]), s_misc[0] = number of rows
// outside the window.  `scratch` = RSH_SCRATCH_BYTES of LDS that are free until the function returns.  Returns the window.
constexpr int RSH_SCRATCH_BYTES = (RSH_BUCKETS + 4) * 4 + (RSH_BUCKETS + 8) * 2 + RS_NW * 16 * RS_NF_MAX * 4 + 256 * 4 + RS_NW * 16 * RS_NF_MAX * 2 +
                                 RSH_EMAX * 6 + 64;
struct RshWindow { int wlo, wn; };
__device__ __forceinline__ RshWindow rsh_build_tile(const int* __restrict__ nbr, const int K, const int KP, const int tile_rows, const int live,
                                                    const int row0, const int n_in, const int s_cap, const bool cube, const bool sort_wanted,
                                                    unsigned short* s_slot, unsigned char* scratch, int* s_extra, int* s_misc,
                                                    unsigned short* s_perm) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int* s_hkey = reinterpret_cast<int*>(scratch);
    unsigned short* s_hslot = reinterpret_cast<unsigned short*>(s_hkey + RSH_BUCKETS + 4);           // (+ a dummy bucket)
    struct { const int* nbr; int dbg; } a = {nbr, sort_wanted ? 0 : 1024};
    // prologue-only arrays behind the hash set (all inside the stage region), and the tile's row order
    int* s_rowkey = reinterpret_cast<int*>(s_hslot + RSH_BUCKETS + 8);                          // tile_rows: sort key of a row
    int* s_hist = s_rowkey + RS_NW * 16 * RS_NF_MAX;                                        // 256 buckets
    unsigned short* s_inv = reinterpret_cast<unsigned short*>(s_hist + 256);                // tile_rows: position of a row
    // ---- P0: hash set cleared, slot table = "no neighbour", masks cleared; strided / transposed maps: the tile's smallest entry ----
    for (int e = tid; e < RSH_BUCKETS + 4; e += RS_TPB) { s_hkey[e] = -1; s_hslot[e] = (unsigned short)RSH_NONE; }
    {
        u32x4* sl4 = reinterpret_cast<u32x4*>(s_slot);
        const int nv = (tile_rows * KP * 2 + 15) >> 4;
        for (int e = tid; e < nv; e += RS_TPB) sl4[e] = u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    }
    for (int e = tid; e < 8 + RS_NW * 32; e += RS_TPB) s_misc[e] = e == 1 ? 0x7fffffff : 0;
    for (int e = tid; e < tile_rows; e += RS_TPB) { s_rowkey[e] = 0; s_perm[e] = (unsigned short)e; s_inv[e] = (unsigned short)e; }
    if (tid < 256) s_hist[tid] = 0;
    const int n_ent = live * K;                                   // rulebook entries of the tile: a contiguous block
    const int* rule = a.nbr ? a.nbr + (size_t)row0 * K : nullptr;
    const bool vec_ok = rule != nullptr && (n_ent & 3) == 0 && (((size_t)row0 * K) & 3) == 0;
    // the tile's entries -> registers (all loads in flight together); entry i of this thread is rulebook element ent(i)
    constexpr int NQ = (RS_NW * 16 * RS_NF_MAX * 32 / 4 + RS_TPB - 1) / RS_TPB;      // int4 loads per thread at most (K <= 32)
    constexpr int NE = NQ * 4;
    int ev[NE];
    auto ent = [&](int i) -> int { return vec_ok ? (tid + RS_TPB * (i >> 2)) * 4 + (i & 3) : tid + RS_TPB * i; };
    if (vec_ok) {
        const int4* src = reinterpret_cast<const int4*>(rule);
        const int nv = n_ent >> 2;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = tid + RS_TPB * q;
            const int4 v = idx < nv ? src[idx] : make_int4(-1, -1, -1, -1);
            ev[q * 4] = v.x; ev[q * 4 + 1] = v.y; ev[q * 4 + 2] = v.z; ev[q * 4 + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + RS_TPB * i;
            ev[i] = e < n_ent ? (rule ? rule[e] : row0 + e) : -1;
        }
    }
    if (!cube && rule) {
        int vmin = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < NE; ++i) if (ev[i] >= 0) vmin = min(vmin, ev[i]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) vmin = min(vmin, __shfl_xor(vmin, o));
        __syncthreads();
        if (lane == 0 && vmin != 0x7fffffff) atomicMin(&s_misc[1], vmin);
    }
    __syncthreads();
    int wlo = cube ? row0 : (rule ? s_misc[1] : row0);
    if (wlo == 0x7fffffff) wlo = 0;
    const int wn = max(0, min(cube || !rule ? tile_rows : s_cap, n_in - wlo));
    // ---- P1: entry -> slot, in PHASES over all of a thread's entries (up to NE independent LDS operations in flight: the phases
    // are throughput-bound, a per-entry loop would pay an LDS round trip per step).  Rows inside the window: slot = row - window
    // low.  Others: LDS hash set (atomicCAS, linear probing); the lane that inserts a row numbers it (window rows + running
    // count) and records it; the finders read the number after a barrier.
    const float inv_k = 1.0f / (float)K;
    const bool sort_rows = cube && K == 27 && !(a.dbg & 1024);
    // Every phase first issues its LDS operations for all NE entries (predicated, results into registers) and consumes them in a
    // second loop: one wait per phase instead of one per entry.
    int eb[NE];          // hash bucket of an entry outside the window | 0x10000: this lane inserted it | 0x20000: keep probing; -1: none
    {   // A: first probes (and the rows' sort keys: the 8 corner offsets of the 3x3x3 cube)
        int old[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int v = ev[i];
            const int e = ent(i);
            const int r = (int)(((float)e + 0.5f) * inv_k), k = e - r * K;
            const bool need = v >= 0 && (unsigned)(v - wlo) >= (unsigned)wn;
            const unsigned b = ((unsigned)v * 2654435761u) >> 21;
            eb[i] = need ? (int)b : -1;
            old[i] = 0;
            if (need) old[i] = atomicCAS(&s_hkey[b], -1, v);
            if (sort_rows && v >= 0) {
                const int ka = k / 9, kb = (k / 3) % 3, kc = k % 3;
                if (ka != 1 && kb != 1 && kc != 1) atomicOr(&s_rowkey[r], 1 << ((ka >> 1) * 4 + (kb >> 1) * 2 + (kc >> 1)));
            }
        }
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (eb[i] >= 0) eb[i] |= old[i] == -1 ? 0x10000 : (old[i] == ev[i] ? 0 : 0x20000);
    }
    // B: collisions (rare: the set is <= a third full)
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        if (eb[i] >= 0 && (eb[i] & 0x20000)) {
            const int v = ev[i];
            unsigned b = ((unsigned)(eb[i] & 0xffff) + 1u) & (RSH_BUCKETS - 1);
            int res = -2;
            for (int probe = 0; probe < RSH_BUCKETS; ++probe) {
                const int old = atomicCAS(&s_hkey[b], -1, v);
                if (old == -1) { res = (int)(b | 0x10000u); break; }
                if (old == v) { res = (int)b; break; }
                b = (b + 1) & (RSH_BUCKETS - 1);
            }
            if (res == -2) atomicAdd(&s_misc[0], RSH_EMAX + 1);      // table full: the tile takes the slow path
            eb[i] = res == -2 ? -1 : res;
        }
    }
    {   // C: the inserting lanes number their rows: one atomicAdd per wave, a lane's rows get consecutive numbers
        int mine = 0;
#pragma unroll
        for (int i = 0; i < NE; ++i) mine += (eb[i] >= 0 && (eb[i] & 0x10000)) ? 1 : 0;
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        int base = 0;
        if (lane == 63) base = atomicAdd(&s_misc[0], incl);
        int idx = __shfl(base, 63) + incl - mine;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const bool ins = eb[i] >= 0 && (eb[i] & 0x10000);
            if (ins) {
                s_hslot[eb[i] & 0xffff] = (unsigned short)min(wn + idx, 0x3ffe);
                if (idx < RSH_EMAX) s_extra[idx] = ev[i];
            }
            idx += ins ? 1 : 0;
        }
    }
    __syncthreads();
    // the numbers the atomics handed out depend on their order; renumber the rows outside the window by VALUE (rank among them):
    // which rows share a stage segment -- and with it the order of the fp32 additions when a stage holds only part of a tile's
    // rows -- must not depend on timing.  (~200 rows per tile on surface scans: 200 x 200 / 512 comparisons per thread)
    {
        int* s_tmp = reinterpret_cast<int*>(s_inv + RS_NW * 16 * RS_NF_MAX);          // RSH_EMAX ints + RSH_EMAX halfwords
        unsigned short* s_rank = reinterpret_cast<unsigned short*>(s_tmp + RSH_EMAX);
        const int ne = min(s_misc[0], RSH_EMAX);
        for (int e = tid; e < ne; e += RS_TPB) {
            const int v = s_extra[e];
            int rk = 0;
            for (int q = 0; q < ne; ++q) rk += s_extra[q] < v ? 1 : 0;
            s_tmp[e] = v;
            s_rank[e] = (unsigned short)rk;
        }
        __syncthreads();
        for (int e = tid; e < ne; e += RS_TPB) s_extra[s_rank[e]] = s_tmp[e];
        for (int b = tid; b < RSH_BUCKETS; b += RS_TPB) {
            const int old = (int)s_hslot[b] - wn;
            if (s_hkey[b] >= 0 && old >= 0 && old < ne) s_hslot[b] = (unsigned short)(wn + s_rank[old]);
        }
        __syncthreads();
    }
    // the tile's row order (cube maps): rows sorted by their 8 corner-offset bits -- rows that share a fragment then share most of
    // their neighbour pattern, and (fragment, offset) pairs without any neighbour (no MFMAs) become 0.62 instead of 0.77 of all
    // pairs at stride 1 (0.68 / 0.85 at stride 2; scripts/analyze_rulebook.py).  Counting sort, 256 buckets; the order inside a
    // bucket is whatever the atomics give: positions only decide which lane computes a row, never its arithmetic.
    if (sort_rows) {
        for (int r = tid; r < live; r += RS_TPB) atomicAdd(&s_hist[s_rowkey[r] & 255], 1);
        __syncthreads();
        if (wave == 0) {
            int h[4], sum = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { h[q] = s_hist[lane * 4 + q]; sum += h[q]; }
            int incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            int run = incl - sum;
#pragma unroll
            for (int q = 0; q < 4; ++q) { s_hist[lane * 4 + q] = run; run += h[q]; }
        }
        __syncthreads();
        for (int r = tid; r < live; r += RS_TPB) {
            const int pos = atomicAdd(&s_hist[s_rowkey[r] & 255], 1);
            s_inv[r] = (unsigned short)pos;
            s_perm[pos] = (unsigned short)r;
        }
        __syncthreads();
    }
    // D: the table (indexed by POSITION) and the fragment masks.  An entry holds the slot's first 16-byte position in the stage,
    // 4 * slot + rotation(slot) -- the operand address is a few VALU operations away -- or 0xffff.
    {
        int ps[NE], hs[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = ent(i);
            const int r = (int)(((float)e + 0.5f) * inv_k);
            ps[i] = 0; hs[i] = 0;
            if (ev[i] >= 0) ps[i] = s_inv[r];
            if (eb[i] >= 0) hs[i] = s_hslot[eb[i] & 0xffff];
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int v = ev[i];
            const int e = ent(i);
            const int r = (int)(((float)e + 0.5f) * inv_k), k = e - r * K;
            unsigned sl = RSH_NONE;
            if ((unsigned)(v - wlo) < (unsigned)wn) sl = (unsigned)(v - wlo);
            else if (eb[i] >= 0) sl = (unsigned)hs[i];
            const int fr = ps[i] >> 4;
            if (v >= 0) {
                s_slot[ps[i] * KP + k] = (unsigned short)(sl >= 0x3fffu ? RSH_NONE : sl * 4u + ((2u * (sl >> 2)) & 3u));
                atomicOr(reinterpret_cast<unsigned*>(&s_misc[8 + (fr & (RS_NW - 1)) * 32 + k]), 1u << (fr / RS_NW));
            }
        }
    }
    __syncthreads();
    return RshWindow{wlo, wn};
}

__global__ __launch_bounds__(RS_TPB) void k_rs_table_build(const int* __restrict__ nbr, const int K, const int* __restrict__ n_out_dev,
                                                            const int n_out, const int n_in, const int tile_rows, const int n_tiles,
                                                            unsigned char* __restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const RsRecord R = rs_record_layout(tile_rows, K);
    // LDS: the record's arrays at their record offsets, the scratch behind them
    int* s_misc = reinterpret_cast<int*>(smem + R.masks);
    unsigned short* s_perm = reinterpret_cast<unsigned short*>(smem + R.perm);
    int* s_extra = reinterpret_cast<int*>(smem + R.extra);
    unsigned short* s_slot = reinterpret_cast<unsigned short*>(smem + R.slots);
    unsigned char* scratch = smem + R.total + 64;
    const int tid = threadIdx.x;
    const int n = n_out_dev ? min(*n_out_dev, n_out) : n_out;
    const int tile = blockIdx.x;
    const int row0 = tile * tile_rows;
    if (row0 >= n) return;
    const int live = min(tile_rows, n - row0);
    for (int e = tid; e < RSH_EMAX; e += RS_TPB) s_extra[e] = 0;
    rsh_build_tile(nbr, K, (int)R.kpg, tile_rows, live, row0, n_in, tile_rows, true, true, s_slot, scratch, s_extra, s_misc, s_perm);
    __syncthreads();
    if (tile == 0 && tid == 0) {
        int* h = reinterpret_cast<int*>(table);
        h[0] = RS_TABLE_MAGIC; h[1] = tile_rows; h[2] = n_tiles; h[3] = K;
    }
    u32x4* dst = reinterpret_cast<u32x4*>(table + RS_TABLE_HEAD + (size_t)tile * R.total);
    const u32x4* src = reinterpret_cast<const u32x4*>(smem);
    for (int e = tid; e < (int)(R.total >> 4); e += RS_TPB) dst[e] = src[e];
}

template <typename T, int NF, int NT, int BO>
__global__ __launch_bounds__(RS_TPB) void k_spconv_rsh(const ConvArgs a, const int tile_rows_launch, const int n_tiles, const int s_cap,
                                                       const int n_in, const int n_in2) {
    static_assert(Tr<T>::ELEMS * sizeof(T) == 16, "one gather vector is 16 bytes");
    static_assert(BO % 4 == 0, "a batch is whole 8-byte groups of slot-table columns");
    constexpr int PWB = (BO * NT + RS_NW - 1) / RS_NW;      // weight pieces per wave and batch
    static_assert(PWB <= BO, "a wave issues at most one weight piece per step");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = a.K, KP = (K + BO - 1) / BO * BO;
    const RshLds L = rsh_layout(tile_rows_launch, K, s_cap, NT, BO);
    unsigned short* s_slot = reinterpret_cast<unsigned short*>(smem + L.slots);
    u32x4* s_w = reinterpret_cast<u32x4*>(smem + L.wring);
    int* s_extra = reinterpret_cast<int*>(smem + L.extra);
    int* s_misc = reinterpret_cast<int*>(smem + L.misc);       // [0] extras, [1] window low; [8 ..]: fragment masks [wave][offset]
    float* s_ss = reinterpret_cast<float*>(smem + L.ss);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, rl = lane & 15;
    const int n = a.n_out_dev ? min(*a.n_out_dev, a.n_out) : a.n_out;
    // capacity form: the launch's tiles share the rows that exist (see k_spconv_rs); the LDS layout stays the launch's
    if (n <= 0) return;                   // (a device-side count of 0: no tile height to cut, nothing to write)
    const int tile_rows = (a.n_out_dev && a.n_sel == a.n_out) ? min(tile_rows_launch, (((n + n_tiles - 1) / n_tiles) + 15) & ~15) : tile_rows_launch;
    const int n_work = (n + tile_rows - 1) / tile_rows;       // tiles that have rows: the XCD-aware map runs over those (see k_spconv_rs)
    if ((int)blockIdx.x >= n_work) return;
    const int tile = xcd_tile(blockIdx.x, n_work);
    const int row0 = tile * tile_rows;
    const int live = min(tile_rows, n - row0);
    const bool cube = (K == 27 || K == 125) && n_in == a.n_out;     // same-level map: the window is the tile's own rows
    RSH_T0
    if (tid < NT * 32) {
        const int c = tid < NT * 16 ? tid : tid - NT * 16;
        const float* src = tid < NT * 16 ? a.scale : a.shift;
        s_ss[tid] = src ? src[c] : (tid < NT * 16 ? 1.0f : 0.0f);
    }
    RshWindow win;
    bool use_tab = false;
    if (a.rs_table) {        // built for this geometry?  (a launch with another tile height ignores the table and builds its own)
        const int* th = reinterpret_cast<const int*>(a.rs_table);
        use_tab = th[0] == RS_TABLE_MAGIC && th[1] == tile_rows && th[2] == n_tiles && th[3] == K && cube;
    }
    if (use_tab) {
        // ---- the tile's tables were built once for the map (pbn_rs_table_build): record -> LDS ----
        const unsigned char* rec = reinterpret_cast<const unsigned char*>(a.rs_table) + RS_TABLE_HEAD + (size_t)tile * rs_table_record_bytes(tile_rows, K);
        const RsRecord R = rs_record_layout(tile_rows, K);
        const int* hdr = reinterpret_cast<const int*>(rec);
        for (int e = tid; e < 8 + RS_NW * 32; e += RS_TPB) s_misc[e] = hdr[e];
        const int ne = min(hdr[0], RSH_EMAX);
        for (int e = tid; e < ne; e += RS_TPB) s_extra[e] = reinterpret_cast<const int*>(rec + R.extra)[e];
        for (int e = tid; e < (tile_rows * 2 + 15) / 16; e += RS_TPB)
            reinterpret_cast<u32x4*>(smem + L.perm)[e] = reinterpret_cast<const u32x4*>(rec + R.perm)[e];
        const int kpg = R.kpg;                      // the record's table pitch (halfwords): whole 8-byte groups
        const int g8 = kpg >> 2, d8 = KP >> 2;
        for (int e = tid; e < tile_rows * d8; e += RS_TPB) {
            const int r = e / d8, q = e - r * d8;
            uint2 v = make_uint2(0xffffffffu, 0xffffffffu);
            if (q < g8) v = reinterpret_cast<const uint2*>(rec + R.slots)[r * g8 + q];
            reinterpret_cast<uint2*>(s_slot)[e] = v;
        }
        win.wlo = row0; win.wn = max(0, min(tile_rows, n_in - row0));
        __syncthreads();
    } else {
        win = rsh_build_tile(a.nbr, K, KP, tile_rows, live, row0, n_in, s_cap, cube, !(a.dbg & 1024), s_slot, smem + L.stage, s_extra, s_misc,
                             reinterpret_cast<unsigned short*>(smem + L.perm));
    }
    const int wlo = win.wlo, wn = win.wn;
    unsigned short* s_perm = reinterpret_cast<unsigned short*>(smem + L.perm);
    RSH_STAMP(1)       /* the tile's tables */
    const int n_extra = s_misc[0];
    const bool direct = n_extra > RSH_EMAX || (a.dbg & 128);       // hash set overflowed: operands lane by lane from global memory
    // lane k of every wave: which of the wave's NF fragments have a neighbour at offset k (lane K: the second source)
    unsigned mreg = lane < K ? (unsigned)s_misc[8 + wave * 32 + lane] : 0u;
    if (lane == K && a.in2) {
#pragma unroll
        for (int f = 0; f < NF; ++f)
            if ((f * RS_NW + wave) * 16 < live) mreg |= 1u << f;
    }
    if (a.dbg & 1) mreg = (lane < K || (lane == K && a.in2)) ? ((1u << NF) - 1u) : 0u;
    __syncthreads();                                               // the prologue arrays (in the stage region) are dead from here
    // zero slot
    if (tid < 16) reinterpret_cast<unsigned*>(smem + L.stage + (size_t)s_cap * 64)[tid] = 0u;
    const int n_act = (a.dbg & 16) ? 0 : K;                        // every offset is walked; a wave skips those its fragments lack
    RSH_STAMP(2)       /* masks */

    f32x4 acc[NF][NT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned long long in_addr = (unsigned long long)a.in, w_addr = (unsigned long long)a.w, in2_addr = (unsigned long long)a.in2;
    const i32x4 rs_in = {(int)(unsigned)in_addr, (int)(unsigned)(in_addr >> 32), (int)a.in_bytes, 0x00020000};
    const i32x4 rs_in2 = {(int)(unsigned)in2_addr, (int)(unsigned)(in2_addr >> 32), (int)a.in2_bytes, 0x00020000};
    const i32x4 rs_w = {(int)(unsigned)w_addr, (int)(unsigned)(w_addr >> 32), (int)a.w_bytes, 0x00020000};
    const unsigned ld_bytes = (unsigned)a.ld_in * (unsigned)sizeof(T), ld2_bytes = (unsigned)a.ld_in2 * (unsigned)sizeof(T);
    const unsigned lds_stage = PBN_RS_LDS_ADDR(smem + L.stage), lds_w = PBN_RS_LDS_ADDR(s_w);
    const unsigned w_lane = (unsigned)lane * 16u;
    constexpr unsigned wslot_bytes = (unsigned)BO * NT * 1024u;
    const int PC = a.vpo >> 2;                                  // 64-byte pieces per input row
    const int PC2 = a.in2 ? (a.vpo2 >> 2) : 0;
    const int s_used = direct ? 0 : wn + n_extra;
    const int nseg = (s_used + s_cap - 1) / s_cap > 1 ? (s_used + s_cap - 1) / s_cap : 1;
    const bool multi = nseg > 1;
    const unsigned zp = (unsigned)s_cap * 4u + ((2u * ((unsigned)s_cap >> 2)) & 3u);      // position of the zero slot (> every resident position)

    // the slow path's resources (compiler-scheduled buffer loads); the rulebook holds n_out * K ints (< 2 GiB / 4: checked by the launcher)
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_in2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in2), 0, (int)a.in2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_nbr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(a.nbr), 0, (int)((unsigned)a.n_out * (unsigned)K * 4u), 0x00020000);
    auto dma = [&](unsigned lds_addr, unsigned voff, const i32x4& rs, unsigned soff) {      // lds_addr, soff: wave-uniform
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
                     : "memory");
    };
    // piece i of this wave of the weights of batch `b` of a pass -> ring slot b & 1.  Batch b holds offsets b * BO .. b * BO + BO - 1;
    // step(k) = k * PC + c for the map, n_main + c for the second source
    auto dma_w_piece = [&](int b, int nb, bool second, int c, int i) {
        if (b >= nb || (a.dbg & 8)) return;
        const int q = wave + RS_NW * i;
        if (q >= BO * NT) return;
        const int j = q / NT, t = q - j * NT;
        const int k = second ? (j == 0 ? K : -1) : (b * BO + j < K ? b * BO + j : -1);
        const int step = second ? a.n_main + c : k * PC + c;
        dma(lds_w + (unsigned)(b & 1) * wslot_bytes + (unsigned)q * 1024u, k >= 0 ? w_lane : RS_OOB, rs_w,
            k >= 0 ? (unsigned)(step * NT + t) * 1024u : 0u);
    };

    // LDS halfword index of each fragment's row in the slot table (fragments past the tile read row 0: their mask bits are 0)
    int rb[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int r = (f * RS_NW + wave) * 16 + rl;
        rb[f] = (r < tile_rows ? r : 0) * KP;
    }
    const unsigned stage_off = L.stage;
    auto pos_of_slot = [](unsigned sl) -> unsigned { return sl * 4u + ((2u * (sl >> 2)) & 3u); };
    // the packed table entries of four columns of this lane's NF rows (one ds_read_b64 per fragment)
    auto load_sv = [&](uint2 (&sv)[NF], int col4) {
#pragma unroll
        for (int f = 0; f < NF; ++f) sv[f] = *reinterpret_cast<const uint2*>(s_slot + rb[f] + col4 * 4);
    };
    // One pass = one 64-byte piece of the input rows against all offsets.  The step loop has ONE shape; the rare cases are folded
    // into what the slot table holds and into scalars:
    //   * a stage that does not hold all rows (segments): the segment's first position is subtracted from every entry -- rows of
    //     lower segments wrap to huge values, rows of higher segments exceed the stage: both clamp to the zero slot;
    //   * the second source: the table's first column group is rewritten (row r -> its own slot r) before those passes, which come
    //     last; the pseudo-offset K is step 0 of a one-batch pass;
    //   * the slow path (hash set overflowed): the table holds every lane's PRIVATE slot (NF * 16 per wave, in the otherwise unused
    //     stage); before a step's operands are read the wave fetches that step's rows itself, lane by lane through buffer
    //     resources (32-bit offsets, compiler-scheduled), and parks them there.
    auto run_pass = [&](const bool second, const int c, const int seg_lo, const int seg_cnt) {
        const int nb = second ? 1 : (n_act + BO - 1) / BO;
        if (nb == 0) return;
        const unsigned seg_pos = 4u * (unsigned)seg_lo;       // seg_lo % 16 == 0: the rotation of a slot is that of its segment-local index
        RSH_STAMP(4)   /* between passes */
        __syncthreads();                                   // everybody is done with the previous pass's stage
        RSH_STAMP(5)   /* pass barrier */
        // ---- stage piece c of the segment's rows: instruction j covers slots 16 j .. 16 j + 15, four lanes per slot ----
        if (!direct && !(a.dbg & 64)) {
            const int cnt = second ? live : seg_cnt;
            for (int j = wave; j * 16 < cnt; j += RS_NW) {
                const int sl = j * 16 + (lane >> 2);        // slot inside the segment
                const int gs = sl + (second ? 0 : seg_lo);  // slot of the tile
                int row = -1;
                if (sl < cnt) row = second ? row0 + sl : (gs < wn ? wlo + gs : s_extra[gs - wn]);       // (second source: slot = row of the tile)
                const unsigned chunk = (unsigned)((lane & 3) - 2 * (sl >> 2)) & 3u;     // position p holds chunk (p - 2 q) & 3, q = slot / 4
                const unsigned voff = row >= 0 ? (unsigned)row * (second ? ld2_bytes : ld_bytes) + chunk * 16u : RS_OOB;
                dma(lds_stage + (unsigned)j * 1024u, voff, second ? rs_in2 : rs_in, (unsigned)c * 64u);
            }
        }
#pragma unroll
        for (int i = 0; i < PWB; ++i) dma_w_piece(0, nb, second, c, i);
        RSH_STAMP(6)   /* stage + first weight batch issued */
        // slow path: this wave's rows of offset k -> its private slots
        auto park = [&](int k, unsigned fm) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int ps = (f * RS_NW + wave) * 16 + rl;                  // position -> row of the tile
                const int r = ps < live ? (int)s_perm[ps] : tile_rows;
                int nbv = -1;
                if (k >= 0 && r < live && ((fm >> f) & 1u))
                    nbv = second ? row0 + r : __builtin_amdgcn_raw_buffer_load_b32(rsrc_nbr, (unsigned)((row0 + r) * K + k) * 4u, 0, 0);
                const unsigned dvoff = nbv >= 0 ? (unsigned)nbv * (second ? ld2_bytes : ld_bytes) + (unsigned)c * 64u + (unsigned)g * 16u : RS_OOB;
                const u32x4 xv = second ? __builtin_amdgcn_raw_buffer_load_b128(rsrc_in2, dvoff, 0, 0)
                                        : __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, dvoff, 0, 0);
                const unsigned t = pos_of_slot((unsigned)((wave * NF + f) * 16 + rl));
                *reinterpret_cast<u32x4*>(smem + stage_off + ((((t + (unsigned)g) & 3u) | (t & ~3u)) << 4)) = xv;
            }
        };
        // the row operands of one step: table entry -> LDS address (position with the lane's chunk rotated in) -> ds_read_b128
        auto fetch_x = [&](u32x4 (&x)[NF], const uint2 (&svx)[BO / 4][NF], int j) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const uint2 w = svx[j >> 2][f];
                const unsigned d = (j & 2) ? w.y : w.x;
                unsigned t = (j & 1) ? (d >> 16) : (d & 0xffffu);            // position of the slot, or 0xffff
                t = min(t - seg_pos, zp);
                const unsigned ad = (((t + (unsigned)g) & 3u) | (t & ~3u)) << 4;
                x[f] = *reinterpret_cast<const u32x4*>(smem + stage_off + ad);
            }
        };
        // software pipeline over the steps of the pass: the operands of step i + 1 are requested before the MFMAs of step i
        // (the stage does not change inside a pass, so the request may cross the batch barrier; weights may not)
        uint2 svc[BO / 4][NF], svn[BO / 4][NF];
        u32x4 xbuf[2][NF];
#pragma unroll
        for (int q = 0; q < BO / 4; ++q) load_sv(svc[q], q);
        for (int b = 0; b < nb; ++b) {
            RSH_STAMP(7)   /* batch tail */
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
            RSH_STAMP(8)   /* wait for own DMA */
            __syncthreads();                               // batch b (and the stage) has landed everywhere; slot (b + 1) & 1 is free
            RSH_STAMP(9)   /* batch barrier */
            const u32x4* wb = s_w + (b & 1) * (BO * NT * 64);
            // the batch's offsets and this wave's fragment masks: scalars; the next batch's table entries: prefetched
            int kk[BO + 1];
            unsigned fmv[BO + 1];
#pragma unroll
            for (int j = 0; j <= BO; ++j) {
                const int k = b * BO + j;
                kk[j] = second ? (j == 0 ? K : -1) : (k < n_act ? k : -1);
                fmv[j] = kk[j] >= 0 ? (unsigned)__builtin_amdgcn_readlane((int)mreg, kk[j] >= 0 ? kk[j] : 0) : 0u;
            }
            if (b + 1 < nb) {
#pragma unroll
                for (int q = 0; q < BO / 4; ++q) load_sv(svn[q], (b + 1) * (BO / 4) + q);
            } else {
#pragma unroll
                for (int q = 0; q < BO / 4; ++q)
#pragma unroll
                    for (int f = 0; f < NF; ++f) svn[q][f] = make_uint2(0xffffffffu, 0xffffffffu);
            }
            if (b == 0) {                                   // the pass's first operands: behind the barrier that says the stage has landed
                if (direct) park(kk[0], fmv[0]);
                fetch_x(xbuf[0], svc, 0);
            }
            RSH_STAMP(11)  /* batch head */
            // the weight fragments of a step are requested one step ahead as well (inside a batch: the next batch's need the barrier)
            u32x4 wfb[2][NT];
            if (fmv[0] != 0u) {
#pragma unroll
                for (int t = 0; t < NT; ++t) wfb[0][t] = wb[t * 64 + lane];
            }
#pragma unroll
            for (int j = 0; j < BO; ++j) {
                // request the next step's operands (next offset of the batch, or the first of the next batch)
                if (direct) park(kk[j + 1], fmv[j + 1]);
                if (j + 1 < BO) fetch_x(xbuf[(j + 1) & 1], svc, j + 1);
                else fetch_x(xbuf[(j + 1) & 1], svn, 0);
                if (j + 1 < BO && fmv[j + 1] != 0u) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        wfb[(j + 1) & 1][t] = wb[((j + 1) * NT + t) * 64 + lane];
                }
                if (j < PWB) dma_w_piece(b + 1, nb, second, c, j);      // the next batch's weights: one piece per step, not a burst
                if (fmv[j] != 0u) {
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        if ((fmv[j] >> f) & 1u) {
#pragma unroll
                            for (int t = 0; t < NT; ++t) mfma_step<T>(wfb[j & 1][t], xbuf[j & 1][f], acc[f][t]);
                        }
                    }
                }
            }
            RSH_STAMP(12)  /* the batch's steps */
#pragma unroll
            for (int q = 0; q < BO / 4; ++q)
#pragma unroll
                for (int f = 0; f < NF; ++f) svc[q][f] = svn[q][f];
        }
    };
    if (direct) {          // the table: every lane's private slot, whatever the offset
        __syncthreads();
        for (int e = tid; e < tile_rows * KP; e += RS_TPB) {
            const int r = e / KP;
            const int fr = r >> 4;
            s_slot[e] = (unsigned short)pos_of_slot((unsigned)(((fr & (RS_NW - 1)) * NF + fr / RS_NW) * 16 + (r & 15)));
        }
    }
    // all passes from ONE call site (one copy of the step loop, one set of accumulator registers): the map's pieces segment by
    // segment, then the second source's pieces
    const int n_main_pass = PC * nseg;
    for (int pass = 0; pass < n_main_pass + PC2; ++pass) {
        const bool second = pass >= n_main_pass;
        const int seg = second ? 0 : pass / PC;
        const int c = second ? pass - n_main_pass : pass - seg * PC;
        if (pass == n_main_pass) {     // the second source: row r reads slot r, one pseudo-offset (column 0 of the table; the map's entries are dead)
            __syncthreads();
            if (!direct)
                for (int r = tid; r < tile_rows; r += RS_TPB)
                    *reinterpret_cast<uint2*>(s_slot + r * KP) = make_uint2(0xffff0000u | (r < live ? pos_of_slot((unsigned)s_perm[r]) : 0xffffu), 0xffffffffu);
        }
        const int seg_lo = seg * s_cap;
        run_pass(second, c, seg_lo, direct ? 0 : min(s_cap, s_used - seg_lo));
    }
    RSH_STAMP(13)
    if (a.dbg & 32) return;
    rs_epilogue<T, NF, NT>(a, acc, s_ss, row0, live, wave, g, rl, s_perm);
    RSH_STAMP(14)      /* epilogue */
    RSH_TWRITE
}

int cu_count() {
    static const int cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return cus;
}

// tile height: the level cut into (a multiple of) one tile per CU, rounded up to whole fragments
struct RsShape { int tile_rows, n_tiles, nf; };
RsShape rs_shape(int n_sel, int force_nf, int force_rows = 0, int n_out = -1) {
    // n_sel: the rows the tile height is chosen for; n_out (>= n_sel: a capacity) the rows the tiles must cover
    if (n_out < 0) n_out = n_sel;
    RsShape s;
    if (force_rows > 0) {                    // tests: an explicit tile height (a multiple of 16, at most 128 rows per fragment count)
        s.tile_rows = ((force_rows + 15) / 16) * 16;
        s.nf = force_nf > 0 ? force_nf : cdiv(s.tile_rows, RS_NW * 16);
        if (s.tile_rows > RS_NW * 16 * s.nf) s.tile_rows = RS_NW * 16 * s.nf;
        s.n_tiles = cdiv(n_out, s.tile_rows);
        return s;
    }
    static const int cus_env = getenv("PBN_RS_CUS") ? atoi(getenv("PBN_RS_CUS")) : 0;
    const int cus = cus_env > 0 ? cus_env : cu_count();
    if (force_nf > 0) {
        s.nf = force_nf;
        s.tile_rows = RS_NW * 16 * force_nf;
        const int rounds = cdiv(cdiv(n_sel, s.tile_rows), cus);
        const int per = cdiv(n_sel, cus * (rounds > 0 ? rounds : 1));
        s.tile_rows = min(s.tile_rows, ((per + 15) / 16) * 16);
    } else {
        for (int rounds = 1;; ++rounds) {
            const int per = cdiv(n_sel, cus * rounds);
            s.tile_rows = ((per + 15) / 16) * 16;
            if (s.tile_rows <= RS_NW * 16 * RS_NF_MAX) break;
        }
        s.nf = cdiv(s.tile_rows, RS_NW * 16);
    }
    if (s.tile_rows < 16) s.tile_rows = 16;
    s.n_tiles = cdiv(n_out, s.tile_rows);
    return s;
}

template <typename T, int NF, int NT, int CG, int RING>
int launch_rs_one(const ConvArgs& a, const RsShape& s, hipStream_t stream) {
    const int KS = a.K | 1;
    const int list_cap = ((a.n_steps / CG + 8 + 3) & ~3);
    const size_t lds = (size_t)RING * CG * NT * 1024 +
                       sizeof(int) * ((size_t)((s.tile_rows * KS + 3) & ~3) + list_cap + 4 + 2 * NT * 16);
    if (lds > 160 * 1024) return PBN_ERR_UNSUPPORTED;
    auto kern = k_spconv_rs<T, NF, NT, CG, RING>;
    if (lds > 64 * 1024)
        PBN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(s.n_tiles), dim3(RS_TPB), lds, stream, a, s.tile_rows, s.n_tiles, list_cap);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

template <typename T, int NT, int CG, int RING>
int launch_rs_nf(const ConvArgs& a, const RsShape& s, hipStream_t stream) {
    switch (s.nf) {
        case 1: return launch_rs_one<T, 1, NT, CG, RING>(a, s, stream);
        case 2: return launch_rs_one<T, 2, NT, CG, RING>(a, s, stream);
        case 3: return launch_rs_one<T, 3, NT, CG, RING>(a, s, stream);
        case 4: return launch_rs_one<T, 4, NT, CG, RING>(a, s, stream);
        case 5: return launch_rs_one<T, 5, NT, CG, RING>(a, s, stream);
        default: return PBN_ERR_UNSUPPORTED;
    }
}

template <typename T>
int launch_rs_t(const ConvArgs& a, const RsShape& s, int cg, hipStream_t stream) {
    const int nt = a.ntiles_total;
    if (nt == 6 && cg == 3) return launch_rs_nf<T, 6, 3, 2>(a, s, stream);
    if (nt == 6 && cg == 4) return launch_rs_nf<T, 6, 4, 2>(a, s, stream);
    if (nt == 2 && cg == 1) return launch_rs_nf<T, 2, 1, 2>(a, s, stream);
    if (nt == 2 && cg == 2) return launch_rs_nf<T, 2, 2, 2>(a, s, stream);
    return PBN_ERR_UNSUPPORTED;
}

template <typename T, int NF, int NT, int BO>
int launch_rsh_one(const ConvArgs& a, const RsShape& s, hipStream_t stream) {
    // stage capacity: what is left of the CU's 160 KiB behind the slot table, the weight ring and the small arrays
    const RshLds fixed = rsh_layout(s.tile_rows, a.K, 0, NT, BO);
    static const int cap_env = getenv("PBN_RSH_SLOTS") ? atoi(getenv("PBN_RSH_SLOTS")) : 0;     // tests: a small stage forces segments
    long long s_cap = ((long long)160 * 1024 - (long long)fixed.total) / 64;
    if (s_cap > 4000) s_cap = 4000;
    if (cap_env > 0 && cap_env < s_cap) s_cap = cap_env;
    s_cap &= ~15LL;                         // whole staging instructions (16 slots each): the last one never writes past the stage
    if (s_cap < s.tile_rows || s_cap < RS_NW * 16 * NF) return PBN_ERR_UNSUPPORTED;   // (the slow path parks NF * 16 rows per wave)
    if ((s_cap + 1) * 64 < RSH_SCRATCH_BYTES) return PBN_ERR_UNSUPPORTED;   // the prologue's arrays live in the stage region
    const RshLds L = rsh_layout(s.tile_rows, a.K, (int)s_cap, NT, BO);
    auto kern = k_spconv_rsh<T, NF, NT, BO>;
    if (L.total > 64 * 1024)
        PBN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total));
    const int esz = (int)sizeof(T);
    const int n_in = a.ld_in > 0 ? (int)(a.in_bytes / ((unsigned)a.ld_in * esz)) : 0;
    const int n_in2 = (a.in2 && a.ld_in2 > 0) ? (int)(a.in2_bytes / ((unsigned)a.ld_in2 * esz)) : 0;
    hipLaunchKernelGGL(kern, dim3(s.n_tiles), dim3(RS_TPB), L.total, stream, a, s.tile_rows, s.n_tiles, (int)s_cap, n_in, n_in2);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

template <typename T, int NT, int BO>
int launch_rsh_nf(const ConvArgs& a, const RsShape& s, hipStream_t stream) {
    switch (s.nf) {
        case 1: return launch_rsh_one<T, 1, NT, BO>(a, s, stream);
        case 2: return launch_rsh_one<T, 2, NT, BO>(a, s, stream);
        case 3: return launch_rsh_one<T, 3, NT, BO>(a, s, stream);
        case 4: return launch_rsh_one<T, 4, NT, BO>(a, s, stream);
        case 5: return launch_rsh_one<T, 5, NT, BO>(a, s, stream);
        default: return PBN_ERR_UNSUPPORTED;
    }
}

template <typename T>
int launch_rsh_t(const ConvArgs& a, const RsShape& s, hipStream_t stream) {
    if (a.ntiles_total == 6) return launch_rsh_nf<T, 6, 4>(a, s, stream);
    if (a.ntiles_total == 2) return launch_rsh_nf<T, 2, 12>(a, s, stream);
    return PBN_ERR_UNSUPPORTED;
}

int rs_cg(const ConvArgs& a) {
    if (a.vpo & 3) return 0;
    const int spo = a.vpo >> 2;
    for (int c = 4; c >= 1; --c)
        if (spo % c == 0) return c;
    return 1;
}

}  // namespace

// Where each form pays (MI355X, bench scene, bf16).  Single layers from a HIP graph (scripts/probe_rs.py, profiles/r05_probe_rs.txt):
// k = 3 cube maps with the map's tables, staged form: 96->96 at 146 k rows 96.5 -> 74.6 us (gather form 82.1), 128->96 135.8 -> 91.6,
// 32->32 at 57 k rows 21.4 -> 16.2; 96 channels at 57 k rows only draws (tiles of 240 rows: the weights are streamed per 240 rows again).
// Inside the pipeline (scripts/op_table.py, profiles/r05_op_table_*.txt; scripts/ab_rs.sh: alternating bench runs on one box) the
// tables do not earn their build: the first layer of a level pays ~20-25 us for them, the three to five layers that follow give
// back 5-8 us each -- convolution ops of one scene 3 490 us (k_spconv) -> 3 332 (tables) / 3 340 (none), and with four scenes in
// flight 330-332 scenes/s (k_spconv) -> 325-335 (tables) / 341-342 (none).  So by default no tables are built
// (PBN_UNET_RS_TABLES=1 builds them), and the automatic choice is:
//   * the gather form for every wide launch it is built for (a few per cent to 20 % ahead of k_spconv: 96->96 at 146 k rows
//     93.7-98.8 -> 79.9-86.3 us, k = 2 transposed 96->96 47.1 -> 40.7, 128->96 at 26 k rows 46.0 -> 35.6), from 20 k rows for 96 output
//     channels and from 40 k rows for 32 (at 26 k rows x 32 channels it loses: 12.4 -> 18.8);
//   * the staged form with in-kernel tables for the one shape whose gather instantiation spills (128 input channels at 5
//     fragments per wave: 128->96 at 146 k rows 163 -> 118.5 us), and for cube maps that come with tables.
bool rs_staged_pays(int n_out, int ntiles_total) {
    static const int min6 = getenv("PBN_RSH_MIN_ROWS6") ? atoi(getenv("PBN_RSH_MIN_ROWS6")) : 100000;
    static const int min2 = getenv("PBN_RSH_MIN_ROWS2") ? atoi(getenv("PBN_RSH_MIN_ROWS2")) : 40000;
    return (ntiles_total == 6 && n_out >= min6) || (ntiles_total == 2 && n_out >= min2);
}

bool rs_family_wanted(const ConvArgs& a, int dtype) {
    static const int env = getenv("PBN_CONV_RS") ? atoi(getenv("PBN_CONV_RS")) : 1;
    static const int min_rows = getenv("PBN_RS_MIN_ROWS") ? atoi(getenv("PBN_RS_MIN_ROWS")) : 20000;
    static const int min_rows2 = getenv("PBN_RS_MIN_ROWS2") ? atoi(getenv("PBN_RS_MIN_ROWS2")) : 40000;
    if (!env || dtype == PBN_F32 || a.row_perm || a.K > 32 || a.n_sel < min_rows) return false;
    const int cg = rs_cg(a);
    if (!cg) return false;
    const int nt = a.ntiles_total;
    return (nt == 6 && (cg == 3 || cg == 4)) || (nt == 2 && cg == 1 && a.n_sel >= min_rows2);
}

// cfg: 0 = automatic (form and tile height); 1..5 = that many fragments per wave (tests, tuning); + 1000: the staged form (rows in
// LDS), + 2000: the gather form; 1..5 alone: PBN_RS_MODE decides (default: staged); + 100000 * (rows / 16): that tile height
int launch_rs(const ConvArgs& a, int dtype, int cfg, hipStream_t stream) {
    const int force_rows = (cfg / 100000) * 16;      // + 100000 * (tile height / 16): an explicit tile height (tests)
    cfg %= 100000;
    static const int mode_env = getenv("PBN_RS_MODE") ? atoi(getenv("PBN_RS_MODE")) : -1;
    const int cg = rs_cg(a);
    int staged = mode_env < 0 ? 1 : mode_env;
    const bool automatic = cfg == 0 && mode_env < 0;
    if (cfg >= 2000) { staged = 0; cfg -= 2000; }
    else if (cfg >= 1000) { staged = 1; cfg -= 1000; }
    if (a.row_perm || a.K > 32 || (a.vpo & 3) || cfg < 0 || cfg > RS_NF_MAX) return PBN_ERR_UNSUPPORTED;
    if (automatic) {
        const RsShape sh = rs_shape(a.n_sel, 0, 0, a.n_out);
        const unsigned esz = dtype == PBN_F32 ? 4u : 2u;
        const bool cube = a.K == 27 && a.ld_in > 0 && (int)(a.in_bytes / ((unsigned)a.ld_in * esz)) == a.n_out;
        const bool gather_ok = !(a.ntiles_total == 6 && cg == 4 && sh.nf == 5);       // (that instantiation spills)
        staged = (cube && a.rs_table && rs_staged_pays(a.n_sel, a.ntiles_total)) || !gather_ok;
    }
    if (staged && a.nbr && (a.ntiles_total == 6 || a.ntiles_total == 2) && (!a.in2 || !(a.vpo2 & 3))) {
        const RsShape sh = rs_shape(a.n_sel, cfg, force_rows, a.n_out);
        int rc = PBN_ERR_UNSUPPORTED;
        switch (dtype) {
            case PBN_BF16: rc = launch_rsh_t<__hip_bfloat16>(a, sh, stream); break;
            case PBN_F16: rc = launch_rsh_t<__half>(a, sh, stream); break;
            case PBN_F32: rc = launch_rsh_t<float>(a, sh, stream); break;
            default: return PBN_ERR_ARG;
        }
        if (rc != PBN_ERR_UNSUPPORTED) return rc;
    }
    if (!a.nbr) return PBN_ERR_UNSUPPORTED;                    // identity maps (1x1 / linear) stay on the other families
    if (a.in2 && ((a.vpo2 & 3) || a.n_main % cg || a.n_steps % cg)) return PBN_ERR_UNSUPPORTED;
    if (!a.in2 && a.n_steps % cg) return PBN_ERR_UNSUPPORTED;
    ConvArgs b = a;
    b.cg = cg;
    const RsShape s = rs_shape(a.n_sel, cfg, force_rows, a.n_out);
    switch (dtype) {
        case PBN_BF16: return launch_rs_t<__hip_bfloat16>(b, s, cg, stream);
        case PBN_F16: return launch_rs_t<__half>(b, s, cg, stream);
        case PBN_F32: return launch_rs_t<float>(b, s, cg, stream);
        default: return PBN_ERR_ARG;
    }
}

}  // namespace pbn

extern "C" size_t pbn_rs_table_bytes(int n_out, int n_offsets) {
    if (n_out <= 0 || n_offsets != 27) return 0;
    const pbn::RsShape sh = pbn::rs_shape(n_out, 0);
    return (size_t)pbn::RS_TABLE_HEAD + (size_t)sh.n_tiles * pbn::rs_table_record_bytes(sh.tile_rows, n_offsets);
}

extern "C" int pbn_rs_table_build(const int32_t* nbr, int n_offsets, const int32_t* n_out_dev, int n_out, void* table,
                                  size_t table_bytes, pbn_stream_t stream_) {
    using namespace pbn;
    if (n_out < 0 || n_offsets != 27) return PBN_ERR_ARG;
    if (n_out == 0) return PBN_OK;
    if (!nbr || !table || (((uintptr_t)table) & 15)) return PBN_ERR_ARG;
    if (pbn_rs_table_bytes(n_out, n_offsets) > table_bytes) return PBN_ERR_WORKSPACE;
    const RsShape sh = rs_shape(n_out, 0);
    const RsRecord R = rs_record_layout(sh.tile_rows, n_offsets);
    const size_t lds = (size_t)R.total + 64 + RSH_SCRATCH_BYTES;
    if (lds > 160 * 1024) return PBN_ERR_UNSUPPORTED;
    if (lds > 64 * 1024)
        PBN_HIP_CHECK(hipFuncSetAttribute((const void*)k_rs_table_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_rs_table_build, dim3(sh.n_tiles), dim3(RS_TPB), lds, (hipStream_t)stream_, nbr, n_offsets, n_out_dev, n_out, n_out,
                       sh.tile_rows, sh.n_tiles, (unsigned char*)table);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

#ifdef PBN_CONV_TIMING
// debug build only: the phase counters of the last k_spconv_rsh launch -> host (RSH_TBLOCKS x 8 waves x 16 counters)
extern "C" int pbn_rsh_timing_read(unsigned long long* host) {
    PBN_HIP_CHECK(hipDeviceSynchronize());
    PBN_HIP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(pbn::g_rsh_timing), sizeof(unsigned long long) * 256 * 8 * 16));
    return PBN_OK;
}
#endif
