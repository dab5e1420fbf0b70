// experiments/spconv_pc.hip -- PAIR-COMPACTED implicit-GEMM sparse convolution for the wide levels (stride 1 / 2) on gfx950.  Round 6.
//
// NOT part of the product library: `make experiments` links it into libpbnet_hip_exp.so (spconv.hip built with -DPBN_EXPERIMENTS
// routes configuration code 13000 + 100000 * (tile rows / 16) here).  It is correct -- bit-identical to k_spconv, tests/test_pc_gpu.py
// runs tests/experiments/pc_cases.py against that library -- and it LOSES: 96 -> 96 at 146 k rows 122-142 us against 82 us for the
// row-stationary gather form.  What was measured (profiles/r06_pc_*.txt) is at the end of this header.
//
// Same arithmetic, operand layouts, packed weights and fused epilogue as spconv.hip / spconv_rs.hip (MinkowskiConvolution /
// ConvolutionTranspose forward, /root/reference/network/Mink.py:221-288,293-350):
//     out[o, :] = epilogue( sum_k in[nbr[o,k], :] @ W[k] )
// What round 5 measured on the output-stationary mappings (profiles/r05_gather_micro_192.txt, r05_ablate_rs_gather.json): the
// CU's vector-memory path is paid per INSTRUCTION whatever its lanes return, and a fragment of 16 CONSECUTIVE output rows is
// populated at 77 % of its offsets while a row has a neighbour at only 28 % of them -- 2.8 x the useful gathers and MFMAs are
// issued.  Here every gather lane and every MFMA row is a real rule pair:
//   * a workgroup of 8 waves owns a tile of R output rows whose fp32 accumulators live in LDS (R x Cout x 4 B: 240 rows at 96
//     channels), not in registers;
//   * prologue: the tile's rulebook rows are staged in LDS (coalesced copy into the region that becomes the output tile) and
//     compacted per offset k, in row order, into lists of pairs (local output row << 23 | input row) by ballot + prefix count;
//   * main loop: one interval per populated offset, offsets ascending.  The pairs of the offset are cut into fragments of 16
//     PAIRS, dealt to the waves round-robin; a wave gathers its fragment's 16 input rows (every lane live), contracts them with
//     W[k] on the matrix cores with the tile's CURRENT values of its 16 output rows as the accumulator input (ds_read_b128)
//     and writes the 16 x Cout result back (ds_write_b128).  Inside one offset an output row occurs at most once, so the rows
//     a wave reads and rewrites in an interval are its own; offsets are separated by the barrier the weight ring needs anyway
//     (behind an lgkmcnt(0) of every wave), so each output element runs through ONE accumulator chain, offsets ascending,
//     channel steps ascending -- k_spconv's summation order: results are bit-identical to it and run to run.
//     (First form of the round: the partials added with ds_add_f32.  The LDS floating-point atomic takes ~78 cycles per
//     wave-instruction on gfx950 -- a lane per cycle --: 96 -> 96 at 146 k rows ran 851 us, profiles/r06_pc_first_form.txt.)
//   * W[k] arrives through a two-slot LDS-DMA ring one offset ahead (18 KiB per offset at 96 -> 96); the gathers of offset
//     k + 1 are issued before the MFMAs of offset k (double-buffered registers);
//   * epilogue: the tile leaves LDS through scale / shift / residual / ReLU as coalesced 8-byte stores.
// Parity: tests/test_pc_gpu.py (<= 1e-4 absolute against the oracle in fp32, equal to k_spconv, run-to-run identical).
// The second source (a BasicBlock's folded 1x1 shortcut, pbn_spconv_forward_dual) is one more, dense, offset whose pairs are
// (row, row); a device-side row count re-cuts the tiles over the rows that exist (as k_spconv_rs).
//
// Why it loses (MI355X, bench scene, bf16, 96 -> 96 at 146 k rows; the budget was 30-45 us):
//   * the fp32 tile bounds the tile height at 240 rows (192 after cutting the level into whole rounds), so an offset holds ~3.4
//     fragments per tile: 27 barrier-separated intervals per tile with one fragment for each of 3-4 waves, three tiles per CU.  With
//     MFMAs, gathers, weight loads and write-back ALL switched off the launch still takes 110 us of 142 (prologue + epilogue 37 us,
//     the interval skeleton 73 us = ~2 200 cycles per interval: barrier, lane-register reads, address arithmetic, the branches of
//     a wave-uniform work split -- instruction issue of two waves per SIMD, profiles/r06_pc_ablate.txt); MFMAs are worth 10 us,
//     the gathers 2 us;
//   * every tile streams all 0.5 MB of weights (18 KiB per interval against ~10 KiB of gathered rows): 3 x the row-stationary
//     form's weight traffic per CU.  Through buffer_load ... lds the copies paced the loop (~25 GB/s per issuing wave);
//   * ds_add_f32 runs a lane per cycle (~78 cycles per wave-instruction): 851 us.  Owner read-modify-write (ds_read_b128 /
//     ds_write_b128, what is here now) is 8 x faster and keeps k_spconv's summation order;
//   * loads whose issue depends on the wave's share of the fragments cannot be hand-counted (inline asm in wave-uniform branches:
//     hundreds of v_mov at the joins, 256 registers) and compiler-counted loads drain the queue at every use.
#include <cstdlib>
#include <type_traits>
#include "../spconv_common.h"

namespace pbn {
namespace {

#define PBN_PC_LDS_ADDR(p) ((unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)(p)))

constexpr int PC_NW = 8;               // waves per workgroup
constexpr int PC_TPB = PC_NW * 64;
constexpr int PC_NFW_MAX = 3;          // fragments per wave and offset: tiles of up to 8 x 3 x 16 = 384 rows (4: the 32-channel shapes spill)
constexpr unsigned PC_OOB = 0x80000000u;
constexpr int PC_SHIFT = 23;           // pair entry = local output row << 23 | input row (input rows < 2^23 - 1)
constexpr unsigned PC_IN_MASK = (1u << PC_SHIFT) - 1u;

#ifdef PBN_CONV_TIMING
constexpr int PC_TBLOCKS = 256;
__device__ unsigned long long g_pc_timing[PC_TBLOCKS * 8 * 16];
#define PC_T0 unsigned long long t_last_ = __builtin_readcyclecounter(); unsigned long long t_acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PC_STAMP(I) { const unsigned long long t_now_ = __builtin_readcyclecounter(); t_acc_[I] += t_now_ - t_last_; t_last_ = t_now_; }
#define PC_TWRITE if (blockIdx.x < PC_TBLOCKS && lane == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) g_pc_timing[(blockIdx.x * 8 + wave) * 16 + i_] = t_acc_[i_]; }
#else
#define PC_T0
#define PC_STAMP(I)
#define PC_TWRITE
#endif

// row pitch of the output tile in LDS: one 16-byte slot more than the row (an odd number of slots: 16 consecutive rows start in
// 16 different slots of the 256-byte bank row -- ds_read_b128 / ds_write_b128 of a fragment's rows conflict only where two of its
// rows are congruent modulo 16 / 8)
__host__ __device__ inline unsigned pc_pitch(int nt) { return (unsigned)nt * 64u + 16u; }
struct PcLds { unsigned w, out, pairs, cnt, list, ss, total; };
__host__ __device__ inline PcLds pc_layout(int rows, int K, int nt, int cg, int list_cap) {
    PcLds L;
    unsigned o = 0;
    L.w = o; o += 2u * (unsigned)cg * (unsigned)nt * 1024u;                 // two ring slots of one interval's weights
    L.out = o; o += (unsigned)(rows + 1) * pc_pitch(nt);                    // fp32 output tile (+ a dump row for padding pairs)
    L.pairs = o; o += (unsigned)rows * (unsigned)K * 4u;                    // [K][rows] compacted pair lists
    L.cnt = o; o += (((unsigned)K + 2u) * 4u + 15u) & ~15u;                 // pairs per offset, [K + 1] = number of intervals
    L.list = o; o += (unsigned)list_cap * 4u;                               // intervals: offset | sub-group << 16
    L.ss = o; o += 2u * (unsigned)nt * 64u;                                 // scale | shift
    L.total = o;
    return L;
}

template <typename T, int NT, int CG, int NFW>
__global__ __launch_bounds__(PC_TPB) void k_spconv_pc(const ConvArgs a, const int rows_launch, const int n_tiles, const int list_cap) {
    static_assert(Tr<T>::ELEMS * sizeof(T) == 16, "one gather vector is 16 bytes");
    constexpr int NPIECE = CG * NT;                           // weight pieces (1 KiB) per interval
    constexpr unsigned ROW_BYTES = (unsigned)NT * 64u + 16u;  // pitch of an output row in the tile (pc_pitch)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = a.K, RL = rows_launch;
    const PcLds L = pc_layout(RL, K, NT, CG, list_cap);
    u32x4* s_w = reinterpret_cast<u32x4*>(smem + L.w);
    float* s_out = reinterpret_cast<float*>(smem + L.out);
    int* s_stage = reinterpret_cast<int*>(smem + L.out);      // the rulebook tile is staged where the output tile will live
    unsigned* s_pairs = reinterpret_cast<unsigned*>(smem + L.pairs);
    int* s_cnt = reinterpret_cast<int*>(smem + L.cnt);
    int* s_list = reinterpret_cast<int*>(smem + L.list);
    float* s_ss = reinterpret_cast<float*>(smem + L.ss);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, rl = lane & 15;
    PC_T0
    const int n = a.n_out_dev ? min(*a.n_out_dev, a.n_out) : a.n_out;
    if (n <= 0) return;
    // capacity form without a rows hint: the launch's tiles share the rows that exist (as k_spconv_rs)
    const int R = (a.n_out_dev && a.n_sel == a.n_out) ? min(RL, (((n + n_tiles - 1) / n_tiles) + 15) & ~15) : RL;
    const int n_work = (n + R - 1) / R;
    if ((int)blockIdx.x >= n_work) return;
    const int tile = xcd_tile(blockIdx.x, n_work);
    const int row0 = tile * R;
    const int live = min(R, n - row0);

    if (tid <= K + 1) s_cnt[tid] = 0;
    if (tid < NT * 32) {
        const int c = tid < NT * 16 ? tid : tid - NT * 16;
        const float* src = tid < NT * 16 ? a.scale : a.shift;
        s_ss[tid] = src ? src[c] : (tid < NT * 16 ? 1.0f : 0.0f);
    }
    // ---- prologue: rulebook rows -> LDS stage (chunks of whole 64-row groups that fit the output region), compacted per offset ----
    {
        const int stage_ints = (int)(((unsigned)(RL + 1) * ROW_BYTES) >> 2);
        int chunk = (stage_ints / K) & ~63;
        if (chunk > R) chunk = (R + 63) & ~63;
        for (int c0 = 0; c0 < live; c0 += chunk) {
            const int crows = min(chunk, live - c0);
            const int nints = crows * K;
            const int* src = a.nbr + (size_t)(row0 + c0) * K;
            if ((((size_t)(row0 + c0) * K) & 3) == 0) {
                const int nv = nints >> 2;
                const int4* src4 = reinterpret_cast<const int4*>(src);
                int4* dst4 = reinterpret_cast<int4*>(s_stage);
#pragma unroll 4
                for (int e = tid; e < nv; e += PC_TPB) dst4[e] = src4[e];
                for (int e = (nv << 2) + tid; e < nints; e += PC_TPB) s_stage[e] = src[e];
            } else {
#pragma unroll 4
                for (int e = tid; e < nints; e += PC_TPB) s_stage[e] = src[e];
            }
            __syncthreads();
            // wave w compacts columns w, w + 8, ...: nobody else touches a column's list or count.  Four 64-row groups per pass:
            // their reads are in flight together
            const bool last = c0 + chunk >= live;
            for (int k = wave; k < K; k += PC_NW) {
                int base = c0 > 0 ? s_cnt[k] : 0;
                for (int i0 = 0; i0 < crows; i0 += 256) {
                    int v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = i0 + u * 64 + lane;
                        v[u] = r < crows ? s_stage[r * K + k] : -1;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool ok = v[u] >= 0;
                        const unsigned long long m = __ballot(ok);
                        const int rank = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        if (ok) s_pairs[k * RL + rank] = ((unsigned)(c0 + i0 + u * 64 + lane) << PC_SHIFT) | ((unsigned)v[u] & PC_IN_MASK);
                        base += __popcll(m);
                    }
                }
                if (lane == 0) s_cnt[k] = base;
                // the list padded to whole fragments: a padding pair reads nothing and lands in the dump row
                if (last && lane < ((16 - (base & 15)) & 15)) s_pairs[k * RL + base + lane] = ((unsigned)RL << PC_SHIFT) | PC_IN_MASK;
            }
            __syncthreads();
        }
    }
    PC_STAMP(0)
    const int gpo = (a.vpo >> 2) / CG;                          // sub-groups (intervals) per offset
    const int n_main_groups = (a.in2 ? a.n_main : a.n_steps) / CG;
    const int n2_groups = a.in2 ? a.n_steps / CG - n_main_groups : 0;
    // ---- the interval list (wave 0: populated offsets in ascending order x their sub-groups, then the second source) ----
    if (wave == 0) {
        int npop = 0;
        for (int kb = 0; kb < K; kb += 64) {
            const int k = kb + lane;
            const int c = k < K ? s_cnt[k] : 0;
            const unsigned long long m = __ballot(c > 0);
            const int rank = npop + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (c > 0)
                for (int sg = 0; sg < gpo; ++sg) s_list[rank * gpo + sg] = k | (sg << 16);
            npop += __popcll(m);
        }
        const int total = npop * gpo;
        if (lane < n2_groups) s_list[total + lane] = K | (lane << 16);
        if (lane == 0) s_cnt[K + 1] = total + n2_groups;
    }
    // ---- zero the output tile (the stage is dead behind the last barrier above) ----
    {
        float4* o4 = reinterpret_cast<float4*>(s_out);
        const int nv = (int)(((unsigned)(RL + 1) * ROW_BYTES) >> 4);
#pragma unroll 4
        for (int e = tid; e < nv; e += PC_TPB) o4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int ng = (a.dbg & 16) ? 0 : __builtin_amdgcn_readfirstlane(s_cnt[K + 1]);
    // every wave keeps the intervals in two registers: lane i (register h: interval 64 h + i) holds the list entry and its pair
    // count -- the loop reads them with v_readlane instead of two dependent LDS round trips per interval
    int meta_pk[2], meta_cnt[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = h * 64 + lane;
        meta_pk[h] = i < ng ? s_list[i] : 0;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = h * 64 + lane;
        const int ko = meta_pk[h] & 0xffff;
        meta_cnt[h] = i < ng ? (ko == K ? live : s_cnt[ko]) : 0;
    }
    PC_STAMP(1)

    // buffer resources: a gather is one instruction with a 32-bit per-lane byte offset, an absent pair reads zeros out of range
    const unsigned w_lane = (unsigned)lane * 16u;
    constexpr unsigned slot_bytes = (unsigned)NPIECE * 1024u;
    const unsigned lds_out = PBN_PC_LDS_ADDR(s_out);

    auto interval_pk = [&](int p) -> int {
        return p < 64 ? __builtin_amdgcn_readlane(meta_pk[0], p) : __builtin_amdgcn_readlane(meta_pk[1], p - 64);
    };
    // the wave's share of interval p's fragments: fragments are dealt in runs, wave w takes [w * per, (w + 1) * per)
    auto interval_share = [&](int p, int& f0) -> int {
        const int cnt = p < 64 ? __builtin_amdgcn_readlane(meta_cnt[0], p) : __builtin_amdgcn_readlane(meta_cnt[1], p - 64);
        const int nfrag = (cnt + 15) >> 4;
        const int per = (nfrag + PC_NW - 1) / PC_NW;
        f0 = wave * per;
        int nact = nfrag - f0;
        nact = nact < 0 ? 0 : (nact > per ? per : nact);
        return nact > NFW ? NFW : nact;
    };
    // Weights of list entry `pk` -> ring slot, piece q = step q / NT, channel tile q % NT (1 KiB each), through REGISTERS: the
    // copies of interval p + 1 are fetched during interval p by the waves that have no fragment in p (`first` = the number of
    // waves that do) -- plus as many of the busy ones, from the top, as it takes to keep a wave's share at PWMAX pieces -- and
    // written to LDS (ds_write_b128) behind the wave's MFMAs.  (Second form of the round: buffer_load ... lds.  The LDS-DMA path
    // lands ~25 GB/s per issuing wave (guide: ldsdma-fill); at 18 KiB of weights per ~1 000-cycle interval the copies, not the
    // gathers, paced the loop: 2 800 cycles per interval, profiles/r06_pc_dma_form.txt.)
    constexpr int PWMAX = NPIECE <= 4 ? 1 : 6;
    constexpr int MIN_ISS = (NPIECE + PWMAX - 1) / PWMAX;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)a.w_bytes, 0x00020000);
    // The loads of the loop are buffer-load BUILTINS, counted by the compiler: which of them a wave issues in an interval depends
    // on its share of the fragments, so hipcc's counters (which must hold on every path) drain the wave's queue -- vmcnt(0) -- in
    // front of the first use.  The loop is ordered so that this costs nothing: what is pending at a wave's first MFMA was issued an
    // interval or more ago.  (Inline-asm loads with hand-counted waits -- the k_spconv_rs way -- were tried: inside wave-uniform
    // branches the register allocator reconciles the tied 128-bit operands with hundreds of v_mov at the joins, 96 -> 256 registers.)
    // returns `first` (the wave is an issuer with rank wave - first) or -1
    auto weights_load = [&](int pk, int first, u32x4 (&wt)[PWMAX]) -> int {
        const int ko = pk & 0xffff, sub = pk >> 16;
        const int gi = ko == K ? n_main_groups + sub : ko * gpo + sub;
        const unsigned gbase = (unsigned)gi * slot_bytes;
        if (first > PC_NW - MIN_ISS) first = PC_NW - MIN_ISS;
        if (wave < first) return -1;
        const int n_iss = PC_NW - first;
        const unsigned wv = (a.dbg & 8) ? PC_OOB : w_lane;
#pragma unroll
        for (int i = 0; i < PWMAX; ++i) {
            const int q = wave - first + i * n_iss;
            if (q < NPIECE) wt[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv, __builtin_amdgcn_readfirstlane(gbase + (unsigned)q * 1024u), 0);
        }
        return first;
    };
    auto weights_store = [&](int first, u32x4* slot, const u32x4 (&wt)[PWMAX]) {
        if (first < 0) return;
        const int n_iss = PC_NW - first;
#pragma unroll
        for (int i = 0; i < PWMAX; ++i) {
            const int q = wave - first + i * n_iss;
            if (q < NPIECE) slot[q * 64 + lane] = wt[i];
        }
    };
    // waves that hold fragments of interval p
    auto interval_busy = [&](int p) -> int {
        if (p >= ng) return 0;
        const int cnt = p < 64 ? __builtin_amdgcn_readlane(meta_cnt[0], p) : __builtin_amdgcn_readlane(meta_cnt[1], p - 64);
        const int nfrag = (cnt + 15) >> 4;
        const int per = (nfrag + PC_NW - 1) / PC_NW;
        return per > 0 ? (nfrag + per - 1) / per : 0;
    };
    // pair entries of the wave's fragments of interval p (nothing waits for them here): local output row << 23 | input row
    auto read_entries = [&](int p, unsigned (&E)[NFW]) -> int {
        if (p >= ng) return 0;
        const int pk = interval_pk(p);
        const int ko = pk & 0xffff;
        int f0;
        const int nact = interval_share(p, f0);
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
            if (j < nact) {
                const int f = f0 + j;
                if (ko != K) E[j] = s_pairs[ko * RL + f * 16 + rl];
                else {
                    const int r = f * 16 + rl;
                    E[j] = r < live ? (((unsigned)r << PC_SHIFT) | (unsigned)(row0 + r)) : (((unsigned)RL << PC_SHIFT) | PC_IN_MASK);
                }
            }
        }
        return nact;
    };
    // gathers of interval p: the input rows of the wave's fragments (every lane a real pair but in a list's last fragment)
    auto gathers = [&](int p, int nact, const unsigned (&E)[NFW], u32x4 (&X)[NFW][CG]) {
        if (nact <= 0) return;
        const int pk = interval_pk(p);
        const int ko = pk & 0xffff, sub = pk >> 16;
        const bool second = ko == K;
        const int nxv = second ? (a.vpo2 >> 2) - sub * CG : CG;             // chunks of this sub-group that exist
        const unsigned cvb = (unsigned)(sub * CG * 4 + g) * 16u;
        // the source of this interval: scalar selects (a select between two resource VALUES goes through memory)
        const unsigned ldb = (unsigned)(second ? a.ld_in2 : a.ld_in) * (unsigned)sizeof(T);
        void* const src_base = const_cast<void*>(second ? a.in2 : a.in);
        const int src_bytes = (int)(second ? a.in2_bytes : a.in_bytes);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(src_base, 0, src_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
            if (j < nact) {
                const unsigned in_row = E[j] & PC_IN_MASK;
                const unsigned voff = (in_row != PC_IN_MASK && !(a.dbg & 4)) ? in_row * ldb + cvb : PC_OOB;
#pragma unroll
                for (int c = 0; c < CG; ++c)
                    X[j][c] = __builtin_amdgcn_raw_buffer_load_b128(rs, (c < nxv) ? voff : PC_OOB, c * 64, 0);
            }
        }
    };
    // one interval, first half: the current values of the wave's output rows -> accumulators (lane (g, rl): row of pair rl, channels
    // 16 t + 4 g .. + 3: the MFMA's C layout with rows = channels, columns = pairs)
    auto load_acc = [&](int nact, const unsigned (&E)[NFW], f32x4 (&acc)[NFW][NT]) {
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
            if (j < nact) {
                const unsigned o = lds_out + (E[j] >> PC_SHIFT) * ROW_BYTES + (unsigned)g * 16u;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[j][t] = *(const f32x4*)((__attribute__((address_space(3))) const f32x4*)(uintptr_t)(o + (unsigned)t * 64u));
            }
        }
    };
    // second half: the wave's `nact` fragments against the weights in `cur`, results back into the tile
    auto compute = [&](const u32x4* cur, int nact, u32x4 (&X)[NFW][CG], const unsigned (&E)[NFW], f32x4 (&acc)[NFW][NT]) {
        if (nact <= 0) return;
        if (!(a.dbg & 2)) {
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                u32x4 wf[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) wf[t] = cur[(c * NT + t) * 64 + lane];
#pragma unroll
                for (int j = 0; j < NFW; ++j) {
                    if (j < nact) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) mfma_step<T>(wf[t], X[j][c], acc[j][t]);      // rows = channels, columns = pairs
                    }
                }
            }
        }
        if (a.dbg & 64) {                    // ablation: no write-back (the accumulators must stay alive)
#pragma unroll
            for (int j = 0; j < NFW; ++j)
#pragma unroll
                for (int t = 0; t < NT; ++t) asm volatile("" : : "v"(acc[j][t]));
            return;
        }
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
            if (j < nact) {
                const unsigned o = lds_out + (E[j] >> PC_SHIFT) * ROW_BYTES + (unsigned)g * 16u;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    *(f32x4*)((__attribute__((address_space(3))) f32x4*)(uintptr_t)(o + (unsigned)t * 64u)) = acc[j][t];
            }
        }
    };

    if (ng > 0) {
        // Software pipeline (a wave's vector-memory instructions issue in order and stall while the CU's address path is busy --
        // ~30 loads of ~30 cycles per interval --, so a wave with fragments runs its MFMAs FIRST and issues loads behind them):
        // at the top of interval p the wave holds the entries of p, p + 1, p + 2 (E0 / E1 / E2); its gathers of p (issued behind the
        // MFMAs of p - 2) have landed in X[p & 1] or are waited for by the first MFMA, those of p + 1 are in flight; the weights of
        // p are in slot p & 1.  Body: accumulator reads, MFMAs and write-back of p; then the gathers of p + 2 into the buffer the
        // MFMAs just released and the entry reads of p + 3.  Waves without fragments fetch the weights of p + 1.
        // Weights: W(q + 2) is fetched into registers at the END of interval q (set q & 1) and written to slot q & 1 -- the slot
        // interval q read -- behind the MFMAs of interval q + 1: no global latency between two barriers.
        u32x4 X0[NFW][CG], X1[NFW][CG];
        u32x4 wtE[PWMAX], wtO[PWMAX];
        unsigned E0[NFW], E1[NFW], E2[NFW], E3[NFW];
        f32x4 acc[NFW][NT];
        int n0 = 0, n1 = 0, n2 = 0, n3 = 0, wfE = -1, wfO = -1;
#pragma unroll
        for (int j = 0; j < NFW; ++j) E0[j] = E1[j] = E2[j] = E3[j] = ((unsigned)RL << PC_SHIFT) | PC_IN_MASK;
        wfE = weights_load(interval_pk(0), 0, wtE);
        if (ng > 1) wfO = weights_load(interval_pk(1), 0, wtO);
        n1 = read_entries(0, E1);
        n2 = read_entries(1, E2);
        n3 = read_entries(2, E3);
        gathers(0, n1, E1, X0);
        gathers(1, n2, E2, X1);
        weights_store(wfE, s_w, wtE);
        for (int pos = 0; pos < ng; pos += 2) {
            // ---- even interval: weights in slot 0, operands in X0 ----
            __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): this wave's LDS writes (tile rows, weight pieces) are done
            PC_STAMP(2)
            __builtin_amdgcn_s_barrier();
            PC_STAMP(3)
#pragma unroll
            for (int j = 0; j < NFW; ++j) { E0[j] = E1[j]; E1[j] = E2[j]; E2[j] = E3[j]; }
            n0 = n1; n1 = n2; n2 = n3;
            load_acc(n0, E0, acc);
            compute(s_w, n0, X0, E0, acc);
            PC_STAMP(5)
            if (pos + 1 < ng) weights_store(wfO, s_w + NPIECE * 64, wtO);
            wfE = pos + 2 < ng ? weights_load(interval_pk(pos + 2), interval_busy(pos), wtE) : -1;
            if (pos + 2 < ng) gathers(pos + 2, n2, E2, X0);
            n3 = read_entries(pos + 3, E3);
            PC_STAMP(8)
            if (pos + 1 >= ng) break;
            // ---- odd interval: weights in slot 1, operands in X1 ----
            __builtin_amdgcn_s_waitcnt(0xc07f);
            PC_STAMP(2)
            __builtin_amdgcn_s_barrier();
            PC_STAMP(3)
#pragma unroll
            for (int j = 0; j < NFW; ++j) { E0[j] = E1[j]; E1[j] = E2[j]; E2[j] = E3[j]; }
            n0 = n1; n1 = n2; n2 = n3;
            load_acc(n0, E0, acc);
            compute(s_w + NPIECE * 64, n0, X1, E0, acc);
            PC_STAMP(5)
            if (pos + 2 < ng) weights_store(wfE, s_w, wtE);
            wfO = pos + 3 < ng ? weights_load(interval_pk(pos + 3), interval_busy(pos + 1), wtO) : -1;
            if (pos + 3 < ng) gathers(pos + 3, n2, E2, X1);
            n3 = read_entries(pos + 4, E3);
            PC_STAMP(8)
        }
    }
    __builtin_amdgcn_s_waitcnt(0);            // vmcnt(0) lgkmcnt(0): the builtin, so that the compiler's own counters know
    __syncthreads();
    PC_STAMP(6)

    // ---- epilogue: a thread takes 4 consecutive channels of a row; consecutive threads consecutive channel groups (coalesced) ----
    if (!(a.dbg & 32)) {
        T* out = reinterpret_cast<T*>(a.out);
        const T* res = reinterpret_cast<const T*>(a.residual);
        const bool has_scale = a.scale != nullptr, has_shift = a.shift != nullptr, relu = a.relu != 0;
        constexpr int VPR = NT * 4;                              // 4-channel vectors per row
        const int total = live * VPR;
        // four vectors per thread and pass: their LDS reads and residual loads are in flight together
        for (int e0 = tid; e0 < total; e0 += 4 * PC_TPB) {
            f32x4 v[4], rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * PC_TPB;
                const int ec = e < total ? e : total - 1;
                const int r = ec / VPR, c0 = (ec - r * VPR) * 4;
                const float4 q = *reinterpret_cast<const float4*>(s_out + r * (int)(ROW_BYTES >> 2) + c0);
                v[u] = f32x4{q.x, q.y, q.z, q.w};
                if (res) rv[u] = load4<T>(res + (size_t)(row0 + r) * a.ld_res + c0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * PC_TPB;
                const int ec = e < total ? e : total - 1;
                const int r = ec / VPR, c0 = (ec - r * VPR) * 4;
                f32x4 w = v[u];
                if (has_scale) {
                    const float4 sc = *reinterpret_cast<const float4*>(s_ss + c0);
                    w[0] *= sc.x; w[1] *= sc.y; w[2] *= sc.z; w[3] *= sc.w;
                }
                if (has_shift) {
                    const float4 sh = *reinterpret_cast<const float4*>(s_ss + NT * 16 + c0);
                    w[0] += sh.x; w[1] += sh.y; w[2] += sh.z; w[3] += sh.w;
                }
                if (res) w += rv[u];
                if (relu) {
                    w[0] = fmaxf(w[0], 0.f); w[1] = fmaxf(w[1], 0.f); w[2] = fmaxf(w[2], 0.f); w[3] = fmaxf(w[3], 0.f);
                }
                if (e < total) store4<T>(out + (size_t)(row0 + r) * a.ld_out + c0, w);
            }
        }
    }
    PC_STAMP(7)
    PC_TWRITE
}

int pc_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

struct PcShape { int rows, n_tiles, nfw, list_cap; unsigned lds; };
// tile height: as many rows as the CU's LDS holds behind the weight ring, then the level cut into whole rounds of one tile per CU
bool pc_shape(const ConvArgs& a, int nt, int cg, int force_rows, PcShape* s) {
    const int K = a.K;
    const int gpo = (a.vpo >> 2) / cg;
    const int n2 = a.in2 ? (a.n_steps - a.n_main) / cg : 0;
    s->list_cap = (K * gpo + n2 + 2 + 3) & ~3;
    if (K * gpo + n2 > 128) return false;            // the kernel keeps the intervals in two 64-lane registers
    const PcLds fixed = pc_layout(0, K, nt, cg, s->list_cap);
    const long long per_row = (long long)pc_pitch(nt) + (long long)K * 4;
    long long rmax = ((long long)160 * 1024 - (long long)fixed.total) / per_row;
    rmax &= ~15LL;
    if (rmax > PC_NW * 16 * PC_NFW_MAX - 16) rmax = PC_NW * 16 * PC_NFW_MAX - 16;     // (the dump row's index must fit the entry's 9 row bits)
    if (rmax < 16) return false;
    // the rulebook stage must hold at least one 64-row group
    if ((long long)(rmax + 1) * pc_pitch(nt) < (long long)64 * K * 4) return false;
    int rows;
    if (force_rows > 0) rows = (int)min((long long)((force_rows + 15) & ~15), rmax);
    else {
        static const int cus_env = getenv("PBN_PC_CUS") ? atoi(getenv("PBN_PC_CUS")) : 0;
        const int cus = cus_env > 0 ? cus_env : pc_cu_count();
        const int rounds = cdiv(cdiv(a.n_sel, (int)rmax), cus);
        const int per = cdiv(a.n_sel, cus * (rounds > 0 ? rounds : 1));
        rows = (per + 15) & ~15;
        if (rows > rmax) rows = (int)rmax;
        if (rows < 16) rows = 16;
    }
    if ((long long)(rows + 1) * pc_pitch(nt) < (long long)64 * K * 4) return false;
    s->rows = rows;
    s->n_tiles = cdiv(a.n_out, rows);
    s->nfw = cdiv(rows, PC_NW * 16);
    s->lds = pc_layout(rows, K, nt, cg, s->list_cap).total;
    return s->lds <= 160u * 1024u;
}

template <typename T, int NT, int CG, int NFW>
int launch_pc_one(const ConvArgs& a, const PcShape& s, hipStream_t stream) {
    auto kern = k_spconv_pc<T, NT, CG, NFW>;
    if (s.lds > 64 * 1024)
        PBN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.lds));
    hipLaunchKernelGGL(kern, dim3(s.n_tiles), dim3(PC_TPB), s.lds, stream, a, s.rows, s.n_tiles, s.list_cap);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

template <typename T, int NT, int CG>
int launch_pc_nfw(const ConvArgs& a, const PcShape& s, hipStream_t stream) {
    switch (s.nfw) {
        case 1: return launch_pc_one<T, NT, CG, 1>(a, s, stream);
        case 2: return launch_pc_one<T, NT, CG, 2>(a, s, stream);
        case 3: if constexpr (NT <= 2) return launch_pc_one<T, NT, CG, 3>(a, s, stream); else return PBN_ERR_UNSUPPORTED;
        default: return PBN_ERR_UNSUPPORTED;
    }
}

int pc_cg(const ConvArgs& a) {
    if (a.vpo & 3) return 0;
    const int spo = a.vpo >> 2;
    for (int c = 4; c >= 1; --c)
        if (spo % c == 0) return c;
    return 1;
}

template <typename T>
int launch_pc_t(const ConvArgs& a, int cg, int force_rows, hipStream_t stream) {
    const int nt = a.ntiles_total;
    PcShape s;
    if (!pc_shape(a, nt, cg, force_rows, &s)) return PBN_ERR_UNSUPPORTED;
    if (nt == 6 && cg == 3) return launch_pc_nfw<T, 6, 3>(a, s, stream);
    if (nt == 6 && cg == 4) return launch_pc_nfw<T, 6, 4>(a, s, stream);
    if (nt == 2 && cg == 1) return launch_pc_nfw<T, 2, 1>(a, s, stream);
    if (nt == 2 && cg == 2) return launch_pc_nfw<T, 2, 2>(a, s, stream);
    return PBN_ERR_UNSUPPORTED;
}

}  // namespace

// Which launches take this family automatically (PBN_CONV_PC: 0 never, 1 where measured to pay -- the default --, 2 wherever built)
bool pc_family_wanted(const ConvArgs& a, int dtype) {
    static const int env = getenv("PBN_CONV_PC") ? atoi(getenv("PBN_CONV_PC")) : 0;
    static const int min_rows = getenv("PBN_PC_MIN_ROWS") ? atoi(getenv("PBN_PC_MIN_ROWS")) : 20000;
    if (!env || dtype == PBN_F32 || !a.nbr || a.row_perm || a.K > 64 || (a.vpo & 3) || a.n_sel < min_rows) return false;
    const int cg = pc_cg(a);
    const int nt = a.ntiles_total;
    return (nt == 6 && (cg == 3 || cg == 4)) || (nt == 2 && (cg == 1 || cg == 2));
}

// cfg: 0 = automatic tile height; otherwise the tile height in rows (tests, tuning)
int launch_pc(const ConvArgs& a, int dtype, int cfg, hipStream_t stream) {
    if (!a.nbr || a.row_perm || a.K > 64 || (a.vpo & 3) || cfg < 0) return PBN_ERR_UNSUPPORTED;
    const int cg = pc_cg(a);
    if (!cg) return PBN_ERR_UNSUPPORTED;
    if (a.in2 && ((a.vpo2 & 3) || a.n_main % cg || a.n_steps % cg)) return PBN_ERR_UNSUPPORTED;
    if (!a.in2 && a.n_steps % cg) return PBN_ERR_UNSUPPORTED;
    // a pair entry holds the input row in 23 bits
    const unsigned esz = dtype == PBN_F32 ? 4u : 2u;
    const unsigned long long n_in = a.ld_in > 0 ? a.in_bytes / ((unsigned long long)a.ld_in * esz) : 0ull;
    if (n_in >= PC_IN_MASK || (unsigned)a.n_out >= PC_IN_MASK) return PBN_ERR_UNSUPPORTED;
    ConvArgs b = a;
    b.cg = cg;
    switch (dtype) {
        case PBN_BF16: return launch_pc_t<__hip_bfloat16>(b, cg, cfg, stream);
        case PBN_F16: return launch_pc_t<__half>(b, cg, cfg, stream);
        case PBN_F32: return launch_pc_t<float>(b, cg, cfg, stream);
        default: return PBN_ERR_ARG;
    }
}

}  // namespace pbn

#ifdef PBN_CONV_TIMING
// debug build only: the phase counters of the last k_spconv_pc launch -> host (PC_TBLOCKS x 8 waves x 16 counters)
extern "C" int pbn_pc_timing_read(unsigned long long* host) {
    PBN_HIP_CHECK(hipDeviceSynchronize());
    PBN_HIP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(pbn::g_pc_timing), sizeof(unsigned long long) * 256 * 8 * 16));
    return PBN_OK;
}
#endif
