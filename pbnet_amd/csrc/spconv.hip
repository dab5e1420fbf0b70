// spconv.hip -- output-stationary implicit-GEMM sparse convolution on gfx950 matrix cores (wave64, MFMA 16x16).
//
// Re-creates the arithmetic the reference delegates to MinkowskiEngine's convolution kernels
// (/root/reference/network/Mink.py:221-288 constructors, :293-350 call sites; PBNet.py:43-82 linear heads):
//     out[o, :] = epilogue( sum_k  in[nbr[o,k], :] @ W[k] )        nbr from coords.hip, -1 = no neighbour
// One kernel serves k=5 / k=3 / k=2,s=2 down / k=2,s=2 transposed / 1x1 / linear: they differ only in the table.
//
// Mapping (MI355X-first, not a translation of gather-GEMM-scatter):
//   * workgroup = 4 waves x NF 16-row fragments = TM output rows (128 in production) x NT*16 output channels;
//     accumulators stay in registers for the whole K*Cin reduction, so there is no scatter, no atomics and a fixed
//     summation order (deterministic);
//   * the reduction axis is the flattened (kernel offset, input channel) axis cut into STEPS of 4 x 16-byte vectors;
//     each lane gathers its MFMA operand (16 B of one neighbour row) straight from the feature slab in HBM/L2 --
//     rows are contiguous, a 16-lane group reads whole 64-B row segments;
//   * the rulebook tile nbr[TM][K] is staged once in LDS; offsets without a neighbour in the tile are dropped from the
//     tile's group list, waves whose fragments are empty for an offset skip its MFMAs;
//   * weights are pre-packed in MFMA-fragment order and reach LDS through the LDS-DMA path (two ring slots, shared by
//     the 4 waves); the main loop is described at k_spconv below;
//   * the products are computed transposed (D = W^T-tile x X^T-tile) so that each lane ends up with 4 CONSECUTIVE
//     output channels of one row: the epilogue (folded BN scale/shift, bias, residual, ReLU, down-cast) is applied
//     in registers and written with 8/16-byte stores;
//   * blockIdx -> tile mapping is XCD-aware (contiguous tile ranges per XCD) so neighbouring tiles, which gather
//     overlapping rows, share an L2.
//
// Numerics: fp32 accumulate always.  f32 slabs use v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain) -- this is the
// parity configuration (1e-4 vs the oracle); bf16/f16 slabs use v_mfma_f32_16x16x32_{bf16,f16}.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "spconv_common.h"

namespace pbn {
namespace {

constexpr int CGMAX = 4;  // channel chunks (steps) per group

// (Round 3: the quad-coalesced row gathers of spconv_wave.hip -- lane 4 r + c fetches chunk c of row r, operand order
// restored by ds_bpermute -- were tried here too and LOSE: L0 96->96 82 -> 105 us with the permutes in front of the MFMAs,
// 113 us with them one chunk ahead.  This kernel's LDS pipe already carries the weight DMA and the fragment reads; the
// crossbar traffic of 24 permutes per group and wave does not fit beside them.)

#define PBN_LDS_ADDR(p) ((unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)(p)))

// Cycle stamps of the main loop (debug build only: make timing -> libpbnet_hip_timing.so, scripts/conv_timing.py):
// the middle workgroup's four waves record s_memtime at 8 points of their first 64 groups.
#ifdef PBN_CONV_TIMING
__device__ unsigned g_conv_timing[4 * 64 * 8 + 8];
#define PBN_STAMP(POS, I)                                                                                             \
    if (timed_ && lane == 0 && (POS) - g_lo < 64)                                                                     \
        s_time[(wave * 64 + ((POS) - g_lo)) * 8 + (I)] = (unsigned)__builtin_readcyclecounter();
#define PBN_TIMING_SKIP_DMA if (a.dbg & 64) continue   /* experiment: no weight copies at all (results are garbage) */
#else
#define PBN_STAMP(POS, I)
#define PBN_TIMING_SKIP_DMA
#endif

// Reduction axis: STEP s = 4 x 16-byte vectors of the flattened (offset, channel) axis; GROUP = cg consecutive steps
// of ONE kernel offset (cg = largest divisor <= 4 of the steps per offset; 1 when an offset is narrower than a step).
//
// Main loop, per group (one barrier):
//   * the group's weight tile (cg x NT KiB, fragment order) is copied global -> LDS by the DMA path
//     (buffer_load ... lds: no staging registers, no ds_write pass) into one of two ring slots, one group ahead;
//   * a lane's gather operands live in ONE register set: as soon as the matrix cores have consumed chunk c of the
//     current group, the same registers are refilled with chunk c of the next group, so every gather has a full
//     group of MFMA work to land behind;
//   * a wave owns NF 16-row fragments: each weight fragment read from LDS feeds NF MFMAs; fragments are read one
//     chunk ahead of the MFMAs that use them.
// Every vector-memory instruction of the loop is issued from inline asm with hand-counted s_waitcnt vmcnt(N): loads
// complete in issue order, so "the DMA of this group has landed" and "chunk c has landed" are fixed counts of the
// loads issued after them -- the barrier never waits for the gathers behind the DMA, an MFMA never waits for a DMA.
// (hipcc's own wait-count insertion is conservative across the loop's branches and drains the queue.)
template <typename T, int NF, int NT, int RING, int NW = 4>
__global__ __launch_bounds__(NW * 64) void k_spconv(const ConvArgs a) {
    static_assert(Tr<T>::ELEMS * sizeof(T) == 16, "one gather vector is 16 bytes");
    constexpr int CONV_TPB = NW * 64;                    // NW = 8 (round 4): one weight ring feeds 8 waves = twice the rows per weight byte
    constexpr int RW = NF * 16;
    constexpr int TM = NW * RW;
    constexpr int NFRAG = TM / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = a.K;
    const int KS = K | 1;  // odd row pitch: conflict-free column reads of the rulebook tile
    const int cg = a.cg;
    const int n_groups = a.n_steps / cg;
    constexpr int DEPTH = RING - 1;                                                 // weight tiles in flight ahead
    u32x4* s_w = reinterpret_cast<u32x4*>(smem);                                    // RING slots x cg * NT * 64 vectors
    int* s_nbr = reinterpret_cast<int*>(smem + (size_t)RING * cg * NT * 1024);      // TM * KS
    int* s_valid = s_nbr + TM * KS;                                                 // K (+1: the second source) fragment masks
    int* s_grp = s_valid + ((K + 4) & ~3);                                          // n_groups + 4 (entry n_groups = count)
    int* s_masks = s_grp + n_groups + 4;                                            // n_groups + 4 fragment masks
    int* s_gko = s_masks + n_groups + 4;                                            // n_groups + 4: offset | sub-group << 16
    float* s_ss = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(s_gko + n_groups + 4) + 15) & ~(uintptr_t)15);  // scale | shift

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform for the compiler
    const int n = a.n_out_dev ? min(*a.n_out_dev, a.n_out) : a.n_out;
    u32x4 pf_sink = {0u, 0u, 0u, 0u};
    if (blockIdx.y == 0 && blockIdx.z == 0) pf_sink = prefetch_next_weights(a, blockIdx.x, gridDim.x, tid, CONV_TPB);
    // the XCD-aware tile map runs over the tiles that HAVE rows (a capacity-sized grid holds more: mapped over the grid, the last
    // XCDs would get only empty tiles and the first ones all the work -- round 5)
    const int nt_work = a.n_out_dev ? (n + TM - 1) / TM : (int)gridDim.x;
    const int row0 = (int)blockIdx.x < nt_work ? xcd_tile(blockIdx.x, nt_work) * TM : n;
#ifdef PBN_CONV_TIMING
    __shared__ unsigned s_time[NW * 64 * 8];
    const bool timed_ = blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == 0;
    if (timed_) for (int e = tid; e < NW * 64 * 8; e += CONV_TPB) s_time[e] = 0u;
#endif
    if (row0 >= n) { prefetch_drain(pf_sink); return; }
    const int tile0 = blockIdx.y * NT;

    for (int k = tid; k <= K; k += CONV_TPB) s_valid[k] = 0;
    // epilogue constants of this workgroup's NT*16 output channels -> LDS (read back after the reduction)
    if (a.ksplit == 1 && tid < NT * 32) {
        const int c = tile0 * 16 + (tid < NT * 16 ? tid : tid - NT * 16);
        const float* src = tid < NT * 16 ? a.scale : a.shift;
        s_ss[tid] = src ? src[c] : (tid < NT * 16 ? 1.0f : 0.0f);
    }
    // rulebook tile -> LDS.  Full tiles of an un-permuted launch are one contiguous, 16-byte aligned block of TM*K
    // ints: copied with independent 16-byte loads; otherwise element-wise (float-reciprocal division, exact here).
    if (a.nbr && !a.row_perm && row0 + TM <= n && KS == K && ((TM * K) & 3) == 0) {
        const int4* src = reinterpret_cast<const int4*>(a.nbr + (size_t)row0 * K);
        int4* dst = reinterpret_cast<int4*>(s_nbr);
        const int nv = (TM * K) >> 2;
#pragma unroll 4
        for (int e = tid; e < nv; e += CONV_TPB) dst[e] = src[e];
    } else {
        const float inv_k = 1.0f / (float)K;
#pragma unroll 2
        for (int e = tid; e < TM * K; e += CONV_TPB) {
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
    // per offset: bitmask of the 16-row fragments that have at least one neighbour there
    for (int e = tid; e < K * NFRAG; e += CONV_TPB) {
        const int k = e / NFRAG, fr = e - k * NFRAG;
        int any = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) any |= (s_nbr[(fr * 16 + i) * KS + k] >= 0) ? 1 : 0;
        if (any) atomicOr(&s_valid[k], 1 << fr);
    }
    __syncthreads();
    // ordered list of the groups that touch at least one populated offset, with their fragment masks (wave 0)
    const int vpo = a.vpo;
    const bool wide = (vpo & 3) == 0;
    const int gpo = wide ? (vpo >> 2) / cg : 1;          // groups per offset (wide layers)
    // second source (a.in2): its steps follow the K offsets as one more "offset" K whose neighbour is the row itself
    const int n_main_groups = a.in2 ? a.n_main / cg : n_groups;
    if (a.in2 && tid == 0) {
        const int live = min(TM, n - row0);
        s_valid[K] = (int)((live >= TM) ? ((NFRAG >= 32) ? 0xffffffffu : ((1u << NFRAG) - 1u)) : ((1u << ((live + 15) >> 4)) - 1u));
    }
    if (a.in2) __syncthreads();
    if (wave == 0) {
        int base = 0;
        for (int g0 = 0; g0 < n_groups; g0 += 64) {
            const int gi = g0 + lane;
            int fm = 0, ko = 0;
            if (gi < n_groups) {
                if (gi >= n_main_groups) {
                    ko = K;
                    fm = s_valid[K];
                } else if (wide) {
                    ko = gi / gpo;
                    fm = s_valid[ko];
                } else {
                    for (int q = 0; q < 4; ++q) {
                        const int kq = (gi * 4 + q) / vpo;
                        if (kq < K) fm |= s_valid[kq];
                    }
                }
            }
            const bool ok = fm != 0;
            const unsigned long long m = __ballot(ok);
            if (ok) {
                const int pos = base + __popcll(m & ((1ULL << lane) - 1ULL));
                s_grp[pos] = gi;
                s_masks[pos] = fm | ((a.dbg & 1) ? 0xffff : 0);
                s_gko[pos] = ko | ((gi >= n_main_groups ? gi - n_main_groups : gi - ko * gpo) << 16);
            }
            base += __popcll(m);
        }
        if (lane == 0) s_grp[n_groups] = base;
    }
    __syncthreads();
    int ng = __builtin_amdgcn_readfirstlane(s_grp[n_groups]);
    int g_lo = 0;
    if (a.ksplit > 1) {  // this workgroup reduces only its slice of the group list
        g_lo = (int)((long long)ng * blockIdx.z / a.ksplit);
        ng = (int)((long long)ng * (blockIdx.z + 1) / a.ksplit);
    }
    const unsigned my_bits = ((1u << NF) - 1u) << (wave * NF);

    f32x4 acc[NF][NT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, rl = lane & 15;

    // Both operands come in through buffer resources: a gather is ONE instruction with a 32-bit per-lane byte offset,
    // and a fragment without a neighbour simply uses an out-of-range offset -- the hardware bounds check returns
    // zeros, so the inner loop has no exec-mask branches and no 64-bit address arithmetic.  num_records is the exact
    // extent of the slab / the packed weights (both < 2 GiB: pbn_spconv_forward returns PBN_ERR_RANGE otherwise), so a
    // corrupt rulebook entry cannot read outside them either.
    const unsigned long long in_addr = (unsigned long long)a.in, w_addr = (unsigned long long)a.w;
    const i32x4 rs_in = {(int)(unsigned)in_addr, (int)(unsigned)(in_addr >> 32), (int)a.in_bytes, 0x00020000};
    const i32x4 rs_w = {(int)(unsigned)w_addr, (int)(unsigned)(w_addr >> 32), (int)a.w_bytes, 0x00020000};
    constexpr unsigned OOB = 0x80000000u;  // >= num_records: the bounds check turns the load into zeros
    const unsigned ld_bytes = (unsigned)a.ld_in * (unsigned)sizeof(T);
    const unsigned long long in2_addr = (unsigned long long)a.in2;
    const i32x4 rs_in2 = {(int)(unsigned)in2_addr, (int)(unsigned)(in2_addr >> 32), (int)a.in2_bytes, 0x00020000};
    const unsigned ld2_bytes = (unsigned)a.ld_in2 * (unsigned)sizeof(T);
    bool nx2_ = false;                     // the group whose rows are being fetched reads the second source
    int nxv_ = 4;                          // ... and this many of its chunks exist
    auto pick_rs = [&](bool second) -> i32x4 {          // wave-uniform choice, made provably scalar for the "s" constraint
        const bool s2 = __builtin_amdgcn_readfirstlane((int)second) != 0;
        return i32x4{__builtin_amdgcn_readfirstlane(s2 ? rs_in2[0] : rs_in[0]), __builtin_amdgcn_readfirstlane(s2 ? rs_in2[1] : rs_in[1]),
                     __builtin_amdgcn_readfirstlane(s2 ? rs_in2[2] : rs_in[2]), 0x00020000};
    };
    const int vshift = (vpo == 2) ? 1 : 0;
    const unsigned w_step_bytes = (unsigned)a.ntiles_total * 1024u;   // one step, all channel tiles
    const unsigned w_lane = (unsigned)lane * 16u;
    const unsigned slot_bytes = (unsigned)cg * NT * 1024u;
    const unsigned lds_w = PBN_LDS_ADDR(s_w);

    // gather byte offsets of group-list entry POS (all out of range when MORE is false): every chunk of a group reads
    // the same neighbour row, +64 B per chunk
#define PBN_GROUP_ROWS(POS, GI, MORE, VOFF, CG)                                                                       \
    {                                                                                                                 \
        int ko_, cv_;                                                                                                 \
        if (wide) {                                                                                                   \
            const int pk_ = __builtin_amdgcn_readfirstlane(s_gko[POS]);                                               \
            ko_ = pk_ & 0xffff; cv_ = (pk_ >> 16) * (CG) * 4 + g;                                                     \
        } else { const int v_ = (GI) * 4 + g; ko_ = v_ >> vshift; cv_ = v_ & (vpo - 1); }                             \
        nx2_ = a.in2 != nullptr && (MORE) && ko_ == K;                                                                \
        /* chunks of the group that exist in the second source's rows (its steps are zero-padded to whole groups: a   */ \
        /* padded chunk must not gather what lies behind the row -- 0 x NaN of an uninitialised neighbour is NaN)      */ \
        nxv_ = nx2_ ? (a.vpo2 >> 2) - (__builtin_amdgcn_readfirstlane(s_gko[POS]) >> 16) * (CG) : (CG);               \
        const unsigned ldb_ = nx2_ ? ld2_bytes : ld_bytes;                                                            \
        _Pragma("unroll") for (int f = 0; f < NF; ++f) {                                                              \
            const int r_ = wave * RW + f * 16 + rl;                                                                   \
            int src_ = ((MORE) && ko_ < K) ? s_nbr[r_ * KS + ko_] : -1;                                               \
            if (nx2_) src_ = row0 + r_ < n ? row0 + r_ : -1;                                                          \
            VOFF[f] = (src_ >= 0 && !(a.dbg & 4)) ? (unsigned)src_ * ldb_ + (unsigned)cv_ * 16u : OOB;                \
        }                                                                                                             \
    }
    // refill chunk C's operand registers in place ("+v": the load lands in the register the MFMAs just read)
#define PBN_LOAD_X(VOFF, C, RS)                                                                                       \
    {                                                                                                                 \
        _Pragma("unroll") for (int f = 0; f < NF; ++f)                                                                \
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen"                                                   \
                         : "+v"(x[C][f]) : "v"((C) < nxv_ ? VOFF[f] : OOB), "s"(RS), "s"((C) * 64));                  \
    }
    // wait until at most N vector-memory loads are outstanding; names chunk C's registers so that their readers
    // are ordered behind the wait
#define PBN_WAIT_X(C, N)                                                                                              \
    {                                                                                                                 \
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x[C][0]) : "n"(N));                                                 \
        _Pragma("unroll") for (int f = 1; f < NF; ++f) asm volatile("" : "+v"(x[C][f]));                              \
    }
    // weight tile of group GI -> ring slot at LDS byte address SLOT: piece p = wave + 4 i is step p / NT, channel tile
    // p % NT; every wave issues the same number of pieces (the tail is clamped onto the last piece: a benign duplicate
    // copy).  MORE false (past the last group): the same instructions with an out-of-range offset: no memory traffic.
#define PBN_DMA_W(GI, MORE, SLOT, CG)                                                                                 \
    {                                                                                                                 \
        const unsigned gbase_ = ((unsigned)(GI) * (CG) * a.ntiles_total + tile0) * 1024u;                             \
        const unsigned wv_ = ((MORE) && !(a.dbg & 8)) ? w_lane : OOB;                                                 \
        _Pragma("unroll") for (int i = 0; i < ((CG) * NT + NW - 1) / NW; ++i) {                                       \
            const int p_ = min(wave + NW * i, (CG) * NT - 1);                                                         \
            const int c_ = p_ / NT, t_ = p_ - c_ * NT;                                                                \
            PBN_TIMING_SKIP_DMA;                                                                                      \
            unsigned keep_;                                                                                           \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"                                        \
                         "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"                               \
                         : "=&s"(keep_)                                                                               \
                         : "s"(__builtin_amdgcn_readfirstlane((SLOT) + (unsigned)p_ * 1024u)), "v"(wv_), "s"(rs_w),   \
                           "s"(__builtin_amdgcn_readfirstlane(gbase_ + c_ * w_step_bytes + t_ * 1024u))               \
                         : "memory");                                                                                 \
        }                                                                                                             \
    }
#define PBN_LOAD_WF(DST, CUR, C)                                                                                      \
    { _Pragma("unroll") for (int t = 0; t < NT; ++t) DST[t] = CUR[((C) * NT + t) * 64 + lane]; }
    // one group.  Issue order per group body j: DMA(j + DEPTH), then x[0..CG-1](j + 1), hence at the top of group g:
    //   barrier  : loads issued behind DMA(g) = CG*NF + (DEPTH-1) * (PW + CG*NF): wait for exactly that count = every
    //              wave's pieces of this group's weight tile have landed (hence the barrier); the slot of group g-1 is free
    //   chunk c  : loads behind x[c](g) = x[c+1..](g), DMA(g + DEPTH), x[..c-1](g + 1) = (CG-1)*NF + PW
#define PBN_GROUP(POS, CG)                                                                                            \
    {                                                                                                                 \
        constexpr int PW_ = ((CG) * NT + NW - 1) / NW;                                                                \
        PBN_STAMP(POS, 0);                                                                                            \
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"((CG) * NF + (DEPTH - 1) * (PW_ + (CG) * NF)) : "memory");          \
        PBN_STAMP(POS, 1);                                                                                            \
        asm volatile("s_barrier" : : : "memory");                                                                     \
        PBN_STAMP(POS, 2);                                                                                            \
        const bool active_ = (fcur & my_bits) && !(a.dbg & 2);                                                        \
        const bool more_ = (POS) + 1 < ng;                                                                            \
        const u32x4* cur_ = s_w + slot * ((CG) * NT * 64);                                                            \
        const unsigned free_ = slot == 0 ? RING - 1 : slot - 1;   /* slot of group POS-1 = slot of group POS+DEPTH */  \
        slot = slot == RING - 1 ? 0 : slot + 1;                                                                       \
        unsigned vnext_[NF];                                                                                          \
        {                                                                                                             \
            const int gdma_ = __builtin_amdgcn_readfirstlane(s_grp[(POS) + DEPTH]);                                   \
            PBN_DMA_W(gdma_, (POS) + DEPTH < ng, lds_w + free_ * slot_bytes, CG);                                     \
            const int gnext_ = __builtin_amdgcn_readfirstlane(s_grp[(POS) + 1]);                                      \
            const unsigned fnext_ = (unsigned)__builtin_amdgcn_readfirstlane(s_masks[(POS) + 1]);                     \
            fcur = more_ ? fnext_ : 0u;                                                                               \
            PBN_GROUP_ROWS((POS) + 1, gnext_, more_, vnext_, CG);                                                     \
        }                                                                                                             \
        const i32x4 rsn_ = pick_rs(nx2_);              /* wave-uniform: the resource of the group being fetched */   \
        PBN_STAMP(POS, 3);                                                                                            \
        /* the asm statements that define x[][] stay on the straight-line path: inside a branch the compiler would */ \
        /* merge them through register copies, i.e. read registers whose loads are still in flight                 */ \
        u32x4 wf_[2][NT];                                                                                             \
        if (active_) PBN_LOAD_WF(wf_[0], cur_, 0);                                                                    \
        _Pragma("unroll") for (int c = 0; c < (CG); ++c) {                                                            \
            PBN_WAIT_X(c, ((CG) - 1) * NF + PW_);                                                                     \
            if (c == 0) { PBN_STAMP(POS, 4); }                                                                        \
            if (c == (CG) - 1) { PBN_STAMP(POS, 5); }                                                                 \
            if (active_) {                                                                                            \
                if (c + 1 < (CG)) PBN_LOAD_WF(wf_[(c + 1) & 1], cur_, c + 1);                                         \
                _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                      \
                    _Pragma("unroll") for (int f = 0; f < NF; ++f) mfma_step<T>(wf_[c & 1][t], x[c][f], acc[f][t]);   \
                }                                                                                                     \
            }                                                                                                         \
            PBN_LOAD_X(vnext_, c, rsn_);                                                                              \
        }                                                                                                             \
        PBN_STAMP(POS, 6);                                                                                            \
    }
    // prologue: the same issue pattern as DEPTH loop bodies (the gathers behind all but the last DMA are dummies with
    // out-of-range offsets) so that the loop's wait counts hold from the first group on
#define PBN_MAINLOOP(CG)                                                                                              \
    {                                                                                                                 \
        u32x4 x[CG][NF];                                                                                              \
        _Pragma("unroll") for (int c = 0; c < (CG); ++c)                                                              \
            _Pragma("unroll") for (int f = 0; f < NF; ++f) x[c][f] = u32x4{0u, 0u, 0u, 0u};                           \
        unsigned fcur = (unsigned)__builtin_amdgcn_readfirstlane(s_masks[g_lo]);                                      \
        unsigned slot = 0;                                                                                            \
        _Pragma("unroll") for (int j = 0; j < DEPTH; ++j) {                                                           \
            const int gj_ = __builtin_amdgcn_readfirstlane(s_grp[g_lo + j]);                                          \
            unsigned v0_[NF];                                                                                         \
            PBN_DMA_W(gj_, g_lo + j < ng, lds_w + (unsigned)j * slot_bytes, CG);                                      \
            const int g0_ = __builtin_amdgcn_readfirstlane(s_grp[g_lo]);                                              \
            PBN_GROUP_ROWS(g_lo, g0_, j == DEPTH - 1, v0_, CG);                                                       \
            const i32x4 rs0_ = pick_rs(nx2_);                                                                         \
            _Pragma("unroll") for (int c = 0; c < (CG); ++c) { PBN_LOAD_X(v0_, c, rs0_); }                            \
        }                                                                                                             \
        for (int pos = g_lo; pos < ng; ++pos) PBN_GROUP(pos, CG);                                                     \
        /* drain: the loads issued for the (non-existent) group past the end still target these registers */         \
        _Pragma("unroll") for (int c = 0; c < (CG); ++c) { PBN_WAIT_X(c, 0); }                                        \
    }

    if (ng > g_lo && !(a.dbg & 16)) {
        switch (cg) {
            case 4: PBN_MAINLOOP(4); break;
            case 3: PBN_MAINLOOP(3); break;
            case 2: PBN_MAINLOOP(2); break;
            default: PBN_MAINLOOP(1); break;
        }
    }
#ifdef PBN_CONV_TIMING
    __syncthreads();
    if (timed_) {
        for (int e = tid; e < 4 * 64 * 8; e += CONV_TPB) g_conv_timing[e] = s_time[e];   // the first four waves
        if (tid == 0) { g_conv_timing[4 * 64 * 8] = (unsigned)(ng - g_lo); g_conv_timing[4 * 64 * 8 + 1] = (unsigned)cg; }
    }
#endif
#undef PBN_MAINLOOP
#undef PBN_GROUP
#undef PBN_LOAD_WF
#undef PBN_DMA_W
#undef PBN_WAIT_X
#undef PBN_GROUP_ROWS
#undef PBN_LOAD_X

    if (a.ksplit > 1) {  // raw fp32 partial sums, tile-position rows; the epilogue runs in k_spconv_reduce
        const int ldp = a.ntiles_total * 16;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int p = row0 + wave * RW + f * 16 + rl;
            if (p >= n) continue;
            float* dst = a.partial + ((size_t)blockIdx.z * a.n_out_pad + p) * ldp;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c0 = (tile0 + t) * 16 + g * 4;
                *reinterpret_cast<float4*>(dst + c0) = make_float4(acc[f][t][0], acc[f][t][1], acc[f][t][2], acc[f][t][3]);
            }
        }
        return;
    }

    // epilogue: lane holds channels c0..c0+3 of output row (wave*RW + f*16 + rl)
    if (a.dbg & 32) return;
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

// second pass of a split-K launch: fixed-order sum of the partial slabs + the fused epilogue
template <typename T>
__global__ __launch_bounds__(256) void k_spconv_reduce(const ConvArgs a) {
    const int ldp = a.ntiles_total * 16;
    const int vec_per_row = ldp >> 2;
    const int n = a.n_out_dev ? min(*a.n_out_dev, a.n_out) : a.n_out;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)n * vec_per_row) return;
    const int p = (int)(e / vec_per_row), c0 = (int)(e - (long long)p * vec_per_row) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < a.ksplit; ++z) {
        const float4 t = *reinterpret_cast<const float4*>(a.partial + ((size_t)z * a.n_out_pad + p) * ldp + c0);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    }
    const int orow = a.row_perm ? a.row_perm[p] : p;
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

template <typename T, int NF, int NT, int RING, int NW = 4>
int launch_ring(ConvArgs a, int ngroups, float* workspace, size_t workspace_bytes, hipStream_t stream) {
    constexpr int TM = 16 * NW * NF;
    constexpr int CONV_TPB = NW * 64;
    const int KS = a.K | 1;
    // steps per barrier group: the largest divisor <= 4 of the steps per offset (1 when offsets are narrower than a step)
    a.cg = 1;
    if ((a.vpo & 3) == 0) {
        const int spo = a.vpo >> 2;
        for (int c = CGMAX; c >= 1; --c)
            if (spo % c == 0) { a.cg = c; break; }
    }
    const int n_groups = a.n_steps / a.cg;
    const size_t lds = (size_t)RING * a.cg * NT * 1024 +
                       sizeof(int) * ((size_t)TM * KS + ((a.K + 4) & ~3) + 3 * ((size_t)n_groups + 4) + 2 * NT * 16 + 4);
    if (lds > 160 * 1024) return PBN_ERR_UNSUPPORTED;
    auto kern = k_spconv<T, NF, NT, RING, NW>;
    if (lds > 64 * 1024)
        PBN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tiles = cdiv(a.n_out, TM);
    // split the reduction over more workgroups when the launch would leave most of the 256 CUs idle
    // (stride-8/16 levels: a few hundred to a few thousand rows but 27 x 256..384 deep reductions)
    a.ksplit = 1;
    a.partial = nullptr;
    a.n_out_pad = tiles * TM;
    const long long wgs = (long long)cdiv(a.n_sel, TM) * ngroups;      // (the split decision follows the rows expected, the grid the capacity)
    // Split-K by a two-term cost model (microseconds, fitted on the bench scene's stride-4..16 levels):
    //   a workgroup's chain of n_groups/ks groups at ~t_group each  +  ks fp32 partial slabs written and read back.
    // The minimum is at ks* = sqrt(n_groups * t_group / slab_cost); more workgroups than ~2 per CU only queue up.
    static const float t_group = getenv("PBN_CONV_TGROUP") ? (float)atof(getenv("PBN_CONV_TGROUP")) : 1.5f;
    static const int max_wgs = getenv("PBN_CONV_SPLIT") ? atoi(getenv("PBN_CONV_SPLIT")) : 512;
    if (workspace && wgs < 256 && n_groups >= 8) {
        const float slab_us = 2.0f * (float)a.n_out_pad * (float)(a.ntiles_total * 16) * 4.0f / 3.0e6f;   // ~3 TB/s
        long long want = (long long)(sqrtf((float)n_groups * t_group / (slab_us > 0.05f ? slab_us : 0.05f)) + 0.5f);
        const long long by_steps = n_groups / 2;
        const long long by_ws = (long long)(workspace_bytes / ((size_t)a.n_out_pad * a.ntiles_total * 16 * sizeof(float)));
        const long long by_wgs = (RING == 3 ? 256 : max_wgs) / wgs;
        if (want > by_wgs) want = by_wgs;
        if (want > by_steps) want = by_steps;
        if (want > by_ws) want = by_ws;
        if (want > 32) want = 32;
        if (want > 1) { a.ksplit = (int)want; a.partial = workspace; }
    }
    hipLaunchKernelGGL(kern, dim3(tiles, ngroups, a.ksplit), dim3(CONV_TPB), lds, stream, a);
    if (a.ksplit > 1) {
        const long long total = (long long)a.n_out * (a.ntiles_total * 4);
        hipLaunchKernelGGL(k_spconv_reduce<T>, dim3(cdiv(total, 256)), dim3(256), 0, stream, a);
    }
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

template <typename T, int NF, int NT>
int launch_one(const ConvArgs& a, int ngroups, float* ws, size_t wsb, hipStream_t stream) {
    // RING = 3 (two weight tiles in flight): experiment switch PBN_CONV_RING=3 for the small levels (< 8k rows)
    static const int ring_env = getenv("PBN_CONV_RING") ? atoi(getenv("PBN_CONV_RING")) : 2;
    // 8-wave workgroups (256 rows at NF = 2 share one weight ring: half the weight DMA per row).  Measured round 4
    // (round 4, PBN_CONV_WAVES8=1): SLOWER everywhere -- L0 96->96 89 -> 106 us, L1 96->96 38 -> 56, L1 32->32 17 ->
    // 26: the ring's barrier then spans 8 waves and the weight stream was not what bound the kernel.  Off by default.
    // PBN_CONV_WAVES8: 0 never (default), 1 always, else the row count from which they are used
    static const long long w8_env = getenv("PBN_CONV_WAVES8") ? atoll(getenv("PBN_CONV_WAVES8")) : 0;
    if constexpr (NF == 2) {
        if (a.K <= 32 && (w8_env == 1 || (w8_env > 1 && a.n_out >= w8_env))) {
            const int rc = launch_ring<T, NF, NT, 2, 8>(a, ngroups, ws, wsb, stream);
            if (rc != PBN_ERR_UNSUPPORTED) return rc;
        }
    }
    if constexpr (NF == 1) {
        if (ring_env == 3 && a.n_out < 8192) {
            const int rc = launch_ring<T, NF, NT, 3>(a, ngroups, ws, wsb, stream);
            if (rc != PBN_ERR_UNSUPPORTED) return rc;
        }
    }
    return launch_ring<T, NF, NT, 2>(a, ngroups, ws, wsb, stream);
}

template <typename T, int NF>
int launch_nt(const ConvArgs& a, float* ws, size_t wsb, hipStream_t stream) {
    const int ntt = a.ntiles_total;
    if (ntt % 8 == 0) return launch_one<T, NF, 8>(a, ntt / 8, ws, wsb, stream);
    if (ntt % 6 == 0) return launch_one<T, NF, 6>(a, ntt / 6, ws, wsb, stream);
    if (ntt % 4 == 0) return launch_one<T, NF, 4>(a, ntt / 4, ws, wsb, stream);
    if (ntt % 2 == 0) return launch_one<T, NF, 2>(a, ntt / 2, ws, wsb, stream);
    return launch_one<T, NF, 1>(a, ntt, ws, wsb, stream);
}

template <typename T>
int launch_t(const ConvArgs& a, int rows_per_wave, float* ws, size_t wsb, hipStream_t stream) {
    if (rows_per_wave == 64) return launch_nt<T, 4>(a, ws, wsb, stream);
    if (rows_per_wave == 32) return launch_nt<T, 2>(a, ws, wsb, stream);
    return launch_nt<T, 1>(a, ws, wsb, stream);
}

// ---- row gather: out[i, :] = in[idx[i], :], zeros for idx < 0  (voxel -> point, PBNet.py:130-134) ----------------------------------
__global__ __launch_bounds__(256) void k_gather_rows(const uint4* __restrict__ in, const long long* __restrict__ idx,
                                                    int n, int vec_per_row, int ld_in_vec, int ld_out_vec,
                                                    uint4* __restrict__ out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)n * vec_per_row) return;
    const int i = (int)(e / vec_per_row), v = (int)(e % vec_per_row);
    const long long r = idx[i];
    out[(size_t)i * ld_out_vec + v] = r >= 0 ? in[(size_t)r * ld_in_vec + v] : make_uint4(0u, 0u, 0u, 0u);
}

}  // namespace
}  // namespace pbn

using namespace pbn;

static int spconv_forward_impl(const void* in_feat, int ld_in, int n_in, const int32_t* nbr, int n_offsets,
                               const int32_t* row_perm, const int32_t* n_out_dev, int n_out, const void* w_packed,
                               int vecs_per_offset, int n_steps, int cout_padded, const float* scale,
                               const float* shift, const void* residual, int ld_res, int relu, void* out_feat,
                               int ld_out, int dtype, int rows_per_wave, void* workspace, size_t workspace_bytes,
                               pbn_stream_t stream_, const void* in2_feat, int ld_in2, int n_in2, int vecs_second) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_out < 0 || n_in < 0 || n_offsets < 1 || vecs_per_offset < 1 || n_steps < 1 || cout_padded < 16 || (cout_padded & 15))
        return PBN_ERR_ARG;
    if (!(vecs_per_offset == 1 || vecs_per_offset == 2 || (vecs_per_offset & 3) == 0)) return PBN_ERR_ARG;
    const int n_main = (n_offsets * vecs_per_offset + 3) / 4;
    if (in2_feat) {
        // second source: wide rows on both sides, a real map, no processing order; its steps follow the map's (padded by the
        // caller to whole barrier groups of the first source: zero weights behind the real channels)
        if ((vecs_per_offset & 3) || vecs_second < 4 || (vecs_second & 3) || !nbr || row_perm || n_in2 < n_out) return PBN_ERR_ARG;
        if (n_steps < n_main + (vecs_second >> 2) || n_steps > n_main + (vecs_second >> 2) + 3) return PBN_ERR_ARG;
    } else if (n_steps != n_main) return PBN_ERR_ARG;
    if (!nbr && n_offsets != 1) return PBN_ERR_ARG;
    if (n_out == 0) return PBN_OK;
    if (!in_feat || !w_packed || !out_feat) return PBN_ERR_ARG;
    const int esz = dtype == PBN_F32 ? 4 : 2;
    if ((ld_in * esz) % 16 || (ld_out * esz) % 8 || (residual && (ld_res * esz) % 8)) return PBN_ERR_ARG;
    if (((uintptr_t)in_feat | (uintptr_t)w_packed) & 15) return PBN_ERR_ARG;
    if (in2_feat && ((ld_in2 * esz) % 16 || ((uintptr_t)in2_feat & 15))) return PBN_ERR_ARG;
    const unsigned long long in2_extent = in2_feat ? (unsigned long long)n_in2 * (unsigned long long)ld_in2 * (unsigned long long)esz : 0ull;
    if (in2_extent >= 0x80000000ull) return PBN_ERR_RANGE;
    // gathers address the slab with 32-bit byte offsets through a buffer resource (k_spconv): a slab or weight block of
    // 2 GiB or more cannot be addressed -- refuse it instead of silently gathering zeros past the limit
    const unsigned long long in_extent = (unsigned long long)n_in * (unsigned long long)ld_in * (unsigned long long)esz;
    const unsigned long long w_extent = (unsigned long long)n_steps * (unsigned long long)(cout_padded / 16) * 1024ull;
    if (in_extent >= 0x80000000ull || w_extent >= 0x80000000ull) return PBN_ERR_RANGE;
    if (!nbr && n_in < n_out) return PBN_ERR_ARG;   // identity map reads row o of the input
    ConvArgs a;
    a.in_bytes = (unsigned)in_extent; a.w_bytes = (unsigned)w_extent;
    a.in = in_feat; a.nbr = nbr; a.row_perm = row_perm; a.n_out_dev = n_out_dev; a.w = w_packed; a.scale = scale;
    a.shift = shift; a.residual = residual; a.out = out_feat; a.ld_in = ld_in; a.ld_res = ld_res; a.ld_out = ld_out;
    a.K = n_offsets; a.vpo = vecs_per_offset; a.n_steps = n_steps; a.ntiles_total = cout_padded / 16; a.n_out = n_out;
    a.relu = relu;
    a.n_sel = (g_rows_hint > 0 && n_out_dev) ? (g_rows_hint < n_out ? g_rows_hint : n_out) : n_out;
    a.ksplit = 1; a.partial = nullptr; a.n_out_pad = 0; a.cg = 1; a.wmajor = 0;
    a.in2 = in2_feat; a.ld_in2 = ld_in2; a.vpo2 = vecs_second; a.n_main = n_main; a.in2_bytes = (unsigned)in2_extent;
    // Next-op weight prefetch, measured round 4 on the bench scene: one scene alone the stride-8 / 16
    // ops gain 0.9 us each (138 convolution ops 3417 -> 3362 us with XCD ownership, 3354 with plain slices), with four scenes
    // in flight the job LOSES 3-8 % (320 / 315 -> 299 / 289 scenes/s with ownership, 311 plain): the touched weights compete
    // with the other scenes' working sets for the same L2s and every op reads 1.1 MB more.  Off by default.
    static const int pf_env = getenv("PBN_CONV_PREFETCH") ? atoi(getenv("PBN_CONV_PREFETCH")) : 0;   // 0 off, 1 with ownership, 2 plain slices
    a.pf_w = nullptr; a.pf_steps = a.pf_ntt = a.pf_nt = a.pf_groups = 0;
    if (pf_env && g_next_weights.w) {
        a.pf_w = g_next_weights.w; a.pf_steps = g_next_weights.steps; a.pf_ntt = g_next_weights.ntt;
        a.pf_nt = g_next_weights.nt; a.pf_groups = pf_env == 1 ? g_next_weights.groups : 0;
    }
    static const int dbg_env = getenv("PBN_CONV_DBG") ? atoi(getenv("PBN_CONV_DBG")) : 0;
    a.dbg = dbg_env;
    // coarse levels (and whatever PBN_CONV_FAMILY selects): the wave-autonomous family of spconv_wave.hip -- K split over
    // the waves of a workgroup instead of over workgroups: no fp32 partial slabs, no second launch
    if (rows_per_wave >= 10000) {             // explicit configuration of the big-tile families: 10000 + 1000 * form + ... (tests, tuning)
        const int c = rows_per_wave - 10000;
#ifdef PBN_EXPERIMENTS
        if ((c % 100000) / 1000 == 3) return launch_pc(a, dtype, (c / 100000) * 16, stream);   // form 3: pair-compacted (experiments/spconv_pc.hip)
#endif
        return launch_rs(a, dtype, c, stream);
    }
    if (rows_per_wave >= 100) return launch_wave(a, dtype, rows_per_wave, stream);   // explicit configuration (tests, tuning)
#ifdef PBN_EXPERIMENTS
    // pair-compacted fragments, fp32 output tile in LDS (experiments/spconv_pc.hip; PBN_CONV_PC=1 in the experiments library)
    if (rows_per_wave == 0 && pc_family_wanted(a, dtype)) {
        const int rc = launch_pc(a, dtype, 0, stream);
        if (rc != PBN_ERR_UNSUPPORTED) return rc;
    }
#endif
    // wide levels (round 5): one tile per CU, weights streamed once per CU (spconv_rs.hip)
    if (rows_per_wave == 0 && rs_family_wanted(a, dtype)) {
        const int rc = launch_rs(a, dtype, 0, stream);
        if (rc != PBN_ERR_UNSUPPORTED) return rc;
    }
    if (rows_per_wave == 0 && wave_family_wanted(a, dtype)) {
        static const int force_cfg = getenv("PBN_WAVE_CFG") ? atoi(getenv("PBN_WAVE_CFG")) : 0;
        const int rc = launch_wave(a, dtype, force_cfg, stream);
        if (rc != PBN_ERR_UNSUPPORTED) return rc;
    }
    float* ws = reinterpret_cast<float*>(workspace);
    if (((uintptr_t)workspace) & 15) ws = nullptr;
    if (rows_per_wave != 16 && rows_per_wave != 32 && rows_per_wave != 64) rows_per_wave = (a.n_sel > 64) ? 32 : 16;
    switch (dtype) {
        case PBN_F32: return launch_t<float>(a, rows_per_wave, ws, workspace_bytes, stream);
        case PBN_BF16: return launch_t<__hip_bfloat16>(a, rows_per_wave, ws, workspace_bytes, stream);
        case PBN_F16: return launch_t<__half>(a, rows_per_wave, ws, workspace_bytes, stream);
        default: return PBN_ERR_ARG;
    }
}

namespace pbn {
thread_local NextWeights g_next_weights = {nullptr, 0, 0, 0, 0};
thread_local int g_rows_hint = 0;
}

extern "C" int pbn_spconv_forward(const void* in_feat, int ld_in, int n_in, const int32_t* nbr, int n_offsets,
                                  const int32_t* row_perm, const int32_t* n_out_dev, int n_out, const void* w_packed,
                                  int vecs_per_offset, int n_steps, int cout_padded, const float* scale,
                                  const float* shift, const void* residual, int ld_res, int relu, void* out_feat,
                                  int ld_out, int dtype, int rows_per_wave, void* workspace, size_t workspace_bytes,
                                  pbn_stream_t stream_) {
    return spconv_forward_impl(in_feat, ld_in, n_in, nbr, n_offsets, row_perm, n_out_dev, n_out, w_packed, vecs_per_offset,
                               n_steps, cout_padded, scale, shift, residual, ld_res, relu, out_feat, ld_out, dtype,
                               rows_per_wave, workspace, workspace_bytes, stream_, nullptr, 0, 0, 0);
}

// Which kernel family the automatic choice of pbn_spconv_forward gives a launch of this shape (0 workgroup-tile k_spconv,
// 1 wave-autonomous k_spconv_wave, 2 row-stationary k_spconv_rs): the same predicates, no launch.  For reports (bench.py's family split).
extern "C" int pbn_spconv_family(int n_out, int n_offsets, int vecs_per_offset, int n_steps, int cout_padded, int dtype, int has_map) {
    if (n_out < 0 || n_offsets < 1 || vecs_per_offset < 1 || n_steps < 1 || cout_padded < 16 || (cout_padded & 15)) return PBN_ERR_ARG;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.nbr = has_map ? reinterpret_cast<const int*>(&a) : nullptr;       // (only tested against null)
    a.K = n_offsets; a.vpo = vecs_per_offset; a.n_steps = n_steps; a.ntiles_total = cout_padded / 16; a.n_out = a.n_sel = n_out;
    if (has_map && rs_family_wanted(a, dtype)) return 2;
    if (wave_family_wanted(a, dtype)) return 1;
    return 0;
}

extern "C" int pbn_spconv_forward_dual(const void* in_feat, int ld_in, int n_in, const int32_t* nbr, int n_offsets,
                                       const int32_t* n_out_dev, int n_out, const void* w_packed, int vecs_per_offset,
                                       int n_steps, int cout_padded, const float* scale, const float* shift,
                                       const void* residual, int ld_res, int relu, void* out_feat, int ld_out, int dtype,
                                       int rows_per_wave, void* workspace, size_t workspace_bytes, const void* in2_feat,
                                       int ld_in2, int n_in2, int vecs_second, pbn_stream_t stream_) {
    if (!in2_feat) return PBN_ERR_ARG;
    return spconv_forward_impl(in_feat, ld_in, n_in, nbr, n_offsets, nullptr, n_out_dev, n_out, w_packed, vecs_per_offset,
                               n_steps, cout_padded, scale, shift, residual, ld_res, relu, out_feat, ld_out, dtype,
                               rows_per_wave, workspace, workspace_bytes, stream_, in2_feat, ld_in2, n_in2, vecs_second);
}

#ifdef PBN_CONV_TIMING
// debug build only: the stamps of the last k_spconv launch -> host (4 waves x 64 groups x 8 stamps, then groups run, cg)
extern "C" int pbn_conv_timing_read(unsigned* host) {
    PBN_HIP_CHECK(hipDeviceSynchronize());
    PBN_HIP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(pbn::g_conv_timing), sizeof(unsigned) * (4 * 64 * 8 + 8)));
    return PBN_OK;
}
#endif

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
