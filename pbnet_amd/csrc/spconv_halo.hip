// spconv_halo.hip -- LDS-staged ("halo") implicit-GEMM sparse convolution for gfx950 (wave64, MFMA 16x16), round 4.
//
// Same arithmetic and operand layouts as spconv.hip (MinkowskiConvolution / ConvolutionTranspose forward,
// /root/reference/network/Mink.py:221-288,293-350):   out[o,:] = epilogue( sum_k in[nbr[o,k],:] @ W[k] ).
//
// What is different: the rows a tile gathers are staged ONCE.  On a Z-ordered lineage a tile of 128 consecutive output rows
// names only 1.4-2.4x as many DISTINCT input rows as it has rows (measured on the bench scene: 1.58x at tensor stride 1, 1.88x
// at stride 2) while it gathers 7.6-15 rows per output row, so the output-stationary kernels of spconv.hip / spconv_wave.hip
// pull every input row 4-7 times through the CU's vector-memory path, in MFMA operand order (the most expensive pattern that
// path has: 62 cycles per KiB against 30 for a contiguous KiB, scripts/micro/gather_layout.hip).  Here:
//   * per kernel map and tile, ONCE per lineage (k_halo_build): the sorted list of distinct input rows of the tile (its
//     "halo"), the map re-expressed as 16-bit slots into that list, and per offset the mask of 16-row fragments that have a
//     neighbour there.  Every layer of the level, in all three networks' passes over it, shares the tables;
//   * the convolution stages a 64-byte piece (one reduction step) of every halo row into LDS with the LDS-DMA path -- four
//     lanes fetch one contiguous 64-byte piece, ~H/16 instructions per piece instead of 8 x 21 gathers -- and all MFMA row
//     operands are ds_read_b128 from there (slot pitch 64 B, 16-byte chunks swizzled by slot bits so that runs of consecutive
//     slots, the common case in Z-order, read conflict-free);
//   * weights travel through an LDS-DMA ring shared by the 4 waves exactly as in spconv.hip, one step per ring slot;
//   * the only vector-memory instructions of the main loop are the ring's DMA pieces: one hand-counted s_waitcnt per step.
// A tile whose halo exceeds the LDS row buffer runs the same loops once per segment of the list (rows outside the segment read
// the zero slot); a tile whose halo exceeded the build's list (PBN_HALO_MAX distinct rows: not reachable with 27 offsets) is
// marked and takes a plain gather loop.  Both are exercised by the tests through small capacities.
#include <cstdlib>
#include <cstring>
#include "spconv_common.h"

namespace pbn {
namespace {

constexpr int HB_TPB = 256;
constexpr int HALO_HT = 8192;         // hash slots of the build (LDS)
constexpr int HALO_MAX = 4096;        // distinct rows a tile's list can hold
constexpr int HALO_JOBS = 16;
constexpr unsigned OOB = 0x80000000u;

struct HaloJobs {
    const int* nbr[HALO_JOBS];
    const int* n_dev[HALO_JOBS];
    int* cnt[HALO_JOBS];
    int* rows[HALO_JOBS];
    unsigned short* loc[HALO_JOBS];
    unsigned short* fmask[HALO_JOBS];
    int n_max[HALO_JOBS], K[HALO_JOBS], tm[HALO_JOBS], pitch[HALO_JOBS], cap[HALO_JOBS];
    int tile_begin[HALO_JOBS + 1];
    int n_jobs;
};

// One workgroup per (map, tile): distinct neighbour rows through an LDS hash set, bitonic sort, slots by binary search.
__global__ __launch_bounds__(HB_TPB) void k_halo_build(const HaloJobs J) {
    __shared__ int s_tab[HALO_HT];
    __shared__ int s_list[HALO_MAX];
    __shared__ int s_fm[128];
    __shared__ int s_cnt[4];
    const int tid = threadIdx.x;
    int job = 0;
    while (job + 1 < J.n_jobs && (int)blockIdx.x >= J.tile_begin[job + 1]) ++job;
    const int tile = (int)blockIdx.x - J.tile_begin[job];
    const int K = J.K[job], tm = J.tm[job], pitch = J.pitch[job];
    const int cap = J.cap[job] < pitch ? J.cap[job] : pitch;
    const int* nbr = J.nbr[job];
    const int n = J.n_dev[job] ? min(*J.n_dev[job], J.n_max[job]) : J.n_max[job];
    const int row0 = tile * tm;
    if (row0 >= n) {                                  // the convolution never visits the tile
        if (tid == 0) J.cnt[job][tile] = 0;
        return;
    }
    const int nrows = min(tm, n - row0);
    for (int e = tid; e < HALO_HT; e += HB_TPB) s_tab[e] = -1;
    if (tid < 128) s_fm[tid] = 0;
    if (tid < 4) s_cnt[tid] = 0;
    __syncthreads();
    const int total = nrows * K;
    const int* src = nbr + (size_t)row0 * K;
    const float inv_k = 1.0f / (float)K;
    for (int e = tid; e < total; e += HB_TPB) {
        const int v = src[e];
        if (v < 0) continue;
        const int r = (int)(((float)e + 0.5f) * inv_k), k = e - r * K;
        atomicOr(&s_fm[k], 1 << (r >> 4));
        unsigned hsh = ((unsigned)v * 2654435761u) >> 19;          // 13 bits
        while (true) {
            if (*(volatile int*)&s_cnt[0] > cap) break;            // overflow: the tile is marked below
            const int old = atomicCAS(&s_tab[hsh], -1, v);
            if (old == -1) { atomicAdd(&s_cnt[0], 1); break; }
            if (old == v) break;
            hsh = (hsh + 1) & (HALO_HT - 1);
        }
    }
    __syncthreads();
    const int H = s_cnt[0];
    unsigned short* fm = J.fmask[job] + (size_t)tile * K;
    for (int k = tid; k < K; k += HB_TPB) fm[k] = (unsigned short)s_fm[k];
    if (H > cap) {                                                 // plain gather loop in the convolution
        if (tid == 0) J.cnt[job][tile] = -1;
        return;
    }
    // compaction (any order), padded to a power of two, bitonic sort
    int P = 64;
    while (P < H) P <<= 1;
    for (int e = tid; e < P; e += HB_TPB) s_list[e] = 0x7fffffff;
    __syncthreads();
    for (int e = tid; e < HALO_HT; e += HB_TPB) {
        const int v = s_tab[e];
        if (v >= 0) s_list[atomicAdd(&s_cnt[1], 1)] = v;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += HB_TPB) {
                const int x = i ^ j;
                if (x > i) {
                    const int a = s_list[i], b = s_list[x];
                    const bool up = (i & k2) == 0;
                    if ((a > b) == up) { s_list[i] = b; s_list[x] = a; }
                }
            }
            __syncthreads();
        }
    int* rows = J.rows[job] + (size_t)tile * pitch;
    for (int e = tid; e < H; e += HB_TPB) rows[e] = s_list[e];
    if (tid == 0) J.cnt[job][tile] = H;
    unsigned short* loc = J.loc[job] + (size_t)tile * tm * K;
    for (int e = tid; e < tm * K; e += HB_TPB) {
        int v = -1;
        if (e < total) v = src[e];
        unsigned short s = 0xffffu;
        if (v >= 0) {
            int lo = 0, hi = H;                                    // lower bound; v is in the list
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_list[mid] <= v) lo = mid; else hi = mid;
            }
            s = (unsigned short)lo;
        }
        loc[e] = s;
    }
}

// Cycle stamps (debug build only: make timing, scripts/halo_timing.py): lane 0 of every wave of the first HT_BLOCKS workgroups
// records s_memtime at up to 48 points.
#ifdef PBN_CONV_TIMING
constexpr int HT_BLOCKS = 256, HT_STAMPS = 48;
__device__ unsigned long long g_halo_timing[HT_BLOCKS * 8 * HT_STAMPS + 8];
#define PBN_HSTAMP(I)                                                                                                 \
    if (lane == 0 && blockIdx.x < HT_BLOCKS && blockIdx.y == 0 && (I) < HT_STAMPS)                                    \
        g_halo_timing[((size_t)blockIdx.x * 8 + wave) * HT_STAMPS + (I)] = __builtin_readcyclecounter();
#else
#define PBN_HSTAMP(I)
#endif

struct HaloArgs {
    const int* cnt;
    const int* rows;
    const unsigned short* loc;
    const unsigned short* fmask;
    int tm, pitch;
    int hs;            // slots of the LDS row buffer (multiple of 16)
    int csp;           // 64-byte pieces (steps) of a row staged per pass: a divisor of the steps per offset
};

#define PBN_LDS_ADDR(p) ((unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)(p)))

// LDS-DMA of 64 x 16 bytes: lane l's 16 bytes land at lds_dst + 16 l
__device__ __forceinline__ void dma16(unsigned lds_dst, unsigned voff, const i32x4& rs, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\t"      /* 5 states behind a readfirstlane of the soffset */
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff)
                 : "memory");
}

// Workgroup = NW waves x 2 fragments = NW*32 output rows x NT*16 channels.  The reduction axis of a pass is the list of UNITS
// (populated offset k, step cc of the staged pieces), offset-major; one loop iteration consumes S units: S*NT weight pieces
// from one ring slot, S*2 row operands per wave, up to S*2*NT MFMAs per wave between two barriers.  Unit word: k | cc << 8 |
// fragment mask << 16; 0 = padding (no fragment: nothing is issued for it).
template <typename T, int NW, int NT, int S, int RING>
__global__ __launch_bounds__(NW * 64) void k_spconv_halo(const ConvArgs a, const HaloArgs h) {
    static_assert(Tr<T>::ELEMS * sizeof(T) == 16, "one gather vector is 16 bytes");
    static_assert(RING == 2 || RING == 3, "one or two iterations of weights in flight");
    constexpr int NF = 2, RW = NF * 16, TM = NW * RW, TPB = NW * 64, D = RING - 1;
    constexpr int NPC = S * NT, PW = (NPC + NW - 1) / NW;                          // weight pieces per iteration / per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = a.K, KS = K | 1, HS = h.hs, CSP = h.csp;
    const int spo = a.vpo >> 2;                                                   // 64-byte pieces (steps) per offset
    const int plane = (HS + 16) * 64;                                             // bytes of one staged piece of all slots
    u32x4* s_w = reinterpret_cast<u32x4*>(smem);                                  // RING x NPC KiB
    unsigned char* s_x = smem + (size_t)RING * NPC * 1024;                         // CSP planes; the last 16 slots of each stay zero
    unsigned short* s_loc = reinterpret_cast<unsigned short*>(s_x + (size_t)CSP * plane);   // TM x KS
    int* s_rows = reinterpret_cast<int*>(s_loc + ((TM * KS + 7) & ~7));           // HS
    int* s_units = s_rows + HS;                                                   // K * CSP + 4 S (padded)
    int* s_grp = s_units + ((K * CSP + 4 * S + 3) & ~3);                          // K + 4 populated offsets, [K] = count
    int* s_fmk = s_grp + ((K + 4 + 3) & ~3);                                      // their fragment masks
    float* s_ss = reinterpret_cast<float*>(s_fmk + ((K + 4 + 3) & ~3));           // scale | shift
    unsigned char* s_dump = reinterpret_cast<unsigned char*>(s_ss + 2 * NT * 16);  // 1 KiB: where the DMA pieces past the end land

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = a.n_out_dev ? min(*a.n_out_dev, a.n_out) : a.n_out;
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int row0 = tile * TM;
    if (row0 >= n) return;
    const int tile0 = blockIdx.y * NT;
    const int g = lane >> 4, rl = lane & 15;
    if (a.dbg & 128) return;
    PBN_HSTAMP(0);
    const int H = h.cnt[tile];

    const unsigned long long in_addr = (unsigned long long)a.in, w_addr = (unsigned long long)a.w;
    const i32x4 rs_in = {(int)(unsigned)in_addr, (int)(unsigned)(in_addr >> 32), (int)a.in_bytes, 0x00020000};
    const i32x4 rs_w = {(int)(unsigned)w_addr, (int)(unsigned)(w_addr >> 32), (int)a.w_bytes, 0x00020000};
    const unsigned ld_bytes = (unsigned)a.ld_in * (unsigned)sizeof(T);

    f32x4 acc[NF][NT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: populated offsets of the tile (wave 0), its slot table, its row list, epilogue constants ----
    if (wave == 0) {
        int base = 0;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + lane;
            const int fmv = k < K ? (int)h.fmask[(size_t)tile * K + k] : 0;
            const unsigned long long m = __ballot(fmv != 0);
            if (fmv != 0) {
                const int pos = base + __popcll(m & ((1ULL << lane) - 1ULL));
                s_grp[pos] = k;
                s_fmk[pos] = fmv;
            }
            base += __popcll(m);
        }
        if (lane == 0) s_grp[K] = base;
    }
    if (tid < NT * 32) {
        const int c = tile0 * 16 + (tid < NT * 16 ? tid : tid - NT * 16);
        const float* src = tid < NT * 16 ? a.scale : a.shift;
        s_ss[tid] = src ? src[c] : (tid < NT * 16 ? 1.0f : 0.0f);
    }
    if (H < 0) {
        // ---- marked tile: plain gathers through the map itself (slow, correct) ----
        __syncthreads();
        const int np = s_grp[K];
        for (int i = 0; i < np; ++i) {
            const int k = s_grp[i];
            for (int c = 0; c < spo; ++c) {
                u32x4 x[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int p = row0 + wave * RW + f * 16 + rl;
                    const int srow = p < n ? a.nbr[(size_t)p * K + k] : -1;
                    x[f] = u32x4{0u, 0u, 0u, 0u};
                    if (srow >= 0)
                        x[f] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(a.in) +
                                                               (size_t)srow * ld_bytes + (size_t)c * 64 + (size_t)g * 16);
                }
                const u32x4* wp = reinterpret_cast<const u32x4*>(a.w) + ((size_t)(k * spo + c) * a.ntiles_total + tile0) * 64 + lane;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const u32x4 wf = wp[(size_t)t * 64];
#pragma unroll
                    for (int f = 0; f < NF; ++f) mfma_step<T>(wf, x[f], acc[f][t]);
                }
            }
        }
    } else {
        {
            const unsigned short* gl = h.loc + (size_t)tile * TM * K;
            if (KS == K && ((TM * K) & 7) == 0) {
                const uint4* s4 = reinterpret_cast<const uint4*>(gl);
                uint4* d4 = reinterpret_cast<uint4*>(s_loc);
                for (int e = tid; e < (TM * K) >> 3; e += TPB) d4[e] = s4[e];
            } else {
                const float inv_k = 1.0f / (float)K;
                for (int e = tid; e < TM * K; e += TPB) {
                    const int r = (int)(((float)e + 0.5f) * inv_k), k = e - r * K;
                    s_loc[r * KS + k] = gl[e];
                }
            }
            const int h0 = H < HS ? H : HS;
            for (int s = tid; s < h0; s += TPB) s_rows[s] = h.rows[(size_t)tile * h.pitch + s];
            for (int e = tid; e < CSP * 64; e += TPB)                                  // the zero slots of every plane
                reinterpret_cast<u32x4*>(s_x + (size_t)(e >> 6) * plane + (size_t)HS * 64)[e & 63] = u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
        PBN_HSTAMP(1);
        if (a.dbg & 16) return;
        const int n_pop = (a.dbg & 32) ? 0 : __builtin_amdgcn_readfirstlane(s_grp[K]);
        const int n_u = n_pop * CSP;
        const int n_it = (n_u + S - 1) / S;
        {   // the unit list: offset-major, the CSP staged pieces of an offset adjacent; padded with empty units
            const float inv_c = 1.0f / (float)CSP;
            for (int u = tid; u < n_it * S + 3 * S; u += TPB) {
                int v = 0;
                if (u < n_u) {
                    const int i = (int)(((float)u + 0.5f) * inv_c), cc = u - i * CSP;
                    v = s_grp[i] | (cc << 8) | (s_fmk[i] << 16);
                }
                s_units[u] = v;
            }
        }
        __syncthreads();
        PBN_HSTAMP(2);
        const int nseg = (H + HS - 1) / HS;
        const int nchunk = spo / CSP;
        const int npass = n_it > 0 ? nseg * nchunk : 0;
        const unsigned lds_w = PBN_LDS_ADDR(s_w), lds_x = PBN_LDS_ADDR(s_x), lds_dump = PBN_LDS_ADDR(s_dump);
        const unsigned w_lane = (unsigned)lane * 16u;

        // S unit words of iteration x as scalars
        struct Units { int u[S]; };
        auto units_at = [&](int x) -> Units {
            const int v = s_units[x * S + (lane < S ? lane : 0)];
            Units r;
#pragma unroll
            for (int s = 0; s < S; ++s) r.u[s] = __builtin_amdgcn_readlane(v, s);
            return r;
        };
        const Units u_first = units_at(0), u_second = units_at(n_it > 1 ? 1 : 0);

        // weights of one iteration -> ring slot: piece pc = unit pc / NT, channel tile pc % NT; empty units and pieces past the
        // end use an out-of-range offset (no traffic)
        int wp_left = npass, wp_j = 0, wp_c0 = 0;              // the next iteration to fetch: passes left, index, first piece of its pass
        unsigned wp_slot = 0;
        auto issue_w = [&](const Units& un) {
#pragma unroll
            for (int j = 0; j < PW; ++j) {
                if (a.dbg & 256) break;
                const int pc = wave + NW * j;
                const int sidx = pc / NT, t = pc - sidx * NT;
                int uw = un.u[S - 1];
#pragma unroll
                for (int q = S - 2; q >= 0; --q) uw = sidx == q ? un.u[q] : uw;
                const bool live = wp_left > 0 && pc < NPC && (uw >> 16) != 0 && !(a.dbg & 1);
                const unsigned step = (unsigned)((uw & 0xff) * spo + wp_c0 + ((uw >> 8) & 0xff));
                const unsigned gsrc = (step * (unsigned)a.ntiles_total + (unsigned)(tile0 + t)) * 1024u;
                // (an out-of-range piece still WRITES its 1 KiB of zeros: pieces past the end go to the dump area)
                dma16(__builtin_amdgcn_readfirstlane(pc < NPC ? lds_w + (wp_slot * NPC + (unsigned)pc) * 1024u : lds_dump),
                      live ? w_lane : OOB, rs_w, __builtin_amdgcn_readfirstlane(live ? gsrc : 0u));
            }
            wp_slot = wp_slot == RING - 1 ? 0 : wp_slot + 1;
            if (++wp_j >= n_it) {
                wp_j = 0;
                --wp_left;
                wp_c0 += CSP;
                if (wp_c0 >= spo) wp_c0 = 0;
            }
        };
        // LDS byte address of the row operand of fragment f for a unit: chunk g of the slot's 64-byte piece in plane cc, chunks
        // swizzled by slot bit 2 (conflict-free for runs of consecutive slots); none / another segment -> the zero slot
        const unsigned short* my_loc = s_loc + (wave * RW + rl) * KS;
        auto slot_addr = [&](int uw, int seg0, int f) -> int {
            int slot = (int)my_loc[f * 16 * KS + (uw & 0xff)] - seg0;
            if ((unsigned)slot >= (unsigned)HS) slot = HS;
            return ((uw >> 8) & 0xff) * plane + slot * 64 + ((g ^ ((slot >> 1) & 2)) << 4);
        };

        if (npass > 0) {
            issue_w(u_first);
            if (D == 2) issue_w(wp_j == 0 ? u_first : u_second);
        }
        unsigned slot_r = 0;
        int seg = 0, c0 = 0;
        PBN_HSTAMP(3);
        for (int p = 0; p < npass; ++p) {
            const int seg0 = seg * HS;
            const int hseg = min(H - seg0, HS);
            if (p > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with the previous pieces
            if (c0 == 0 && seg > 0) {
                for (int s = tid; s < hseg; s += TPB) s_rows[s] = h.rows[(size_t)tile * h.pitch + seg0 + s];
                __syncthreads();
            }
            if (!(a.dbg & 8)) {   // stage pieces c0 .. c0+CSP-1 of the segment's rows: block = 16 slots x 64 B, lane 4 s + j' fetches chunk j' ^ swizzle(slot)
                const int nblk = (hseg + 15) >> 4;
                for (int e = wave; e < nblk * CSP; e += NW) {
                    const int cc = e / nblk, b = e - cc * nblk;
                    const int slot = b * 16 + (lane >> 2);
                    const int row = slot < hseg ? s_rows[slot] : -1;
                    const unsigned voff = row >= 0 ? (unsigned)row * ld_bytes + (unsigned)(c0 + cc) * 64u + (unsigned)((((lane & 3) ^ ((slot >> 1) & 2))) << 4) : OOB;
                    dma16(__builtin_amdgcn_readfirstlane(lds_x + (unsigned)cc * (unsigned)plane + (unsigned)b * 1024u), voff, rs_in, 0u);
                }
            }
            if (p == 0) { PBN_HSTAMP(4); }
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (p == 0) { PBN_HSTAMP(5); }
            // software pipeline over the iterations of the pass: the row operands of iteration j+1 and the slots of iteration
            // j+2 are fetched under the MFMAs of iteration j
            Units uc = u_first, un = u_second, u2 = units_at(n_it > 2 ? 2 : n_it - 1);
            u32x4 bc[S][NF], bn[S][NF];
            int an[S][NF];
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    bc[s][f] = *reinterpret_cast<const u32x4*>(s_x + slot_addr(uc.u[s], seg0, f));
                    an[s][f] = slot_addr(un.u[s], seg0, f);
                }
            int vnext = s_units[(n_it > 3 ? 3 : n_it - 1) * S + (lane < S ? lane : 0)];       // units of iteration 3
            if (p == 0) { PBN_HSTAMP(6); }
            for (int j = 0; j < n_it; ++j) {
                if (p == 0) { PBN_HSTAMP(8 + 4 * j); }
                // the weights of this iteration have landed (loads complete in issue order: the D-1 iterations behind it may
                // still be in flight), and every wave has left the previous iteration: its ring slot is free
                if (!(a.dbg & 4)) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((D - 1) * PW) : "memory");
                if (p == 0) { PBN_HSTAMP(9 + 4 * j); }
                {   // fetch the weights D iterations ahead (it may belong to the next pass: the list is the same)
                    const int jd = j + D;
                    if (jd < n_it) issue_w(D == 1 ? un : u2);
                    else issue_w(jd - n_it == 0 || n_it == 1 ? u_first : u_second);
                }
                if (p == 0) { PBN_HSTAMP(10 + 4 * j); }
                const u32x4* cur = s_w + slot_r * (NPC * 64) + lane;
                slot_r = slot_r == RING - 1 ? 0 : slot_r + 1;
                // row operands of the next iteration, slots of the one after
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int f = 0; f < NF; ++f) bn[s][f] = *reinterpret_cast<const u32x4*>(s_x + ((a.dbg & 512) ? 0 : an[s][f]));
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int f = 0; f < NF; ++f) an[s][f] = slot_addr(u2.u[s], seg0, f);
                Units u3;
#pragma unroll
                for (int s = 0; s < S; ++s) u3.u[s] = __builtin_amdgcn_readlane(vnext, s);
                vnext = s_units[(j + 4 < n_it ? j + 4 : n_it - 1) * S + (lane < S ? lane : 0)];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const unsigned fm = ((unsigned)uc.u[s] >> 16) >> (wave * NF);
                    if ((fm & ((1u << NF) - 1u)) && !(a.dbg & 2)) {
                        u32x4 wf[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) wf[t] = cur[(s * NT + t) * 64];
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            if ((fm >> f) & 1u) {
#pragma unroll
                                for (int t = 0; t < NT; ++t) mfma_step<T>(wf[t], bc[s][f], acc[f][t]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int f = 0; f < NF; ++f) bc[s][f] = bn[s][f];
                uc = un; un = u2; u2 = u3;
                if (p == 0) { PBN_HSTAMP(11 + 4 * j); }
            }
            if (p == 0) { PBN_HSTAMP(7); }
            c0 += CSP;
            if (c0 >= spo) { c0 = 0; ++seg; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing (dummy) weight fetches
        PBN_HSTAMP(44);
    }

    // ---- epilogue: lane holds channels c0..c0+3 of output row (wave*RW + f*16 + rl) ----
    if (a.dbg & 64) return;
    T* out = reinterpret_cast<T*>(a.out);
    const T* res = reinterpret_cast<const T*>(a.residual);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int p = row0 + wave * RW + f * 16 + rl;
        if (p >= n) continue;
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
            if (res) v += load4<T>(res + (size_t)p * a.ld_res + c0);
            if (a.relu) {
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            }
            store4<T>(out + (size_t)p * a.ld_out + c0, v);
        }
    }
    PBN_HSTAMP(45);
}

template <typename T, int NW, int NT, int S, int RING>
int launch_halo_cfg(const ConvArgs& a, const HaloArgs& h, hipStream_t stream) {
    constexpr int TM = 32 * NW;
    if (h.tm != TM || a.ntiles_total % NT) return PBN_ERR_UNSUPPORTED;
    const int KS = a.K | 1;
    const size_t kw = (size_t)((a.K + 4 + 3) & ~3);
    const size_t lds = (size_t)RING * S * NT * 1024 + (size_t)h.csp * (h.hs + 16) * 64 +
                       sizeof(unsigned short) * (size_t)((TM * KS + 7) & ~7) +
                       sizeof(int) * ((size_t)h.hs + (size_t)((a.K * h.csp + 4 * S + 3) & ~3) + 2 * kw) + sizeof(float) * 2 * NT * 16 + 1024;
    if (lds > 160 * 1024) return PBN_ERR_UNSUPPORTED;
    auto kern = k_spconv_halo<T, NW, NT, S, RING>;
    if (lds > 64 * 1024)
        PBN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(cdiv(a.n_out, TM), a.ntiles_total / NT), dim3(NW * 64), lds, stream, a, h);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

// cfg = 1000 * NW + 100 * S + 10 * RING + (NT code: channel tiles per workgroup, 0 = all when <= 8)
template <typename T>
int launch_halo_t(const ConvArgs& a, const HaloArgs& h, int nw, int s, int ring, hipStream_t stream) {
    const int ntt = a.ntiles_total;
    int nt = ntt <= 8 ? ntt : (ntt % 8 == 0 ? 8 : (ntt % 6 == 0 ? 6 : (ntt % 4 == 0 ? 4 : 2)));
#define PBN_HALO_CASE(NWV, NTV, SV, RV) \
    if (nw == NWV && nt == NTV && s == SV && ring == RV) return launch_halo_cfg<T, NWV, NTV, SV, RV>(a, h, stream);
#define PBN_HALO_NT(NWV, SV, RV) PBN_HALO_CASE(NWV, 2, SV, RV) PBN_HALO_CASE(NWV, 4, SV, RV) PBN_HALO_CASE(NWV, 6, SV, RV) PBN_HALO_CASE(NWV, 8, SV, RV)
    // built: 3 or 4 units per iteration, two ring slots (the experiment's other shapes -- S = 2, RING = 3 -- measured the same
    // or worse and are no longer instantiated: they tripled the build)
    PBN_HALO_NT(4, 3, 2) PBN_HALO_NT(4, 4, 2)
    PBN_HALO_NT(8, 3, 2) PBN_HALO_NT(8, 4, 2)
#undef PBN_HALO_NT
#undef PBN_HALO_CASE
    return PBN_ERR_UNSUPPORTED;
}

}  // namespace

bool halo_supported(const ConvArgs& a, int tm) {
    return (tm == 128 || tm == 256) && (a.vpo & 3) == 0 && a.K <= 128 && a.nbr != nullptr && a.row_perm == nullptr &&
           (a.ntiles_total & 1) == 0;
}

// cfg: 0 = automatic; else 100 * S + 10 * RING + CSP (units per iteration, ring slots, steps staged per pass)
int launch_halo(const ConvArgs& a, int dtype, const int* cnt, const int* rows, const unsigned short* loc,
                const unsigned short* fmask, int tm, int pitch, int hs, int cfg, hipStream_t stream) {
    if (!halo_supported(a, tm)) return PBN_ERR_UNSUPPORTED;
    HaloArgs h;
    h.cnt = cnt; h.rows = rows; h.loc = loc; h.fmask = fmask; h.tm = tm; h.pitch = pitch;
    static const int hs_env = getenv("PBN_HALO_SLOTS") ? atoi(getenv("PBN_HALO_SLOTS")) : 0;
    static const int cfg_env = getenv("PBN_HALO_CFG") ? atoi(getenv("PBN_HALO_CFG")) : 0;
    if (cfg <= 0) cfg = cfg_env;
    const int nw = tm / 32;
    if (hs <= 0) hs = hs_env > 0 ? hs_env : (a.K > 27 ? 3 * tm : (tm == 256 ? 480 : 352));
    h.hs = (hs + 15) & ~15;
    if (h.hs > HALO_MAX) h.hs = HALO_MAX;
    const int spo = a.vpo >> 2;
    int s = cfg > 0 ? cfg / 100 : 0, ring = cfg > 0 ? (cfg / 10) % 10 : 0, csp = cfg > 0 ? cfg % 10 : 0;
    ring = 2;
    const int ntt = a.ntiles_total, nt = ntt <= 8 ? ntt : (ntt % 8 == 0 ? 8 : (ntt % 6 == 0 ? 6 : (ntt % 4 == 0 ? 4 : 2)));
    const int KS = a.K | 1;
    auto lds_of = [&](int sv, int cv) -> size_t {
        return (size_t)ring * sv * nt * 1024 + (size_t)cv * (h.hs + 16) * 64 + 2 * (size_t)((tm * KS + 7) & ~7) +
               4 * ((size_t)h.hs + (size_t)a.K * cv + 4 * sv + 2 * (a.K + 8)) + 8 * nt * 16 + 64 + 1024;
    };
    if (csp <= 0 || spo % csp) {
        // as many steps per pass as fit (a divisor of the steps per offset, at most 4), 3 or 4 units per iteration
        csp = 1;
        for (int c = 4; c >= 1; --c)
            if (spo % c == 0 && lds_of(c == 3 ? 3 : 4, c) <= (size_t)(tm == 256 ? 156 : 120) * 1024) { csp = c; break; }
    }
    if (s < 3 || s > 4) s = (csp == 3 || (csp == 1 && spo == 3)) ? 3 : 4;
    h.csp = csp;
    switch (dtype) {
        case PBN_F32: return launch_halo_t<float>(a, h, nw, s, ring, stream);
        case PBN_BF16: return launch_halo_t<__hip_bfloat16>(a, h, nw, s, ring, stream);
        case PBN_F16: return launch_halo_t<__half>(a, h, nw, s, ring, stream);
        default: return PBN_ERR_ARG;
    }
}

}  // namespace pbn

#ifdef PBN_CONV_TIMING
// debug build only: the stamps of the last k_spconv_halo launch -> host
extern "C" int pbn_halo_timing_read(unsigned long long* host) {
    PBN_HIP_CHECK(hipDeviceSynchronize());
    PBN_HIP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(pbn::g_halo_timing), sizeof(unsigned long long) * (pbn::HT_BLOCKS * 8 * pbn::HT_STAMPS + 8)));
    return PBN_OK;
}
#endif

using namespace pbn;

extern "C" size_t pbn_halo_bytes(int n_out, int n_offsets, int tile_rows, pbn_halo_layout* L) {
    if (n_out < 0 || n_offsets < 1 || n_offsets > 128 || tile_rows < 16 || tile_rows > 256 || (tile_rows & 15) || !L) return 0;
    const size_t tiles = (size_t)cdiv(n_out > 0 ? n_out : 1, tile_rows);
    int pitch = tile_rows * n_offsets;
    if (pitch > HALO_MAX) pitch = HALO_MAX;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return (int64_t)o; };
    L->cnt = take(tiles * sizeof(int));
    L->rows = take(tiles * (size_t)pitch * sizeof(int));
    L->loc = take(tiles * (size_t)tile_rows * n_offsets * sizeof(unsigned short));
    L->fmask = take(tiles * (size_t)n_offsets * sizeof(unsigned short));
    L->tile_rows = tile_rows; L->n_offsets = n_offsets; L->pitch = pitch; L->tiles = (int32_t)tiles;
    return off;
}

extern "C" int pbn_halo_build(const pbn_halo_job* jobs, int n_jobs, pbn_stream_t stream) {
    if (!jobs || n_jobs < 1 || n_jobs > HALO_JOBS) return PBN_ERR_ARG;
    HaloJobs J;
    memset(&J, 0, sizeof(J));
    int tiles = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const pbn_halo_job& q = jobs[j];
        if (!q.nbr || !q.table || q.n_out < 0 || q.layout.tile_rows < 16 || q.layout.n_offsets < 1 || q.layout.n_offsets > 128)
            return PBN_ERR_ARG;
        pbn_halo_layout chk;
        if (pbn_halo_bytes(q.n_out, q.layout.n_offsets, q.layout.tile_rows, &chk) == 0 || chk.pitch != q.layout.pitch ||
            chk.tiles != q.layout.tiles)
            return PBN_ERR_ARG;
        char* A = (char*)q.table;
        J.nbr[j] = q.nbr; J.n_dev[j] = q.n_out_dev; J.n_max[j] = q.n_out; J.K[j] = q.layout.n_offsets; J.tm[j] = q.layout.tile_rows;
        J.pitch[j] = q.layout.pitch;
        J.cap[j] = q.max_rows < 1 ? HALO_MAX : (q.max_rows > HALO_MAX ? HALO_MAX : q.max_rows);
        J.cnt[j] = (int*)(A + q.layout.cnt); J.rows[j] = (int*)(A + q.layout.rows);
        J.loc[j] = (unsigned short*)(A + q.layout.loc); J.fmask[j] = (unsigned short*)(A + q.layout.fmask);
        J.tile_begin[j] = tiles;
        tiles += q.n_out > 0 ? q.layout.tiles : 0;
    }
    for (int j = n_jobs; j <= HALO_JOBS; ++j) J.tile_begin[j] = tiles;
    J.n_jobs = n_jobs;
    if (tiles == 0) return PBN_OK;
    hipLaunchKernelGGL(k_halo_build, dim3(tiles), dim3(HB_TPB), 0, (hipStream_t)stream, J);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_spconv_forward_halo(const void* in_feat, int ld_in, int n_in, const int32_t* nbr, int n_offsets,
                                       const int32_t* n_out_dev, int n_out, const void* w_packed, int vecs_per_offset,
                                       int n_steps, int cout_padded, const float* scale, const float* shift,
                                       const void* residual, int ld_res, int relu, void* out_feat, int ld_out, int dtype,
                                       const void* halo_table, const pbn_halo_layout* halo, int lds_slots, int cfg,
                                       pbn_stream_t stream) {
    if (n_out < 0 || n_in < 0 || n_offsets < 1 || vecs_per_offset < 4 || (vecs_per_offset & 3) || n_steps < 1 ||
        cout_padded < 16 || (cout_padded & 15) || !halo_table || !halo || !nbr)
        return PBN_ERR_ARG;
    if (n_steps != n_offsets * (vecs_per_offset >> 2) || halo->n_offsets != n_offsets) return PBN_ERR_ARG;
    {   // the table must have been laid out for THIS map: a table of another lineage or capacity would be indexed out of bounds
        pbn_halo_layout chk;
        if (!pbn_halo_bytes(n_out, n_offsets, halo->tile_rows, &chk) || memcmp(&chk, halo, sizeof(chk)) != 0) return PBN_ERR_ARG;
    }
    if (n_out == 0) return PBN_OK;
    if (!in_feat || !w_packed || !out_feat) return PBN_ERR_ARG;
    const int esz = dtype == PBN_F32 ? 4 : 2;
    if ((ld_in * esz) % 16 || (ld_out * esz) % 8 || (residual && (ld_res * esz) % 8)) return PBN_ERR_ARG;
    if (((uintptr_t)in_feat | (uintptr_t)w_packed) & 15) return PBN_ERR_ARG;
    const unsigned long long in_extent = (unsigned long long)n_in * (unsigned long long)ld_in * (unsigned long long)esz;
    const unsigned long long w_extent = (unsigned long long)n_steps * (unsigned long long)(cout_padded / 16) * 1024ull;
    if (in_extent >= 0x80000000ull || w_extent >= 0x80000000ull) return PBN_ERR_RANGE;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in_bytes = (unsigned)in_extent; a.w_bytes = (unsigned)w_extent;
    a.in = in_feat; a.nbr = nbr; a.row_perm = nullptr; a.n_out_dev = n_out_dev; a.w = w_packed; a.scale = scale;
    a.shift = shift; a.residual = residual; a.out = out_feat; a.ld_in = ld_in; a.ld_res = ld_res; a.ld_out = ld_out;
    a.K = n_offsets; a.vpo = vecs_per_offset; a.n_steps = n_steps; a.ntiles_total = cout_padded / 16; a.n_out = n_out;
    a.relu = relu; a.ksplit = 1; a.cg = 1;
    static const int dbg_env = getenv("PBN_HALO_DBG") ? atoi(getenv("PBN_HALO_DBG")) : 0;   // ablations (results are garbage): 1 no weight DMA, 2 no MFMA, 4 no barriers, 8 no row staging
    a.dbg = dbg_env;
    const char* A = (const char*)halo_table;
    if (cfg >= 0) {
        // the wave-autonomous family with staged rows (spconv_wave_halo.hip); 0 = by tile height and channel tiles
        int depth = cfg / 10000, wcfg = cfg % 10000;
        if (wcfg == 0) {
            const int ntt = a.ntiles_total, tr = halo->tile_rows;
            const int nt = ntt % 8 == 0 ? 8 : (ntt % 6 == 0 ? 6 : (ntt % 4 == 0 ? 4 : (ntt % 2 == 0 ? 2 : 1)));
            if (tr == 32 || tr == 64) {
                // K-split: the narrowest channel tile that still gives ~a workgroup per CU keeps the weight stream per workgroup short
                const long long tiles = (n_out + tr - 1) / tr;
                int pick = nt;
                for (int c : {8, 4, 2, 1})
                    if (ntt % c == 0 && c <= nt) { pick = c; if (tiles * (ntt / c) >= 180) break; }
                wcfg = 1000 + (tr / 16) * 100 + pick;
            } else if (tr == 128 || tr == 256) {
                wcfg = (tr / 64) * 100 + (nt == 1 ? 2 : nt);
            } else return PBN_ERR_UNSUPPORTED;
        }
        return launch_wave_halo(a, dtype, (const int*)(A + halo->cnt), (const int*)(A + halo->rows),
                                (const unsigned short*)(A + halo->loc), (const unsigned short*)(A + halo->fmask), halo->tile_rows,
                                halo->pitch, lds_slots, wcfg, depth, (hipStream_t)stream);
    }
    cfg = -cfg;
    return launch_halo(a, dtype, (const int*)(A + halo->cnt), (const int*)(A + halo->rows),
                       (const unsigned short*)(A + halo->loc), (const unsigned short*)(A + halo->fmask), halo->tile_rows,
                       halo->pitch, lds_slots, cfg, (hipStream_t)stream);
}
