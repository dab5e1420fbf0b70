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

// (Round 5 also had a STAGED form here -- k_spconv_rsh + k_rs_table_build: the tile's distinct input rows staged in LDS, operands
// by ds_read_b128, ~900 lines.  It served ONE launch per forward (128 -> 96 at 146 k rows, where this form's 5-fragment instantiation
// spills); round 6 measured this form at 3 fragments per wave / 288-row tiles / two rounds on that launch at 109.6 us against the
// staged form's 112.9 (profiles/r06_probe_rs_shapes.txt) and removed the staged form, its per-map tables and their C ABI
// entries; the code is in the history: git show 794bf2f:pbnet_amd/csrc/spconv_rs.hip.)

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
RsShape rs_shape(int n_sel, int force_nf, int force_rows = 0, int n_out = -1, int nf_max = RS_NF_MAX) {
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
            if (s.tile_rows <= RS_NW * 16 * nf_max) break;
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

// fragments per wave an instantiation is built for: 5 (640-row tiles) where the accumulators fit, 3 at 128 output channels
// (8 channel tiles x 4 registers per fragment) -- round 6: the 64- and 128-channel shapes, for the levels that reach 20 k rows when
// several scenes share a forward (pbnet_amd/serving.py) or the scene is dense (configs[3])
constexpr int rs_nf_max(int nt, int cg) { return nt >= 8 ? 3 : ((nt == 6 && cg == 4) ? 4 : 5); }    // (6 tiles x 4 steps at 5 fragments spills)

template <typename T, int NT, int CG, int RING>
int launch_rs_nf(const ConvArgs& a, const RsShape& s, hipStream_t stream) {
    switch (s.nf) {
        case 1: return launch_rs_one<T, 1, NT, CG, RING>(a, s, stream);
        case 2: return launch_rs_one<T, 2, NT, CG, RING>(a, s, stream);
        case 3: return launch_rs_one<T, 3, NT, CG, RING>(a, s, stream);
        case 4: if constexpr (rs_nf_max(NT, CG) >= 4) return launch_rs_one<T, 4, NT, CG, RING>(a, s, stream); else return PBN_ERR_UNSUPPORTED;
        case 5: if constexpr (rs_nf_max(NT, CG) >= 5) return launch_rs_one<T, 5, NT, CG, RING>(a, s, stream); else return PBN_ERR_UNSUPPORTED;
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
    if constexpr (!std::is_same<T, float>::value) {        // (16-bit slabs only: the parity configuration never reaches these levels' row counts)
        if (nt == 4 && cg == 1) return launch_rs_nf<T, 4, 1, 2>(a, s, stream);
        if (nt == 4 && cg == 2) return launch_rs_nf<T, 4, 2, 2>(a, s, stream);
        if (nt == 8 && cg == 2) return launch_rs_nf<T, 8, 2, 2>(a, s, stream);
        if (nt == 8 && cg == 3) return launch_rs_nf<T, 8, 3, 2>(a, s, stream);
        if (nt == 8 && cg == 4) return launch_rs_nf<T, 8, 4, 2>(a, s, stream);
    }
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

// Where this family pays (MI355X, bench scene, bf16; scripts/probe_rs.py, profiles/r05_probe_rs.txt, r06_probe_rs_shapes.txt): a few per
// cent to 20 % ahead of k_spconv from 20 k rows for 96 output channels and from 40 k rows for 32 (96->96 at 146 k rows 93.7-98.8 ->
// 79.9-86.3 us, k = 2 transposed 96->96 47.1 -> 40.7, 128->96 at 26 k rows 46.0 -> 35.6; at 26 k rows x 32 channels it loses: 12.4 -> 18.8).
bool rs_family_wanted(const ConvArgs& a, int dtype) {
    static const int env = getenv("PBN_CONV_RS") ? atoi(getenv("PBN_CONV_RS")) : 1;
    static const int min_rows = getenv("PBN_RS_MIN_ROWS") ? atoi(getenv("PBN_RS_MIN_ROWS")) : 20000;
    static const int min_rows2 = getenv("PBN_RS_MIN_ROWS2") ? atoi(getenv("PBN_RS_MIN_ROWS2")) : 40000;
    if (!env || dtype == PBN_F32 || a.row_perm || a.K > 32 || a.n_sel < min_rows) return false;
    const int cg = rs_cg(a);
    if (!cg) return false;
    const int nt = a.ntiles_total;
    static const int wide = getenv("PBN_RS_WIDE") ? atoi(getenv("PBN_RS_WIDE")) : 1;       // round 6: 64 / 128 output channels
    if (wide && ((nt == 4 && cg <= 2) || (nt == 8 && cg >= 2))) return true;
    return (nt == 6 && (cg == 3 || cg == 4)) || (nt == 2 && cg == 1 && a.n_sel >= min_rows2);
}

// cfg: 0 = automatic tile height; 1..5 = that many fragments per wave (tests, tuning); + 100000 * (rows / 16): that tile height
// (+ 2000, the round-5 code of this form, is accepted and ignored)
int launch_rs(const ConvArgs& a, int dtype, int cfg, hipStream_t stream) {
    const int force_rows = (cfg / 100000) * 16;
    cfg %= 100000;
    if (cfg >= 2000) cfg -= 2000;
    const int cg = rs_cg(a);
    if (!cg || !a.nbr || a.row_perm || a.K > 32 || cfg < 0 || cfg > RS_NF_MAX) return PBN_ERR_UNSUPPORTED;   // (identity maps stay on the other families)
    if (a.in2 && ((a.vpo2 & 3) || a.n_main % cg || a.n_steps % cg)) return PBN_ERR_UNSUPPORTED;
    if (!a.in2 && a.n_steps % cg) return PBN_ERR_UNSUPPORTED;
    if (cfg > rs_nf_max(a.ntiles_total, cg)) return PBN_ERR_UNSUPPORTED;
    ConvArgs b = a;
    b.cg = cg;
    const RsShape s = rs_shape(a.n_sel, cfg, force_rows, a.n_out, rs_nf_max(a.ntiles_total, cg));
    switch (dtype) {
        case PBN_BF16: return launch_rs_t<__hip_bfloat16>(b, s, cg, stream);
        case PBN_F16: return launch_rs_t<__half>(b, s, cg, stream);
        case PBN_F32: return launch_rs_t<float>(b, s, cg, stream);
        default: return PBN_ERR_ARG;
    }
}

}  // namespace pbn
