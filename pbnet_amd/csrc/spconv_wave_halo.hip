// spconv_wave_halo.hip -- the wave-autonomous convolution family of spconv_wave.hip with its ROW operands staged in LDS
// (round 4).  Same arithmetic (MinkowskiConvolution / ConvolutionTranspose forward, /root/reference/network/Mink.py:221-288,
// 293-350):   out[o,:] = epilogue( sum_k in[nbr[o,k],:] @ W[k] ).
//
// Why: spconv_wave.hip runs at the CU's vector-memory rate -- every wave pulls NT KiB of weights AND NF KiB of gathered rows
// per unit through the same path, and a row is gathered once per offset that names it (7-15 times).  On a Z-ordered lineage a
// tile of 32..256 consecutive output rows names only 1.3-2.5x as many DISTINCT input rows as it has rows (the "halo";
// pbn_halo_build lists them per tile once per lineage, csrc/spconv_halo.hip).  Here a workgroup stages the whole rows of its
// halo ONCE (LDS-DMA, four lanes per 64-byte piece, ~H x C_in / 1 KiB instructions) and every MFMA row operand is a
// ds_read_b128; the main loop's only vector-memory traffic is the weight stream, still global/L2 -> registers per wave with
// one hand-counted s_waitcnt per unit, no barrier, no M0 traffic -- the loop of spconv_wave.hip minus its gathers and its
// ds_bpermute transposes (operand order comes out of LDS directly).
//   * K-SPLIT (coarse levels): the KW waves share one tile of NF*16 rows and take the reduction steps round-robin; partial
//     tiles are summed through LDS in wave order (the buffer overlays the staged rows), epilogue once;
//   * ROW-SPLIT (wide levels): each wave owns NF*16 rows of a KW*NF*16-row tile and walks the steps of ITS populated offsets
//     (fragment masks of the halo tables).
// LDS layout of the staged rows: plane c = the c-th 64-byte piece of every slot, slot pitch 64 B, the four 16-byte chunks of
// a piece swizzled by slot bit 2 (runs of consecutive slots, the common case in Z-order, read conflict-free).  A tile whose
// halo exceeds the buffer runs once per segment of its list; a tile marked by the build (-1) takes a plain gather loop.
#include <cstdlib>
#include <cstring>
#include "spconv_common.h"

namespace pbn {
namespace {

constexpr unsigned OOB = 0x80000000u;
constexpr int U_SENTINEL = 0x10000;     // unit word: offset | piece << 8; sentinel = a unit past the end (zero weights, zero rows)
constexpr int U_TAIL = 8;               // sentinels behind a unit list

struct WhArgs {
    const int* cnt;
    const int* rows;
    const unsigned short* loc;
    const unsigned short* fmask;
    int tm, pitch;
    int hs;            // slots of the LDS row buffer (multiple of 16)
};

#define PBN_LDS_ADDR(p) ((unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)(p)))

__device__ __forceinline__ void wh_dma16(unsigned lds_dst, unsigned voff, const i32x4& rs, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ void wh_load_asm(u32x4& dst, const i32x4& rs, unsigned voff, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}

constexpr int wh_round_tiles(int kw, int tm, int nt) { return (kw * tm * nt * 64 <= 64 * 1024) ? nt : (nt < 2 ? nt : 2); }
constexpr int wh_red_pitch(int ntb) { return ntb * 16 + 4; }

struct WhTile { int row_tile, group; bool valid; };
__device__ __forceinline__ WhTile wh_map_block(const ConvArgs& a, int n_row_tiles, int n_groups) {
    WhTile m;
    const int b = blockIdx.x;
    if (!a.wmajor) {
        m.group = b / n_row_tiles;
        m.row_tile = xcd_tile(b - m.group * n_row_tiles, n_row_tiles);
        m.valid = m.group < n_groups;
        return m;
    }
    const int xcd = b & 7, idx = b >> 3;
    if (n_groups >= 8) {
        const int gl = idx / n_row_tiles;
        m.group = xcd + 8 * gl;
        m.row_tile = idx - gl * n_row_tiles;
        m.valid = m.group < n_groups;
    } else {
        const int share = 8 / n_groups;
        m.group = xcd % n_groups;
        m.row_tile = (xcd / n_groups) + share * idx;
        m.valid = m.row_tile < n_row_tiles;
    }
    return m;
}

template <typename T, int NF, int NT, int KW, bool KSPLIT, int B>
__global__ __launch_bounds__(KW * 64) void k_spconv_wh(const ConvArgs a, const WhArgs h) {
    static_assert(Tr<T>::ELEMS * sizeof(T) == 16, "one gather vector is 16 bytes");
    constexpr int RW = NF * 16, TM = KSPLIT ? RW : KW * RW, TPB = KW * 64;
    constexpr int NTB = wh_round_tiles(KW, TM, NT);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = a.K, KS = K | 1, HS = h.hs;
    const int spo = a.vpo >> 2;                                       // 64-byte pieces per row (= steps per offset)
    const int plane = (HS + 16) * 64;
    const int n_steps = a.n_steps;
    const int upw = (KSPLIT ? (n_steps + KW - 1) / KW : n_steps) + U_TAIL;       // unit words per wave
    size_t x_bytes = (size_t)spo * plane;
    if (KSPLIT) {
        const size_t red = sizeof(float) * (size_t)KW * TM * wh_red_pitch(NTB);
        if (red > x_bytes) x_bytes = red;
    }
    unsigned char* s_x = smem;                                                                     // staged rows | reduction buffer
    unsigned short* s_loc = reinterpret_cast<unsigned short*>(smem + ((x_bytes + 15) & ~(size_t)15));   // TM x KS
    int* s_rows = reinterpret_cast<int*>(s_loc + ((TM * KS + 7) & ~7));                            // HS
    int* s_units = s_rows + HS;                                                                    // KW x upw
    float* s_ss = reinterpret_cast<float*>(s_units + ((KW * upw + 3) & ~3));                       // scale | shift (K-split)
    float* s_red = reinterpret_cast<float*>(s_x);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_groups = a.ntiles_total / NT;
    const int n = a.n_out_dev ? min(*a.n_out_dev, a.n_out) : a.n_out;
    const WhTile tm = wh_map_block(a, (a.n_out + TM - 1) / TM, n_groups);
    if (!tm.valid) return;
    const int tile = tm.row_tile;
    const int row0 = tile * TM;
    if (row0 >= n) return;
    const int tile0 = tm.group * NT;
    const int g = lane >> 4, rl = lane & 15;
    const int wrow0 = KSPLIT ? 0 : wave * RW;
    const int H = h.cnt[tile];

    const unsigned long long in_addr = (unsigned long long)a.in, w_addr = (unsigned long long)a.w;
    const i32x4 rs_in = {(int)(unsigned)in_addr, (int)(unsigned)(in_addr >> 32), (int)a.in_bytes, 0x00020000};
    const i32x4 rs_w = {(int)(unsigned)w_addr, (int)(unsigned)(w_addr >> 32), (int)a.w_bytes, 0x00020000};
    const unsigned ld_bytes = (unsigned)a.ld_in * (unsigned)sizeof(T);
    const unsigned w_lane = (unsigned)lane * 16u;
    const unsigned w_tile0 = (unsigned)tile0 * 1024u;
    const unsigned w_step_bytes = (unsigned)a.ntiles_total * 1024u;

    f32x4 acc[NF][NT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (KSPLIT) {
        if (tid < NT * 32) {
            const int c = tile0 * 16 + (tid < NT * 16 ? tid : tid - NT * 16);
            const float* src = tid < NT * 16 ? a.scale : a.shift;
            s_ss[tid] = src ? src[c] : (tid < NT * 16 ? 1.0f : 0.0f);
        }
    }

    if (H < 0) {
        // ---- marked tile: plain gathers through the map itself (slow, correct) ----
        for (int s = KSPLIT ? wave : 0; s < n_steps; s += KSPLIT ? KW : 1) {
            const int k = s / spo, c = s - k * spo;
            u32x4 x[NF];
            bool any = false;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int p = row0 + wrow0 + f * 16 + rl;
                const int srow = p < n ? a.nbr[(size_t)p * K + k] : -1;
                x[f] = u32x4{0u, 0u, 0u, 0u};
                if (srow >= 0) {
                    any = true;
                    x[f] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(a.in) + (size_t)srow * ld_bytes +
                                                           (size_t)c * 64 + (size_t)g * 16);
                }
            }
            if (!__any(any)) continue;
            const u32x4* wp = reinterpret_cast<const u32x4*>(a.w) + ((size_t)s * a.ntiles_total + tile0) * 64 + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const u32x4 wf = wp[(size_t)t * 64];
#pragma unroll
                for (int f = 0; f < NF; ++f) mfma_step<T>(wf, x[f], acc[f][t]);
            }
        }
        __syncthreads();
    } else {
        // ---- slot table, row list, zero slots ----
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
        }
        // ---- this wave's unit list ----
        int* my_units = s_units + wave * upw;
        int n_units;
        if constexpr (KSPLIT) {
            n_units = n_steps > wave ? (n_steps - wave + KW - 1) / KW : 0;
            const float inv_s = 1.0f / (float)spo;
            for (int i = lane; i < n_units + U_TAIL; i += 64) {
                int v = U_SENTINEL;
                if (i < n_units) {
                    const int s = wave + i * KW;
                    const int k = (int)(((float)s + 0.5f) * inv_s);
                    v = k | ((s - k * spo) << 8);
                }
                my_units[i] = v;
            }
        } else {
            // populated offsets of this wave's fragments (fragment masks of the halo tables), spo steps each, in order
            const unsigned wb = ((1u << NF) - 1u) << (wave * NF);
            int base = 0;
            for (int k0 = 0; k0 < K; k0 += 64) {
                const int k = k0 + lane;
                const bool ok = k < K && ((unsigned)h.fmask[(size_t)tile * K + k] & wb) != 0u;
                const unsigned long long m = __ballot(ok);
                if (ok) {
                    const int pos = (base + __popcll(m & ((1ULL << lane) - 1ULL))) * spo;
                    for (int c = 0; c < spo; ++c) my_units[pos + c] = k | (c << 8);
                }
                base += __popcll(m);
            }
            n_units = base * spo;
            if (lane < U_TAIL) my_units[n_units + lane] = U_SENTINEL;
        }
        __syncthreads();

        const unsigned lds_x = PBN_LDS_ADDR(s_x);
        const unsigned short* my_loc = s_loc + (wrow0 + rl) * KS;
        const int nseg = (H + HS - 1) / HS;
        for (int seg = 0; seg < nseg; ++seg) {
            const int seg0 = seg * HS;
            const int hseg = min(H - seg0, HS);
            if (seg > 0) {
                __syncthreads();                                          // every wave is done with the previous segment's rows
                for (int s = tid; s < hseg; s += TPB) s_rows[s] = h.rows[(size_t)tile * h.pitch + seg0 + s];
                __syncthreads();
            }
            {   // stage every 64-byte piece of the segment's rows: block = 16 slots of one piece, lane 4 s + j' fetches chunk
                // j' ^ swizzle(slot); an out-of-range lane writes zeros (slots behind the list, incl. the zero slots at HS)
                const int nblk = ((hseg + 15) >> 4);
                const int nb1 = (HS >> 4) + 1;                            // blocks of a plane incl. the zero block
                const float inv_b = 1.0f / (float)nb1;
                for (int e = wave; e < nb1 * spo; e += KW) {
                    const int c = (int)(((float)e + 0.5f) * inv_b), b = e - c * nb1;
                    if (b >= nblk && b != nb1 - 1) continue;              // wave-uniform
                    const int slot = b * 16 + (lane >> 2);
                    const int row = slot < hseg ? s_rows[slot] : -1;
                    const unsigned voff = row >= 0 ? (unsigned)row * ld_bytes + (unsigned)c * 64u + (unsigned)((((lane & 3) ^ ((slot >> 1) & 2))) << 4) : OOB;
                    wh_dma16(__builtin_amdgcn_readfirstlane(lds_x + (unsigned)c * (unsigned)plane + (unsigned)b * 1024u), voff, rs_in, 0u);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();

            // ---- main loop: B weight stages in registers, row operands one unit ahead, slots two units ahead ----
            struct Stage { u32x4 w[NT]; };
            Stage st[B];
#pragma unroll
            for (int s = 0; s < B; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t) st[s].w[t] = u32x4{0u, 0u, 0u, 0u};
            auto issue_w = [&](Stage& sg, int uw) {
                const bool live = !(uw & U_SENTINEL);
                const unsigned step = (unsigned)((uw & 0xff) * spo + ((uw >> 8) & 0xff));
                const unsigned soff = (live ? step : 0u) * w_step_bytes + w_tile0;
                const unsigned voff = live ? w_lane : OOB;
#pragma unroll
                for (int t = 0; t < NT; ++t) wh_load_asm(sg.w[t], rs_w, voff, soff + (unsigned)t * 1024u);
            };
            auto slot_addr = [&](int uw, int f) -> int {
                int slot = (int)my_loc[f * 16 * KS + (uw & 0xff)] - seg0;
                if ((unsigned)slot >= (unsigned)HS || (uw & U_SENTINEL)) slot = HS;           // none / other segment / past the end: zeros
                return ((uw >> 8) & 0xff) * plane + slot * 64 + ((g ^ ((slot >> 1) & 2)) << 4);
            };
            constexpr int BEHIND = (B - 1) * NT;
            auto wait_unit = [&](Stage& u) {
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(u.w[0]) : "n"(BEHIND));
#pragma unroll
                for (int t = 1; t < NT; ++t) asm volatile("" : "+v"(u.w[t]));
            };
            // unit words: uq[0] = unit i, uq[j] = unit i + j; one more is fetched per unit
            int uq[B + 2];
#pragma unroll
            for (int j = 0; j < B + 2; ++j) uq[j] = __builtin_amdgcn_readfirstlane(my_units[j]);
            int raw = my_units[B + 2];
#pragma unroll
            for (int s = 0; s < B; ++s) issue_w(st[s], uq[s]);
            u32x4 bx[NF], bn[NF];
            int ad[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                bx[f] = *reinterpret_cast<const u32x4*>(s_x + slot_addr(uq[0], f));
                ad[f] = slot_addr(uq[1], f);
            }
            for (int i = 0; i < n_units; i += B) {
#pragma unroll
                for (int s = 0; s < B; ++s) {
                    // rows of unit i+s+1, slots of unit i+s+2, the word of unit i+s+B+3
#pragma unroll
                    for (int f = 0; f < NF; ++f) bn[f] = *reinterpret_cast<const u32x4*>(s_x + ad[f]);
#pragma unroll
                    for (int f = 0; f < NF; ++f) ad[f] = slot_addr(uq[2], f);
                    const int nxt = __builtin_amdgcn_readfirstlane(raw);
                    raw = my_units[min(i + s + B + 3, n_units + U_TAIL - 1)];
                    wait_unit(st[s]);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int f = 0; f < NF; ++f) mfma_step<T>(st[s].w[t], bx[f], acc[f][t]);
                    __builtin_amdgcn_sched_barrier(0);     // the refill stays behind the unit's own MFMAs
                    issue_w(st[s], uq[B]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int f = 0; f < NF; ++f) bx[f] = bn[f];
#pragma unroll
                    for (int j = 0; j < B + 1; ++j) uq[j] = uq[j + 1];
                    uq[B + 1] = nxt;
                }
            }
            // the trailing issues were sentinels; drain them, naming every stage register (they stay allocated up to here)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int s = 0; s < B; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(st[s].w[t]));
        }
        __syncthreads();       // K-split: the reduction buffer overlays the staged rows
    }

    if constexpr (!KSPLIT) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int p = row0 + wave * RW + f * 16 + rl;
            if (p >= n) continue;
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
                if (a.residual) v += load4<T>(reinterpret_cast<const T*>(a.residual) + (size_t)p * a.ld_res + c0);
                if (a.relu) {
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                store4<T>(reinterpret_cast<T*>(a.out) + (size_t)p * a.ld_out + c0, v);
            }
        }
    } else {
        // fixed-order sum of the KW partial tiles through LDS, NTB channel tiles per round, then the epilogue
        constexpr int RP = NTB * 16;
        constexpr int RPP = wh_red_pitch(NTB);
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
            for (int e = tid; e < TM * (RP / 4); e += TPB) {
                const int r = e / (RP / 4), q = e - r * (RP / 4);
                const int p = row0 + r;
                if (p >= n) continue;
                const int cl = t0 * 16 + q * 4;
                f32x4 rv = f32x4{0.f, 0.f, 0.f, 0.f};
                if (res) rv = load4<T>(res + (size_t)p * a.ld_res + tile0 * 16 + cl);
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
                store4<T>(out + (size_t)p * a.ld_out + tile0 * 16 + cl, v);
            }
            if (t0 + NTB < NT) __syncthreads();
        }
    }
}

template <typename T, int NF, int NT, int KW, bool KSPLIT, int B>
int launch_wh_cfg(ConvArgs a, const WhArgs& h, hipStream_t stream) {
    constexpr int RW = NF * 16;
    constexpr int TM = KSPLIT ? RW : KW * RW;
    constexpr int NTB = wh_round_tiles(KW, TM, NT);
    if (a.ntiles_total % NT || h.tm != TM) return PBN_ERR_UNSUPPORTED;
    const int KS = a.K | 1, spo = a.vpo >> 2;
    const int upw = (KSPLIT ? (a.n_steps + KW - 1) / KW : a.n_steps) + U_TAIL;
    size_t x_bytes = (size_t)spo * (h.hs + 16) * 64;
    if (KSPLIT) {
        const size_t red = sizeof(float) * (size_t)KW * TM * wh_red_pitch(NTB);
        if (red > x_bytes) x_bytes = red;
    }
    const size_t lds = ((x_bytes + 15) & ~(size_t)15) + sizeof(unsigned short) * (size_t)((TM * KS + 7) & ~7) +
                       sizeof(int) * ((size_t)h.hs + (size_t)((KW * upw + 3) & ~3)) + sizeof(float) * 2 * NT * 16;
    if (lds > 160 * 1024 || a.K > 128 || spo > 255) return PBN_ERR_UNSUPPORTED;
    auto kern = k_spconv_wh<T, NF, NT, KW, KSPLIT, B>;
    if (lds > 64 * 1024)
        PBN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int row_tiles = cdiv(a.n_out, TM), groups = a.ntiles_total / NT;
    const bool shape_ok = groups >= 8 ? (groups % 8 == 0) : (groups == 1 || groups == 2 || groups == 4);
    a.wmajor = (KSPLIT && a.w_bytes > a.in_bytes && shape_ok && groups > 1) ? 1 : 0;
    int blocks = row_tiles * groups;
    if (a.wmajor && groups < 8) blocks = 8 * cdiv(row_tiles, 8 / groups);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(KW * 64), lds, stream, a, h);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

// cfg = 1000 * ksplit + 100 * NF + NT (as spconv_wave.hip); depth = weight stages in flight per wave (2 or 3)
template <typename T>
int launch_wh_by_cfg(const ConvArgs& a, const WhArgs& h, int cfg, int depth, hipStream_t stream) {
#define PBN_WH_CASE(CODE, NFV, NTV, KWV, KS)                                                        \
    case CODE: return depth == 3 ? launch_wh_cfg<T, NFV, NTV, KWV, KS, 3>(a, h, stream) : launch_wh_cfg<T, NFV, NTV, KWV, KS, 2>(a, h, stream);
#define PBN_WH_CASE2(CODE, NFV, NTV, KWV, KS) case CODE: return launch_wh_cfg<T, NFV, NTV, KWV, KS, 2>(a, h, stream);   /* depth 3 would spill */
    switch (cfg) {
        PBN_WH_CASE(402, 4, 2, 4, false) PBN_WH_CASE(404, 4, 4, 4, false) PBN_WH_CASE(406, 4, 6, 4, false) PBN_WH_CASE(408, 4, 8, 4, false)
        PBN_WH_CASE(202, 2, 2, 4, false) PBN_WH_CASE(204, 2, 4, 4, false) PBN_WH_CASE(206, 2, 6, 4, false) PBN_WH_CASE(208, 2, 8, 4, false)
        PBN_WH_CASE(1401, 4, 1, 8, true) PBN_WH_CASE(1402, 4, 2, 8, true) PBN_WH_CASE(1404, 4, 4, 8, true) PBN_WH_CASE2(1408, 4, 8, 8, true)
        PBN_WH_CASE(1201, 2, 1, 8, true) PBN_WH_CASE(1202, 2, 2, 8, true) PBN_WH_CASE(1204, 2, 4, 8, true) PBN_WH_CASE(1208, 2, 8, 8, true)
        default: return PBN_ERR_UNSUPPORTED;
    }
#undef PBN_WH_CASE
#undef PBN_WH_CASE2
}

}  // namespace

// rows per tile of a configuration
int wh_tile_rows(int cfg) {
    const int nf = (cfg / 100) % 10;
    return cfg >= 1000 ? nf * 16 : 4 * nf * 16;
}

int launch_wave_halo(const ConvArgs& a, int dtype, const int* cnt, const int* rows, const unsigned short* loc,
                     const unsigned short* fmask, int tile_rows, int pitch, int lds_slots, int cfg, int depth,
                     hipStream_t stream) {
    if ((a.vpo & 3) || a.K > 128 || !a.nbr || a.row_perm || cfg <= 0 || wh_tile_rows(cfg) != tile_rows) return PBN_ERR_UNSUPPORTED;
    WhArgs h;
    h.cnt = cnt; h.rows = rows; h.loc = loc; h.fmask = fmask; h.tm = tile_rows; h.pitch = pitch;
    static const int hs_env = getenv("PBN_WH_SLOTS") ? atoi(getenv("PBN_WH_SLOTS")) : 0;
    static const int depth_env = getenv("PBN_WH_DEPTH") ? atoi(getenv("PBN_WH_DEPTH")) : 0;
    if (lds_slots <= 0) lds_slots = hs_env;
    if (lds_slots <= 0) {
        // default: the largest halo seen on the bench scene's levels plus a margin, cut down until the staged rows fit
        lds_slots = tile_rows <= 32 ? 144 : (tile_rows <= 64 ? 256 : (tile_rows <= 128 ? 384 : 560));
        if (a.K > 27) lds_slots *= 2;
        const int spo = a.vpo >> 2;
        while (lds_slots > 48 && (size_t)spo * (lds_slots + 16) * 64 > (size_t)120 * 1024) lds_slots -= 16;
    }
    h.hs = (lds_slots + 15) & ~15;
    if (h.hs > 4096) h.hs = 4096;
    if (depth != 2 && depth != 3) depth = depth_env == 3 ? 3 : 2;
    switch (dtype) {
        case PBN_F32: return launch_wh_by_cfg<float>(a, h, cfg, depth, stream);
        case PBN_BF16: return launch_wh_by_cfg<__hip_bfloat16>(a, h, cfg, depth, stream);
        case PBN_F16: return launch_wh_by_cfg<__half>(a, h, cfg, depth, stream);
        default: return PBN_ERR_ARG;
    }
}

}  // namespace pbn
