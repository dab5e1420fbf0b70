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
constexpr unsigned UNIT_NONE = 0xffff0000u;   // step 0 (any valid weight address), offset 255 (no such offset): adds zeros

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
constexpr int pipe_depth(int nf, int nt) {
    (void)nf; (void)nt;
    return 2;
}

template <int NF, int NT>
struct Stage {
    u32x4 w[NT];
    u32x4 x[NF];
};

// unit descriptor (built once per wave, kept in LDS): step | offset of lane group 0 << 16 | its channel vector << 24
__device__ __forceinline__ unsigned pack_unit(int step, int ko, int cv0) {
    return (unsigned)step | ((unsigned)ko << 16) | ((unsigned)cv0 << 24);
}

__device__ __forceinline__ u32x4 buf_load(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0));
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

template <typename T, int NF, int NT, int KW, bool KSPLIT>
__global__ __launch_bounds__(KW * 64) void k_spconv_wave(const ConvArgs a) {
    static_assert(Tr<T>::ELEMS * sizeof(T) == 16, "one gather vector is 16 bytes");
    constexpr int RW = NF * 16;                          // rows per wave
    constexpr int TM = KSPLIT ? RW : KW * RW;            // rows per workgroup
    constexpr int NTB = NT < 2 ? NT : 2;                 // channel tiles per LDS reduction round (K-split)
    constexpr int TPB = KW * 64;
    constexpr int B = pipe_depth(NF, NT);                // units in flight per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = a.K;
    const int KS = K | 1;
    int* s_nbr = reinterpret_cast<int*>(smem);                                        // TM * KS
    unsigned* s_units = reinterpret_cast<unsigned*>(s_nbr + ((TM * KS + 3) & ~3));    // KW * (n_steps + 2)
    const int units_pitch = a.n_steps + 2 * MAX_DEPTH;
    float* s_red = reinterpret_cast<float*>(s_units + ((KW * units_pitch + 3) & ~3)); // K-split: KW * TM * NTB*16 floats

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PBN_CONV_TIMING
    const int wt_blk = blockIdx.y * gridDim.x + blockIdx.x;
    if (tid == 0 && wt_blk == 0) { g_wave_timing[WT_BLOCKS * 64] = gridDim.x; g_wave_timing[WT_BLOCKS * 64 + 1] = gridDim.y; }
#endif
    PBN_WSTAMP(7);
    PBN_WSTAMP(0);
    const int n = a.n_out_dev ? min(*a.n_out_dev, a.n_out) : a.n_out;
    const int row0 = xcd_tile(blockIdx.x, gridDim.x) * TM;
    if (row0 >= n) return;
    const int tile0 = blockIdx.y * NT;
    const int g = lane >> 4, rl = lane & 15;

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
    __syncthreads();
    PBN_WSTAMP(1);

    // ---- which offsets are populated among this wave's rows (K <= 128: two ballots) ----
    const int wrow0 = KSPLIT ? 0 : wave * RW;
    unsigned long long pop0, pop1;
    if constexpr (KSPLIT) {
        // coarse levels: 93-98 % of the (64-row tile, offset) pairs are populated (scripts/analyze_rulebook.py) -- visiting
        // every offset costs a few per cent of MFMA work and takes the population scan (2 x RW LDS reads per lane) out of
        // the prologue of a launch whose whole main loop is a few microseconds
        pop0 = pop1 = ~0ull;
    } else {
        bool any0 = false, any1 = false;
        if (lane < K)
            for (int r = 0; r < RW; ++r) any0 |= s_nbr[(wrow0 + r) * KS + lane] >= 0;
        if (64 + lane < K)
            for (int r = 0; r < RW; ++r) any1 |= s_nbr[(wrow0 + r) * KS + 64 + lane] >= 0;
        pop0 = __ballot(any0);
        pop1 = __ballot(any1);
    }
    auto populated = [&](int k) -> bool { return ((k < 64 ? pop0 >> k : pop1 >> (k - 64)) & 1ull) != 0ull; };

    // ---- this wave's unit list: populated steps, in order; K-split: every KW-th of them ----
    const int vpo = a.vpo;
    const bool wide = (vpo & 3) == 0;
    const int spo = wide ? (vpo >> 2) : 1;
    const int vshift = (vpo == 2) ? 1 : 0;
    unsigned* my_units = s_units + wave * units_pitch;
    int n_units;
    {
        int base = 0;   // populated steps seen so far (all waves of a K-split workgroup agree on it)
        for (int s0 = 0; s0 < a.n_steps; s0 += 64) {
            const int s = s0 + lane;
            bool ok = false;
            int ko = 0, cv0 = 0;
            if (s < a.n_steps) {
                if (wide) {
                    ko = s / spo;
                    cv0 = (s - ko * spo) * 4;
                    ok = populated(ko);
                } else {                      // a step spans 4 / vpo offsets; lane group g reads offset (4 s + g) >> vshift
                    ko = (s * 4) >> vshift;
                    const int span = 4 >> vshift;
                    for (int q = 0; q < span; ++q)
                        if (ko + q < K) ok |= populated(ko + q);
                }
            }
            const unsigned long long m = __ballot(ok);
            if (ok) {
                const int pos = base + __popcll(m & ((1ULL << lane) - 1ULL));
                if (!KSPLIT) my_units[pos] = pack_unit(s, ko, cv0);
                else if (pos % KW == wave) my_units[pos / KW] = pack_unit(s, ko, cv0);
            }
            base += __popcll(m);
        }
        n_units = KSPLIT ? (base > wave ? (base - wave + KW - 1) / KW : 0) : base;
        // sentinels: the pipeline below always has B units in flight and needs no tail handling
        if (lane < 2 * B) my_units[n_units + lane] = UNIT_NONE;
    }
    // the list is wave-private and a wave's LDS operations execute in order: the reads below follow the writes above
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    PBN_WSTAMP(2);

    f32x4 acc[NF][NT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)a.w_bytes, 0x00020000);
    const unsigned ld_bytes = (unsigned)a.ld_in * (unsigned)sizeof(T);
    const unsigned w_lane = (unsigned)lane * 16u;
    const unsigned w_tile0 = (unsigned)tile0 * 1024u;
    const unsigned w_step_bytes = (unsigned)a.ntiles_total * 1024u;
    const int my_row = wrow0 + rl;

    // all loads of one unit: NT weight fragments (wave-uniform address + lane * 16) and NF gathered row fragments (one
    // 16-byte vector of one neighbour row per lane; no neighbour / past the last offset -> out-of-range offset -> zeros)
    auto issue = [&](Stage<NF, NT>& st, int i) {
        const unsigned u = (unsigned)__builtin_amdgcn_readfirstlane((int)my_units[i]);
        const unsigned step = u & 0xffffu, ko0 = (u >> 16) & 0xffu, cv0 = u >> 24;
        int ko, cv;
        if (wide) {
            ko = (int)ko0;
            cv = (int)cv0 + g;
        } else {
            const int v = (int)step * 4 + g;
            ko = ko0 == 0xffu ? 255 : (v >> vshift);
            cv = v & (vpo - 1);
        }
        const unsigned w_soff = step * w_step_bytes + w_tile0;
#pragma unroll
        for (int t = 0; t < NT; ++t) st.w[t] = buf_load(rs_w, w_lane, w_soff + (unsigned)t * 1024u);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int src = ko < K ? s_nbr[(my_row + f * 16) * KS + ko] : -1;
            const unsigned voff = src >= 0 ? (unsigned)src * ld_bytes + (unsigned)cv * 16u : OOB;
            st.x[f] = buf_load(rs_in, voff, 0u);
        }
    };
    auto compute = [&](const Stage<NF, NT>& st) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int f = 0; f < NF; ++f) mfma_step<T>(st.w[t], st.x[f], acc[f][t]);
    };

    {
        // B units in flight: a unit's registers are refilled with the unit B positions ahead as soon as its MFMAs are
        // issued (loads retire in order, so the compiler's counted vmcnt waits are exact); a coarse-level wave-unit is
        // ~100 cycles of MFMA against ~1500 cycles of L2 latency, hence the depth
        Stage<NF, NT> st[B];
#pragma unroll
        for (int s = 0; s < B; ++s) issue(st[s], s);
        PBN_WSTAMP(3);
        for (int i = 0; i < n_units; i += B) {
#pragma unroll
            for (int s = 0; s < B; ++s) {
                compute(st[s]);
                issue(st[s], i + B + s);
            }
        }
        // the trailing issues read sentinels: harmless loads that are never consumed
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
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += NTB) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int tt = 0; tt < NTB; ++tt) {
                    const f32x4 v = acc[f][t0 + tt];
                    *reinterpret_cast<float4*>(s_red + ((size_t)(wave * TM + f * 16 + rl) * RP + tt * 16 + g * 4)) =
                        make_float4(v[0], v[1], v[2], v[3]);
                }
            __syncthreads();
            if (t0 == 0) { PBN_WSTAMP(5); }
            for (int e = tid; e < TM * (RP / 4); e += TPB) {
                const int r = e / (RP / 4), q = e - r * (RP / 4);
                const int p = row0 + r;
                if (p >= n) continue;
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < KW; ++w) {
                    const float4 s = *reinterpret_cast<const float4*>(s_red + ((size_t)(w * TM + r) * RP + q * 4));
                    v[0] += s.x; v[1] += s.y; v[2] += s.z; v[3] += s.w;
                }
                const int orow = a.row_perm ? a.row_perm[p] : p;
                epilogue_store<T>(a, v, orow, (tile0 + t0) * 16 + q * 4);
            }
            if (t0 + NTB < NT) __syncthreads();
        }
    }
    PBN_WSTAMP(6);
}

template <typename T, int NF, int NT, int KW, bool KSPLIT>
int launch_cfg(const ConvArgs& a, hipStream_t stream) {
    constexpr int RW = NF * 16;
    constexpr int TM = KSPLIT ? RW : KW * RW;
    constexpr int NTB = NT < 2 ? NT : 2;
    if (a.ntiles_total % NT) return PBN_ERR_UNSUPPORTED;
    const int KS = a.K | 1;
    size_t lds = sizeof(int) * (size_t)((TM * KS + 3) & ~3) +
                 sizeof(unsigned) * (size_t)((KW * (a.n_steps + 2 * MAX_DEPTH) + 3) & ~3);
    if (KSPLIT) lds += sizeof(float) * (size_t)KW * TM * NTB * 16;
    if (lds > 160 * 1024 || a.K > 128 || a.n_steps > 0xfffe) return PBN_ERR_UNSUPPORTED;
    auto kern = k_spconv_wave<T, NF, NT, KW, KSPLIT>;
    if (lds > 64 * 1024)
        PBN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(cdiv(a.n_out, TM), a.ntiles_total / NT), dim3(KW * 64), lds, stream, a);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

// cfg = 1000 * ksplit + 100 * NF + NT
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
        default: return PBN_ERR_UNSUPPORTED;
    }
}

// Pick (mode, NF, NT).  The wave-level cost is load bytes through the CU's 64 B/clk return path: a workgroup moves
// (all populated K) x (NT KiB of weights + NF KiB of rows) per wave-slice; the chip wants >= 256 workgroups.  K-split
// spreads ONE row tile over 8 waves, so it is the mode of choice whenever row-split tiles alone cannot fill the CUs.
int pick_cfg(const ConvArgs& a) {
    const int ntt = a.ntiles_total;
    const long long rows = a.n_out;
    // stem-like layers (a 16/32-byte input row, K = 125): 32 rows x 32 channels per workgroup measured best at every size
    // (146 k rows: 44.8 us against 68.1 for 64 rows and 72.6 for the workgroup-tile kernel)
    if (a.vpo <= 2 && a.K >= 64 && ntt % 2 == 0) return 1202;
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
// Measured crossover (scripts/probe_wave.py, HIP-graph replay): the K-split wave kernel is level with or ahead of the
// workgroup-tile kernel + its split-K reduce launch up to ~6e9 dense MACs (rows x K x C_in x C_out) on levels below 20 k
// rows, and behind it above (L3 384->256, L2 128->128) and on every wide level.
bool wave_family_wanted(const ConvArgs& a, int dtype) {
    (void)dtype;
    static const int fam = getenv("PBN_CONV_FAMILY") ? atoi(getenv("PBN_CONV_FAMILY")) : 2;
    static const int max_rows = getenv("PBN_WAVE_MAX_ROWS") ? atoi(getenv("PBN_WAVE_MAX_ROWS")) : 20000;
    static const double max_macs = getenv("PBN_WAVE_MAX_GMACS") ? atof(getenv("PBN_WAVE_MAX_GMACS")) * 1e9 : 6.3e9;
    if (a.K > 128) return false;
    if (fam == 0) return false;
    if (fam == 1) return true;
    if (a.vpo <= 2 && a.K >= 64 && !a.row_perm) return true;   // the k = 5 stem: rulebook-bound, no weight reuse to speak of
    const double elems_per_step = 4.0 * (dtype == PBN_F32 ? 4.0 : 8.0);
    const double dense = (double)a.n_out * a.n_steps * elems_per_step * a.ntiles_total * 16.0;
    return a.n_out < max_rows && dense <= max_macs;
}

}  // namespace pbn
