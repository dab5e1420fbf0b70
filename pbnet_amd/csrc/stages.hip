// stages.hip -- the glue between the big kernels of PBNet.forward (/root/reference/network/PBNet.py:113-280), fused:
// every entry point replaces a run of small tensor ops (index / cat / where / floor / arange ...) of the reference by ONE
// launch with the same integer results and the same fp32 arithmetic.  Nothing here is arithmetic-heavy; the point is
// launch count (host and device) on the inference path, where the stages between U-Nets are latency bound.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include "pbn_common.h"

namespace pbn {
namespace {

constexpr int TPB = 256;

// largest e in [0, n) with start[e] <= r  (start ascending, start[0] = 0, start[n] = total > r)
__device__ __forceinline__ int upper_entry(const int* __restrict__ start, int n, int r) {
    int lo = 0, hi = n;  // invariant: start[lo] <= r < start[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (start[mid] <= r) lo = mid; else hi = mid;
    }
    return lo;
}

// ---- local-scene rows (PBNet.py:182-247) ---------------------------------------------------------------------------
// One thread per (row, 32-bit word of the output feature row).  Word w of a row of ESZ-byte elements:
//   elements [0, C)      : point_feat[p, :]
//   element  C           : sem_score[p, sem_pred[p]]      (PBNet.py:162-163: softmax score of the point's own class)
//   element  C + 1       : entry weight                   (PBNet.py:194,230)
//   elements [C+2, ld)   : 0
template <int ESZ>
__global__ __launch_bounds__(TPB) void k_local_scene_rows(
    const int* __restrict__ ent_row_start, const int* __restrict__ ent_member_start, const int* __restrict__ ent_scene,
    const float* __restrict__ ent_weight, int n_ent_cap, int n_rows_cap, const int* __restrict__ n_ent_dev,
    const int* __restrict__ n_rows_dev, const int* __restrict__ member_idx,
    const long long* __restrict__ ins_ind, const float* __restrict__ xyz, float inv_voxel,
    const unsigned char* __restrict__ point_feat, int ld_feat, int channels, const unsigned char* __restrict__ sem_score,
    int ld_sem, const long long* __restrict__ sem_pred, int dtype, long long* __restrict__ point_idx,
    long long* __restrict__ row_scene, int* __restrict__ coords, unsigned* __restrict__ feat_out, int words_per_row) {
    const int n_ent = n_ent_dev ? min(*n_ent_dev, n_ent_cap) : n_ent_cap;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_cap) : n_rows_cap;
    const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
    if (e >= (long long)n_rows * words_per_row) return;
    const int r = (int)(e / words_per_row), w = (int)(e - (long long)r * words_per_row);
    const int ent = upper_entry(ent_row_start, n_ent, r);
    const int local = member_idx[ent_member_start[ent] + (r - ent_row_start[ent])];
    const long long p = ins_ind[local];
    if (w == 0) {
        point_idx[r] = p;
        row_scene[r] = ent_scene[ent];
        // the reference divides a device tensor by a host scalar: that is a multiplication by the fp32 reciprocal
        const float x = xyz[3 * p + 0] * inv_voxel, y = xyz[3 * p + 1] * inv_voxel, z = xyz[3 * p + 2] * inv_voxel;
        reinterpret_cast<int4*>(coords)[r] = make_int4(ent_scene[ent], (int)floorf(x), (int)floorf(y), (int)floorf(z));
    }
    constexpr int EPW = 4 / ESZ;  // elements per word
    unsigned out = 0;
#pragma unroll
    for (int k = 0; k < EPW; ++k) {
        const int c = w * EPW + k;
        unsigned bits = 0;
        if (c < channels) {
            if (ESZ == 4) bits = *reinterpret_cast<const unsigned*>(point_feat + ((size_t)p * ld_feat + c) * 4);
            else bits = *reinterpret_cast<const unsigned short*>(point_feat + ((size_t)p * ld_feat + c) * 2);
        } else if (c == channels) {
            const size_t o = (size_t)p * ld_sem + (sem_pred ? (size_t)sem_pred[p] : 0);
            if (ESZ == 4) bits = *reinterpret_cast<const unsigned*>(sem_score + o * 4);
            else bits = *reinterpret_cast<const unsigned short*>(sem_score + o * 2);
        } else if (c == channels + 1) {
            const float wt = ent_weight[ent];
            if (ESZ == 4) bits = __float_as_uint(wt);
            else if (dtype == PBN_BF16) bits = __builtin_bit_cast(unsigned short, __float2bfloat16(wt));
            else bits = __builtin_bit_cast(unsigned short, __float2half(wt));
        }
        out |= bits << (8 * ESZ * k);
    }
    feat_out[e] = out;
}

// ---- out[i, :C] = in[idx[i], :C], out[i, C:ld_out] = 0  (32-bit words) -----------------------------------------------
__global__ __launch_bounds__(TPB) void k_gather_pad_rows(const unsigned* __restrict__ in, int ld_in_w, int row_w,
                                                        const long long* __restrict__ idx,
                                                        const long long* __restrict__ idx2, int n_cap,
                                                        const int* __restrict__ n_dev, unsigned* __restrict__ out,
                                                        int ld_out_w) {
    const int n = n_dev ? min(*n_dev, n_cap) : n_cap;
    const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
    if (e >= (long long)n * ld_out_w) return;
    const int i = (int)(e / ld_out_w), w = (int)(e - (long long)i * ld_out_w);
    unsigned v = 0;
    if (w < row_w) {
        long long r = idx ? idx[i] : i;
        if (idx2) r = idx2[r];
        v = in[(size_t)r * ld_in_w + w];
    }
    out[e] = v;
}


// ---- two-layer head: out = act(W2 . prelu(bn(W1 . x)) + b2) per row (PBNet.py:43-82 heads, eval mode) ----------------
// One thread per output row; the row is read through up to two index levels (row = idx_b[idx_a[i]]), so "gather the
// voxel features to the points" and "undo the Z-order" cost nothing extra.  All arithmetic in fp32, weights are uniform
// (scalar loads), one rounding at the end.
template <typename T> struct RowIO;
template <> struct RowIO<float> {
    static constexpr int E = 4;
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct RowIO<__hip_bfloat16> {
    static constexpr int E = 8;
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }
};
template <> struct RowIO<__half> {
    static constexpr int E = 8;
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 t = __half22float2(__builtin_bit_cast(__half2, w[i]));
            f[2 * i] = t.x; f[2 * i + 1] = t.y;
        }
    }
    static __device__ __forceinline__ void store(__half* p, float v) { *p = __float2half(v); }
};

template <typename T, int C, int H>
__global__ __launch_bounds__(TPB) void k_mlp_rows(const T* __restrict__ in, int ld_in, const long long* __restrict__ idx_a,
                                                 const long long* __restrict__ idx_b, int n_cap,
                                                 const int* __restrict__ n_dev, long long in_rows,
                                                 const float* __restrict__ w1, const float* __restrict__ scale,
                                                 const float* __restrict__ shift, const float* __restrict__ slope,
                                                 const float* __restrict__ w2, const float* __restrict__ b2, int n_out,
                                                 int sigmoid, T* __restrict__ out, int ld_out) {
    constexpr int E = RowIO<T>::E;
    const int n = n_dev ? min(*n_dev, n_cap) : n_cap;
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    long long row = idx_a ? idx_a[i] : i;
    if (idx_b && row >= 0) row = idx_b[row];
    float x[C];
    // a resolved row outside the slab (capacity-planned mode: a level overflowed its capacity and the index tables name
    // rows that were never computed) reads as zeros instead of memory past the slab; the overflow itself is flagged elsewhere
    const bool inside = row >= 0 && (in_rows < 0 || row < in_rows);
    const uint4* src = reinterpret_cast<const uint4*>(in + (size_t)(inside ? row : 0) * ld_in);
#pragma unroll
    for (int v = 0; v < C / E; ++v) {
        RowIO<T>::unpack(inside ? src[v] : make_uint4(0u, 0u, 0u, 0u), x + v * E);
    }
    float h[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) a = fmaf(w1[k * C + c], x[c], a);
        a = fmaf(a, scale[k], shift[k]);
        h[k] = a >= 0.f ? a : a * slope[k];
    }
    for (int o = 0; o < n_out; ++o) {
        float a = b2 ? b2[o] : 0.f;
#pragma unroll
        for (int k = 0; k < H; ++k) a = fmaf(w2[o * H + k], h[k], a);
        if (sigmoid) a = 1.0f / (1.0f + expf(-a));
        RowIO<T>::store(out + (size_t)i * ld_out + o, a);
    }
}

// The same heads on the matrix cores (round 4).  k_mlp_rows is latency-bound per wave: every thread streams the 1 024 + H n_out
// weights through scalar loads for 1 700 dependent FMAs (38 us for 161 k rows of a 32 -> 32 -> 32 head whose arithmetic is 3 us
// of the vector ALUs).  Here a wave takes 16 rows per round on v_mfma_f32_16x16x4_f32 (exact fp32: an fmaf chain), both layers
// TRANSPOSED so that no operand ever changes lanes:
//   layer 1:  D1[h][row] = sum_c W1[h][c] x[row][c]      A = W1 (lane (kq, h) holds w1[h][8 kq + kb] for step kb),
//                                                        B = x^T (lane (kq, row) holds the 8 channels 8 kq .. 8 kq + 7 of its row:
//                                                        ONE 16-byte load for 16-bit slabs); D1: lane (q, row) = hidden 4 q + i
//   layer 2:  D2[o][row] = sum_h W2[o][h] hid[row][h]    B = the D1 registers as they are (step (t, i): hidden 16 t + 4 kq + i),
//                                                        A = W2 indexed accordingly; D2: lane (q, row) = outputs 4 q + i of its row
// Weights, BatchNorm constants and slopes live in registers for the wave's whole loop over row blocks.  Summation order differs from
// k_mlp_rows (channels 0, 8, 16, 24, 1, 9, ...): fp32 results agree to rounding (tests: 1e-5).
typedef float mlp_f32x4 __attribute__((ext_vector_type(4)));
template <typename T, int H, int U>
__global__ __launch_bounds__(TPB) void k_mlp_rows_mfma(const T* __restrict__ in, int ld_in, const long long* __restrict__ idx_a,
                                                      const long long* __restrict__ idx_b, int n_cap,
                                                      const int* __restrict__ n_dev, long long in_rows,
                                                      const float* __restrict__ w1, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, const float* __restrict__ slope,
                                                      const float* __restrict__ w2, const float* __restrict__ b2, int n_out,
                                                      int sigmoid, T* __restrict__ out, int ld_out) {
    constexpr int C = 32, HT = H / 16;
    const int n = n_dev ? min(*n_dev, n_cap) : n_cap;
    const int lane = threadIdx.x & 63, kq = lane >> 4, r = lane & 15;
    const int wave = (int)((blockIdx.x * TPB + threadIdx.x) >> 6), n_waves = (int)((gridDim.x * TPB) >> 6);
    // ---- per-wave constants ----
    float a1[HT][8];                 // layer-1 A operands: w1[16 t + r][8 kq + kb]
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) a1[t][kb] = w1[(16 * t + r) * C + 8 * kq + kb];
    float sc[HT][4], sh[HT][4], sl[HT][4];      // of hidden 16 t + 4 kq + i (this lane's D1 registers)
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hh = 16 * t + 4 * kq + i;
            sc[t][i] = scale[hh]; sh[t][i] = shift[hh]; sl[t][i] = slope[hh];
        }
    float a2[U][HT][4];              // layer-2 A operands: w2[16 u + r][16 t + 4 kq + i]
    float bias[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int o = 16 * u + r;
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) a2[u][t][i] = o < n_out ? w2[o * H + 16 * t + 4 * kq + i] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int oo = 16 * u + 4 * kq + i;
            bias[u][i] = (b2 && oo < n_out) ? b2[oo] : 0.0f;
        }
    }
    for (int blk = wave; blk * 16 < n; blk += n_waves) {
        const int i_row = blk * 16 + r;
        long long row = -1;
        if (i_row < n) {
            row = idx_a ? idx_a[i_row] : i_row;
            if (idx_b && row >= 0) row = idx_b[row];
        }
        const bool inside = row >= 0 && (in_rows < 0 || row < in_rows);
        float x[8];
        {
            const unsigned char* src = reinterpret_cast<const unsigned char*>(in + (size_t)(inside ? row : 0) * ld_in) + (size_t)kq * 8 * sizeof(T);
            if constexpr (sizeof(T) == 4) {
                const uint4 v0 = inside ? reinterpret_cast<const uint4*>(src)[0] : make_uint4(0u, 0u, 0u, 0u);
                const uint4 v1 = inside ? reinterpret_cast<const uint4*>(src)[1] : make_uint4(0u, 0u, 0u, 0u);
                RowIO<T>::unpack(v0, x);
                RowIO<T>::unpack(v1, x + 4);
            } else {
                RowIO<T>::unpack(inside ? reinterpret_cast<const uint4*>(src)[0] : make_uint4(0u, 0u, 0u, 0u), x);
            }
        }
        mlp_f32x4 d1[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            d1[t] = mlp_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) d1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t][kb], x[kb], d1[t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = fmaf(d1[t][i], sc[t][i], sh[t][i]);
                d1[t][i] = v >= 0.f ? v : v * sl[t][i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mlp_f32x4 d2 = mlp_f32x4{bias[u][0], bias[u][1], bias[u][2], bias[u][3]};
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[u][t][i], d1[t][i], d2, 0, 0, 0);
            if (i_row < n) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int oo = 16 * u + 4 * kq + i;
                    if (oo < n_out) {
                        float v = d2[i];
                        if (sigmoid) v = 1.0f / (1.0f + expf(-v));
                        RowIO<T>::store(out + (size_t)i_row * ld_out + oo, v);
                    }
                }
            }
        }
    }
}

// ---- semantic argmax + own-class softmax score + [class, batch] population table (PBNet.py:134,151-163) -------------
// Block = SEL_BLOCK consecutive points.  Besides the global table it leaves the per-block class histogram that
// k_select_points turns into stable output positions.
constexpr int SEL_BLOCK = 1024;
constexpr int SEL_MAX_CLASSES = 32;

template <typename T>
__global__ __launch_bounds__(TPB) void k_sem_argmax_table(const T* __restrict__ score, int ld, int n_cls,
                                                         const int* __restrict__ batch, int nb, int n,
                                                         long long* __restrict__ sem_pred, T* __restrict__ sem_prob,
                                                         int* __restrict__ table, int* __restrict__ block_hist) {
    __shared__ int s_tab[SEL_MAX_CLASSES * 8];
    __shared__ int s_cls[SEL_MAX_CLASSES];
    for (int e = threadIdx.x; e < n_cls * nb; e += TPB) s_tab[e] = 0;
    if (threadIdx.x < n_cls) s_cls[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * SEL_BLOCK;
    for (int k = 0; k < SEL_BLOCK / TPB; ++k) {
        const int i = base + k * TPB + threadIdx.x;
        if (i >= n) break;
        const T* s = score + (size_t)i * ld;
        float best = -__builtin_inff();
        int arg = 0;
        for (int c = 0; c < n_cls; ++c) {
            float v;
            if constexpr (sizeof(T) == 4) v = s[c];
            else if constexpr (__is_same(T, __hip_bfloat16)) v = __bfloat162float(s[c]);
            else v = __half2float(s[c]);
            if (v > best) { best = v; arg = c; }       // first maximum wins
        }
        float sum = 0.f;
        for (int c = 0; c < n_cls; ++c) {
            float v;
            if constexpr (sizeof(T) == 4) v = s[c];
            else if constexpr (__is_same(T, __hip_bfloat16)) v = __bfloat162float(s[c]);
            else v = __half2float(s[c]);
            sum += expf(v - best);
        }
        sem_pred[i] = arg;
        if (sem_prob) RowIO<T>::store(sem_prob + i, 1.0f / sum);
        atomicAdd(&s_cls[arg], 1);
        const int b = batch ? batch[i] : 0;
        if (b >= 0 && b < nb) atomicAdd(&s_tab[arg * nb + b], 1);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n_cls * nb; e += TPB)
        if (s_tab[e]) atomicAdd(&table[e], s_tab[e]);
    if (threadIdx.x < n_cls) block_hist[blockIdx.x * n_cls + threadIdx.x] = s_cls[threadIdx.x];
}

// ---- stable class-major selection of the points of the kept classes + their grouping inputs (PBNet.py:151-170) -------
// Output position of point i of class c = class_base[c] + #(points j < i of class c): blocks in front come from the
// per-block histograms, waves in front from a per-wave histogram, lanes in front from ballots.  Deterministic.
template <typename T>
__global__ __launch_bounds__(TPB) void k_select_points(const long long* __restrict__ sem_pred, int n, int n_cls,
                                                      const int* __restrict__ class_base,
                                                      const int* __restrict__ block_hist, const float* __restrict__ xyz,
                                                      const T* __restrict__ offset, int ld_off,
                                                      long long* __restrict__ ins_ind, float* __restrict__ ins_orig,
                                                      float* __restrict__ ins_off, int* __restrict__ ins_sem) {
    __shared__ int s_base[SEL_MAX_CLASSES];
    __shared__ int s_wave[4][SEL_MAX_CLASSES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < n_cls) {
        int t = class_base[threadIdx.x];
        if (t >= 0)
            for (int b = 0; b < (int)blockIdx.x; ++b) t += block_hist[b * n_cls + threadIdx.x];
        else
            t = -(1 << 30);                           // dropped class: stays negative whatever is added to it
        s_base[threadIdx.x] = t;
    }
    for (int e = threadIdx.x; e < 4 * SEL_MAX_CLASSES; e += TPB) (&s_wave[0][0])[e] = 0;
    __syncthreads();
    const int wbase = blockIdx.x * SEL_BLOCK + wave * (SEL_BLOCK / 4);
    int cls[SEL_BLOCK / 256];
#pragma unroll
    for (int k = 0; k < SEL_BLOCK / 256; ++k) {
        const int i = wbase + k * 64 + lane;
        cls[k] = i < n ? (int)sem_pred[i] : -1;
        if (cls[k] >= 0) atomicAdd(&s_wave[wave][cls[k]], 1);
    }
    __syncthreads();
    // running position per class for this wave (lane c owns class c)
    int run = 0;
    if (lane < n_cls) {
        run = s_base[lane];
        for (int w = 0; w < wave; ++w) run += s_wave[w][lane];
    }
#pragma unroll
    for (int k = 0; k < SEL_BLOCK / 256; ++k) {
        const int i = wbase + k * 64 + lane;
        const int c = cls[k];
        int pos = -1;
        unsigned long long todo = __ballot(c >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int lc = __shfl(c, leader, 64);
            const unsigned long long m = __ballot(c == lc);
            const int start = __shfl(run, lc, 64);
            if (c == lc) pos = start >= 0 ? start + __popcll(m & ((1ULL << lane) - 1ULL)) : -1;
            if (lane == lc) run += (run >= 0) ? __popcll(m) : 0;
            todo &= ~m;
        }
        if (pos >= 0) {
            const float x = xyz[3 * (size_t)i + 0], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
            float ox, oy, oz;
            const T* o = offset + (size_t)i * ld_off;
            if constexpr (sizeof(T) == 4) { ox = o[0]; oy = o[1]; oz = o[2]; }
            else if constexpr (__is_same(T, __hip_bfloat16)) { ox = __bfloat162float(o[0]); oy = __bfloat162float(o[1]); oz = __bfloat162float(o[2]); }
            else { ox = __half2float(o[0]); oy = __half2float(o[1]); oz = __half2float(o[2]); }
            ins_ind[pos] = i;
            ins_orig[3 * (size_t)pos + 0] = x; ins_orig[3 * (size_t)pos + 1] = y; ins_orig[3 * (size_t)pos + 2] = z;
            ins_off[3 * (size_t)pos + 0] = x + ox; ins_off[3 * (size_t)pos + 1] = y + oy; ins_off[3 * (size_t)pos + 2] = z + oz;
            ins_sem[pos] = c;
        }
    }
}


// ---- proposals (PBNet.py:317-347 + 240-252): rows whose mask score passes the threshold, order preserved ------------
template <typename T> __device__ __forceinline__ float ld_f32(const T* p);
template <> __device__ __forceinline__ float ld_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_f32<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }
template <> __device__ __forceinline__ float ld_f32<__half>(const __half* p) { return __half2float(*p); }

// pass 1: kept rows per block of SEL_BLOCK rows and per local scene (wave-aggregated atomics: rows are grouped by scene)
template <typename T>
__global__ __launch_bounds__(TPB) void k_mask_count(const T* __restrict__ score, int ld, float thd,
                                                   const long long* __restrict__ row_scene, int n_cap,
                                                   const int* __restrict__ n_dev, int n_scenes,
                                                   int* __restrict__ per_scene, int* __restrict__ block_cnt) {
    const int n = n_dev ? min(*n_dev, n_cap) : n_cap;
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int base = blockIdx.x * SEL_BLOCK;
    int mine = 0;
    for (int k = 0; k < SEL_BLOCK / TPB; ++k) {
        const int i = base + k * TPB + threadIdx.x;
        const bool keep = i < n && ld_f32<T>(score + (size_t)i * ld) > thd;
        const int sc = keep ? (int)row_scene[i] : -1;
        unsigned long long todo = __ballot(keep);
        mine += keep ? 1 : 0;
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int ls = __shfl(sc, leader, 64);
            const unsigned long long m = __ballot(sc == ls);
            if (lane == leader && ls >= 0 && ls < n_scenes) atomicAdd(&per_scene[ls], __popcll(m));
            todo &= ~m;
        }
    }
    mine = wave_reduce_add(mine);
    if (lane == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_cnt;
}

// pass 2: compaction.  Thread t of a block owns SEL_BLOCK/TPB CONSECUTIVE rows, so positions follow the row order.
template <typename T>
__global__ __launch_bounds__(TPB) void k_proposal_rows(const T* __restrict__ score, int ld, float thd,
                                                      const long long* __restrict__ row_scene,
                                                      const long long* __restrict__ point_idx, int n_cap,
                                                      const int* __restrict__ n_dev,
                                                      const int* __restrict__ dense_of, const int* __restrict__ block_cnt,
                                                      const float* __restrict__ xyz, float scale, float inv_voxel,
                                                      const uint4* __restrict__ point_feat, int ld_feat_vec, int vpr,
                                                      long long* __restrict__ prop_idx, T* __restrict__ prop_ms,
                                                      int* __restrict__ coords, uint4* __restrict__ feat) {
    constexpr int PER = SEL_BLOCK / TPB;
    const int n = n_dev ? min(*n_dev, n_cap) : n_cap;
    __shared__ int wtot[TPB / 64];
    __shared__ int s_base;
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += TPB) part += block_cnt[b];
    part = wave_reduce_add(part);
    if ((threadIdx.x & 63) == 0) wtot[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) s_base = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();
    const int block_base = s_base;
    const int r0 = blockIdx.x * SEL_BLOCK + threadIdx.x * PER;
    bool keep[PER];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = r0 + k;
        keep[k] = i < n && ld_f32<T>(score + (size_t)i * ld) > thd;
        cnt += keep[k] ? 1 : 0;
    }
    // exclusive prefix of cnt over the block
    int incl = cnt;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) wtot[threadIdx.x >> 6] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wtot[w];
    int pos = block_base + woff + incl - cnt;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (!keep[k]) continue;
        const int i = r0 + k;
        const long long p = point_idx[i];
        const int dense = dense_of[(int)row_scene[i]];
        prop_idx[2 * (size_t)pos + 0] = dense;
        prop_idx[2 * (size_t)pos + 1] = p;
        prop_ms[pos] = score[(size_t)i * ld];
        if (coords) {
            // (xyz * scale) / voxel with device-tensor-by-host-scalar semantics: two fp32 multiplications
            const float x = (xyz[3 * p + 0] * scale) * inv_voxel, y = (xyz[3 * p + 1] * scale) * inv_voxel,
                        z = (xyz[3 * p + 2] * scale) * inv_voxel;
            reinterpret_cast<int4*>(coords)[pos] = make_int4(dense, (int)floorf(x), (int)floorf(y), (int)floorf(z));
        }
        if (feat)
            for (int v = 0; v < vpr; ++v) feat[(size_t)pos * vpr + v] = point_feat[(size_t)p * ld_feat_vec + v];
        ++pos;
    }
}


// ---- wgrad operand: out[v, j, :] = in[nbr[v, k0 + j], :] (zeros where there is no neighbour), 32-bit words ----------
__global__ __launch_bounds__(TPB) void k_gather_rulebook_rows(const unsigned* __restrict__ in, int ld_in_w, int row_w,
                                                             const int* __restrict__ nbr, int K, int k0, int kc, int n,
                                                             unsigned* __restrict__ out) {
    const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
    const long long per_row = (long long)kc * row_w;
    if (e >= (long long)n * per_row) return;
    const int v = (int)(e / per_row);
    const int r = (int)(e - (long long)v * per_row);
    const int j = r / row_w, w = r - j * row_w;
    const int src = nbr[(size_t)v * K + k0 + j];
    out[e] = src >= 0 ? in[(size_t)src * ld_in_w + w] : 0u;
}


// ---- weight packing: fp32 master [K, A, B] -> MFMA-fragment order (include/pbnet_hip.h, pbn_spconv_forward) ----------
// element (k, ci, co) of the convolution = src[flip ? K-1-k : k][ci][co], or [..][co][ci] when `transpose` (the dgrad
// weights of a forward kernel).  One thread per packed element; E = elements per 16-byte vector.
template <typename T>
__device__ __forceinline__ void pack_weight_element(const float* __restrict__ src, int K, int A, int B, int flip, int transpose,
                                                    int cin, int cout, int cin_p, int cout_p, long long e, T* __restrict__ out) {
    constexpr int E = 16 / (int)sizeof(T);
    const int ntt = cout_p / 16;
    const int j = (int)(e % E);
    const int lane = (int)((e / E) % 64);
    const int tt = (int)((e / (E * 64)) % ntt);
    const int st = (int)(e / ((long long)E * 64 * ntt));
    const int g = lane >> 4, c = lane & 15;
    const int r = st * 4 * E + g * E + j;           // row of the flattened (offset, channel) axis
    const int k = r / cin_p, ci = r - k * cin_p;
    const int co = tt * 16 + c;
    float v = 0.f;
    if (k < K && ci < cin && co < cout) {
        const int ks = flip ? K - 1 - k : k;
        v = transpose ? src[((size_t)ks * A + co) * B + ci] : src[((size_t)ks * A + ci) * B + co];
    }
    RowIO<T>::store(out + e, v);
}

template <typename T>
__global__ __launch_bounds__(TPB) void k_pack_weight(const float* __restrict__ src, int K, int A, int B, int flip,
                                                    int transpose, int cin, int cout, int cin_p, int cout_p, int n_steps,
                                                    T* __restrict__ out) {
    constexpr int E = 16 / (int)sizeof(T);
    const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
    const long long total = (long long)n_steps * (cout_p / 16) * 64 * E;
    if (e >= total) return;
    pack_weight_element<T>(src, K, A, B, flip, transpose, cin, cout, cin_p, cout_p, e, out);
}

// every layer's weights (both the forward and the input-gradient form) in ONE launch: blockIdx.y = job of a device table
template <typename T>
__global__ __launch_bounds__(TPB) void k_pack_weights_batch(const pbn_pack_job* __restrict__ jobs) {
    constexpr int E = 16 / (int)sizeof(T);
    const pbn_pack_job jb = jobs[blockIdx.y];
    const long long total = (long long)jb.n_steps * (jb.cout_p / 16) * 64 * E;
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB)
        pack_weight_element<T>(jb.src, jb.n_offsets, jb.dim_a, jb.dim_b, jb.flip, jb.transpose, jb.cin, jb.cout, jb.cin_p,
                               jb.cout_p, e, (T*)jb.out);
}

}  // namespace
}  // namespace pbn

using namespace pbn;

static int local_scene_rows_impl(const int32_t* ent_row_start, const int32_t* ent_member_start,
                                    const int32_t* ent_scene, const float* ent_weight, int n_ent, int n_rows,
                                    const int32_t* n_ent_dev, const int32_t* n_rows_dev,
                                    const int32_t* member_idx, const int64_t* ins_ind, const float* xyz, float inv_voxel,
                                    const void* point_feat, int ld_feat, int channels, const void* sem_score, int ld_sem,
                                    const int64_t* sem_pred, int dtype, int64_t* point_idx, int64_t* row_scene,
                                    int32_t* coords, void* feat_out, int ld_out, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_ent < 0 || n_rows < 0 || channels < 0 || ld_out < channels + 2 || ld_feat < channels) return PBN_ERR_ARG;
    if (n_rows == 0) return PBN_OK;
    if (n_ent == 0 || !ent_row_start || !ent_member_start || !ent_scene || !ent_weight || !member_idx || !ins_ind || !xyz ||
        !point_feat || !sem_score || !point_idx || !row_scene || !coords || !feat_out)
        return PBN_ERR_ARG;
    const int esz = dtype == PBN_F32 ? 4 : 2;
    if ((ld_out * esz) % 4 || ((uintptr_t)feat_out & 3) || ((uintptr_t)coords & 15)) return PBN_ERR_ARG;
    const int wpr = ld_out * esz / 4;
    const long long total = (long long)n_rows * wpr;
    const dim3 grid(cdiv(total, TPB));
#define PBN_ARGS                                                                                                        \
    ent_row_start, ent_member_start, ent_scene, ent_weight, n_ent, n_rows, n_ent_dev, n_rows_dev, member_idx,            \
        (const long long*)ins_ind, xyz,                                                                                  \
        inv_voxel, (const unsigned char*)point_feat, ld_feat, channels, (const unsigned char*)sem_score, ld_sem,         \
        (const long long*)sem_pred, dtype, (long long*)point_idx, (long long*)row_scene, coords, (unsigned*)feat_out, wpr
    if (esz == 4) hipLaunchKernelGGL(k_local_scene_rows<4>, grid, dim3(TPB), 0, stream, PBN_ARGS);
    else hipLaunchKernelGGL(k_local_scene_rows<2>, grid, dim3(TPB), 0, stream, PBN_ARGS);
#undef PBN_ARGS
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_local_scene_rows(const int32_t* ent_row_start, const int32_t* ent_member_start,
                                    const int32_t* ent_scene, const float* ent_weight, int n_ent, int n_rows,
                                    const int32_t* member_idx, const int64_t* ins_ind, const float* xyz, float inv_voxel,
                                    const void* point_feat, int ld_feat, int channels, const void* sem_score, int ld_sem,
                                    const int64_t* sem_pred, int dtype, int64_t* point_idx, int64_t* row_scene,
                                    int32_t* coords, void* feat_out, int ld_out, pbn_stream_t stream) {
    return local_scene_rows_impl(ent_row_start, ent_member_start, ent_scene, ent_weight, n_ent, n_rows, nullptr, nullptr,
                                 member_idx, ins_ind, xyz, inv_voxel, point_feat, ld_feat, channels, sem_score, ld_sem,
                                 sem_pred, dtype, point_idx, row_scene, coords, feat_out, ld_out, stream);
}

extern "C" int pbn_local_scene_rows_dev(const int32_t* ent_row_start, const int32_t* ent_member_start,
                                        const int32_t* ent_scene, const float* ent_weight, int n_ent_cap, int n_rows_cap,
                                        const int32_t* n_ent_dev, const int32_t* n_rows_dev, const int32_t* member_idx,
                                        const int64_t* ins_ind, const float* xyz, float inv_voxel, const void* point_feat,
                                        int ld_feat, int channels, const void* sem_score, int ld_sem,
                                        const int64_t* sem_pred, int dtype, int64_t* point_idx, int64_t* row_scene,
                                        int32_t* coords, void* feat_out, int ld_out, pbn_stream_t stream) {
    if (!n_ent_dev || !n_rows_dev) return PBN_ERR_ARG;
    return local_scene_rows_impl(ent_row_start, ent_member_start, ent_scene, ent_weight, n_ent_cap, n_rows_cap, n_ent_dev,
                                 n_rows_dev, member_idx, ins_ind, xyz, inv_voxel, point_feat, ld_feat, channels, sem_score,
                                 ld_sem, sem_pred, dtype, point_idx, row_scene, coords, feat_out, ld_out, stream);
}

static int gather_pad_rows_impl(const void* in, int ld_in_bytes, int row_bytes, const int64_t* idx, const int64_t* idx2,
                                int n, const int32_t* n_dev, void* out, int ld_out_bytes, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || row_bytes <= 0 || (row_bytes & 3) || (ld_in_bytes & 3) || (ld_out_bytes & 3) || ld_out_bytes < row_bytes ||
        ld_in_bytes < row_bytes)
        return PBN_ERR_ARG;
    if (n == 0) return PBN_OK;
    if (!in || !out || (((uintptr_t)in | (uintptr_t)out) & 3)) return PBN_ERR_ARG;
    const long long total = (long long)n * (ld_out_bytes / 4);
    hipLaunchKernelGGL(k_gather_pad_rows, dim3(cdiv(total, TPB)), dim3(TPB), 0, stream, (const unsigned*)in, ld_in_bytes / 4,
                       row_bytes / 4, (const long long*)idx, (const long long*)idx2, n, n_dev, (unsigned*)out,
                       ld_out_bytes / 4);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_gather_pad_rows(const void* in, int ld_in_bytes, int row_bytes, const int64_t* idx, int n, void* out,
                                   int ld_out_bytes, pbn_stream_t stream) {
    return gather_pad_rows_impl(in, ld_in_bytes, row_bytes, idx, nullptr, n, nullptr, out, ld_out_bytes, stream);
}

extern "C" int pbn_gather_pad_rows_dev(const void* in, int ld_in_bytes, int row_bytes, const int64_t* idx,
                                       const int64_t* idx2, int n_cap, const int32_t* n_dev, void* out, int ld_out_bytes,
                                       pbn_stream_t stream) {
    return gather_pad_rows_impl(in, ld_in_bytes, row_bytes, idx, idx2, n_cap, n_dev, out, ld_out_bytes, stream);
}

static int mlp_rows_impl(const void* in, int ld_in, long long in_rows, int channels, const int64_t* idx_a, const int64_t* idx_b, int n,
                            const int32_t* n_dev, const float* w1, const float* scale, const float* shift, const float* slope, int hidden,
                            const float* w2, const float* b2, int n_out, int sigmoid, void* out, int ld_out, int dtype,
                            pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || n_out < 1 || ld_out < n_out || ld_in < channels) return PBN_ERR_ARG;
    if (channels != 32 || (hidden != 16 && hidden != 32)) return PBN_ERR_UNSUPPORTED;
    if (n == 0) return PBN_OK;
    if (!in || !w1 || !scale || !shift || !slope || !w2 || !out) return PBN_ERR_ARG;
    const int esz = dtype == PBN_F32 ? 4 : 2;
    if ((ld_in * esz) % 16 || ((uintptr_t)in & 15)) return PBN_ERR_ARG;
    // matrix-core form (default; PBN_MLP_FORM=0: the scalar kernel, kept as the cross-check): 16 rows per wave and round
    static const int form_env = getenv("PBN_MLP_FORM") ? atoi(getenv("PBN_MLP_FORM")) : 1;
    if (form_env != 0 && n_out <= 32) {
        const long long blocks16 = cdiv(n, 16);
        // ~4 row blocks per wave: the ~60 register-resident weight words of a wave pay off (scripts/probe_heads.py, HIP-graph replay,
        // 161 517 gathered rows, bf16: 32->32->32 35.3 -> 24.4 us, 32->16->20 16.8 -> 12.7, the 3- and 1-output heads 8.6 either way;
        // one block per wave: no gain at all).  PBN_MLP_BLOCKS: row blocks per wave (measurement)
        static const int bpw_env = getenv("PBN_MLP_BLOCKS") ? atoi(getenv("PBN_MLP_BLOCKS")) : 4;
        long long wgs = cdiv(blocks16, (TPB / 64) * (bpw_env > 0 ? bpw_env : 1));
        if (wgs > 8192) wgs = 8192;
        if (wgs < 1) wgs = 1;
        const dim3 g((unsigned)wgs);
#define PBN_MLPM(TT, HH, UU)                                                                                            \
        hipLaunchKernelGGL((k_mlp_rows_mfma<TT, HH, UU>), g, dim3(TPB), 0, stream, (const TT*)in, ld_in,                 \
                           (const long long*)idx_a, (const long long*)idx_b, n, n_dev, in_rows, w1, scale, shift, slope, w2, b2, \
                           n_out, sigmoid, (TT*)out, ld_out)
#define PBN_MLPM_T(TT)                                                                                                  \
        { if (hidden == 16) { if (n_out <= 16) PBN_MLPM(TT, 16, 1); else PBN_MLPM(TT, 16, 2); }                          \
          else { if (n_out <= 16) PBN_MLPM(TT, 32, 1); else PBN_MLPM(TT, 32, 2); } }
        if (dtype == PBN_F32) PBN_MLPM_T(float)
        else if (dtype == PBN_BF16) PBN_MLPM_T(__hip_bfloat16)
        else if (dtype == PBN_F16) PBN_MLPM_T(__half)
        else return PBN_ERR_ARG;
#undef PBN_MLPM_T
#undef PBN_MLPM
        PBN_LAUNCH_CHECK();
        return PBN_OK;
    }
    const dim3 grid(cdiv(n, TPB));
#define PBN_MLP(TT, HH)                                                                                                 \
    hipLaunchKernelGGL((k_mlp_rows<TT, 32, HH>), grid, dim3(TPB), 0, stream, (const TT*)in, ld_in,                        \
                       (const long long*)idx_a, (const long long*)idx_b, n, n_dev, in_rows, w1, scale, shift, slope, w2, b2, \
                       n_out,                                                                                            \
                       sigmoid, (TT*)out, ld_out)
    if (dtype == PBN_F32) { if (hidden == 16) PBN_MLP(float, 16); else PBN_MLP(float, 32); }
    else if (dtype == PBN_BF16) { if (hidden == 16) PBN_MLP(__hip_bfloat16, 16); else PBN_MLP(__hip_bfloat16, 32); }
    else if (dtype == PBN_F16) { if (hidden == 16) PBN_MLP(__half, 16); else PBN_MLP(__half, 32); }
    else return PBN_ERR_ARG;
#undef PBN_MLP
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_mlp_rows(const void* in, int ld_in, int channels, const int64_t* idx_a, const int64_t* idx_b, int n,
                            const float* w1, const float* scale, const float* shift, const float* slope, int hidden,
                            const float* w2, const float* b2, int n_out, int sigmoid, void* out, int ld_out, int dtype,
                            pbn_stream_t stream) {
    return mlp_rows_impl(in, ld_in, -1, channels, idx_a, idx_b, n, nullptr, w1, scale, shift, slope, hidden, w2, b2, n_out,
                         sigmoid, out, ld_out, dtype, stream);
}

extern "C" int pbn_mlp_rows_dev(const void* in, int ld_in, int in_rows, int channels, const int64_t* idx_a, const int64_t* idx_b,
                                int n_cap, const int32_t* n_dev, const float* w1, const float* scale, const float* shift,
                                const float* slope, int hidden, const float* w2, const float* b2, int n_out, int sigmoid,
                                void* out, int ld_out, int dtype, pbn_stream_t stream) {
    if (in_rows < 0) return PBN_ERR_ARG;
    return mlp_rows_impl(in, ld_in, in_rows, channels, idx_a, idx_b, n_cap, n_dev, w1, scale, shift, slope, hidden, w2, b2, n_out,
                         sigmoid, out, ld_out, dtype, stream);
}

extern "C" int pbn_select_blocks(int n) { return n > 0 ? cdiv(n, SEL_BLOCK) : 0; }

extern "C" int pbn_sem_argmax_table(const void* score, int ld, int n_cls, const int32_t* batch, int nb, int n, int dtype,
                                    int64_t* sem_pred, void* sem_prob, int32_t* table, int32_t* block_hist,
                                    pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || n_cls < 1 || n_cls > SEL_MAX_CLASSES || nb < 1 || nb > 8 || ld < n_cls || !table) return PBN_ERR_ARG;
    { const int frc_ = fill_bytes(table, 0, sizeof(int) * (size_t)n_cls * nb, stream); if (frc_ != PBN_OK) return frc_; }
    if (n == 0) return PBN_OK;
    if (!score || !sem_pred || !block_hist) return PBN_ERR_ARG;
    const dim3 grid(cdiv(n, SEL_BLOCK));
    if (dtype == PBN_F32)
        hipLaunchKernelGGL(k_sem_argmax_table<float>, grid, dim3(TPB), 0, stream, (const float*)score, ld, n_cls, batch, nb,
                           n, (long long*)sem_pred, (float*)sem_prob, table, block_hist);
    else if (dtype == PBN_BF16)
        hipLaunchKernelGGL(k_sem_argmax_table<__hip_bfloat16>, grid, dim3(TPB), 0, stream, (const __hip_bfloat16*)score, ld,
                           n_cls, batch, nb, n, (long long*)sem_pred, (__hip_bfloat16*)sem_prob, table, block_hist);
    else if (dtype == PBN_F16)
        hipLaunchKernelGGL(k_sem_argmax_table<__half>, grid, dim3(TPB), 0, stream, (const __half*)score, ld, n_cls, batch,
                           nb, n, (long long*)sem_pred, (__half*)sem_prob, table, block_hist);
    else
        return PBN_ERR_ARG;
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_select_points(const int64_t* sem_pred, int n, int n_cls, const int32_t* class_base,
                                 const int32_t* block_hist, const float* xyz, const void* offset, int ld_off, int dtype,
                                 int64_t* ins_ind, float* ins_orig, float* ins_off, int32_t* ins_sem,
                                 pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || n_cls < 1 || n_cls > SEL_MAX_CLASSES || ld_off < 3) return PBN_ERR_ARG;
    if (n == 0) return PBN_OK;
    if (!sem_pred || !class_base || !block_hist || !xyz || !offset || !ins_ind || !ins_orig || !ins_off || !ins_sem)
        return PBN_ERR_ARG;
    const dim3 grid(cdiv(n, SEL_BLOCK));
    if (dtype == PBN_F32)
        hipLaunchKernelGGL(k_select_points<float>, grid, dim3(TPB), 0, stream, (const long long*)sem_pred, n, n_cls,
                           class_base, block_hist, xyz, (const float*)offset, ld_off, (long long*)ins_ind, ins_orig, ins_off,
                           ins_sem);
    else if (dtype == PBN_BF16)
        hipLaunchKernelGGL(k_select_points<__hip_bfloat16>, grid, dim3(TPB), 0, stream, (const long long*)sem_pred, n, n_cls,
                           class_base, block_hist, xyz, (const __hip_bfloat16*)offset, ld_off, (long long*)ins_ind, ins_orig,
                           ins_off, ins_sem);
    else if (dtype == PBN_F16)
        hipLaunchKernelGGL(k_select_points<__half>, grid, dim3(TPB), 0, stream, (const long long*)sem_pred, n, n_cls,
                           class_base, block_hist, xyz, (const __half*)offset, ld_off, (long long*)ins_ind, ins_orig, ins_off,
                           ins_sem);
    else
        return PBN_ERR_ARG;
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

static int mask_count_impl(const void* mask_score, int ld, float thd, const int64_t* row_scene, int n,
                           const int32_t* n_dev, int n_scenes, int dtype, int32_t* per_scene, int32_t* block_cnt,
                           pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || n_scenes < 0 || ld < 1) return PBN_ERR_ARG;
    if (n_scenes > 0) {
        if (!per_scene) return PBN_ERR_ARG;
        { const int frc_ = fill_bytes(per_scene, 0, sizeof(int) * (size_t)n_scenes, stream); if (frc_ != PBN_OK) return frc_; }
    }
    if (n == 0) return PBN_OK;
    if (!mask_score || !row_scene || !block_cnt) return PBN_ERR_ARG;
    const dim3 grid(cdiv(n, SEL_BLOCK));
    if (dtype == PBN_F32)
        hipLaunchKernelGGL(k_mask_count<float>, grid, dim3(TPB), 0, stream, (const float*)mask_score, ld, thd,
                           (const long long*)row_scene, n, n_dev, n_scenes, per_scene, block_cnt);
    else if (dtype == PBN_BF16)
        hipLaunchKernelGGL(k_mask_count<__hip_bfloat16>, grid, dim3(TPB), 0, stream, (const __hip_bfloat16*)mask_score, ld,
                           thd, (const long long*)row_scene, n, n_dev, n_scenes, per_scene, block_cnt);
    else if (dtype == PBN_F16)
        hipLaunchKernelGGL(k_mask_count<__half>, grid, dim3(TPB), 0, stream, (const __half*)mask_score, ld, thd,
                           (const long long*)row_scene, n, n_dev, n_scenes, per_scene, block_cnt);
    else
        return PBN_ERR_ARG;
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_mask_count(const void* mask_score, int ld, float thd, const int64_t* row_scene, int n, int n_scenes,
                              int dtype, int32_t* per_scene, int32_t* block_cnt, pbn_stream_t stream) {
    return mask_count_impl(mask_score, ld, thd, row_scene, n, nullptr, n_scenes, dtype, per_scene, block_cnt, stream);
}

extern "C" int pbn_mask_count_dev(const void* mask_score, int ld, float thd, const int64_t* row_scene, int n_cap,
                                  const int32_t* n_dev, int n_scenes_cap, int dtype, int32_t* per_scene, int32_t* block_cnt,
                                  pbn_stream_t stream) {
    return mask_count_impl(mask_score, ld, thd, row_scene, n_cap, n_dev, n_scenes_cap, dtype, per_scene, block_cnt, stream);
}

static int proposal_rows_impl(const void* mask_score, int ld, float thd, const int64_t* row_scene,
                                 const int64_t* point_idx, int n, const int32_t* n_dev, const int32_t* dense_of,
                                 const int32_t* block_cnt,
                                 const float* xyz, float scale, float inv_voxel, const void* point_feat, int ld_feat,
                                 int channels, int dtype, int64_t* proposals_idx, void* proposals_ms, int32_t* coords,
                                 void* feat_out, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || ld < 1) return PBN_ERR_ARG;
    if (n == 0) return PBN_OK;
    if (!mask_score || !row_scene || !point_idx || !dense_of || !block_cnt || !proposals_idx || !proposals_ms)
        return PBN_ERR_ARG;
    const int esz = dtype == PBN_F32 ? 4 : 2;
    int vpr = 0;
    if (feat_out) {
        if (!point_feat || channels < 1 || (channels * esz) % 16 || (ld_feat * esz) % 16 ||
            (((uintptr_t)point_feat | (uintptr_t)feat_out) & 15))
            return PBN_ERR_ARG;
        vpr = channels * esz / 16;
    }
    if (coords && (!xyz || ((uintptr_t)coords & 15))) return PBN_ERR_ARG;
    const dim3 grid(cdiv(n, SEL_BLOCK));
#define PBN_PR(TT)                                                                                                      \
    hipLaunchKernelGGL(k_proposal_rows<TT>, grid, dim3(TPB), 0, stream, (const TT*)mask_score, ld, thd,                    \
                       (const long long*)row_scene, (const long long*)point_idx, n, n_dev, dense_of, block_cnt, xyz,     \
                       scale,                                                                                            \
                       inv_voxel, (const uint4*)point_feat, feat_out ? ld_feat * esz / 16 : 0, vpr,                      \
                       (long long*)proposals_idx, (TT*)proposals_ms, coords, (uint4*)feat_out)
    if (dtype == PBN_F32) PBN_PR(float);
    else if (dtype == PBN_BF16) PBN_PR(__hip_bfloat16);
    else if (dtype == PBN_F16) PBN_PR(__half);
    else return PBN_ERR_ARG;
#undef PBN_PR
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_proposal_rows(const void* mask_score, int ld, float thd, const int64_t* row_scene,
                                 const int64_t* point_idx, int n, const int32_t* dense_of, const int32_t* block_cnt,
                                 const float* xyz, float scale, float inv_voxel, const void* point_feat, int ld_feat,
                                 int channels, int dtype, int64_t* proposals_idx, void* proposals_ms, int32_t* coords,
                                 void* feat_out, pbn_stream_t stream) {
    return proposal_rows_impl(mask_score, ld, thd, row_scene, point_idx, n, nullptr, dense_of, block_cnt, xyz, scale,
                              inv_voxel, point_feat, ld_feat, channels, dtype, proposals_idx, proposals_ms, coords, feat_out,
                              stream);
}

extern "C" int pbn_proposal_rows_dev(const void* mask_score, int ld, float thd, const int64_t* row_scene,
                                     const int64_t* point_idx, int n_cap, const int32_t* n_dev, const int32_t* dense_of,
                                     const int32_t* block_cnt, const float* xyz, float scale, float inv_voxel,
                                     const void* point_feat, int ld_feat, int channels, int dtype, int64_t* proposals_idx,
                                     void* proposals_ms, int32_t* coords, void* feat_out, pbn_stream_t stream) {
    return proposal_rows_impl(mask_score, ld, thd, row_scene, point_idx, n_cap, n_dev, dense_of, block_cnt, xyz, scale,
                              inv_voxel, point_feat, ld_feat, channels, dtype, proposals_idx, proposals_ms, coords, feat_out,
                              stream);
}

// ---- rulebook pairs, offset-major (training: operands of the weight gradient) -----------------------------------------
// pair lists of an output-stationary map nbr[n, K]: for every offset k the (input row, output row) pairs with
// nbr[o, k] >= 0, output rows ascending, cut into segments of `seg` pairs.  Three launches: per-block column counts,
// a column-wise exclusive scan over the blocks (+ column totals), the fill.  Positions are prefix counts, never atomics:
// the lists -- and with them the summation order of the weight gradient -- are the same on every run.
namespace pbn {
namespace {
constexpr int PAIR_ROWS = 256;   // rows per block = TPB
constexpr int PAIR_KC = 32;      // offsets per LDS tile
constexpr int PAIR_PITCH = PAIR_KC + 1;

// A block owns 256 rows.  The [256 x 32-offset] piece of the table is read ONCE, along the rows (coalesced), into LDS
// (pitch 33: the column reads below are conflict-free); every wave then owns the columns w, w + 4, ... of the piece and walks
// its 256 rows as four ballots -- no barrier per offset (round 2 read the table column by column with two barriers each).
__device__ __forceinline__ void pair_tile_load(const int* __restrict__ nbr, int n, int K, int row0, int k0, int kc, int* s_tile) {
    for (int e = threadIdx.x; e < PAIR_ROWS * kc; e += TPB) {
        const int r = e / kc, cc = e - r * kc;
        s_tile[r * PAIR_PITCH + cc] = row0 + r < n ? nbr[(size_t)(row0 + r) * K + k0 + cc] : -1;
    }
}

__global__ __launch_bounds__(TPB) void k_pair_count(const int* __restrict__ nbr, int n, int K, int* __restrict__ table) {
    __shared__ int s_tile[PAIR_ROWS * PAIR_PITCH];
    const int row0 = blockIdx.x * PAIR_ROWS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k0 = 0; k0 < K; k0 += PAIR_KC) {
        const int kc = min(PAIR_KC, K - k0);
        pair_tile_load(nbr, n, K, row0, k0, kc, s_tile);
        __syncthreads();
        for (int cc = wave; cc < kc; cc += TPB / 64) {
            int cnt = 0;
#pragma unroll
            for (int sub = 0; sub < PAIR_ROWS / 64; ++sub) cnt += __popcll(__ballot(s_tile[(sub * 64 + lane) * PAIR_PITCH + cc] >= 0));
            if (lane == 0) table[(size_t)blockIdx.x * K + k0 + cc] = cnt;
        }
        __syncthreads();
    }
}

// one wave per column: table[b][k] <- pairs of column k in blocks before b; totals[k] <- column total
__global__ __launch_bounds__(64) void k_pair_scan(int* __restrict__ table, int n_blocks, int K, int* __restrict__ totals) {
    const int k = blockIdx.x, lane = threadIdx.x;
    int run = 0;
    for (int b0 = 0; b0 < n_blocks; b0 += 64) {
        const int b = b0 + lane;
        const int v = b < n_blocks ? table[(size_t)b * K + k] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (b < n_blocks) table[(size_t)b * K + k] = run + inc - v;
        run += __shfl(inc, 63);
    }
    if (lane == 0) totals[k] = run;
}

// seg_start NULL (device lists): every block derives the first segment of every offset from the totals itself (one wave,
// K <= a few hundred) and block 0 publishes them as seg_begin_out[K + 1] -- no separate launch, no read-back
__device__ __forceinline__ void pair_fill_body(const int* __restrict__ nbr, int n, int K, const int* __restrict__ table,
                                               const int* __restrict__ seg_start, const int* __restrict__ totals,
                                               int* __restrict__ seg_begin_out, int seg, int* __restrict__ in_idx,
                                               int* __restrict__ out_idx, long long* __restrict__ seg_offset, int blk,
                                               int* s_tile, int* s_seg) {
    const int row0 = blk * PAIR_ROWS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (seg_start) {
        for (int k = threadIdx.x; k < K; k += TPB) s_seg[k] = seg_start[k];
    } else if (wave == 0) {
        int run = 0;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + lane;
            const int v = k < K ? (totals[k] + seg - 1) / seg : 0;
            int inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            if (k < K) { s_seg[k] = run + inc - v; if (blk == 0) seg_begin_out[k] = run + inc - v; }
            run += __shfl(inc, 63);
        }
        if (lane == 0 && blk == 0) seg_begin_out[K] = run;
    }
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += PAIR_KC) {
        const int kc = min(PAIR_KC, K - k0);
        pair_tile_load(nbr, n, K, row0, k0, kc, s_tile);
        __syncthreads();
        for (int cc = wave; cc < kc; cc += TPB / 64) {
            const int k = k0 + cc;
            int pos0 = table[(size_t)blk * K + k];
            const int first = s_seg[k];
#pragma unroll
            for (int sub = 0; sub < PAIR_ROWS / 64; ++sub) {
                const int src = s_tile[(sub * 64 + lane) * PAIR_PITCH + cc];
                const unsigned long long m = __ballot(src >= 0);
                if (src >= 0) {
                    const int pos = pos0 + __popcll(m & ((1ULL << lane) - 1ULL));
                    const int q = pos / seg;
                    const long long slot = (long long)(first + q) * seg + (pos - q * seg);
                    in_idx[slot] = src;
                    out_idx[slot] = row0 + sub * 64 + lane;
                    if (pos == q * seg) seg_offset[first + q] = k;
                }
                pos0 += __popcll(m);
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(TPB) void k_pair_fill(const int* __restrict__ nbr, int n, int K, const int* __restrict__ table,
                                                  const int* __restrict__ seg_start, const int* __restrict__ totals,
                                                  int* __restrict__ seg_begin_out, int seg, int* __restrict__ in_idx,
                                                  int* __restrict__ out_idx, long long* __restrict__ seg_offset) {
    __shared__ int s_tile[PAIR_ROWS * PAIR_PITCH];
    __shared__ int s_seg[512];
    pair_fill_body(nbr, n, K, table, seg_start, totals, seg_begin_out, seg, in_idx, out_idx, seg_offset, blockIdx.x, s_tile, s_seg);
}


// ---- all maps of a pyramid in three launches (the training executor builds 14 lists per lineage; most maps are small and
// a launch costs more than their work): a job table by value, blocks -> jobs through the prefix of their block counts -------
constexpr int PAIR_MAX_JOBS = 16;
struct PairJobs {
    const int* nbr[PAIR_MAX_JOBS]; int* table[PAIR_MAX_JOBS]; int* totals[PAIR_MAX_JOBS]; int* seg_begin[PAIR_MAX_JOBS];
    int* in_idx[PAIR_MAX_JOBS]; int* out_idx[PAIR_MAX_JOBS]; long long* seg_offset[PAIR_MAX_JOBS];
    int n[PAIR_MAX_JOBS], K[PAIR_MAX_JOBS];
    int block_begin[PAIR_MAX_JOBS + 1];      // row blocks (count / fill)
    int col_begin[PAIR_MAX_JOBS + 1];        // columns (scan)
    int n_jobs, seg;
};
__device__ __forceinline__ int pair_job_of(const int* begin, int n_jobs, int b) {
    int j = 0;
    while (j + 1 < n_jobs && b >= begin[j + 1]) ++j;
    return j;
}

__device__ __forceinline__ void pair_count_body(const int* __restrict__ nbr, int n, int K, int* __restrict__ table, int blk, int* s_tile) {
    const int row0 = blk * PAIR_ROWS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k0 = 0; k0 < K; k0 += PAIR_KC) {
        const int kc = min(PAIR_KC, K - k0);
        pair_tile_load(nbr, n, K, row0, k0, kc, s_tile);
        __syncthreads();
        for (int cc = wave; cc < kc; cc += TPB / 64) {
            int cnt = 0;
#pragma unroll
            for (int sub = 0; sub < PAIR_ROWS / 64; ++sub) cnt += __popcll(__ballot(s_tile[(sub * 64 + lane) * PAIR_PITCH + cc] >= 0));
            if (lane == 0) table[(size_t)blk * K + k0 + cc] = cnt;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(TPB) void k_pair_count_multi(const PairJobs J) {
    __shared__ int s_tile[PAIR_ROWS * PAIR_PITCH];
    const int j = pair_job_of(J.block_begin, J.n_jobs, blockIdx.x);
    pair_count_body(J.nbr[j], J.n[j], J.K[j], J.table[j], blockIdx.x - J.block_begin[j], s_tile);
}
__global__ __launch_bounds__(64) void k_pair_scan_multi(const PairJobs J) {
    const int j = pair_job_of(J.col_begin, J.n_jobs, blockIdx.x);
    const int k = blockIdx.x - J.col_begin[j], lane = threadIdx.x, K = J.K[j];
    const int n_blocks = J.block_begin[j + 1] - J.block_begin[j];
    int* table = J.table[j];
    int run = 0;
    for (int b0 = 0; b0 < n_blocks; b0 += 64) {
        const int b = b0 + lane;
        const int v = b < n_blocks ? table[(size_t)b * K + k] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (b < n_blocks) table[(size_t)b * K + k] = run + inc - v;
        run += __shfl(inc, 63);
    }
    if (lane == 0) J.totals[j][k] = run;
}
__global__ __launch_bounds__(TPB) void k_pair_fill_multi(const PairJobs J) {
    __shared__ int s_tile[PAIR_ROWS * PAIR_PITCH];
    __shared__ int s_seg[512];
    const int j = pair_job_of(J.block_begin, J.n_jobs, blockIdx.x);
    pair_fill_body(J.nbr[j], J.n[j], J.K[j], J.table[j], nullptr, J.totals[j], J.seg_begin[j], J.seg, J.in_idx[j], J.out_idx[j],
                   J.seg_offset[j], blockIdx.x - J.block_begin[j], s_tile, s_seg);
}
}  // namespace
}  // namespace pbn

// pbn_rulebook_pairs_multi: pbn_rulebook_pair_counts + pbn_rulebook_pair_fill_dev of up to 16 maps in three launches
extern "C" int pbn_rulebook_pairs_multi(const pbn_pair_job* jobs, int n_jobs, int segment, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!jobs || n_jobs < 1 || n_jobs > PAIR_MAX_JOBS || segment < 1) return PBN_ERR_ARG;
    PairJobs J;
    J.n_jobs = n_jobs; J.seg = segment;
    int blocks = 0, cols = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const pbn_pair_job& q = jobs[j];
        if (q.n < 0 || q.n_offsets < 1 || q.n_offsets > 512 || !q.nbr || !q.table || !q.totals || !q.seg_begin || !q.in_idx ||
            !q.out_idx || !q.seg_offset)
            return PBN_ERR_ARG;
        J.nbr[j] = q.nbr; J.table[j] = q.table; J.totals[j] = q.totals; J.seg_begin[j] = q.seg_begin;
        J.in_idx[j] = q.in_idx; J.out_idx[j] = q.out_idx; J.seg_offset[j] = (long long*)q.seg_offset;
        J.n[j] = q.n; J.K[j] = q.n_offsets;
        J.block_begin[j] = blocks; J.col_begin[j] = cols;
        blocks += q.n > 0 ? (q.n + PAIR_ROWS - 1) / PAIR_ROWS : 1;      // an empty map keeps one (empty) block: it publishes seg_begin
        cols += q.n_offsets;
    }
    for (int j = n_jobs; j <= PAIR_MAX_JOBS; ++j) { J.block_begin[j] = blocks; J.col_begin[j] = cols; }
    hipLaunchKernelGGL(k_pair_count_multi, dim3(blocks), dim3(TPB), 0, stream, J);
    hipLaunchKernelGGL(k_pair_scan_multi, dim3(cols), dim3(64), 0, stream, J);
    hipLaunchKernelGGL(k_pair_fill_multi, dim3(blocks), dim3(TPB), 0, stream, J);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_rulebook_pair_blocks(int n) { return n > 0 ? (n + pbn::PAIR_ROWS - 1) / pbn::PAIR_ROWS : 0; }

extern "C" int pbn_rulebook_pair_counts(const int32_t* nbr, int n, int n_offsets, int32_t* table, int32_t* totals,
                                        pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || n_offsets < 1) return PBN_ERR_ARG;
    if (!totals) return PBN_ERR_ARG;
    if (n == 0) { { const int frc_ = fill_bytes(totals, 0, sizeof(int32_t) * n_offsets, stream); if (frc_ != PBN_OK) return frc_; } return PBN_OK; }
    if (!nbr || !table) return PBN_ERR_ARG;
    const int nb = pbn_rulebook_pair_blocks(n);
    hipLaunchKernelGGL(k_pair_count, dim3(nb), dim3(TPB), 0, stream, nbr, n, n_offsets, table);
    hipLaunchKernelGGL(k_pair_scan, dim3(n_offsets), dim3(64), 0, stream, table, nb, n_offsets, totals);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_rulebook_pair_fill(const int32_t* nbr, int n, int n_offsets, const int32_t* table, const int32_t* seg_start,
                                      int seg, int n_segments, int32_t* in_idx, int32_t* out_idx, int64_t* seg_offset,
                                      pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || n_offsets < 1 || seg < 1 || n_segments < 0) return PBN_ERR_ARG;
    if (n_segments == 0) return PBN_OK;
    if (!in_idx || !out_idx || !seg_offset) return PBN_ERR_ARG;
    // padding slots (and surplus segments) read as -1 / offset 0
    { const int frc_ = fill_bytes(in_idx, 0xff, sizeof(int32_t) * (size_t)n_segments * seg, stream); if (frc_ != PBN_OK) return frc_; }
    { const int frc_ = fill_bytes(out_idx, 0xff, sizeof(int32_t) * (size_t)n_segments * seg, stream); if (frc_ != PBN_OK) return frc_; }
    { const int frc_ = fill_bytes(seg_offset, 0, sizeof(int64_t) * (size_t)n_segments, stream); if (frc_ != PBN_OK) return frc_; }
    if (n == 0) return PBN_OK;
    if (!nbr || !table || !seg_start) return PBN_ERR_ARG;
    if (n_offsets > 512) return PBN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_pair_fill, dim3(pbn_rulebook_pair_blocks(n)), dim3(TPB), 0, stream, nbr, n, n_offsets, table, seg_start,
                       (const int*)nullptr, (int*)nullptr, seg, in_idx, out_idx, (long long*)seg_offset);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_rulebook_pair_fill_dev(const int32_t* nbr, int n, int n_offsets, const int32_t* table, const int32_t* totals,
                                          int seg, int32_t* seg_begin, int32_t* in_idx, int32_t* out_idx, int64_t* seg_offset,
                                          pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || n_offsets < 1 || seg < 1 || !totals || !seg_begin) return PBN_ERR_ARG;
    if (n_offsets > 512) return PBN_ERR_UNSUPPORTED;
    if (!nbr || !table || !in_idx || !out_idx || !seg_offset) return PBN_ERR_ARG;
    // n == 0: one (empty) block still publishes seg_begin.  The tails of the last segments are NOT padded: the consumer
    // bounds every offset by its pair count (pbn_spconv_wgrad's pair_counts = `totals`)
    hipLaunchKernelGGL(k_pair_fill, dim3(n > 0 ? pbn_rulebook_pair_blocks(n) : 1), dim3(TPB), 0, stream, nbr, n, n_offsets, table,
                       (const int*)nullptr, totals, seg_begin, seg, in_idx, out_idx, (long long*)seg_offset);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_gather_rulebook_rows(const void* in, int ld_in_bytes, int row_bytes, const int32_t* nbr, int n_offsets,
                                        int k0, int kc, int n, void* out, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || row_bytes <= 0 || (row_bytes & 3) || (ld_in_bytes & 3) || ld_in_bytes < row_bytes || k0 < 0 || kc < 1 ||
        k0 + kc > n_offsets)
        return PBN_ERR_ARG;
    if (n == 0) return PBN_OK;
    if (!in || !nbr || !out || (((uintptr_t)in | (uintptr_t)out) & 3)) return PBN_ERR_ARG;
    const long long total = (long long)n * kc * (row_bytes / 4);
    hipLaunchKernelGGL(k_gather_rulebook_rows, dim3(cdiv(total, TPB)), dim3(TPB), 0, stream, (const unsigned*)in,
                       ld_in_bytes / 4, row_bytes / 4, nbr, n_offsets, k0, kc, n, (unsigned*)out);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_pack_weight(const float* src, int n_offsets, int dim_a, int dim_b, int flip, int transpose, int dtype,
                               int vecs_per_offset, int n_steps, int cout_padded, void* out, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!src || !out || n_offsets < 1 || dim_a < 1 || dim_b < 1 || vecs_per_offset < 1 || n_steps < 1 || cout_padded < 16 ||
        (cout_padded & 15))
        return PBN_ERR_ARG;
    const int esz = dtype == PBN_F32 ? 4 : 2;
    const int e = 16 / esz;
    const int cin = transpose ? dim_b : dim_a, cout = transpose ? dim_a : dim_b;
    const int cin_p = vecs_per_offset * e;
    if (cin_p < cin || cout_padded < cout || (long long)n_steps * 4 * e < (long long)n_offsets * cin_p) return PBN_ERR_ARG;
    const long long total = (long long)n_steps * (cout_padded / 16) * 64 * e;
    const dim3 grid(cdiv(total, TPB));
    if (dtype == PBN_F32)
        hipLaunchKernelGGL(k_pack_weight<float>, grid, dim3(TPB), 0, stream, src, n_offsets, dim_a, dim_b, flip, transpose, cin,
                           cout, cin_p, cout_padded, n_steps, (float*)out);
    else if (dtype == PBN_BF16)
        hipLaunchKernelGGL(k_pack_weight<__hip_bfloat16>, grid, dim3(TPB), 0, stream, src, n_offsets, dim_a, dim_b, flip,
                           transpose, cin, cout, cin_p, cout_padded, n_steps, (__hip_bfloat16*)out);
    else if (dtype == PBN_F16)
        hipLaunchKernelGGL(k_pack_weight<__half>, grid, dim3(TPB), 0, stream, src, n_offsets, dim_a, dim_b, flip, transpose,
                           cin, cout, cin_p, cout_padded, n_steps, (__half*)out);
    else
        return PBN_ERR_ARG;
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_pack_weights_batch(const pbn_pack_job* jobs_dev, int n_jobs, int max_vectors, int dtype, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_jobs < 0 || max_vectors < 0) return PBN_ERR_ARG;
    if (n_jobs == 0 || max_vectors == 0) return PBN_OK;
    if (!jobs_dev || n_jobs > 65535) return PBN_ERR_ARG;
    const int e = dtype == PBN_F32 ? 4 : 8;
    long long bx = cdiv((long long)max_vectors * e, TPB);
    if (bx > 512) bx = 512;
    const dim3 grid((unsigned)bx, (unsigned)n_jobs);
    if (dtype == PBN_F32) hipLaunchKernelGGL(k_pack_weights_batch<float>, grid, dim3(TPB), 0, stream, jobs_dev);
    else if (dtype == PBN_BF16) hipLaunchKernelGGL(k_pack_weights_batch<__hip_bfloat16>, grid, dim3(TPB), 0, stream, jobs_dev);
    else if (dtype == PBN_F16) hipLaunchKernelGGL(k_pack_weights_batch<__half>, grid, dim3(TPB), 0, stream, jobs_dev);
    else return PBN_ERR_ARG;
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}
