// bnorm.hip -- train-mode batch normalisation of a feature slab [n, C] (ME.MinkowskiBatchNorm = nn.BatchNorm1d on .F,
// /root/reference/network/Mink.py:71-73,224; training step of BASELINE configs[2]).  HBM-bound byte work: every pass
// streams the slab with 16-byte vectors, per-channel sums are kept in registers by the thread that owns a channel vector,
// merged inside the block through LDS in a fixed order, across blocks by a second tiny launch in double precision --
// no atomics, so statistics and gradients are the same on every run.
//   forward : partial (sum x, sum x^2) -> mean, biased var, invstd, running-stat update -> y = (x - mean) invstd w + b
//   backward: partial (sum dy, sum dy (x - mean)) -> dbias, dweight, dx = w invstd (dy - mean(dy) - xhat mean(dy xhat))
// The block tail of the network (Mink.py:293-350: conv -> bn -> relu, conv -> bn -> += residual -> relu) rides along: the
// forward apply pass adds the residual and clamps, the backward passes mask dy with (y > 0) while they read it and hand the
// masked gradient to the residual branch -- the ReLU / add passes over the slab and their launches disappear.
#include "pbn_common.h"

#include <cstdlib>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

namespace pbn {
namespace {

constexpr int BN_TPB = 256;
constexpr int BN_MAX_BLOCKS = 1024;

template <typename T> struct Vec;
template <> struct Vec<float> { static constexpr int W = 4; };
template <> struct Vec<__hip_bfloat16> { static constexpr int W = 8; };
template <> struct Vec<__half> { static constexpr int W = 8; };

__device__ __forceinline__ void unpack(const uint4& v, float (&f)[4], float*) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
__device__ __forceinline__ void unpack(const uint4& v, float (&f)[8], __hip_bfloat16*) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ void unpack(const uint4& v, float (&f)[8], __half*) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __half2 h = *reinterpret_cast<const __half2*>(&w[i]);
        f[2 * i] = __low2float(h); f[2 * i + 1] = __high2float(h);
    }
}
__device__ __forceinline__ uint4 pack(const float (&f)[4], float*) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
__device__ __forceinline__ uint4 pack(const float (&f)[8], __hip_bfloat16*) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __hip_bfloat16 lo = __float2bfloat16(f[2 * i]), hi = __float2bfloat16(f[2 * i + 1]);   // round to nearest even
        w[i] = (unsigned)*reinterpret_cast<const unsigned short*>(&lo) | ((unsigned)*reinterpret_cast<const unsigned short*>(&hi) << 16);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ uint4 pack(const float (&f)[8], __half*) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __half2 h = __floats2half2_rn(f[2 * i], f[2 * i + 1]);
        w[i] = *reinterpret_cast<const unsigned*>(&h);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// rows [lo, hi) of this block; a thread owns channel vector cv and every rows_per_iter-th row
struct Span { int lo, hi, cv, r, rows_per_iter, active; };
__device__ __forceinline__ Span span_of(int n, int vpr) {
    Span s;
    s.rows_per_iter = BN_TPB / vpr;
    s.active = threadIdx.x < s.rows_per_iter * vpr;
    s.r = threadIdx.x / vpr;
    s.cv = threadIdx.x - s.r * vpr;
    const int per_block = (n + gridDim.x - 1) / gridDim.x;
    s.lo = blockIdx.x * per_block;
    s.hi = min(n, s.lo + per_block);
    return s;
}

// partial[b][0][c] = sum over the block's rows of A, partial[b][1][c] = sum of B, with
//   forward : A = x, B = x^2;   backward: A = dy, B = dy * (x - mean[c])
// The second (merge) step of a pass, folded into the partial launch on small slabs (<= FUSE_MAX_BLOCKS partial blocks = 16 k
// rows: the stride-4 / 8 / 16 levels, two thirds of the layers): the LAST workgroup to arrive (a ticket counter in the
// workspace, left at zero again) merges the block partials, one thread per channel, in block order and in double -- the same
// result whichever workgroup is last.  Saves the merge launch; an experiment, off by default (see fuse_final()).
constexpr int FUSE_MAX_BLOCKS = 128;
struct FinalArgs {
    int* counter;              // nullptr: no fused merge (the separate k_bn_stats_final / k_bn_bwd_final launch follows)
    float eps, momentum;
    float* running_mean; float* running_var; float* save_mean; float* save_invstd;      // forward
    const float* invstd; float* dweight; float* dbias; float* coef;                     // backward
};

__device__ __forceinline__ void stats_from_sums(double s1, double s2, int ch, int n, const float* shift, const FinalArgs& f) {
    const double dm = s1 / n;                           // mean of (x - k)
    const double mean = (double)shift[ch] + dm;
    double var = s2 / n - dm * dm;                      // biased (normalisation)
    if (var < 0.0) var = 0.0;
    f.save_mean[ch] = (float)mean;
    f.save_invstd[ch] = (float)(1.0 / sqrt(var + (double)f.eps));
    if (f.running_mean) f.running_mean[ch] = (float)((1.0 - f.momentum) * f.running_mean[ch] + f.momentum * mean);
    if (f.running_var) {
        const double unbiased = n > 1 ? var * n / (n - 1) : var;
        f.running_var[ch] = (float)((1.0 - f.momentum) * f.running_var[ch] + f.momentum * unbiased);
    }
}
__device__ __forceinline__ void coefs_from_sums(double s1, double s2, int ch, int n, int c, const FinalArgs& f) {
    const double is = f.invstd[ch];
    if (f.dbias) f.dbias[ch] = (float)s1;
    if (f.dweight) f.dweight[ch] = (float)(s2 * is);
    f.coef[ch] = (float)(s1 / n);
    f.coef[c + ch] = (float)(s2 * is * is / n);         // multiplies (x - mean)
}

template <typename T, bool BWD>
__global__ __launch_bounds__(BN_TPB) void k_bn_partial(const T* __restrict__ x, int ld_x, const T* __restrict__ dy, int ld_dy,
                                                      const T* __restrict__ ymask, int ld_y,
                                                      int n, int c, const float* __restrict__ mean, float* __restrict__ partial,
                                                      float* __restrict__ shift_out, const FinalArgs fa) {
    constexpr int W = Vec<T>::W;
    extern __shared__ float s_acc[];          // [BN_TPB][2 * W]
    const int vpr = c / W;
    const Span s = span_of(n, vpr);
    float a[W], b[W], mu[W];
#pragma unroll
    for (int i = 0; i < W; ++i) { a[i] = 0.f; b[i] = 0.f; mu[i] = (BWD && s.active) ? mean[s.cv * W + i] : 0.f; }
    if (!BWD && s.active) {
        // forward sums are taken about the slab's first row (sum (x - k), sum (x - k)^2): channel means far from zero
        // would otherwise cancel in fp32.  Block 0 publishes k for the final merge.
        unpack(*reinterpret_cast<const uint4*>(x + s.cv * W), mu, (T*)nullptr);
        if (blockIdx.x == 0 && s.r == 0) {
#pragma unroll
            for (int i = 0; i < W; ++i) shift_out[s.cv * W + i] = mu[i];
        }
    }
    if (s.active) {
        for (int row = s.lo + s.r; row < s.hi; row += s.rows_per_iter) {
            float xv[W];
            unpack(*reinterpret_cast<const uint4*>(x + (size_t)row * ld_x + s.cv * W), xv, (T*)nullptr);
            if (BWD) {
                float gv[W];
                unpack(*reinterpret_cast<const uint4*>(dy + (size_t)row * ld_dy + s.cv * W), gv, (T*)nullptr);
                if (ymask) {   // fused ReLU: the gradient only flows where the block's output was positive
                    float yv[W];
                    unpack(*reinterpret_cast<const uint4*>(ymask + (size_t)row * ld_y + s.cv * W), yv, (T*)nullptr);
#pragma unroll
                    for (int i = 0; i < W; ++i) gv[i] = yv[i] > 0.f ? gv[i] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < W; ++i) { a[i] += gv[i]; b[i] = fmaf(gv[i], xv[i] - mu[i], b[i]); }
            } else {
#pragma unroll
                for (int i = 0; i < W; ++i) { const float d = xv[i] - mu[i]; a[i] += d; b[i] = fmaf(d, d, b[i]); }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < W; ++i) { s_acc[threadIdx.x * 2 * W + i] = a[i]; s_acc[threadIdx.x * 2 * W + W + i] = b[i]; }
    __syncthreads();
    // fixed-order merge over the rows_per_iter row slots: one thread per (A|B, channel)
    for (int e = threadIdx.x; e < 2 * c; e += BN_TPB) {
        const int which = e / c, ch = e - which * c;
        const int cv = ch / W, i = ch - cv * W;
        float t = 0.f;
        for (int r = 0; r < s.rows_per_iter; ++r) t += s_acc[(r * vpr + cv) * 2 * W + which * W + i];
        partial[((size_t)blockIdx.x * 2 + which) * c + ch] = t;
    }
    if (!fa.counter) return;
    // ---- fused merge: the last workgroup to finish ----
    __shared__ int s_last;
    __threadfence();                                   // this workgroup's partials (and block 0's shift) are visible device-wide ...
    __syncthreads();                                   // ... for every thread of it, before its ticket is drawn
    if (threadIdx.x == 0) s_last = atomicAdd(fa.counter, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                   // acquire: the other workgroups' partials
    if (threadIdx.x == 0) *fa.counter = 0;             // ready for the next launch on this workspace (stream order)
    const int blocks = gridDim.x;
    const float* pin = partial;
    for (int ch = threadIdx.x; ch < c; ch += BN_TPB) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
        for (int b = 0; b < blocks; ++b) {
            s1 += (double)__builtin_nontemporal_load(pin + ((size_t)b * 2 + 0) * c + ch);
            s2 += (double)__builtin_nontemporal_load(pin + ((size_t)b * 2 + 1) * c + ch);
        }
        if (BWD) coefs_from_sums(s1, s2, ch, n, c, fa);
        else stats_from_sums(s1, s2, ch, n, shift_out, fa);
    }
}

// one wave per channel: lane l sums the block partials l, l + 64, ... in double, then a fixed butterfly over the lanes
__device__ __forceinline__ void channel_sums(const float* __restrict__ partial, int blocks, int c, int ch, double& s1, double& s2) {
    const int lane = threadIdx.x & 63;
    s1 = 0.0; s2 = 0.0;
    // loads four block steps ahead of the (ordered) double additions: the merge is one latency chain per channel otherwise
    int b = lane;
    for (; b + 192 < blocks; b += 256) {
        float p0[4], p1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            p0[u] = partial[((size_t)(b + 64 * u) * 2 + 0) * c + ch];
            p1[u] = partial[((size_t)(b + 64 * u) * 2 + 1) * c + ch];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s1 += p0[u]; s2 += p1[u]; }
    }
    for (; b < blocks; b += 64) { s1 += partial[((size_t)b * 2 + 0) * c + ch]; s2 += partial[((size_t)b * 2 + 1) * c + ch]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
}

__global__ __launch_bounds__(BN_TPB) void k_bn_stats_final(const float* __restrict__ partial, const float* __restrict__ shift,
                                                          int blocks, int n, int c, float eps,
                                                          float momentum, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ save_mean,
                                                          float* __restrict__ save_invstd) {
    const int ch = blockIdx.x * (BN_TPB / 64) + (threadIdx.x >> 6);
    if (ch >= c) return;
    double s1, s2;
    channel_sums(partial, blocks, c, ch, s1, s2);
    if (threadIdx.x & 63) return;
    FinalArgs f{};
    f.eps = eps; f.momentum = momentum; f.running_mean = running_mean; f.running_var = running_var;
    f.save_mean = save_mean; f.save_invstd = save_invstd;
    stats_from_sums(s1, s2, ch, n, shift, f);
}

// coef[0][c] = mean(dy), coef[1][c] = mean(dy * xhat) * invstd  (the two projections of dx), dweight, dbias
__global__ __launch_bounds__(BN_TPB) void k_bn_bwd_final(const float* __restrict__ partial, int blocks, int n, int c,
                                                        const float* __restrict__ invstd, float* __restrict__ dweight,
                                                        float* __restrict__ dbias, float* __restrict__ coef) {
    const int ch = blockIdx.x * (BN_TPB / 64) + (threadIdx.x >> 6);
    if (ch >= c) return;
    double s1, s2;
    channel_sums(partial, blocks, c, ch, s1, s2);
    if (threadIdx.x & 63) return;
    FinalArgs f{};
    f.invstd = invstd; f.dweight = dweight; f.dbias = dbias; f.coef = coef;
    coefs_from_sums(s1, s2, ch, n, c, f);
}

// forward apply: y = act((x - mean) * (invstd * w) + b [+ residual]) ; backward apply: g = dy masked by (y > 0),
// dx = (g - coef0 - (x - mean) * coef1) * (invstd * w), residual gradient = g.
// Round 4: a thread owns ONE channel vector (its 8 scales / means / coefficients live in registers for the whole launch) and
// walks rows, two in flight; round 3's form -- a thread per (row, vector), found by a 64-bit division, re-reading 32-40
// per-channel floats for every 16 bytes of slab -- ran the stride-1 slabs at 1.2 TB/s (97 us for 146 k x 96 backward).
template <typename T, bool BWD>
__global__ __launch_bounds__(BN_TPB) void k_bn_apply(const T* __restrict__ x, int ld_x, const T* __restrict__ dy, int ld_dy, int n,
                                                    int c, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                    const float* __restrict__ weight, const float* __restrict__ bias,
                                                    const float* __restrict__ coef, T* __restrict__ out, int ld_out,
                                                    const T* __restrict__ aux, int ld_aux, int relu, T* __restrict__ dres,
                                                    int ld_dres) {
    constexpr int W = Vec<T>::W;
    constexpr int U = 2;                      // rows in flight per thread
    const int vpr = c / W;
    const Span s = span_of(n, vpr);
    if (!s.active) return;
    float sc[W], mu[W], k0[W], k1[W];
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const int ch = s.cv * W + i;
        sc[i] = invstd[ch] * (weight ? weight[ch] : 1.f);
        mu[i] = mean[ch];
        k0[i] = BWD ? coef[ch] : (bias ? bias[ch] : 0.f);
        k1[i] = BWD ? coef[c + ch] : 0.f;
    }
    const size_t col = (size_t)s.cv * W;
    for (int row0 = s.lo + s.r; row0 < s.hi; row0 += U * s.rows_per_iter) {
        uint4 xr[U], gr[U], ar[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = row0 + u * s.rows_per_iter;
            live[u] = row < s.hi;
            xr[u] = gr[u] = ar[u] = make_uint4(0u, 0u, 0u, 0u);
            if (live[u]) {
                xr[u] = *reinterpret_cast<const uint4*>(x + (size_t)row * ld_x + col);
                if (BWD) gr[u] = *reinterpret_cast<const uint4*>(dy + (size_t)row * ld_dy + col);
                if (aux) ar[u] = *reinterpret_cast<const uint4*>(aux + (size_t)row * ld_aux + col);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            const int row = row0 + u * s.rows_per_iter;
            float xv[W], o[W];
            unpack(xr[u], xv, (T*)nullptr);
            if (BWD) {
                float gv[W];
                unpack(gr[u], gv, (T*)nullptr);
                if (aux) {   // aux = the forward output y: ReLU mask
                    float yv[W];
                    unpack(ar[u], yv, (T*)nullptr);
#pragma unroll
                    for (int i = 0; i < W; ++i) gv[i] = yv[i] > 0.f ? gv[i] : 0.f;
                }
                if (dres) *reinterpret_cast<uint4*>(dres + (size_t)row * ld_dres + col) = pack(gv, (T*)nullptr);
#pragma unroll
                for (int i = 0; i < W; ++i) o[i] = (gv[i] - k0[i] - (xv[i] - mu[i]) * k1[i]) * sc[i];
            } else {
#pragma unroll
                for (int i = 0; i < W; ++i) o[i] = fmaf(xv[i] - mu[i], sc[i], k0[i]);
                if (aux) {   // aux = the residual branch
                    float rv[W];
                    unpack(ar[u], rv, (T*)nullptr);
#pragma unroll
                    for (int i = 0; i < W; ++i) o[i] += rv[i];
                }
                if (relu) {
#pragma unroll
                    for (int i = 0; i < W; ++i) o[i] = fmaxf(o[i], 0.f);
                }
            }
            *reinterpret_cast<uint4*>(out + (size_t)row * ld_out + col) = pack(o, (T*)nullptr);
        }
    }
}

// grid of an apply pass: rows_per_iter rows per block step, up to 8 steps per block (fewer on small slabs: more blocks)
int apply_blocks(int n, int vpr) {
    const int rpi = BN_TPB / vpr;
    int iters = n / (rpi * 1024);
    iters = iters < 1 ? 1 : (iters > 8 ? 8 : iters);
    return (int)cdiv((long long)n, (long long)rpi * iters);
}

// workspace layout: 4 ints (the merge ticket counter: at a FIXED place, layers of different widths share the workspace) |
// [BN_MAX_BLOCKS][2][c] block partials | [2][c] shift / coefficients.  `ws` below points behind the counter.
int* counter_of(float* ws, int c) { (void)c; return reinterpret_cast<int*>(ws - 4); }
bool fuse_final() {
    // OFF by default: measured on the configs[2] step (round 4), the last workgroup's tail -- fence, ticket, acquire, merge, double
    // division / square root, all on one latency chain -- adds 8.5 us to a fused partial launch where the separate merge
    // launch costs 5: 17.5 -> 21.2 ms of batch-norm kernels per 4 steps, step time level (31.7 ms both ways, 220 launches fewer)
    static const bool on = getenv("PBN_BN_FUSE_FINAL") && atoi(getenv("PBN_BN_FUSE_FINAL")) == 1;
    return on;
}

int blocks_for(int n) {
    int b = (n + 127) / 128;
    return b < 1 ? 1 : (b > BN_MAX_BLOCKS ? BN_MAX_BLOCKS : b);
}

template <typename T>
bool layout_ok(const void* p, int ld, int c) {
    constexpr int W = Vec<T>::W;
    return p && c % W == 0 && c / W >= 1 && c / W <= BN_TPB && ld >= c && (ld * sizeof(T)) % 16 == 0 && ((uintptr_t)p & 15) == 0;
}

template <typename T>
int forward_t(const void* x, int ld_x, int n, int c, const float* weight, const float* bias, float eps, float momentum,
              float* running_mean, float* running_var, const void* residual, int ld_res, int relu, void* y, int ld_y,
              float* save_mean, float* save_invstd, float* ws, hipStream_t stream) {
    if (!layout_ok<T>(x, ld_x, c) || !layout_ok<T>(y, ld_y, c) || (residual && !layout_ok<T>(residual, ld_res, c)))
        return PBN_ERR_UNSUPPORTED;
    constexpr int W = Vec<T>::W;
    const int blocks = blocks_for(n);
    FinalArgs fa{};
    const bool fuse = fuse_final() && blocks <= FUSE_MAX_BLOCKS;
    if (fuse) {
        fa.counter = counter_of(ws, c);
        fa.eps = eps; fa.momentum = momentum; fa.running_mean = running_mean; fa.running_var = running_var;
        fa.save_mean = save_mean; fa.save_invstd = save_invstd;
    }
    hipLaunchKernelGGL((k_bn_partial<T, false>), dim3(blocks), dim3(BN_TPB), BN_TPB * 2 * W * sizeof(float), stream, (const T*)x,
                       ld_x, (const T*)nullptr, 0, (const T*)nullptr, 0, n, c, (const float*)nullptr, ws,
                       ws + (size_t)BN_MAX_BLOCKS * 2 * c, fa);
    if (!fuse)
        hipLaunchKernelGGL(k_bn_stats_final, dim3(cdiv(c, BN_TPB / 64)), dim3(BN_TPB), 0, stream, ws,
                           ws + (size_t)BN_MAX_BLOCKS * 2 * c, blocks, n, c, eps, momentum,
                           running_mean, running_var, save_mean, save_invstd);
    hipLaunchKernelGGL((k_bn_apply<T, false>), dim3(apply_blocks(n, c / W)), dim3(BN_TPB), 0, stream, (const T*)x, ld_x,
                       (const T*)nullptr, 0, n, c, save_mean, save_invstd, weight, bias, (const float*)nullptr, (T*)y, ld_y,
                       (const T*)residual, ld_res, relu, (T*)nullptr, 0);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

template <typename T>
int backward_t(const void* x, int ld_x, const void* dy, int ld_dy, const void* y, int ld_y, int n, int c, const float* weight,
               const float* save_mean, const float* save_invstd, void* dx, int ld_dx, void* dres, int ld_dres, float* dweight,
               float* dbias, float* ws, hipStream_t stream) {
    if (!layout_ok<T>(x, ld_x, c) || !layout_ok<T>(dy, ld_dy, c) || !layout_ok<T>(dx, ld_dx, c) ||
        (y && !layout_ok<T>(y, ld_y, c)) || (dres && !layout_ok<T>(dres, ld_dres, c)))
        return PBN_ERR_UNSUPPORTED;
    constexpr int W = Vec<T>::W;
    const int blocks = blocks_for(n);
    float* coef = ws + (size_t)BN_MAX_BLOCKS * 2 * c;
    FinalArgs fa{};
    const bool fuse = fuse_final() && blocks <= FUSE_MAX_BLOCKS;
    if (fuse) {
        fa.counter = counter_of(ws, c);
        fa.invstd = save_invstd; fa.dweight = dweight; fa.dbias = dbias; fa.coef = coef;
    }
    hipLaunchKernelGGL((k_bn_partial<T, true>), dim3(blocks), dim3(BN_TPB), BN_TPB * 2 * W * sizeof(float), stream, (const T*)x,
                       ld_x, (const T*)dy, ld_dy, (const T*)y, ld_y, n, c, save_mean, ws, (float*)nullptr, fa);
    if (!fuse)
        hipLaunchKernelGGL(k_bn_bwd_final, dim3(cdiv(c, BN_TPB / 64)), dim3(BN_TPB), 0, stream, ws, blocks, n, c, save_invstd,
                           dweight, dbias, coef);
    hipLaunchKernelGGL((k_bn_apply<T, true>), dim3(apply_blocks(n, c / W)), dim3(BN_TPB), 0, stream, (const T*)x, ld_x,
                       (const T*)dy, ld_dy, n, c, save_mean, save_invstd, weight, (const float*)nullptr, coef, (T*)dx, ld_dx,
                       (const T*)y, ld_y, 0, (T*)dres, ld_dres);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

}  // namespace
}  // namespace pbn

using namespace pbn;

extern "C" size_t pbn_bn_workspace_bytes(int channels) {
    return channels > 0 ? sizeof(float) * ((size_t)BN_MAX_BLOCKS * 2 * channels + 2 * (size_t)channels) + 16 : 0;
}

extern "C" int pbn_bn_act_train_forward(const void* x, int ld_x, int n, int channels, int dtype, const float* weight,
                                        const float* bias, float eps, float momentum, float* running_mean, float* running_var,
                                        const void* residual, int ld_res, int relu, void* y, int ld_y, float* save_mean,
                                        float* save_invstd, void* workspace, size_t workspace_bytes, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 1 || channels < 1 || !save_mean || !save_invstd) return PBN_ERR_ARG;
    if (!workspace || workspace_bytes < pbn_bn_workspace_bytes(channels) || ((uintptr_t)workspace & 15)) return PBN_ERR_WORKSPACE;
    float* ws = (float*)workspace + 4;
    switch (dtype) {
        case PBN_F32: return forward_t<float>(x, ld_x, n, channels, weight, bias, eps, momentum, running_mean, running_var,
                                              residual, ld_res, relu, y, ld_y, save_mean, save_invstd, ws, stream);
        case PBN_BF16: return forward_t<__hip_bfloat16>(x, ld_x, n, channels, weight, bias, eps, momentum, running_mean,
                                                        running_var, residual, ld_res, relu, y, ld_y, save_mean, save_invstd, ws,
                                                        stream);
        case PBN_F16: return forward_t<__half>(x, ld_x, n, channels, weight, bias, eps, momentum, running_mean, running_var,
                                               residual, ld_res, relu, y, ld_y, save_mean, save_invstd, ws, stream);
        default: return PBN_ERR_ARG;
    }
}

extern "C" int pbn_bn_train_forward(const void* x, int ld_x, int n, int channels, int dtype, const float* weight, const float* bias,
                                    float eps, float momentum, float* running_mean, float* running_var, void* y, int ld_y,
                                    float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes,
                                    pbn_stream_t stream_) {
    return pbn_bn_act_train_forward(x, ld_x, n, channels, dtype, weight, bias, eps, momentum, running_mean, running_var, nullptr, 0,
                                    0, y, ld_y, save_mean, save_invstd, workspace, workspace_bytes, stream_);
}

extern "C" int pbn_bn_act_train_backward(const void* x, int ld_x, const void* dy, int ld_dy, const void* y, int ld_y, int n,
                                         int channels, int dtype, const float* weight, const float* save_mean,
                                         const float* save_invstd, void* dx, int ld_dx, void* dres, int ld_dres, float* dweight,
                                         float* dbias, void* workspace, size_t workspace_bytes, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 1 || channels < 1 || !save_mean || !save_invstd) return PBN_ERR_ARG;
    if (!workspace || workspace_bytes < pbn_bn_workspace_bytes(channels) || ((uintptr_t)workspace & 15)) return PBN_ERR_WORKSPACE;
    float* ws = (float*)workspace + 4;
    switch (dtype) {
        case PBN_F32: return backward_t<float>(x, ld_x, dy, ld_dy, y, ld_y, n, channels, weight, save_mean, save_invstd, dx, ld_dx,
                                               dres, ld_dres, dweight, dbias, ws, stream);
        case PBN_BF16: return backward_t<__hip_bfloat16>(x, ld_x, dy, ld_dy, y, ld_y, n, channels, weight, save_mean, save_invstd,
                                                         dx, ld_dx, dres, ld_dres, dweight, dbias, ws, stream);
        case PBN_F16: return backward_t<__half>(x, ld_x, dy, ld_dy, y, ld_y, n, channels, weight, save_mean, save_invstd, dx, ld_dx,
                                                dres, ld_dres, dweight, dbias, ws, stream);
        default: return PBN_ERR_ARG;
    }
}

extern "C" int pbn_bn_train_backward(const void* x, int ld_x, const void* dy, int ld_dy, int n, int channels, int dtype,
                                     const float* weight, const float* save_mean, const float* save_invstd, void* dx, int ld_dx,
                                     float* dweight, float* dbias, void* workspace, size_t workspace_bytes, pbn_stream_t stream_) {
    return pbn_bn_act_train_backward(x, ld_x, dy, ld_dy, nullptr, 0, n, channels, dtype, weight, save_mean, save_invstd, dx, ld_dx,
                                     nullptr, 0, dweight, dbias, workspace, workspace_bytes, stream_);
}
