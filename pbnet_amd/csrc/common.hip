// common.hip -- version / error plumbing and the device-wide scan used by the grouping and coordinate kernels.
#include "pbn_common.h"

namespace pbn {

thread_local int g_last_hip_error = 0;

__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// Block-wide exclusive scan of one value per thread (SCAN_THREADS threads); returns exclusive prefix, sets total.
__device__ __forceinline__ int block_excl_scan(int v, int* lds_wave_tot, int& block_total) {
    const int lane = lane_id();
    const int wid = threadIdx.x >> 6;
    const int nw = SCAN_THREADS / 64;
    int incl = wave_incl_scan(v);
    if (lane == 63) lds_wave_tot[wid] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < nw; ++w) {
        int t = lds_wave_tot[w];
        if (w < wid) woff += t;
        tot += t;
    }
    block_total = tot;
    __syncthreads();
    return woff + incl - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(const int* __restrict__ in, int n,
                                                                  int* __restrict__ sums) {
    __shared__ int wtot[SCAN_THREADS / 64];
    const long long base = (long long)blockIdx.x * SCAN_TILE;
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        long long i = base + (long long)k * SCAN_THREADS + threadIdx.x;
        if (i < n) s += in[i];
    }
    s = wave_reduce_add(s);
    if (lane_id() == 0) wtot[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < SCAN_THREADS / 64; ++w) t += wtot[w];
        sums[blockIdx.x] = t;
    }
}

// Single block: exclusive scan of nb block sums in place; total -> sums[nb] and *total.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sums(int* __restrict__ sums, int nb, int* __restrict__ total) {
    __shared__ int wtot[SCAN_THREADS / 64];
    int carry = 0;
    for (int base = 0; base < nb; base += SCAN_THREADS) {
        int i = base + threadIdx.x;
        int v = (i < nb) ? sums[i] : 0;
        int tot;
        int ex = block_excl_scan(v, wtot, tot);
        if (i < nb) sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        sums[nb] = carry;
        if (total) *total = carry;
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const int* in, int* out, int n,
                                                             const int* __restrict__ sums) {
    __shared__ int wtot[SCAN_THREADS / 64];
    // thread t owns SCAN_ITEMS consecutive elements: base + t*SCAN_ITEMS + k
    const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        long long i = base + k;
        v[k] = (i < n) ? in[i] : 0;
        s += v[k];
    }
    int tot;
    int ex = block_excl_scan(s, wtot, tot) + sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        long long i = base + k;
        if (i < n) out[i] = ex;
        ex += v[k];
    }
}

// Up to 8 aligned vector ranges and 16 byte ranges (unaligned heads / tails, small ranges) in one launch.  No
// hipMemsetAsync anywhere on the capturable paths: memset NODES of a HIP graph were observed not to re-execute after an
// explicit stream / device synchronisation between two replays on this ROCm runtime (a 240-byte table kept accumulating,
// hash tables stayed full -> endless probe loops; scripts/debug_stop2.py), kernel nodes always do.
struct FillArgs {
    uint4* p[8]; unsigned long long vecs[8]; unsigned pattern[8];
    unsigned char* bp[16]; unsigned blen[16]; unsigned char bval[16];
    int n, nb;
};
__global__ __launch_bounds__(256) void k_fill_ranges(const FillArgs a) {
    if (blockIdx.x == 0) {   // the byte ranges: a few bytes to a few hundred each
        for (int r = 0; r < a.nb; ++r)
            for (unsigned i = threadIdx.x; i < a.blen[r]; i += 256) a.bp[r][i] = a.bval[r];
    }
    unsigned long long total = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) total += r < a.n ? a.vecs[r] : 0ull;
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (unsigned long long)gridDim.x * 256) {
        unsigned long long off = e;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r >= a.n) break;
            if (off < a.vecs[r]) { a.p[r][off] = make_uint4(a.pattern[r], a.pattern[r], a.pattern[r], a.pattern[r]); break; }
            off -= a.vecs[r];
        }
    }
}

static void fill_flush(FillArgs& a, hipStream_t stream) {
    if (a.n == 0 && a.nb == 0) return;
    unsigned long long total = 0;
    for (int r = 0; r < a.n; ++r) total += a.vecs[r];
    long long blocks = cdiv((long long)total, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_fill_ranges, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    a.n = 0; a.nb = 0;
}

int fill_ranges(const FillRange* ranges, int n, hipStream_t stream) {
    FillArgs a;
    a.n = 0; a.nb = 0;
    auto add_bytes = [&](unsigned char* p, size_t len, unsigned char v) {
        while (len) {
            if (a.nb == 16) fill_flush(a, stream);
            const size_t take = len > 0xffffffffull ? 0xffffffffull : len;
            a.bp[a.nb] = p; a.blen[a.nb] = (unsigned)take; a.bval[a.nb] = v; ++a.nb;
            p += take; len -= take;
        }
    };
    for (int i = 0; i < n; ++i) {
        const FillRange& r = ranges[i];
        if (!r.p || r.bytes == 0) continue;
        unsigned char* p = (unsigned char*)r.p;
        size_t len = r.bytes;
        if (len < 64) { add_bytes(p, len, r.value); continue; }
        const size_t head = (16 - ((uintptr_t)p & 15)) & 15;
        if (head) { add_bytes(p, head, r.value); p += head; len -= head; }
        const size_t tail = len & 15;
        if (len >= 16) {
            if (a.n == 8) fill_flush(a, stream);
            a.p[a.n] = (uint4*)p; a.vecs[a.n] = len / 16; a.pattern[a.n] = 0x01010101u * r.value; ++a.n;
        }
        if (tail) add_bytes(p + (len - tail), tail, r.value);
    }
    fill_flush(a, stream);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

// Single-launch form: a chained (decoupled look-back) scan over the workgroups.  state = one 64-bit word per workgroup,
// flag << 32 | value (flag 1: the workgroup's own sum, 2: the inclusive prefix over workgroups 0..b) + a ticket word, all
// ZEROED by the caller before the launch; every word is written by ONE agent-scope atomic store (payload and flag share the
// 8-byte granule: no fences), read by agent-scope atomic loads, 64 predecessors per poll (lane j looks at workgroup p - j);
// polls are bounded: a scan that gives up raises *status bit 16 instead of spinning.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_chained(const int* in, int* out, int n, unsigned long long* state,
                                                               int* ticket, int* __restrict__ total, int* __restrict__ status) {
    __shared__ int wtot[SCAN_THREADS / 64];
    __shared__ int s_blk, s_ex;
    if (threadIdx.x == 0) s_blk = atomicAdd(ticket, 1);
    __syncthreads();
    const int blk = s_blk;
    const long long base = (long long)blk * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long long i = base + k;
        v[k] = (i < n) ? in[i] : 0;
        s += v[k];
    }
    int tot;
    const int ex_in_block = block_excl_scan(s, wtot, tot);
    const int lane = lane_id();
    if (threadIdx.x < 64) {
        if (lane == 0)
            __hip_atomic_store(state + blk, (blk == 0 ? 2ull : 1ull) << 32 | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ex = 0, p = blk - 1;
        unsigned spins = 0;
        while (p >= 0) {
            const int pp = p - lane;
            const unsigned long long w = pp >= 0 ? __hip_atomic_load(state + pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            const bool ready = pp < 0 || (w >> 32) != 0;
            const unsigned long long not_ready = __ballot(!ready), is_prefix = __ballot(pp >= 0 && (w >> 32) == 2);
            int upto = not_ready ? __ffsll((long long)not_ready) - 1 : 64;
            bool stop = false;
            if (is_prefix) { const int fp = __ffsll((long long)is_prefix) - 1; if (fp < upto) { upto = fp + 1; stop = true; } }
            ex += wave_reduce_add((lane < upto && pp >= 0) ? (int)(unsigned)w : 0);
            if (stop) break;
            if (upto == 0) {
                if (++spins > (1u << 24)) { if (lane == 0 && status) atomicOr(status, 16); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            p -= upto;
        }
        if (lane == 0) {
            if (blk > 0)
                __hip_atomic_store(state + blk, 2ull << 32 | (unsigned)(ex + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ex = ex;
        }
    }
    __syncthreads();
    int ex = s_ex + ex_in_block;
    if (total && threadIdx.x == SCAN_THREADS - 1 && (long long)(blk + 1) * SCAN_TILE >= n && (long long)blk * SCAN_TILE < n)
        *total = ex + s;                                   // the last workgroup that holds elements: its last thread's inclusive sum
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long long i = base + k;
        if (i < n) out[i] = ex;
        ex += v[k];
    }
}

size_t scan_chained_state_words(long long n) { return (size_t)cdiv(n, SCAN_TILE) + 2; }   // 64-bit words: states + ticket

int scan_exclusive_i32_chained(const int* in, int* out, int n, unsigned long long* state_zeroed, int* total, int* status,
                               hipStream_t stream) {
    if (n <= 0) return PBN_OK;                             // *total was zeroed with the state by the caller
    const int nb = cdiv(n, SCAN_TILE);
    hipLaunchKernelGGL(k_scan_chained, dim3(nb), dim3(SCAN_THREADS), 0, stream, in, out, n, state_zeroed,
                       (int*)(state_zeroed + nb), total, status);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

int scan_exclusive_i32(const int* in, int* out, int n, int* tmp, int* total, hipStream_t stream) {
    if (n <= 0) {
        if (total) { const int frc_ = fill_bytes(total, 0, sizeof(int), stream); if (frc_ != PBN_OK) return frc_; }
        return PBN_OK;
    }
    const int nb = cdiv(n, SCAN_TILE);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(SCAN_THREADS), 0, stream, in, n, tmp);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, stream, tmp, nb, total);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(SCAN_THREADS), 0, stream, in, out, n, tmp);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

}  // namespace pbn

extern "C" const char* pbn_version(void) { return "pbnet_hip 0.1 (gfx950)"; }
extern "C" int pbn_last_hip_error(void) { return pbn::g_last_hip_error; }
