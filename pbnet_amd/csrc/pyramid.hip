// pyramid.hip -- the Z-ordered coordinate lineage of pbn_coords_prepare, hand-written and hash-free above the de-duplication:
// what MinkowskiEngine's coordinate manager does behind ME.SparseTensor + MinkUNetBase.forward
// (/root/reference/network/PBNet.py:117,240-247,265-271; network/Mink.py:293-350), re-designed around ONE ordering.
//
//   1. de-duplication of the input rows in a temporary hash table (first occurrence wins, survivors numbered in ascending
//      input order = the external row order); the same launches reduce the bounding box and write each survivor's Z-order
//      key  (batch - b_min) << 3e | interleave(x - x0, y - y0, z - z0)  (x0.. = box minimum rounded down to a multiple of 16,
//      e = bits of the largest extent) together with the digit histograms of the sort;
//   2. stable LSD radix sort of (key, row) -- own kernels: 11-bit digits, one launch per digit with a chained (decoupled
//      look-back) scan over the workgroups, only the digits the box needs (24..33 bits for a room at 2 cm -> 3 launches);
//   3. ONE pass over the sorted rows builds all four coarser levels: in Z-order the children of a voxel are contiguous, so
//      "first row of a level-l voxel" is a comparison with the previous row, the level's numbering is a 4-wide scan
//      (chained like the sort), and coordinates, parent / child tables and the transposed-convolution tables follow -- no
//      hash table per level, no insert, no renumbering;
//   4. kernel maps top-down: levels 2..4 (a few thousand rows) by binary search in their sorted keys; level 1 and level 0
//      (k = 3 and the k = 5 stem map) from the parent level's map -- the neighbour at offset d of a voxel with parity p in
//      its parent is child (p + d) & 1 of the parent's neighbour at (p + d) >> 1: two reads of small, local tables instead
//      of a probe sequence through a table of random 64-byte lines per (row, offset).
// Every output equals the one the hash-table pipeline of coords.hip produces for the same rows (tests/test_pyramid_gpu.py).
#include "coords_dev.h"

namespace pbn {
namespace {

typedef unsigned long long u64;

constexpr int RADIX_BITS = 11, RADIX = 1 << RADIX_BITS, MAX_PASSES = 4;
constexpr int SORT_ITEMS = 16, SORT_TILE = TPB * SORT_ITEMS;       // 4096 keys per workgroup, 1024 per wave (contiguous)
constexpr int PYR_ITEMS = 8, PYR_TILE = TPB * PYR_ITEMS;           // 2048 rows per workgroup, 8 consecutive per thread
constexpr unsigned SPIN_LIMIT = 1u << 24;                          // every chained-scan poll is bounded

// plan words (int32, device): bounding box, tickets, error flag
enum { PL_BMIN = 0, PL_XMIN, PL_YMIN, PL_ZMIN, PL_BMAX, PL_XMAX, PL_YMAX, PL_ZMAX, PL_TICKET0 = 8, PL_TICKET_PYR = 12,
       PL_ERROR = 13, PL_WORDS = 16 };
constexpr int PBN_STATUS_KEY_BITS = 4;     // the box needs more key bits than MAX_PASSES digits hold
constexpr int PBN_STATUS_SPIN = 8;         // a chained scan gave up waiting for a predecessor

struct KeyPlan { int b0, x0, y0, z0, e, nbits; };
__device__ __forceinline__ KeyPlan key_plan(const int* __restrict__ plan) {
    KeyPlan k;
    k.b0 = plan[PL_BMIN];
    k.x0 = plan[PL_XMIN] & ~15; k.y0 = plan[PL_YMIN] & ~15; k.z0 = plan[PL_ZMIN] & ~15;
    const int ext = max(max(plan[PL_XMAX] - k.x0, plan[PL_YMAX] - k.y0), plan[PL_ZMAX] - k.z0);
    k.e = ext > 0 ? 32 - __clz(ext) : 1;
    const int bext = plan[PL_BMAX] - k.b0;
    k.nbits = 3 * k.e + (bext > 0 ? 32 - __clz(bext) : 0);
    return k;
}
__device__ __forceinline__ u64 key_of(const KeyPlan& k, int b, int x, int y, int z) {
    return ((u64)(unsigned)(b - k.b0) << (3 * k.e)) | spread3((unsigned)(x - k.x0)) | (spread3((unsigned)(y - k.y0)) << 1) |
           (spread3((unsigned)(z - k.z0)) << 2);
}

// ---- 1. de-duplication -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_insert_bbox(const int* __restrict__ coords, const int* n_dev, int n_max,
                                                    u64* __restrict__ keys, int* __restrict__ vals, unsigned mask,
                                                    int* __restrict__ slot_of_row, int* __restrict__ status,
                                                    int* __restrict__ plan) {
    __shared__ int s_red[8][TPB / 64];
    const int n = real_n(n_dev, n_max);
    int lo[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[4] = {(int)0x80000000, (int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        if (!in_range(c.x, c.y, c.z, c.w)) atomicOr(status, PBN_STATUS_RANGE);
        slot_of_row[i] = table_insert_min(keys, vals, mask, pack4(c.x, c.y, c.z, c.w), i, status);
        lo[0] = min(lo[0], c.x); lo[1] = min(lo[1], c.y); lo[2] = min(lo[2], c.z); lo[3] = min(lo[3], c.w);
        hi[0] = max(hi[0], c.x); hi[1] = max(hi[1], c.y); hi[2] = max(hi[2], c.z); hi[3] = max(hi[3], c.w);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = min(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = max(hi[a], __shfl_xor(hi[a], o, 64));
        }
        if (lane_id() == 0) { s_red[a][threadIdx.x >> 6] = lo[a]; s_red[4 + a][threadIdx.x >> 6] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const int a = threadIdx.x;
        int v = s_red[a][0];
        for (int w = 1; w < TPB / 64; ++w) v = a < 4 ? min(v, s_red[a][w]) : max(v, s_red[a][w]);
        if (a < 4) atomicMin(&plan[PL_BMIN + a], v); else atomicMax(&plan[PL_BMAX + a - 4], v);
    }
}

// survivors: external-order tables (32- and 64-bit forms), their coordinates, their sort key + row id, and the digit
// histograms of every radix pass (LDS per workgroup, then one atomicAdd per populated bin)
__global__ __launch_bounds__(TPB) void k_unique_keys(const int* __restrict__ coords, const int* __restrict__ slot_of_row,
                                                    const int* __restrict__ first_row, const int* __restrict__ newid,
                                                    const int* n_dev, int n_max, int* __restrict__ uidx32,
                                                    int* __restrict__ inv32, long long* __restrict__ uidx64,
                                                    long long* __restrict__ inv64, int* __restrict__ ucoords,
                                                    const int* __restrict__ plan, int* __restrict__ status,
                                                    u64* __restrict__ sort_keys, int* __restrict__ sort_vals,
                                                    unsigned* __restrict__ ghist) {
    __shared__ unsigned s_hist[MAX_PASSES][RADIX];
    for (int e = threadIdx.x; e < MAX_PASSES * RADIX; e += TPB) (&s_hist[0][0])[e] = 0u;
    __syncthreads();
    const int n = real_n(n_dev, n_max);
    const KeyPlan kp = key_plan(plan);
    if (blockIdx.x == 0 && threadIdx.x == 0 && kp.nbits > MAX_PASSES * RADIX_BITS) atomicOr(status, PBN_STATUS_KEY_BITS);
    const int passes = min(MAX_PASSES, (kp.nbits + RADIX_BITS - 1) / RADIX_BITS);
    for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
        const int first = first_row[i];
        const int id_first = newid[first];
        inv32[i] = id_first;
        inv64[i] = id_first;
        if (first == i) {
            const int4 c = reinterpret_cast<const int4*>(coords)[i];
            uidx32[id_first] = i;
            uidx64[id_first] = i;
            reinterpret_cast<int4*>(ucoords)[id_first] = c;
            const u64 key = key_of(kp, c.x, c.y, c.z, c.w);
            sort_keys[id_first] = key;
            sort_vals[id_first] = id_first;
            for (int p = 0; p < passes; ++p) atomicAdd(&s_hist[p][(unsigned)(key >> (p * RADIX_BITS)) & (RADIX - 1)], 1u);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < passes * RADIX; e += TPB) {
        const unsigned v = (&s_hist[0][0])[e];
        if (v) atomicAdd(&ghist[e], v);
    }
}

// ---- 2. radix sort: one digit per launch, chained scan over the workgroups ---------------------------------------------------
// state[blk][RADIX / 2]: two bins per 64-bit word, each  flag << 30 | count  (flag 1: the workgroup's own count, 2: the
// inclusive prefix over workgroups 0..blk); a word is written by ONE agent-scope atomic store and read by agent-scope
// atomic loads (the payload and its flag share the 8-byte granule: no fences).  Workgroup numbers are drawn from a ticket
// so that every predecessor of a running workgroup has started.
__device__ __forceinline__ u64 ld_state(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_state(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(TPB) void k_sort_pass(const u64* __restrict__ kin, u64* __restrict__ kout,
                                                  const int* __restrict__ vin, int* __restrict__ vout, const int* n_dev,
                                                  int n_max, int pass, int* __restrict__ plan, int* __restrict__ status,
                                                  const unsigned* __restrict__ ghist, u64* __restrict__ state) {
    __shared__ unsigned s_cnt[TPB / 64][RADIX];      // per wave and bin: counts, then exclusive wave bases
    __shared__ unsigned s_base[RADIX];               // where the bin's keys of this workgroup start in the output
    __shared__ unsigned s_wtot[TPB / 64];
    __shared__ int s_blk;
    const KeyPlan kp = key_plan(plan);
    if (pass * RADIX_BITS >= kp.nbits || kp.nbits > MAX_PASSES * RADIX_BITS) return;   // this digit is constant
    const int n = real_n(n_dev, n_max);
    if (n <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_blk = atomicAdd(&plan[PL_TICKET0 + pass], 1);
    for (int e = tid; e < (TPB / 64) * RADIX; e += TPB) (&s_cnt[0][0])[e] = 0u;
    __syncthreads();
    const int blk = s_blk;
    const long long base = (long long)blk * SORT_TILE;
    if (base >= n) return;
    const int shift = pass * RADIX_BITS;

    // A. stable rank of every key inside (wave, bin): a wave owns 1024 consecutive keys, 64 per round
    u64 key[SORT_ITEMS];
    unsigned rank[SORT_ITEMS];
    volatile unsigned* my_cnt = s_cnt[wave];
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const long long i = base + wave * (SORT_ITEMS * 64) + j * 64 + lane;
        const bool ok = i < n;
        key[j] = ok ? kin[i] : ~0ull;
        const unsigned d = (unsigned)(key[j] >> shift) & (RADIX - 1);
        u64 peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const u64 m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        unsigned old = 0;
        const int leader = ok ? __ffsll((long long)peers) - 1 : lane;
        if (ok && lane == leader) {
            old = my_cnt[d];
            my_cnt[d] = old + (unsigned)__popcll(peers);
        }
        old = (unsigned)__shfl((int)old, leader, 64);
        rank[j] = old + (unsigned)__popcll(peers & lt);
    }
    __syncthreads();

    // B. per bin: wave bases, the workgroup's count, the chained scan over the workgroups, the bin's global start
    {
        const int b0 = tid * (RADIX / TPB);            // 8 consecutive bins per thread
        unsigned tot[RADIX / TPB], gsum = 0;
#pragma unroll
        for (int q = 0; q < RADIX / TPB; ++q) {
            unsigned run = 0;
#pragma unroll
            for (int w = 0; w < TPB / 64; ++w) {
                const unsigned c = s_cnt[w][b0 + q];
                s_cnt[w][b0 + q] = run;
                run += c;
            }
            tot[q] = run;
            gsum += ghist[pass * RADIX + b0 + q];
        }
        // publish this workgroup's counts (or, for the first one, its prefixes)
        u64* mine = state + (size_t)blk * (RADIX / 2) + b0 / 2;
        const u64 fl = blk == 0 ? 2ull : 1ull;
#pragma unroll
        for (int q = 0; q < RADIX / TPB; q += 2)
            st_state(mine + q / 2, (fl << 30 | tot[q]) | ((fl << 30 | tot[q + 1]) << 32));
        // exclusive prefix over the workgroups in front of this one
        unsigned excl[RADIX / TPB];
#pragma unroll
        for (int q = 0; q < RADIX / TPB; ++q) excl[q] = 0;
        unsigned open = blk > 0 ? (1u << (RADIX / TPB)) - 1u : 0u;       // bins still looking back
        unsigned spins = 0;
        for (int p = blk - 1; p >= 0 && open; ) {
            const u64* src = state + (size_t)p * (RADIX / 2) + b0 / 2;
            u64 w[RADIX / TPB / 2];
#pragma unroll
            for (int q = 0; q < RADIX / TPB / 2; ++q) w[q] = ld_state(src + q);
            bool all = true;
#pragma unroll
            for (int q = 0; q < RADIX / TPB; ++q)
                if ((open >> q) & 1u) all &= (((unsigned)(w[q / 2] >> ((q & 1) * 32)) >> 30) != 0u);
            if (!all) {
                if (++spins > SPIN_LIMIT) { atomicOr(status, PBN_STATUS_SPIN); break; }
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
#pragma unroll
            for (int q = 0; q < RADIX / TPB; ++q) {
                if (!((open >> q) & 1u)) continue;
                const unsigned v = (unsigned)(w[q / 2] >> ((q & 1) * 32));
                excl[q] += v & 0x3fffffffu;
                if ((v >> 30) == 2u) open &= ~(1u << q);
            }
            --p;
        }
        if (blk > 0) {
#pragma unroll
            for (int q = 0; q < RADIX / TPB; q += 2)
                st_state(mine + q / 2, (2ull << 30 | (excl[q] + tot[q])) | ((2ull << 30 | (excl[q + 1] + tot[q + 1])) << 32));
        }
        // bucket starts: exclusive scan of the global histogram over the bins (8 per thread, then over the threads)
        const unsigned incl = (unsigned)wave_incl_scan_i((int)gsum);
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        unsigned woff = 0;
        for (int w = 0; w < wave; ++w) woff += s_wtot[w];
        unsigned start = woff + incl - gsum;
#pragma unroll
        for (int q = 0; q < RADIX / TPB; ++q) {
            s_base[b0 + q] = start + excl[q];
            start += ghist[pass * RADIX + b0 + q];
        }
    }
    __syncthreads();

    // C. scatter
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const long long i = base + wave * (SORT_ITEMS * 64) + j * 64 + lane;
        if (i < n) {
            const unsigned d = (unsigned)(key[j] >> shift) & (RADIX - 1);
            const unsigned pos = s_base[d] + s_cnt[wave][d] + rank[j];
            kout[pos] = key[j];
            vout[pos] = vin[i];
        }
    }
}

// ---- 3. all levels from the sorted rows ------------------------------------------------------------------------------------
struct PyrOut {
    int* coords[5];
    int* parent_row[4]; int* child_k[4]; int* nbr_down[4]; int* up[4];
    u64* lkeys[3];             // sorted keys of levels 2, 3, 4 (for the binary-search maps)
    int* counts;               // 5 row counts
    long long* perm; long long* inv_perm;
};

__device__ __forceinline__ void write_up_row(int* up, int row, int k, int parent) {
    int4 lo = make_int4(-1, -1, -1, -1), hi = lo;
    int* v = (k < 4) ? &lo.x : &hi.x;
    v[k & 3] = parent;
    reinterpret_cast<int4*>(up)[2 * (size_t)row + 0] = lo;
    reinterpret_cast<int4*>(up)[2 * (size_t)row + 1] = hi;
}

__global__ __launch_bounds__(TPB) void k_pyramid(const u64* __restrict__ keys_a, const u64* __restrict__ keys_b,
                                                const int* __restrict__ vals_a, const int* __restrict__ vals_b,
                                                const int* __restrict__ ucoords, const int* n_dev, int n_max,
                                                int* __restrict__ plan, int* __restrict__ status,
                                                u64* __restrict__ state, const PyrOut o) {
    __shared__ int s_wtot[TPB / 64][4];
    __shared__ int s_ex[4];
    __shared__ int s_blk;
    const int n = real_n(n_dev, n_max);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_blk = atomicAdd(&plan[PL_TICKET_PYR], 1);
    __syncthreads();
    const int blk = s_blk;
    const bool bad = *status != 0;
    if (blk == 0 && tid == 0 && (n <= 0 || bad)) {
        for (int l = 0; l < 5; ++l) o.counts[l] = bad ? -1 : 0;
    }
    if (n <= 0 || bad) return;
    const int base = blk * PYR_TILE;
    if (base >= n) return;
    const KeyPlan kp = key_plan(plan);
    const int passes = (kp.nbits + RADIX_BITS - 1) / RADIX_BITS;       // sorted data lives in buffer (passes & 1)
    const u64* keys = (passes & 1) ? keys_b : keys_a;
    const int* vals = (passes & 1) ? vals_b : vals_a;

    const int i0 = base + tid * PYR_ITEMS;
    int4 c[PYR_ITEMS];
    int r[PYR_ITEMS];
    int4 prev = make_int4(-1, 0, 0, 0);
    if (i0 > 0 && i0 < n) prev = reinterpret_cast<const int4*>(ucoords)[vals[i0 - 1]];
    int f[PYR_ITEMS];                      // bit l-1: row is the first one of its level-l voxel
    int s[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < PYR_ITEMS; ++k) {
        const int i = i0 + k;
        f[k] = 0;
        if (i < n) {
            r[k] = vals[i];
            c[k] = reinterpret_cast<const int4*>(ucoords)[r[k]];
            const int4 p = k == 0 ? prev : c[k - 1];
#pragma unroll
            for (int l = 1; l <= 4; ++l) {
                const bool first = i == 0 || p.x != c[k].x || (p.y >> l) != (c[k].y >> l) || (p.z >> l) != (c[k].z >> l) ||
                                   (p.w >> l) != (c[k].w >> l);
                f[k] |= first ? (1 << (l - 1)) : 0;
                s[l - 1] += first ? 1 : 0;
            }
        }
    }
    // workgroup scan of the four counters
    int incl[4], woff[4], tot[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        incl[l] = wave_incl_scan_i(s[l]);
        if (lane == 63) s_wtot[wave][l] = incl[l];
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        woff[l] = 0; tot[l] = 0;
        for (int w = 0; w < TPB / 64; ++w) { const int t = s_wtot[w][l]; if (w < wave) woff[l] += t; tot[l] += t; }
    }
    // chained scan over the workgroups: two words per workgroup, flag << 62 | a << 31 | b
    if (tid == 0) {
        u64* mine = state + (size_t)blk * 2;
        const u64 fl = blk == 0 ? 2ull : 1ull;
        st_state(mine + 0, fl << 62 | (u64)(unsigned)tot[0] << 31 | (u64)(unsigned)tot[1]);
        st_state(mine + 1, fl << 62 | (u64)(unsigned)tot[2] << 31 | (u64)(unsigned)tot[3]);
        int ex[4] = {0, 0, 0, 0};
        unsigned spins = 0;
        for (int p = blk - 1; p >= 0; ) {
            const u64 a = ld_state(state + (size_t)p * 2), b = ld_state(state + (size_t)p * 2 + 1);
            if ((a >> 62) == 0 || (b >> 62) == 0 || (a >> 62) != (b >> 62)) {
                if (++spins > SPIN_LIMIT) { atomicOr(status, PBN_STATUS_SPIN); break; }
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            ex[0] += (int)((a >> 31) & 0x7fffffffu); ex[1] += (int)(a & 0x7fffffffu);
            ex[2] += (int)((b >> 31) & 0x7fffffffu); ex[3] += (int)(b & 0x7fffffffu);
            if ((a >> 62) == 2) break;
            --p;
        }
        if (blk > 0) {
            st_state(mine + 0, 2ull << 62 | (u64)(unsigned)(ex[0] + tot[0]) << 31 | (u64)(unsigned)(ex[1] + tot[1]));
            st_state(mine + 1, 2ull << 62 | (u64)(unsigned)(ex[2] + tot[2]) << 31 | (u64)(unsigned)(ex[3] + tot[3]));
        }
        for (int l = 0; l < 4; ++l) s_ex[l] = ex[l];
        if (base + PYR_TILE >= n) {                          // the last workgroup publishes the row counts
            o.counts[0] = n;
            for (int l = 0; l < 4; ++l) o.counts[l + 1] = ex[l] + tot[l];
        }
    }
    __syncthreads();
    int id[4];                                              // level-l row of the row in front of this thread's first one
#pragma unroll
    for (int l = 0; l < 4; ++l) id[l] = s_ex[l] + woff[l] + incl[l] - s[l] - 1;
#pragma unroll
    for (int k = 0; k < PYR_ITEMS; ++k) {
        const int i = i0 + k;
        if (i >= n) break;
#pragma unroll
        for (int l = 0; l < 4; ++l) id[l] += (f[k] >> l) & 1;
        const int4 cc = c[k];
        reinterpret_cast<int4*>(o.coords[0])[i] = cc;
        o.perm[i] = r[k];
        o.inv_perm[r[k]] = i;
        const u64 key = keys[i];
        // level l-1 row `row` (this row's ancestor) hangs below level-l row id[l-1]: written by the first row of the ancestor
        int row = i;
#pragma unroll
        for (int l = 1; l <= 4; ++l) {
            const bool mine = l == 1 || ((f[k] >> (l - 2)) & 1);        // first row of the level-(l-1) voxel
            if (mine) {
                const int kk = ((cc.y >> (l - 1)) & 1) + 2 * ((cc.z >> (l - 1)) & 1) + 4 * ((cc.w >> (l - 1)) & 1);
                const int par = id[l - 1];
                o.parent_row[l - 1][row] = par;
                o.child_k[l - 1][row] = kk;
                o.nbr_down[l - 1][(size_t)par * 8 + kk] = row;
                write_up_row(o.up[l - 1], row, kk, par);
            }
            if ((f[k] >> (l - 1)) & 1) {                                 // first row of the level-l voxel: its coordinates
                const int m = ~((1 << l) - 1);
                reinterpret_cast<int4*>(o.coords[l])[id[l - 1]] = make_int4(cc.x, cc.y & m, cc.z & m, cc.w & m);
                if (l >= 2) o.lkeys[l - 2][id[l - 1]] = key >> (3 * l);
            }
            row = id[l - 1];
        }
    }
}

// ---- 4. kernel maps ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cube_offset(int k, int ksize, int x_fastest, int& dx, int& dy, int& dz) {
    const int c0 = (ksize & 1) ? ksize / 2 : 0;
    const int a = k % ksize - c0, b = (k / ksize) % ksize - c0, c = k / (ksize * ksize) - c0;
    dx = x_fastest ? a : c; dy = b; dz = x_fastest ? c : a;
}

// levels 2..4 (k = 3): binary search in the level's sorted keys
struct TopJobs { const int* coords[3]; const u64* lkeys[3]; int* nbr[3]; const int* counts; int n_max, x_fastest; };
__global__ __launch_bounds__(TPB) void k_maps_top(const TopJobs jb, const int* __restrict__ plan) {
    const KeyPlan kp = key_plan(plan);
    long long first[4];
    first[0] = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int c = jb.counts[2 + j]; first[j + 1] = first[j] + (long long)(c > 0 ? min(c, jb.n_max) : 0) * 27; }
    for (long long g = (long long)blockIdx.x * TPB + threadIdx.x; g < first[3]; g += (long long)gridDim.x * TPB) {
        const int j = (g >= first[1]) + (g >= first[2]);
        const int l = 2 + j;
        const long long e = g - first[j];
        const int row = (int)(e / 27), k = (int)(e % 27);
        int dx, dy, dz;
        cube_offset(k, 3, jb.x_fastest, dx, dy, dz);
        const int4 cc = reinterpret_cast<const int4*>(jb.coords[j])[row];
        const int x = cc.y + (dx << l) - kp.x0, y = cc.z + (dy << l) - kp.y0, z = cc.w + (dz << l) - kp.z0;
        int res = -1;
        const int lim = 1 << kp.e;
        if (x >= 0 && y >= 0 && z >= 0 && x < lim && y < lim && z < lim) {
            const u64 want = (((u64)(unsigned)(cc.x - kp.b0) << (3 * kp.e)) | spread3((unsigned)x) | (spread3((unsigned)y) << 1) |
                              (spread3((unsigned)z) << 2)) >> (3 * l);
            const u64* keys = jb.lkeys[j];
            int lo = 0, hi = min(jb.counts[l], jb.n_max);
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (keys[mid] < want) lo = mid + 1; else hi = mid;
            }
            if (lo < min(jb.counts[l], jb.n_max) && keys[lo] == want) res = lo;
        }
        jb.nbr[j][e] = res;
    }
}

// level l (k = 3, and k = 5 on level 0) from the k = 3 map of level l + 1
struct DownJob {
    const int* parent_row; const int* child_k; const int* nbr_down; const int* k3_up;   // tables of (l, l + 1)
    const int* n_dev; int* k3; int* k5; int n_max, x_fastest;
};
__global__ __launch_bounds__(TPB) void k_maps_down(const DownJob jb) {
    const int n = real_n(jb.n_dev, jb.n_max);
    const int per_row = 27 + (jb.k5 ? 125 : 0);
    const long long total = (long long)n * per_row;
    for (long long g = (long long)blockIdx.x * TPB + threadIdx.x; g < total; g += (long long)gridDim.x * TPB) {
        const int row = (int)(g / per_row);
        int k = (int)(g - (long long)row * per_row);
        const bool five = k >= 27;
        if (five) k -= 27;
        int dx, dy, dz;
        cube_offset(k, five ? 5 : 3, jb.x_fastest, dx, dy, dz);
        const int ck = jb.child_k[row], par = jb.parent_row[row];
        const int tx = (ck & 1) + dx, ty = ((ck >> 1) & 1) + dy, tz = ((ck >> 2) & 1) + dz;
        const int px = tx >> 1, py = ty >> 1, pz = tz >> 1;                      // parent's neighbour, each in -1..1
        const int kp3 = jb.x_fastest ? (px + 1) + 3 * (py + 1) + 9 * (pz + 1) : (pz + 1) + 3 * (py + 1) + 9 * (px + 1);
        const int q = jb.k3_up[(size_t)par * 27 + kp3];
        int res = -1;
        if (q >= 0) res = jb.nbr_down[(size_t)q * 8 + ((tx & 1) + 2 * (ty & 1) + 4 * (tz & 1))];
        if (five) jb.k5[(size_t)row * 125 + k] = res; else jb.k3[(size_t)row * 27 + k] = res;
    }
}

}  // namespace

// scratch of the sorted pipeline inside the prepare arena's `sort_temp` block
size_t pyramid_scratch_bytes(int n) {
    const size_t N = (size_t)(n > 0 ? n : 1);
    const size_t sort_blocks = (size_t)cdiv((long long)N, SORT_TILE), pyr_blocks = (size_t)cdiv((long long)N, PYR_TILE);
    size_t b = 0;
    b += align_up(PL_WORDS * sizeof(int), 256);                               // plan
    b += align_up((size_t)MAX_PASSES * RADIX * sizeof(unsigned), 256);       // digit histograms
    b += align_up((size_t)MAX_PASSES * sort_blocks * (RADIX / 2) * sizeof(u64), 256);   // sort scan state
    b += align_up(pyr_blocks * 2 * sizeof(u64), 256);                         // pyramid scan state
    b += 3 * align_up(N * sizeof(u64), 256);                                  // level keys 2..4
    return b + 256;
}

int coords_prepare_sorted(const int32_t* coords, const int32_t* n_dev, int n, int want_k5, int x_fastest, void* arena,
                          const pbn_prepare_layout* P, hipStream_t st) {
    char* A = (char*)arena;
    const pbn_coords_layout* L = &P->pyramid;
    auto I = [&](int64_t o) { return (int32_t*)(A + o); };
    const size_t N = (size_t)(n > 0 ? n : 1);
    const size_t sort_blocks = (size_t)cdiv((long long)N, SORT_TILE), pyr_blocks = (size_t)cdiv((long long)N, PYR_TILE);
    // carve the scratch block
    char* S = A + P->sort_temp;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = S + off; off += align_up(bytes, 256); return p; };
    int* plan = (int*)take(PL_WORDS * sizeof(int));
    unsigned* ghist = (unsigned*)take((size_t)MAX_PASSES * RADIX * sizeof(unsigned));
    u64* sort_state = (u64*)take((size_t)MAX_PASSES * sort_blocks * (RADIX / 2) * sizeof(u64));
    u64* pyr_state = (u64*)take(pyr_blocks * 2 * sizeof(u64));
    u64* lkeys[3];
    for (int j = 0; j < 3; ++j) lkeys[j] = (u64*)take(N * sizeof(u64));
    if (off > (size_t)P->sort_temp_bytes) return PBN_ERR_WORKSPACE;
    int32_t* n_unique = I(P->n_unique);
    int32_t* status = n_unique + 8;
    int32_t* counts = I(L->counts);
    // one fill launch: counts / status zero, temporary table empty, nbr_down "no child", plan (box minima 0x7f.., maxima 0x80..,
    // tickets and error zero), histograms and scan states zero
    {
        const FillRange fr[] = {{A + P->n_unique, 16 * sizeof(int), 0},
                                {A + P->tmp_keys, (size_t)(P->tmp_vals - P->tmp_keys), 0xff},
                                {A + P->tmp_vals, (size_t)(P->unique_index - P->tmp_vals), 0x7f},
                                {A + L->counts, 16 * sizeof(int), 0},
                                {A + L->nbr_down[0], (size_t)(L->vals[0] - L->nbr_down[0]), 0xff},
                                {plan + PL_BMIN, 4 * sizeof(int), 0x7f},
                                {plan + PL_BMAX, 4 * sizeof(int), 0x80},
                                {plan + PL_TICKET0, (PL_WORDS - PL_TICKET0) * sizeof(int), 0},
                                {ghist, (size_t)((char*)lkeys[0] - (char*)ghist), 0}};
        const int frc = fill_ranges(fr, 9, st);
        if (frc != PBN_OK) return frc;
    }
    if (n == 0) return PBN_OK;
    if (!coords) return PBN_ERR_ARG;
    // 1. de-duplication (workspace of the pyramid: slot / first row / new id / block sums)
    Carver cv(A + L->workspace, (size_t)L->workspace_bytes);
    int* slot_of_row = cv.take<int>(N);
    int* first_row = cv.take<int>(N);
    int* newid = cv.take<int>(N);
    int* scan_tmp = cv.take<int>(scan_tmp_ints((long long)N));
    if (!cv.ok) return PBN_ERR_WORKSPACE;
    const int cap = pbn_hash_capacity(n);
    const int gb = (int)(cdiv(n, TPB * 4) < 2048 ? cdiv(n, TPB * 4) : 2048);
    hipLaunchKernelGGL(k_insert_bbox, dim3(gb), dim3(TPB), 0, st, coords, n_dev, n, (u64*)(A + P->tmp_keys), I(P->tmp_vals),
                       (unsigned)cap - 1, slot_of_row, status, plan);
    int rc = coords_number_first(slot_of_row, I(P->tmp_vals), n_dev, n, scan_tmp, newid, first_row, n_unique, status, st);
    if (rc != PBN_OK) return rc;
    u64* keys_a = (u64*)(A + P->sort_keys);
    u64* keys_b = keys_a + N;
    int* vals_a = I(P->sort_vals);
    int* vals_b = vals_a + N;
    hipLaunchKernelGGL(k_unique_keys, dim3(gb), dim3(TPB), 0, st, coords, slot_of_row, first_row, newid, n_dev, n, I(P->uidx32),
                       I(P->inv32), (long long*)(A + P->unique_index), (long long*)(A + P->inverse), I(P->ucoords), plan, status,
                       keys_a, vals_a, ghist);
    // 2. sort: pass p reads buffer p & 1
    for (int p = 0; p < MAX_PASSES; ++p)
        hipLaunchKernelGGL(k_sort_pass, dim3((unsigned)sort_blocks), dim3(TPB), 0, st, (p & 1) ? keys_b : keys_a,
                           (p & 1) ? keys_a : keys_b, (p & 1) ? vals_b : vals_a, (p & 1) ? vals_a : vals_b, n_unique, n, p, plan,
                           status, ghist, sort_state + (size_t)p * sort_blocks * (RADIX / 2));
    // 3. every level
    PyrOut o;
    for (int l = 0; l < 5; ++l) o.coords[l] = I(L->coords[l]);
    for (int l = 0; l < 4; ++l) {
        o.parent_row[l] = I(L->parent_row[l]); o.child_k[l] = I(L->child_k[l]); o.nbr_down[l] = I(L->nbr_down[l]);
        o.up[l] = I(L->up[l]);
    }
    for (int j = 0; j < 3; ++j) o.lkeys[j] = lkeys[j];
    o.counts = counts;
    o.perm = (long long*)(A + P->perm);
    o.inv_perm = (long long*)(A + P->inv_perm);
    hipLaunchKernelGGL(k_pyramid, dim3((unsigned)pyr_blocks), dim3(TPB), 0, st, keys_a, keys_b, vals_a, vals_b, I(P->ucoords),
                       n_unique, n, plan, status, pyr_state, o);
    // 4. maps
    {
        TopJobs tj;
        for (int j = 0; j < 3; ++j) { tj.coords[j] = I(L->coords[2 + j]); tj.lkeys[j] = lkeys[j]; tj.nbr[j] = I(L->k3[2 + j]); }
        tj.counts = counts; tj.n_max = n; tj.x_fastest = x_fastest;
        long long blocks = cdiv((long long)n * 27 / 4 + 1, TPB);      // the coarse levels hold a small fraction of the rows
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_maps_top, dim3((unsigned)blocks), dim3(TPB), 0, st, tj, plan);
        for (int l = 1; l >= 0; --l) {
            DownJob dj;
            dj.parent_row = I(L->parent_row[l]); dj.child_k = I(L->child_k[l]); dj.nbr_down = I(L->nbr_down[l]);
            dj.k3_up = I(L->k3[l + 1]); dj.n_dev = counts + l; dj.k3 = I(L->k3[l]);
            dj.k5 = (l == 0 && want_k5) ? I(L->k5) : nullptr; dj.n_max = n; dj.x_fastest = x_fastest;
            long long bl = cdiv((long long)n * (dj.k5 ? 152 : 27), TPB);
            if (bl > 8192) bl = 8192;
            hipLaunchKernelGGL(k_maps_down, dim3((unsigned)bl), dim3(TPB), 0, st, dj);
        }
    }
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

}  // namespace pbn
