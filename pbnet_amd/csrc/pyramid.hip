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
#ifndef PBN_SORT_ITEMS
#define PBN_SORT_ITEMS 8
#endif
constexpr int SORT_ITEMS = PBN_SORT_ITEMS, SORT_TILE = TPB * SORT_ITEMS;   // keys per workgroup; a wave's are contiguous
constexpr int PYR_ITEMS = 4, PYR_TILE = TPB * PYR_ITEMS;           // 1024 rows per workgroup, striped over the threads
constexpr unsigned SPIN_LIMIT = 1u << 24;                          // every chained-scan poll is bounded
constexpr size_t SORT_LDS_BYTES = (size_t)SORT_TILE * 12 + (size_t)(TPB / 64 + 2) * RADIX * 4;

// plan words (int32, device): bounding box, tickets, error flag
enum { PL_BMIN = 0, PL_XMIN, PL_YMIN, PL_ZMIN, PL_BMAX, PL_XMAX, PL_YMAX, PL_ZMAX, PL_TICKET0 = 8, PL_TICKET_PYR = 12,
       PL_ERROR = 13, PL_STAMPS = 16, PL_WORDS = 64 };
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

// digits of the sort: see "2. radix sort"
struct SortPlan { int passes, dbits; };
__device__ __forceinline__ SortPlan sort_plan(int nbits) {
    SortPlan sp;
    sp.passes = nbits <= 3 * RADIX_BITS ? min(3, (nbits + 7) / 8) : MAX_PASSES;
    sp.dbits = max(8, (nbits + sp.passes - 1) / sp.passes);
    return sp;
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
    const SortPlan sp = sort_plan(min(kp.nbits, MAX_PASSES * RADIX_BITS));
    const int passes = sp.passes;
    const unsigned dmask = (1u << sp.dbits) - 1u;
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
            for (int p = 0; p < passes; ++p) atomicAdd(&s_hist[p][(unsigned)(key >> (p * sp.dbits)) & dmask], 1u);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < passes * RADIX; e += TPB) {
        const unsigned v = (&s_hist[0][0])[e];
        if (v) atomicAdd(&ghist[e], v);
    }
}

// ---- 2. radix sort: one digit per launch, chained scan over the workgroups ---------------------------------------------------
// Digit width: the box needs nbits key bits; they are cut into the fewest digits of at most 11 bits (3 up to 33 bits, 4
// up to 44), each as NARROW as that allows (>= 8 bits): a workgroup's 4096 keys then fall into few enough bins that, after
// a local reorder through LDS, runs of keys leave for consecutive addresses (a 24-bit room: 256 bins, runs of 16 keys =
// whole cache lines; the scattered 8- and 4-byte stores of a direct scatter were 2/3 of a pass).
// state[blk][bins / 2]: two bins per 64-bit word, each  flag << 30 | count  (flag 1: the workgroup's own count, 2: the
// inclusive prefix over workgroups 0..blk); a word is written by ONE agent-scope atomic store and read by agent-scope
// atomic loads (the payload and its flag share the 8-byte granule: no fences).  Workgroup numbers are drawn from a ticket
// so that every predecessor of a running workgroup has started.
__device__ __forceinline__ u64 ld_state(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the same load without the compiler's wait behind it: a batch of them is issued back to back and waited for once
// (ld_state_wait); the agent-scope atomic form above costs one memory round trip PER load (measured: 8 of them 3.4 us)
__device__ __forceinline__ void ld_state_issue(u64& dst, const u64* p) {
    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void ld_state_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void st_state(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifdef PBN_SORT_STAMPS
#define SORT_STAMP(I) if (threadIdx.x == 0 && pass == 0 && (blk == 0 || blk == 20)) plan[PL_STAMPS + (blk ? 16 : 0) + (I)] = (int)__builtin_readcyclecounter();
#else
#define SORT_STAMP(I)
#endif
struct SortShared {
    u64* s_key; int* s_val; unsigned (*s_cnt)[RADIX]; unsigned* s_base; unsigned* s_local;
};

// One digit of the sort for a workgroup, BPT = bins per thread (digit width 8 + log2(BPT) bits) as a compile-time constant:
// every register array below is indexed by constants.
template <int BPT>
__device__ __forceinline__ void sort_pass_body(const u64* __restrict__ kin, u64* __restrict__ kout, const int* __restrict__ vin,
                                               int* __restrict__ vout, int n, int pass, int* __restrict__ plan,
                                               int* __restrict__ status, const unsigned* __restrict__ ghist,
                                               u64* __restrict__ state, const SortShared& sh, int blk, unsigned (*s_wtot)[TPB / 64]) {
    constexpr int DBITS = (BPT == 1 ? 8 : BPT == 2 ? 9 : BPT == 4 ? 10 : 11), R = 1 << DBITS;
    constexpr int W = BPT >= 2 ? BPT / 2 : 1;        // 64-bit state words of one workgroup this thread reads
    constexpr int LB = 32 / W;                       // predecessors per look-back batch (32 loads in flight per thread)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = (long long)blk * SORT_TILE;
    const int shift = pass * DBITS;
    constexpr unsigned dmask = (unsigned)R - 1u;
    SORT_STAMP(0);

    // A. stable rank of every key inside (wave, bin): a wave owns a run of consecutive keys, 64 per round.  Every key and
    // value of the thread is loaded first (independent loads, all in flight together): a load left inside the ranking
    // rounds would wait out its latency once per round.
    u64 key[SORT_ITEMS];
    int val[SORT_ITEMS];
    unsigned rank[SORT_ITEMS];
    unsigned* my_cnt = sh.s_cnt[wave];
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const long long i = base + wave * (SORT_ITEMS * 64) + j * 64 + lane;
        key[j] = i < n ? kin[i] : ~0ull;
        val[j] = i < n ? vin[i] : 0;
    }
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const long long i = base + wave * (SORT_ITEMS * 64) + j * 64 + lane;
        const bool ok = i < n;
        const unsigned d = (unsigned)(key[j] >> shift) & dmask;
        u64 peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < DBITS; ++b) {
            const u64 m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        unsigned old = 0;
        const int leader = ok ? __ffsll((long long)peers) - 1 : lane;
        if (ok && lane == leader) old = atomicAdd(&my_cnt[d], (unsigned)__popcll(peers));     // own row of s_cnt: wave-private
        old = (unsigned)__shfl((int)old, leader, 64);
        rank[j] = old + (unsigned)__popcll(peers & lt);
    }
    SORT_STAMP(1);
    __syncthreads();
    SORT_STAMP(2);

    // B. per bin: wave bases, the workgroup's count, the chained scan over the workgroups, the bin's global and local start
    const int b0 = tid * BPT;                        // BPT consecutive bins per thread (threads past the last bin idle: BPT >= 1)
    unsigned tot[BPT], gh[BPT], gsum = 0, lsum = 0;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        unsigned run = 0;
#pragma unroll
        for (int w = 0; w < TPB / 64; ++w) {
            const unsigned c = sh.s_cnt[w][b0 + q];
            sh.s_cnt[w][b0 + q] = run;
            run += c;
        }
        tot[q] = run;
        gh[q] = ghist[pass * RADIX + b0 + q];
        gsum += gh[q];
        lsum += run;
    }
    // publish this workgroup's counts (or, for the first one, its prefixes); BPT == 1: lanes 2i, 2i+1 share a word
    u64* mine = state + (size_t)blk * (RADIX / 2);
    auto publish = [&](const unsigned* v, u64 flag) {
        if constexpr (BPT == 1) {
            const unsigned other = (unsigned)__shfl_xor((int)v[0], 1, 64);
            if (!(tid & 1)) st_state(mine + tid / 2, (flag << 30 | v[0]) | ((flag << 30 | other) << 32));
        } else {
#pragma unroll
            for (int q = 0; q < BPT; q += 2) st_state(mine + (b0 + q) / 2, (flag << 30 | v[q]) | ((flag << 30 | v[q + 1]) << 32));
        }
    };
    publish(tot, blk == 0 ? 2ull : 1ull);
    SORT_STAMP(3);
    // exclusive prefix over the workgroups in front of this one, in batches of LB predecessors: their words are loaded
    // together (32 independent loads, one round trip: the workgroups of a small sort all start at once, so a walk that
    // waited for one predecessor at a time would be as long as the grid), then consumed nearest first until a prefix flag
    // ends the bin's walk
    unsigned excl[BPT];
#pragma unroll
    for (int q = 0; q < BPT; ++q) excl[q] = 0;
    unsigned open = blk > 0 ? (1u << BPT) - 1u : 0u;                 // bins still looking back
    unsigned spins = 0;
    int p = blk - 1;
    while (p >= 0 && open) {
        u64 wq[LB][W];
#pragma unroll
        for (int d = 0; d < LB; ++d) {
            const int pp = p - d;
#pragma unroll
            for (int q2 = 0; q2 < W; ++q2) {
                wq[d][q2] = 0ull;
                ld_state_issue(wq[d][q2], state + (size_t)(pp >= 0 ? pp : 0) * (RADIX / 2) + b0 / 2 + q2);   // block 0 stands in for pp < 0
            }
        }
        ld_state_wait();
        int used = 0;                                                // predecessors of this batch consumed
#pragma unroll
        for (int d = 0; d < LB; ++d) {
#pragma unroll
            for (int q2 = 0; q2 < W; ++q2) asm volatile("" : "+v"(wq[d][q2]));
            if (p - d < 0 || !open || used != d) continue;
            unsigned v[BPT];
            if constexpr (BPT == 1) v[0] = (unsigned)(wq[d][0] >> ((tid & 1) * 32));
            else {
#pragma unroll
                for (int q = 0; q < BPT; q += 2) { v[q] = (unsigned)wq[d][q / 2]; v[q + 1] = (unsigned)(wq[d][q / 2] >> 32); }
            }
            bool ready = true;
#pragma unroll
            for (int q = 0; q < BPT; ++q)
                if ((open >> q) & 1u) ready &= ((v[q] >> 30) != 0u);
            if (!ready) continue;
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                if (!((open >> q) & 1u)) continue;
                excl[q] += v[q] & 0x3fffffffu;
                if ((v[q] >> 30) == 2u) open &= ~(1u << q);
            }
            used = d + 1;
        }
        p -= used;
        if (used == 0) {
            if (++spins > SPIN_LIMIT) { atomicOr(status, PBN_STATUS_SPIN); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    SORT_STAMP(4);
    if (blk > 0) {
        unsigned incl[BPT];
#pragma unroll
        for (int q = 0; q < BPT; ++q) incl[q] = excl[q] + tot[q];
        publish(incl, 2ull);
    }
    // bucket starts (exclusive scan of the global histogram over the bins) and the bins' starts inside this workgroup
    const unsigned g_incl = (unsigned)wave_incl_scan_i((int)gsum), l_incl = (unsigned)wave_incl_scan_i((int)lsum);
    if (lane == 63) { s_wtot[0][wave] = g_incl; s_wtot[1][wave] = l_incl; }
    __syncthreads();
    unsigned gstart = g_incl - gsum, lstart = l_incl - lsum;
    for (int w = 0; w < wave; ++w) { gstart += s_wtot[0][w]; lstart += s_wtot[1][w]; }
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        sh.s_base[b0 + q] = gstart + excl[q];
        sh.s_local[b0 + q] = lstart;
        gstart += gh[q];
        lstart += tot[q];
    }
    __syncthreads();
    SORT_STAMP(5);

    // C. the workgroup's keys in sorted order through LDS, then out: consecutive threads write consecutive addresses of a bin
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const long long i = base + wave * (SORT_ITEMS * 64) + j * 64 + lane;
        if (i < n) {
            const unsigned d = (unsigned)(key[j] >> shift) & dmask;
            const unsigned lp = sh.s_local[d] + sh.s_cnt[wave][d] + rank[j];
            sh.s_key[lp] = key[j];
            sh.s_val[lp] = val[j];
        }
    }
    __syncthreads();
    SORT_STAMP(6);
    const int here = (int)min((long long)SORT_TILE, (long long)n - base);
    for (int e = tid; e < here; e += TPB) {
        const u64 k = sh.s_key[e];
        const unsigned d = (unsigned)(k >> shift) & dmask;
        const unsigned pos = sh.s_base[d] + ((unsigned)e - sh.s_local[d]);
        kout[pos] = k;
        vout[pos] = sh.s_val[e];
    }
    SORT_STAMP(7);
}

__global__ __launch_bounds__(TPB) void k_sort_pass(const u64* __restrict__ kin, u64* __restrict__ kout,
                                                  const int* __restrict__ vin, int* __restrict__ vout, const int* n_dev,
                                                  int n_max, int pass, int* __restrict__ plan, int* __restrict__ status,
                                                  const unsigned* __restrict__ ghist, u64* __restrict__ state) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_smem[];      // SORT_LDS_BYTES
    __shared__ unsigned s_wtot[2][TPB / 64];
    __shared__ int s_blk;
    SortShared sh;
    sh.s_key = reinterpret_cast<u64*>(sort_smem);                                   // SORT_TILE keys in the workgroup's sorted order
    sh.s_val = reinterpret_cast<int*>(sh.s_key + SORT_TILE);
    sh.s_cnt = reinterpret_cast<unsigned (*)[RADIX]>(sh.s_val + SORT_TILE);         // per wave and bin: counts, then wave bases
    sh.s_base = &sh.s_cnt[TPB / 64][0];          // where the bin's keys of this workgroup start in the output
    sh.s_local = sh.s_base + RADIX;              // ... and in the workgroup's own sorted order
    const KeyPlan kp = key_plan(plan);
    if (kp.nbits > MAX_PASSES * RADIX_BITS) return;
    const SortPlan sp = sort_plan(kp.nbits);
    if (pass >= sp.passes) return;                    // this digit does not exist
    const int n = real_n(n_dev, n_max);
    if (n <= 0) return;
    const int tid = threadIdx.x;
    if (tid == 0) s_blk = atomicAdd(&plan[PL_TICKET0 + pass], 1);
    for (int e = tid; e < (TPB / 64) * RADIX; e += TPB) (&sh.s_cnt[0][0])[e] = 0u;
    __syncthreads();
    const int blk = s_blk;
    if ((long long)blk * SORT_TILE >= n) return;
    switch (sp.dbits) {
        case 8: sort_pass_body<1>(kin, kout, vin, vout, n, pass, plan, status, ghist, state, sh, blk, s_wtot); break;
        case 9: sort_pass_body<2>(kin, kout, vin, vout, n, pass, plan, status, ghist, state, sh, blk, s_wtot); break;
        case 10: sort_pass_body<4>(kin, kout, vin, vout, n, pass, plan, status, ghist, state, sh, blk, s_wtot); break;
        default: sort_pass_body<8>(kin, kout, vin, vout, n, pass, plan, status, ghist, state, sh, blk, s_wtot); break;
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
    const int passes = sort_plan(kp.nbits).passes;                     // sorted data lives in buffer (passes & 1)
    const u64* keys = (passes & 1) ? keys_b : keys_a;
    const int* vals = (passes & 1) ? vals_b : vals_a;

    // Rows are taken STRIPED (round r: row base + r * 256 + thread): every load and store of a wave touches consecutive
    // rows.  All loads first (independent), then per round the four "first row of its level-l voxel" flags and their scan.
    int r[PYR_ITEMS], rp = 0;
    int4 c[PYR_ITEMS];
    u64 key[PYR_ITEMS];
#pragma unroll
    for (int k = 0; k < PYR_ITEMS; ++k) {
        const int i = base + k * TPB + tid;
        r[k] = i < n ? vals[i] : 0;
        key[k] = i < n ? keys[i] : 0ull;
    }
    const int ip = base + wave * 64 - 1;                                // lane 0 of a wave also needs the row in front of round 0 ...
    int rpk[PYR_ITEMS];
#pragma unroll
    for (int k = 0; k < PYR_ITEMS; ++k) rpk[k] = (lane == 0 && ip + k * TPB >= 0 && ip + k * TPB < n) ? vals[ip + k * TPB] : -1;
    (void)rp;
#pragma unroll
    for (int k = 0; k < PYR_ITEMS; ++k) c[k] = reinterpret_cast<const int4*>(ucoords)[r[k]];
    int4 cp[PYR_ITEMS];
#pragma unroll
    for (int k = 0; k < PYR_ITEMS; ++k)
        cp[k] = rpk[k] >= 0 ? reinterpret_cast<const int4*>(ucoords)[rpk[k]] : make_int4(-1, 0, 0, 0);
    int f[PYR_ITEMS], wid[PYR_ITEMS][4], carry[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < PYR_ITEMS; ++k) {
        const int i = base + k * TPB + tid;
        int4 p;
        p.x = __shfl_up(c[k].x, 1, 64); p.y = __shfl_up(c[k].y, 1, 64); p.z = __shfl_up(c[k].z, 1, 64); p.w = __shfl_up(c[k].w, 1, 64);
        if (lane == 0) p = cp[k];
        f[k] = 0;
        int inc[4];
#pragma unroll
        for (int l = 1; l <= 4; ++l) {
            const bool first = i < n && (i == 0 || p.x != c[k].x || (p.y >> l) != (c[k].y >> l) || (p.z >> l) != (c[k].z >> l) ||
                                         (p.w >> l) != (c[k].w >> l));
            f[k] |= first ? (1 << (l - 1)) : 0;
            inc[l - 1] = wave_incl_scan_i(first ? 1 : 0);
            if (lane == 63) s_wtot[wave][l - 1] = inc[l - 1];
        }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            int woff = 0, tot = 0;
            for (int w = 0; w < TPB / 64; ++w) { const int t = s_wtot[w][l]; if (w < wave) woff += t; tot += t; }
            wid[k][l] = carry[l] + woff + inc[l];            // rows of level l+1 up to and including this row, inside the workgroup
            carry[l] += tot;
        }
        __syncthreads();
    }
    const int* tot = carry;
    // chained scan over the workgroups: two words per workgroup, flag << 62 | a << 31 | b.  Wave 0 looks back 64
    // predecessors at a time (lane j polls workgroup p - j): the nearest prefix flag ends the walk, the counts in front
    // of it are summed across the lanes.
    if (wave == 0) {
        u64* mine = state + (size_t)blk * 2;
        if (lane == 0) {
            const u64 fl = blk == 0 ? 2ull : 1ull;
            st_state(mine + 0, fl << 62 | (u64)(unsigned)tot[0] << 31 | (u64)(unsigned)tot[1]);
            st_state(mine + 1, fl << 62 | (u64)(unsigned)tot[2] << 31 | (u64)(unsigned)tot[3]);
        }
        int ex[4] = {0, 0, 0, 0};
        unsigned spins = 0;
        int p = blk - 1;
        while (p >= 0) {
            const int pp = p - lane;
            u64 a = 0, b = 0;
            if (pp >= 0) { a = ld_state(state + (size_t)pp * 2); b = ld_state(state + (size_t)pp * 2 + 1); }
            const bool ready = pp < 0 || ((a >> 62) != 0 && (a >> 62) == (b >> 62));
            const u64 not_ready = __ballot(!ready);
            const u64 is_prefix = __ballot(pp >= 0 && ready && (a >> 62) == 2);
            // lanes usable this round: 0 .. first - 1 all ready, up to and including the first prefix
            int upto = not_ready ? __ffsll((long long)not_ready) - 1 : 64;      // first lane that is not ready
            int stop = 64;
            if (is_prefix) { const int fp = __ffsll((long long)is_prefix) - 1; if (fp < upto) { stop = fp; upto = fp + 1; } }
            const bool take = lane < upto && pp >= 0;
            int v[4] = {take ? (int)((a >> 31) & 0x7fffffffu) : 0, take ? (int)(a & 0x7fffffffu) : 0,
                        take ? (int)((b >> 31) & 0x7fffffffu) : 0, take ? (int)(b & 0x7fffffffu) : 0};
#pragma unroll
            for (int l = 0; l < 4; ++l) ex[l] += wave_reduce_add(v[l]);
            if (stop < 64) break;                          // a prefix was consumed
            if (upto == 0) {
                if (++spins > SPIN_LIMIT) { if (lane == 0) atomicOr(status, PBN_STATUS_SPIN); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            p -= upto;
        }
        if (lane == 0) {
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
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PYR_ITEMS; ++k) {
        const int i = base + k * TPB + tid;
        if (i >= n) continue;
        int id[4];                                              // the row's ancestor at levels 1..4
#pragma unroll
        for (int l = 0; l < 4; ++l) id[l] = s_ex[l] + wid[k][l] - 1;
        const int4 cc = c[k];
        reinterpret_cast<int4*>(o.coords[0])[i] = cc;
        o.perm[i] = r[k];
        o.inv_perm[r[k]] = i;
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
                if (l >= 2) o.lkeys[l - 2][id[l - 1]] = key[k] >> (3 * l);
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

// level l (k = 3, and k = 5 on level 0) from the k = 3 map of level l + 1.  One WAVE per parent voxel: the child tables of
// the parent's 27 neighbours (27 x 8 rows) are gathered once into LDS, then every entry of every child's map (children are
// contiguous rows: their maps are one contiguous block of the output) is a lookup in that table.
struct DownJob {
    const int* nbr_down; const int* k3_up;            // child table of (l, l + 1), k = 3 map of level l + 1
    const int* n_parents_dev; int* k3; int* k5; int n_max, x_fastest;
};
__global__ __launch_bounds__(TPB) void k_maps_down(const DownJob jb) {
    __shared__ int s_tab[TPB / 64][27 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_par = real_n(jb.n_parents_dev, jb.n_max);
    int* tab = s_tab[wave];
    for (int par = blockIdx.x * (TPB / 64) + wave; par < n_par; par += gridDim.x * (TPB / 64)) {
        const int q = lane < 27 ? jb.k3_up[(size_t)par * 27 + lane] : -1;
#pragma unroll
        for (int e0 = 0; e0 < 27 * 8; e0 += 64) {                        // every lane takes part in the shuffle
            const int e = e0 + lane;
            const int qq = __shfl(q, (e >> 3) < 27 ? (e >> 3) : 0, 64);
            if (e < 27 * 8) tab[e] = qq >= 0 ? jb.nbr_down[(size_t)qq * 8 + (e & 7)] : -1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the children are consecutive rows (Z-order), so their maps are ONE contiguous block of each output: the lanes
        // stride over (child, offset) entries; child j of the block has parity cks[j]
        int first = -1, nchild = 0;
        unsigned cks = 0;                                               // 3 bits per present child, in row order
#pragma unroll
        for (int ck = 0; ck < 8; ++ck) {
            const int row = tab[13 * 8 + ck];                           // the parent is its own neighbour 13
            if (row >= 0) { if (first < 0) first = row; cks |= (unsigned)ck << (3 * nchild); ++nchild; }
        }
        auto entry = [&](int j, int k, int ksize) -> int {
            const int ck = (int)(cks >> (3 * j)) & 7;
            int dx, dy, dz;
            cube_offset(k, ksize, jb.x_fastest, dx, dy, dz);
            const int tx = (ck & 1) + dx, ty = ((ck >> 1) & 1) + dy, tz = ((ck >> 2) & 1) + dz;
            const int px = tx >> 1, py = ty >> 1, pz = tz >> 1;          // parent's neighbour, each in -1..1
            const int kp3 = jb.x_fastest ? (px + 1) + 3 * (py + 1) + 9 * (pz + 1) : (pz + 1) + 3 * (py + 1) + 9 * (px + 1);
            return tab[kp3 * 8 + ((tx & 1) + 2 * (ty & 1) + 4 * (tz & 1))];
        };
        for (int e = lane; e < nchild * 27; e += 64) {
            const int j = e / 27;
            jb.k3[(size_t)first * 27 + e] = entry(j, e - j * 27, 3);
        }
        if (jb.k5)
            for (int e = lane; e < nchild * 125; e += 64) {
                const int j = e / 125;
                jb.k5[(size_t)first * 125 + e] = entry(j, e - j * 125, 5);
            }
        __builtin_amdgcn_wave_barrier();
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
    const int gi = (int)(cdiv(n, TPB) < 8192 ? cdiv(n, TPB) : 8192);        // one row per thread: the insert is a chain of atomics
    hipLaunchKernelGGL(k_insert_bbox, dim3(gi), dim3(TPB), 0, st, coords, n_dev, n, (u64*)(A + P->tmp_keys), I(P->tmp_vals),
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
    static const bool lds_attr = (hipFuncSetAttribute((const void*)k_sort_pass, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      (int)SORT_LDS_BYTES) == hipSuccess);
    if (!lds_attr) return PBN_ERR_HIP;
    for (int p = 0; p < MAX_PASSES; ++p)
        hipLaunchKernelGGL(k_sort_pass, dim3((unsigned)sort_blocks), dim3(TPB), SORT_LDS_BYTES, st, (p & 1) ? keys_b : keys_a,
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
            dj.nbr_down = I(L->nbr_down[l]); dj.k3_up = I(L->k3[l + 1]); dj.n_parents_dev = counts + l + 1; dj.k3 = I(L->k3[l]);
            dj.k5 = (l == 0 && want_k5) ? I(L->k5) : nullptr; dj.n_max = n; dj.x_fastest = x_fastest;
            long long bl = cdiv((long long)n, (TPB / 64) * 2);         // a level holds at most half as many parents as rows... of its children
            if (bl > 8192) bl = 8192;
            if (bl < 1) bl = 1;
            hipLaunchKernelGGL(k_maps_down, dim3((unsigned)bl), dim3(TPB), 0, st, dj);
        }
    }
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

}  // namespace pbn
