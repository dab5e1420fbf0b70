// cluster.hip -- point-wise binarization + neighbour clustering on gfx950 (wave64).
//
// Replaces the Solver pipeline of /root/reference/lib/PB_lib/src/pbnet/{cluster.cu,binary.cu,binary_cuda_functions.cu}
// with a batch-of-segments pipeline that never leaves the device and never materialises the adjacency:
//
//   seg offsets -> uniform grid hash (cell = 0.5625 r: any two points of one cell are within r) -> cell-sorted slab
//   -> exact r-ball count over the 5x5x5 neighbourhood (+HP flag) -> lock-free union-find (min-index roots) driven by
//   CELLS, not pairs: HPs of a cell are chained to the cell's representative, two cells need ONE witnessed edge
//   -> border LP = max adjacent seed (one witness per cell)
//   -> sizes -> keep flags -> scan = final ids -> exact NN for unassigned points -> ordered members + centres
//
// Determinism: every output is a function of the input only (atomics are used on integers where the result is
// order independent: min, max, add).  Floating point follows oracle/pb_cluster_ref.c exactly: unfused binary32
// d2 = (dx*dx+dy*dy)+dz*dz, d2 <= r*r, sequential running mean with IEEE division.  This file MUST be compiled
// with -ffp-contract=off (see csrc/Makefile); the pragma below is a second line of defence.
#include "pbn_common.h"

#pragma clang fp contract(off)

namespace pbn {
namespace {

constexpr int TPB = 256;
constexpr unsigned long long EMPTY_KEY = ~0ULL;
constexpr int CELL_BIAS = 32768;
constexpr float CELL_CLAMP = 32764.0f;  // +-2 neighbour offsets must stay inside 16 bits
constexpr int CELL_R = 2;                // cells of side 0.5625 r: r-neighbours are at most 2 cells away per axis
constexpr int BIG = 0x7f7f7f7f;  // memset(0x7f) pattern: "no seed yet"

// lib/PB_lib/src/pbnet/binary.cu:229 ("mean count from HAIS"), indexed by sem-2
__constant__ float c_mean_count[18] = {3917.0f, 12056.0f, 2303.0f, 8331.0f, 3948.0f, 3166.0f, 5629.0f, 11719.0f, 1003.0f,
                                       3317.0f, 4912.0f,  10221.0f, 3889.0f, 4136.0f, 2120.0f, 945.0f,  3967.0f, 2589.0f};

// binary_cuda_functions.cu:305-308 in the fixed unfused form
__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    const float a = __fmul_rn(dx, dx), b = __fmul_rn(dy, dy), c = __fmul_rn(dz, dz);
    return __fadd_rn(__fadd_rn(a, b), c);
}

__device__ __forceinline__ int cls_of(int sem) { return min(max(sem - 2, 0), 17); }  // index into 18-entry tables

__device__ __forceinline__ int cell_coord(float v, float inv_cell) {
    float f = floorf(v * inv_cell);
    f = fminf(fmaxf(f, -CELL_CLAMP), CELL_CLAMP);  // also maps NaN to -CELL_CLAMP
    return (int)f;
}

__device__ __forceinline__ unsigned long long pack_key(int seg, int cx, int cy, int cz) {
    return ((unsigned long long)(unsigned)seg << 48) | ((unsigned long long)(unsigned)(cx + CELL_BIAS) << 32) |
           ((unsigned long long)(unsigned)(cy + CELL_BIAS) << 16) | (unsigned long long)(unsigned)(cz + CELL_BIAS);
}

__device__ __forceinline__ int hash_lookup(const unsigned long long* __restrict__ hkeys, unsigned hmask,
                                           unsigned long long key) {
    unsigned h = hash64(key) & hmask;
    for (unsigned probes = 0; probes <= hmask; ++probes) {     // bounded by the capacity: a full / corrupted table ends the probe
        unsigned long long k = hkeys[h];
        if (k == key) return (int)h;
        if (k == EMPTY_KEY) return -1;
        h = (h + 1) & hmask;
    }
    return -1;
}

// ---- segment offsets: exclusive scan of seg_len by one wave; status != 0 when the lengths do not add up -------
__global__ void k_seg_offsets(const int* __restrict__ seg_len, int n_seg, int n, int capacity, int* __restrict__ seg_off,
                              int* __restrict__ status) {
    const int lane = lane_id();
    int carry = 0;
    int bad = 0;
    for (int base = 0; base < n_seg; base += 64) {
        int i = base + lane;
        int v = (i < n_seg) ? seg_len[i] : 0;
        if (v < 0) bad = 1;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (i < n_seg) seg_off[i] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
    bad = __any(bad);
    if (lane == 0) {
        seg_off[n_seg] = carry;
        if ((capacity ? carry > n : carry != n) || bad) atomicOr(status, 1);
    }
}

// Point count of a launch: `cap` sizes the grids and the workspace; in capacity mode (pbn_binary_cluster flags bit 1) the
// number of points that exist is the sum of the segment lengths, read from device memory (seg_off[n_seg])
struct NRef {
    int cap;
    const int* dev;
    __device__ __forceinline__ int get() const { return dev ? min(cap, *dev) : cap; }
};

__device__ __forceinline__ int find_segment(const int* __restrict__ seg_off, int n_seg, int i) {
    int lo = 0, hi = n_seg;  // largest s with seg_off[s] <= i
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (seg_off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// ---- grid hash build: one slot per occupied (segment, cell); per-slot population ------------------------------
__global__ __launch_bounds__(TPB) void k_cell_insert(const float* __restrict__ off_xyz, const int* __restrict__ sem,
                                                    const int* __restrict__ seg_off, int n_seg, NRef n_ref, float inv_cell,
                                                    unsigned long long* __restrict__ hkeys, int* __restrict__ hcount,
                                                    unsigned hmask, int* __restrict__ slot_of_pt,
                                                    int* __restrict__ seg_of_pt, int* __restrict__ status) {
    const int n = n_ref.get();
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int s = sem[i];
    if (s < 2 || s > 19) atomicOr(status, 2);
    const int seg = find_segment(seg_off, n_seg, i);
    seg_of_pt[i] = seg;
    const float x = off_xyz[3 * i + 0], y = off_xyz[3 * i + 1], z = off_xyz[3 * i + 2];
    const unsigned long long key = pack_key(seg, cell_coord(x, inv_cell), cell_coord(y, inv_cell), cell_coord(z, inv_cell));
    unsigned h = hash64(key) & hmask;
    bool placed = false;
    for (unsigned probes = 0; probes <= hmask; ++probes) {     // bounded by the capacity (2x the points: never full unless corrupted)
        unsigned long long prev = atomicCAS(&hkeys[h], EMPTY_KEY, key);
        if (prev == EMPTY_KEY || prev == key) { placed = true; break; }
        h = (h + 1) & hmask;
    }
    if (!placed) { atomicOr(status, 4); h = 0; }               // status != 0: the launch reports invalid input (n_clusters = -1)
    atomicAdd(&hcount[h], 1);
    slot_of_pt[i] = (int)h;
}

// ---- scatter points into the cell-sorted slab: float4(x, y, z, bits(original index)) --------------------------
__global__ __launch_bounds__(TPB) void k_cell_scatter(const float* __restrict__ off_xyz, NRef n_ref,
                                                     const int* __restrict__ slot_of_pt, const int* __restrict__ hstart,
                                                     int* __restrict__ hcursor, const int* __restrict__ seg_of_pt,
                                                     float4* __restrict__ spt, int* __restrict__ sseg,
                                                     int* __restrict__ sslot) {
    const int n = n_ref.get();
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int slot = slot_of_pt[i];
    const int pos = hstart[slot] + atomicAdd(&hcursor[slot], 1);
    spt[pos] = make_float4(off_xyz[3 * i + 0], off_xyz[3 * i + 1], off_xyz[3 * i + 2], __int_as_float(i));
    sseg[pos] = seg_of_pt[i];
    sslot[pos] = slot;
}

__device__ __forceinline__ bool cell_trusted(int cx, int cy, int cz) {
    // a clamped border cell collects arbitrarily distant points: its members are NOT known to be within r of each other
    const int lim = (int)CELL_CLAMP;
    return cx > -lim && cx < lim && cy > -lim && cy < lim && cz > -lim && cz < lim;
}

// Enumerate the 5x5x5 cells around point (x,y,z) of segment seg; f(beg, end, slot, trusted, same) per occupied cell.
// The 125 cells of a point are dealt round-robin to the `stride` lanes that share the point (lane `first` takes cells
// first, first+stride, ...): the hash probes, not the distance tests, are what these sweeps spend their time on.
template <typename F>
__device__ __forceinline__ void for_each_neighbour_cell(float x, float y, float z, int seg, float inv_cell,
                                                        const unsigned long long* __restrict__ hkeys, unsigned hmask,
                                                        const int* __restrict__ hstart, const int* __restrict__ hcount,
                                                        int first, int stride, F&& f) {
    const int cx = cell_coord(x, inv_cell), cy = cell_coord(y, inv_cell), cz = cell_coord(z, inv_cell);
    constexpr int W = 2 * CELL_R + 1;
    for (int c = first; c < W * W * W; c += stride) {
        const int dz = c / (W * W) - CELL_R, dy = (c / W) % W - CELL_R, dx = c % W - CELL_R;
        const int slot = hash_lookup(hkeys, hmask, pack_key(seg, cx + dx, cy + dy, cz + dz));
        if (slot < 0) continue;
        const int beg = hstart[slot];
        f(beg, beg + hcount[slot], slot, cell_trusted(cx + dx, cy + dy, cz + dz), (dx | dy | dz) == 0);
    }
}

// ---- a10/a12: exact r-ball population (self excluded).  NB_Q adjacent lanes share one cell-sorted position: per round
// every lane probes ONE of the 125 cells, then the group walks the found cells together, lane q taking candidates
// beg+q, beg+q+NB_Q, ... (shifted coordinates pile up near instance centres: single cells hold hundreds of points)
constexpr int NB_Q = 8;
__device__ __forceinline__ bool group_any(bool pred) {
    const unsigned long long m = __ballot(pred);
    return ((m >> (threadIdx.x & 63 & ~(NB_Q - 1))) & ((1ULL << NB_Q) - 1ULL)) != 0ULL;
}
// f(beg, end, slot, trusted, same) is called by ALL lanes of the group for every occupied cell (group-uniform arguments)
template <typename F>
__device__ __forceinline__ void for_each_neighbour_cell_group(float x, float y, float z, int seg, float inv_cell,
                                                              const unsigned long long* __restrict__ hkeys, unsigned hmask,
                                                              const int* __restrict__ hstart,
                                                              const int* __restrict__ hcount, int q, F&& f) {
    const int cx = cell_coord(x, inv_cell), cy = cell_coord(y, inv_cell), cz = cell_coord(z, inv_cell);
    constexpr int W = 2 * CELL_R + 1;
    const int gbase = (threadIdx.x & 63) & ~(NB_Q - 1);
    for (int r = 0; r < W * W * W; r += NB_Q) {
        const int c = r + q;
        int slot = -1, beg = 0, end = 0;
        if (c < W * W * W) {
            const int dz = c / (W * W) - CELL_R, dy = (c / W) % W - CELL_R, dx = c % W - CELL_R;
            slot = hash_lookup(hkeys, hmask, pack_key(seg, cx + dx, cy + dy, cz + dz));
            if (slot >= 0) { beg = hstart[slot]; end = beg + hcount[slot]; }
        }
#pragma unroll
        for (int k = 0; k < NB_Q; ++k) {
            const int sk = __shfl(slot, gbase + k, 64);
            if (sk < 0) continue;
            const int bk = __shfl(beg, gbase + k, 64), ek = __shfl(end, gbase + k, 64);
            const int ck = r + k;
            const int dz = ck / (W * W) - CELL_R, dy = (ck / W) % W - CELL_R, dx = ck % W - CELL_R;
            f(bk, ek, sk, cell_trusted(cx + dx, cy + dy, cz + dz), (dx | dy | dz) == 0);
        }
    }
}
__device__ __forceinline__ int group_sum(int v) {
#pragma unroll
    for (int o = 1; o < NB_Q; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int group_max(int v) {
#pragma unroll
    for (int o = 1; o < NB_Q; o <<= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__global__ __launch_bounds__(TPB) void k_count(const float4* __restrict__ spt, const int* __restrict__ sseg, NRef n_ref,
                                              float inv_cell, float r2, const unsigned long long* __restrict__ hkeys,
                                              unsigned hmask, const int* __restrict__ hstart,
                                              const int* __restrict__ hcount, int* __restrict__ den) {
    // The 8 points of a wave are neighbours in the cell-sorted slab: in 82 % of the waves (90 % of the tests) they share a
    // cell, hence the 125 candidate ranges.  Then the wave loads every candidate ONCE -- 64 per load, one per lane,
    // through a per-wave LDS line, the next 64 already in flight -- and each point's 8 lanes test them from there instead
    // of 8 groups fetching the same candidates from L2; waves whose groups look at different cells keep the per-group
    // walk.  Integer counts: order-free.  Bench scene (64 k points, 3e8 tests): 298 -> 274 us, of which the hash probes are
    // 30.  Also measured, none of them better: packed fp32 tests (298), a per-cell table of the 125 neighbour slots (269, + 16
    // to build and 500 B of workspace per point), cell-by-cell passes so that every test takes the shared path (378), a
    // cell-tiled kernel with candidates through the scalar cache (477).
    __shared__ float4 s_cand[TPB / 64][64];
    const int n = n_ref.get();
    if (n <= 0) return;
    const long long t = (long long)blockIdx.x * TPB + threadIdx.x;
    // capacity mode: a wave whose first point is past the count has nothing to do (round 5: its lanes used to be folded onto point
    // n - 1 and repeat that point's whole neighbourhood walk -- 0.27 -> 1.9 ms at 7.5 x capacity); only the wave that straddles
    // the count keeps its idle lane groups, for the shuffles
    if ((t - (threadIdx.x & 63)) / NB_Q >= n) return;
    int p = (int)(t / NB_Q);
    const int q = (int)(t % NB_Q);
    const bool live = p < n;
    if (!live) p = n - 1;  // keep the lane group complete for the shuffles
    const float4 me = spt[p];
    const int seg = sseg[p];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gbase = lane & ~(NB_Q - 1);
    float4* line = s_cand[wave];
    int cnt = 0;
    const int cx = cell_coord(me.x, inv_cell), cy = cell_coord(me.y, inv_cell), cz = cell_coord(me.z, inv_cell);
    constexpr int W = 2 * CELL_R + 1;
    for (int r = 0; r < W * W * W; r += NB_Q) {
        const int c = r + q;
        int slot = -1, beg = 0, end = 0;
        if (c < W * W * W) {
            const int dz = c / (W * W) - CELL_R, dy = (c / W) % W - CELL_R, dx = c % W - CELL_R;
            slot = hash_lookup(hkeys, hmask, pack_key(seg, cx + dx, cy + dy, cz + dz));
            if (slot >= 0) { beg = hstart[slot]; end = beg + hcount[slot]; }
        }
        // do all 8 groups of the wave look at the same 8 ranges?
        const bool same = beg == __shfl(beg, q, 64) && end == __shfl(end, q, 64);
        if (__all(same)) {
#pragma unroll 1
            for (int k = 0; k < NB_Q; ++k) {
                const int bk = __shfl(beg, k, 64), ek = __shfl(end, k, 64);   // wave-uniform
                float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bk + lane < ek) nxt = spt[bk + lane];
                for (int cb = bk; cb < ek; cb += 64) {
                    line[lane] = nxt;
                    if (cb + 64 + lane < ek) nxt = spt[cb + 64 + lane];      // lands behind this chunk's tests
                    const int m = min(64, ek - cb);
#pragma unroll
                    for (int i = 0; i < 64 / NB_Q; ++i) {
                        const int idx = q + NB_Q * i;
                        if (idx < m) {
                            const float4 v = line[idx];
                            cnt += (sqdist(me.x, me.y, me.z, v.x, v.y, v.z) <= r2) ? 1 : 0;
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NB_Q; ++k) {
                const int sk = __shfl(slot, gbase + k, 64);
                if (sk < 0) continue;
                const int bk = __shfl(beg, gbase + k, 64), ek = __shfl(end, gbase + k, 64);
                int j = bk + q;
                for (; j + NB_Q < ek; j += 2 * NB_Q) {  // two independent loads in flight
                    const float4 q0 = spt[j], q1 = spt[j + NB_Q];
                    cnt += (sqdist(me.x, me.y, me.z, q0.x, q0.y, q0.z) <= r2) ? 1 : 0;
                    cnt += (sqdist(me.x, me.y, me.z, q1.x, q1.y, q1.z) <= r2) ? 1 : 0;
                }
                for (; j < ek; j += NB_Q) {
                    const float4 q0 = spt[j];
                    cnt += (sqdist(me.x, me.y, me.z, q0.x, q0.y, q0.z) <= r2) ? 1 : 0;
                }
            }
        }
    }
    cnt = group_sum(cnt);
    if (live && q == 0) den[__float_as_int(me.w)] = cnt - 1;  // binary_cuda_functions.cu:88
}

// tag the HP flag into bit 31 of the index word of the sorted slab; init union-find parents
__global__ __launch_bounds__(TPB) void k_tag_hp(float4* __restrict__ spt, NRef n_ref, const int* __restrict__ den, int min_pts,
                                               int* __restrict__ parent, int* __restrict__ lab,
                                               const int* __restrict__ sslot, int* __restrict__ cell_rep) {
    const int n = n_ref.get();
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= n) return;
    const int i = __float_as_int(spt[p].w);
    const int hp = den[i] >= min_pts;  // binary_cuda_functions.cu:185
    spt[p].w = __int_as_float(i | (hp ? (int)0x80000000 : 0));
    parent[i] = i;
    lab[i] = -1;
    if (hp) atomicMin(&cell_rep[sslot[p]], i);  // representative HP of the cell (BIG = the cell has no HP)
}

// ---- lock-free union-find; roots are always the smallest index of their set -----------------------------------
__device__ __forceinline__ int uf_load(int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int uf_find(int* __restrict__ parent, int x) {
    while (true) {
        const int p = uf_load(&parent[x]);
        if (p == x) return x;
        const int gp = uf_load(&parent[p]);
        if (gp == p) return p;
        atomicMin(&parent[x], gp);  // path halving; monotone, so concurrent hooks are never lost
        x = gp;
    }
}

__device__ __forceinline__ int uf_union(int* __restrict__ parent, int a, int b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return a;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&parent[a], b);  // hook the larger root under the smaller
        if (old == a) return b;
        a = old;  // a had been hooked meanwhile: keep merging its new parent with b
    }
}

// a13: connected components of the HP graph, cell driven.
//  * own cell, trusted: every HP of the cell is within r of every other -> one union with the cell representative;
//  * other trusted cell: its HPs form one set already (their own chain), so ONE witnessed edge (first HP within r)
//    merges the two cells, and the whole cell is skipped when its representative already shares our root;
//  * untrusted (clamped border) cells fall back to pairwise unions, each edge handled by its larger endpoint.
// Every HP-HP edge (i,j) ends with find(i) == find(j): either both cells are trusted and chained to representatives
// that some witness joined, or the pairwise fallback handled the edge itself.
__global__ __launch_bounds__(TPB) void k_union(const float4* __restrict__ spt, const int* __restrict__ sseg, NRef n_ref,
                                              float inv_cell, float r2, const unsigned long long* __restrict__ hkeys,
                                              unsigned hmask, const int* __restrict__ hstart,
                                              const int* __restrict__ hcount, int* parent,
                                              const int* __restrict__ cell_rep, const int* __restrict__ sslot, int pass) {
    const int n = n_ref.get();
    // pass 0: every HP chains to its cell representative; only the representatives (and points of untrusted cells)
    //         look across cells -- a few hundred threads do almost all merging without contention;
    // (k_compress flattens the forest in between)
    // pass 1: every HP repeats the cross-cell search, which is now a cached parent compare for all merged pairs and
    //         only does real work for the edges a representative could not witness.
    // NB_Q lanes share a point and split its 125 cells; each lane keeps its own cached root (the forest is the only
    // shared state and it is lock-free), which also shortens the serial chain of the few representatives of pass 0
    const long long t = (long long)blockIdx.x * TPB + threadIdx.x;
    const int p = (int)(t / NB_Q), q = (int)(t % NB_Q);
    if (p >= n) return;
    const float4 me = spt[p];
    const int wi = __float_as_int(me.w);
    if (wi >= 0) return;  // LP: never expands (binary_cuda_functions.cu:209)
    const int i = wi & 0x7fffffff;
    const int my_rep = cell_rep[sslot[p]];
    const bool own_trusted = cell_trusted(cell_coord(me.x, inv_cell), cell_coord(me.y, inv_cell), cell_coord(me.z, inv_cell));
    int ri = (pass == 0) ? i : parent[i];
    if (pass == 0 && own_trusted) {
        if (my_rep != i) {  // chained; the representative does the rest
            if (q == 0) uf_union(parent, i, my_rep);
            return;
        }
    }
    for_each_neighbour_cell(me.x, me.y, me.z, sseg[p], inv_cell, hkeys, hmask, hstart, hcount, q, NB_Q,
                            [&](int beg, int end, int slot, bool trusted, bool same) {
        const int rep = cell_rep[slot];
        if (rep == BIG) return;  // no HP in that cell
        if (trusted) {
            if (same) return;    // own trusted cell: chained in pass 0
            // plain (possibly stale) read: a stale parent is a former ancestor, i.e. provably the same set
            if (rep == ri || parent[rep] == ri) return;
            ri = uf_find(parent, ri);
            if (uf_find(parent, rep) == ri) return;
            for (int j = beg; j < end; ++j) {
                const float4 q = spt[j];
                if (__float_as_int(q.w) < 0 && sqdist(me.x, me.y, me.z, q.x, q.y, q.z) <= r2) {
                    ri = uf_union(parent, ri, __float_as_int(q.w) & 0x7fffffff);
                    return;
                }
            }
            return;
        }
        for (int j = beg; j < end; ++j) {
            const float4 q = spt[j];
            const int wj = __float_as_int(q.w);
            const int jj = wj & 0x7fffffff;
            if (wj < 0 && jj < i && sqdist(me.x, me.y, me.z, q.x, q.y, q.z) <= r2) {
                if (parent[jj] != ri) ri = uf_union(parent, ri, jj);
            }
        }
    });
}

// flatten the forest between the two union passes: afterwards parent[i] is i's root for every HP
__global__ __launch_bounds__(TPB) void k_compress(const float4* __restrict__ spt, NRef n_ref, int* parent) {
    const int n = n_ref.get();
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= n) return;
    const int wi = __float_as_int(spt[p].w);
    if (wi >= 0) return;
    const int i = wi & 0x7fffffff;
    const int r = uf_find(parent, i);
    if (r != i) atomicMin(&parent[i], r);
}

// HP: lab = root (general mode: also register the smallest same-class HP of the component)
__global__ __launch_bounds__(TPB) void k_flatten(const float4* __restrict__ spt, NRef n_ref, int* __restrict__ parent,
                                                const int* __restrict__ sem, int general, int* __restrict__ semseed,
                                                int* __restrict__ lab) {
    const int n = n_ref.get();
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= n) return;
    const int wi = __float_as_int(spt[p].w);
    if (wi >= 0) return;
    const int i = wi & 0x7fffffff;
    const int r = uf_find(parent, i);
    lab[i] = r;
    if (general) atomicMin(&semseed[(size_t)cls_of(sem[i]) * n + r], i);
}

// general mode: HP label = seed of (component, class)
__global__ __launch_bounds__(TPB) void k_hp_seed_general(const float4* __restrict__ spt, NRef n_ref, const int* __restrict__ sem,
                                                        const int* __restrict__ semseed, int* __restrict__ lab) {
    const int n = n_ref.get();
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= n) return;
    const int wi = __float_as_int(spt[p].w);
    if (wi >= 0) return;
    const int i = wi & 0x7fffffff;
    lab[i] = semseed[(size_t)cls_of(sem[i]) * n + lab[i]];
}

// a13 (border rule): an LP within r of HPs takes the LAST cluster that reached it = the largest seed index among
// clusters (component of an adjacent HP, class of the LP) -- binary.cu:206-209.  `root` holds component roots of HPs.
// All HPs of a trusted cell share one component, so the cell contributes one candidate seed: it is skipped when that
// seed cannot raise the maximum, taken without any distance test in the LP's own cell, and otherwise needs one witness.
__global__ __launch_bounds__(TPB) void k_border(const float4* __restrict__ spt, const int* __restrict__ sseg, NRef n_ref,
                                               float inv_cell, float r2, const unsigned long long* __restrict__ hkeys,
                                               unsigned hmask, const int* __restrict__ hstart,
                                               const int* __restrict__ hcount, const int* __restrict__ sem, int general,
                                               const int* __restrict__ semseed, const int* __restrict__ root,
                                               int* __restrict__ lab, const int* __restrict__ cell_rep) {
    const int n = n_ref.get();
    const long long t = (long long)blockIdx.x * TPB + threadIdx.x;
    if (n <= 0 || (t - (threadIdx.x & 63)) / NB_Q >= n) return;   // (as in k_count: whole waves past the count leave)
    int p = (int)(t / NB_Q);
    const int q = (int)(t % NB_Q);
    const bool live = p < n;
    if (!live) p = n - 1;  // keep the lane group complete for the shuffles
    const float4 me = spt[p];
    const int wi = __float_as_int(me.w);
    if (wi < 0) return;  // HP (the whole lane group shares the point, so it leaves together)
    const int i = wi;
    const int my_cls = cls_of(sem[i]);
    int best = -1;
    auto seed_of = [&](int hp_index) {
        int s = root[hp_index];
        if (general) {
            s = semseed[(size_t)my_cls * n + s];
            if (s == BIG) s = -1;
        }
        return s;
    };
    // `best` is kept identical on all lanes of the group; candidates of a cell are tested NB_Q at a time
    for_each_neighbour_cell_group(me.x, me.y, me.z, sseg[p], inv_cell, hkeys, hmask, hstart, hcount, q,
                                  [&](int beg, int end, int slot, bool trusted, bool same) {
        const int rep = cell_rep[slot];
        if (rep == BIG) return;
        if (trusted) {
            const int s = seed_of(rep);
            if (s <= best) return;
            if (same) { best = s; return; }
            for (int j0 = beg; j0 < end; j0 += NB_Q) {
                const int j = j0 + q;
                bool hit = false;
                if (j < end) {
                    const float4 c = spt[j];
                    hit = __float_as_int(c.w) < 0 && sqdist(me.x, me.y, me.z, c.x, c.y, c.z) <= r2;
                }
                if (group_any(hit)) { best = s; return; }
            }
            return;
        }
        for (int j0 = beg; j0 < end; j0 += NB_Q) {
            const int j = j0 + q;
            int cand = -1;
            if (j < end) {
                const float4 c = spt[j];
                const int wj = __float_as_int(c.w);
                if (wj < 0 && sqdist(me.x, me.y, me.z, c.x, c.y, c.z) <= r2) cand = seed_of(wj & 0x7fffffff);
            }
            best = max(best, group_max(cand));
        }
    });
    best = group_max(best);  // the maximum over the lanes' cells = the maximum over all cells
    if (live && q == 0) lab[i] = best;
}

__global__ __launch_bounds__(TPB) void k_copy_i32(const int* __restrict__ src, int* __restrict__ dst, NRef n_ref) {
    const int n = n_ref.get();
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// a14: cluster population (HPs + border LPs) per seed
__global__ __launch_bounds__(TPB) void k_sizes(const int* __restrict__ lab, NRef n_ref, int* __restrict__ size) {
    const int n = n_ref.get();
    // a wave's lanes mostly share a handful of seeds: one atomic per distinct seed per wave instead of one per point
    const int i = blockIdx.x * TPB + threadIdx.x;
    int s = (i < n) ? lab[i] : -1;
    unsigned long long todo = __ballot(s >= 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int ls = __shfl(s, leader, 64);
        const unsigned long long same = __ballot(s == ls);
        if (lane_id() == leader) atomicAdd(&size[ls], __popcll(same));
        todo &= ~same;
    }
}

// a14: keep flag per seed: dropped iff float(size) < mean_count[sem-2] * para_f (binary.cu:255-256)
__global__ __launch_bounds__(TPB) void k_keep(const int* __restrict__ lab, const int* __restrict__ size,
                                             const int* __restrict__ sem, NRef n_ref, float para_f, int* __restrict__ keep) {
    const int n = n_ref.get();
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    int k = 0;
    if (lab[i] == i) {
        const int cs = cls_of(sem[i]);
        const float thr = __fmul_rn(c_mean_count[cs], para_f);
        k = ((float)size[i] < thr) ? 0 : 1;
    }
    keep[i] = k;
}

// final ids after filtering; candidates for the NN search; per-cluster bookkeeping
__global__ __launch_bounds__(TPB) void k_relabel(const int* __restrict__ lab, const int* __restrict__ keep,
                                                const int* __restrict__ newid, const int* __restrict__ sem,
                                                const int* __restrict__ seg_of_pt, const float* __restrict__ org_xyz,
                                                NRef n_ref, int* __restrict__ lab2, int* __restrict__ cluster_id,
                                                int* __restrict__ clt_sem, int* __restrict__ clt_seg,
                                                int* __restrict__ last_assigned, int* __restrict__ fsize,
                                                int* __restrict__ noise_flag, float4* __restrict__ cand,
                                                const int* __restrict__ size) {
    const int n = n_ref.get();
    if (n <= 0) return;   // capacity mode with nothing selected
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i - (int)(threadIdx.x & 63) >= n) return;   // whole waves past the count leave
    const bool live = i < n;
    if (!live) i = n - 1;  // keep whole waves alive for the shuffles below; duplicates of the last point are harmless
    const int s = lab[i];
    const int id = (s >= 0 && keep[s]) ? newid[s] : -1;
    if (live) {
        lab2[i] = id;
        cluster_id[i] = id;
        noise_flag[i] = id < 0;
    }
    if (keep[i]) {
        clt_sem[newid[i]] = sem[i];
        clt_seg[newid[i]] = seg_of_pt[i];
        fsize[newid[i]] = size[i];  // HPs + border LPs; unassigned points are added by k_noise_nn
    }
    {   // highest assigned index per segment: one atomic per wave when the wave sits in one segment (the usual case)
        const int seg = seg_of_pt[i];
        const int cand_i = (id >= 0) ? i : -1;
        const int seg0 = __shfl(seg, 0, 64);
        if (__all(seg == seg0)) {
            int m = cand_i;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
            if (lane_id() == 0 && m >= 0) atomicMax(&last_assigned[seg0], m);
        } else if (cand_i >= 0) {
            atomicMax(&last_assigned[seg], cand_i);
        }
    }
    if (live)
        cand[i] = make_float4(org_xyz[3 * i + 0], org_xyz[3 * i + 1], org_xyz[3 * i + 2],
                              __int_as_float(id >= 0 ? ((seg_of_pt[i] << 8) | sem[i]) : -1));
}

__global__ void k_cluster_num(const int* __restrict__ newid, const int* __restrict__ seg_off, int n_seg, NRef n_ref,
                              const int* __restrict__ total, int* __restrict__ cluster_num,
                              const int* __restrict__ status, int* __restrict__ n_clusters) {
    const int n = n_ref.get();
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) *n_clusters = (*status != 0) ? -1 : *total;
    if (b >= n_seg) return;
    const int lo = seg_off[b], hi = seg_off[b + 1];
    const int a = (lo < n) ? newid[lo] : *total;
    const int e = (hi < n) ? newid[hi] : *total;
    cluster_num[b] = e - a;
}

__global__ __launch_bounds__(TPB) void k_compact_noise(const int* __restrict__ noise_flag, const int* __restrict__ pos,
                                                      NRef n_ref, int* __restrict__ noise_list) {
    const int n = n_ref.get();
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i < n && noise_flag[i]) noise_list[pos[i]] = i;
}

// a15: exact nearest assigned point of the same class, ORIGINAL coordinates, ties -> highest index
// (binary_cuda_functions.cu:258-302: ascending scan with `<=`, i.e. the lexicographic optimum (min d2, max index)).
// One WAVE per unassigned point: the 64 lanes stride over the point's segment with coalesced 16-byte candidate reads,
// keep a private optimum and combine with shuffles -- order independent, hence identical to the sequential scan.
// cand.w = (segment << 8 | class) for assigned points, -1 otherwise.
__global__ __launch_bounds__(TPB) void k_noise_nn(const int* __restrict__ noise_list, const int* __restrict__ n_noise,
                                                 const float4* __restrict__ cand, const int* __restrict__ sem,
                                                 const int* __restrict__ seg_of_pt, const int* __restrict__ seg_off,
                                                 const int* __restrict__ lab2, const int* __restrict__ last_assigned,
                                                 int* __restrict__ cluster_id, int* __restrict__ fsize) {
    const int lane = lane_id();
    const int w = (int)(((long long)blockIdx.x * TPB + threadIdx.x) >> 6);
    if (w >= *n_noise) return;
    const int i = noise_list[w];
    const int seg = seg_of_pt[i];
    const int beg = seg_off[seg], end = seg_off[seg + 1];
    const float4 me = cand[i];
    const int key = (seg << 8) | sem[i];
    float best = __builtin_inff();
    int bi = -1;
    for (int j = beg + lane; j < end; j += 64) {
        const float4 q = cand[j];
        if (__float_as_int(q.w) == key) {
            const float d = sqdist(me.x, me.y, me.z, q.x, q.y, q.z);
            if (d <= best) { best = d; bi = j; }  // ascending j per lane
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (oi >= 0 && (bi < 0 || ob < best || (ob == best && oi > bi))) { best = ob; bi = oi; }
    }
    if (lane != 0) return;
    if (bi < 0) bi = last_assigned[seg];  // no assigned point of this class: binary_cuda_functions.cu:287-299
    const int id = (bi >= 0) ? lab2[bi] : -1;
    cluster_id[i] = id;
    if (id >= 0) atomicAdd(&fsize[id], 1);
}

// RN(d / n) for an integer-valued n < 2^24 - 1 given y = RN(1/n): reciprocal multiply + one FMA correction step
// (Markstein; bit-for-bit evidence in oracle/fastdiv_check.c).  Exact when no intermediate leaves the normal range,
// which k_centers guarantees by a per-chunk magnitude check (it falls back to the IEEE division otherwise).
__device__ __forceinline__ float exact_quotient(float d, float n, float y) {
    const float q0 = __fmul_rn(d, y);
    const float r = __fmaf_rn(-q0, n, d);
    return __fmaf_rn(r, y, q0);
}

constexpr int CTR_TPB = 64;
constexpr int CTR_SUB = 4;                       // 64-point slices per chunk
constexpr int CTR_CHUNK = CTR_TPB * CTR_SUB;     // segment points compacted per round
__global__ __launch_bounds__(CTR_TPB) void k_centers(const int* __restrict__ cluster_id, const float* __restrict__ off_xyz,
                                                    const int* __restrict__ clt_seg, const int* __restrict__ seg_off,
                                                    const int* __restrict__ n_clusters_total,
                                                    const int* __restrict__ member_start, int* __restrict__ member_idx,
                                                    float* __restrict__ centers) {
    // a16 + members CSR: one 64-lane workgroup per final cluster walks its segment in index order.  Each 256-point
    // chunk is compacted (ballot prefix) into LDS; lanes 0..2 then advance the sequential running mean
    // M += (p - M)/N (binary_cuda_functions.cu:237-239) of x, y, z.  A single wave is issue-bound, so the chain is kept
    // to five instructions per member: the reciprocals 1/N of a chunk are computed by all lanes at once and the
    // quotient is a multiply + two FMAs (exact, see exact_quotient).  Round 4: the members reach the chain in registers,
    // sixteen at a time and one group AHEAD of the chain (the LDS round trip of a group used to sit in front of every four
    // members), and a chunk is 256 segment points (the compaction, its two barriers and the reciprocals once per ~120
    // members instead of once per ~30).  The next chunk's global loads are issued before the chain.
    __shared__ __attribute__((aligned(16))) float s_xyz[3][CTR_CHUNK + 32];
    __shared__ __attribute__((aligned(16))) float s_rcp[CTR_CHUNK + 32];
    const int lane = threadIdx.x;
    const int C = *n_clusters_total;
    for (int c = blockIdx.x; c < C; c += gridDim.x) {
        const int seg = clt_seg[c];
        const int beg = seg_off[seg], end = seg_off[seg + 1];
        int wpos = member_start[c];
        int N = 0;
        float m = 0.f;  // lane 0: x, lane 1: y, lane 2: z
        bool hit_n[CTR_SUB];
        float nx[CTR_SUB], ny[CTR_SUB], nz[CTR_SUB];
#pragma unroll
        for (int j = 0; j < CTR_SUB; ++j) {
            const int i = beg + j * CTR_TPB + lane;
            hit_n[j] = (i < end) && (cluster_id[i] == c);
            nx[j] = ny[j] = nz[j] = 0.f;
            if (hit_n[j]) { nx[j] = off_xyz[3 * i + 0]; ny[j] = off_xyz[3 * i + 1]; nz[j] = off_xyz[3 * i + 2]; }
        }
        for (int base = beg; base < end; base += CTR_CHUNK) {
            bool hit[CTR_SUB];
            float px[CTR_SUB], py[CTR_SUB], pz[CTR_SUB];
#pragma unroll
            for (int j = 0; j < CTR_SUB; ++j) { hit[j] = hit_n[j]; px[j] = nx[j]; py[j] = ny[j]; pz[j] = nz[j]; }
#pragma unroll
            for (int j = 0; j < CTR_SUB; ++j) {   // prefetch the next chunk
                const int i = base + CTR_CHUNK + j * CTR_TPB + lane;
                hit_n[j] = (i < end) && (cluster_id[i] == c);
                if (hit_n[j]) { nx[j] = off_xyz[3 * i + 0]; ny[j] = off_xyz[3 * i + 1]; nz[j] = off_xyz[3 * i + 2]; }
            }
            int cnt = 0;
            bool ok = true;
#pragma unroll
            for (int j = 0; j < CTR_SUB; ++j) {
                const unsigned long long mask = __ballot(hit[j]);
                const int rank = cnt + __popcll(mask & ((1ULL << lane) - 1ULL));
                cnt += __popcll(mask);
                // magnitudes for which the correction-step quotient is provably exact: 0 or [1e-20, 1e20]
                const float ax = fabsf(px[j]), ay = fabsf(py[j]), az = fabsf(pz[j]);
                ok = ok && (!hit[j] || ((ax == 0.f || (ax >= 1e-20f && ax <= 1e20f)) && (ay == 0.f || (ay >= 1e-20f && ay <= 1e20f)) &&
                                        (az == 0.f || (az >= 1e-20f && az <= 1e20f))));
                if (hit[j]) {
                    s_xyz[0][rank] = px[j]; s_xyz[1][rank] = py[j]; s_xyz[2][rank] = pz[j];
                    if (member_idx) member_idx[wpos + rank] = base + j * CTR_TPB + lane;
                }
            }
            if (cnt == 0) continue;        // wave-uniform
            const bool fast = __all(ok);
#pragma unroll
            for (int j = 0; j < CTR_SUB; ++j)     // RN(1/n) for the next 256 member counts, all lanes at once
                if (j * CTR_TPB < cnt) s_rcp[j * CTR_TPB + lane] = __fdiv_rn(1.0f, (float)(N + 1 + j * CTR_TPB + lane));
            wpos += cnt;
            __syncthreads();
            if (lane < 3) {
                const float* v = s_xyz[lane];
                if (fast) {
                    // two register groups of 16 members, ping-pong: group B is read from LDS while the chain walks group A
                    float4 va[4], ya[4], vb[4], yb[4];
                    auto load16 = [&](float4 (&vv)[4], float4 (&yy)[4], int k) {   // reads past cnt stay inside the padded arrays
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            vv[q] = *reinterpret_cast<const float4*>(v + k + 4 * q);
                            yy[q] = *reinterpret_cast<const float4*>(s_rcp + k + 4 * q);
                        }
                    };
                    auto chain16 = [&](const float4 (&vv)[4], const float4 (&yy)[4], int k) {
                        const float n0 = (float)(N + k + 1);
                        const int left = cnt - k;
                        if (left >= 16) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                m = __fadd_rn(m, exact_quotient(__fsub_rn(vv[q].x, m), n0 + (float)(4 * q + 0), yy[q].x));
                                m = __fadd_rn(m, exact_quotient(__fsub_rn(vv[q].y, m), n0 + (float)(4 * q + 1), yy[q].y));
                                m = __fadd_rn(m, exact_quotient(__fsub_rn(vv[q].z, m), n0 + (float)(4 * q + 2), yy[q].z));
                                m = __fadd_rn(m, exact_quotient(__fsub_rn(vv[q].w, m), n0 + (float)(4 * q + 3), yy[q].w));
                            }
                        } else {
                            for (int j = 0; j < left; ++j)      // the last group of a chunk: from LDS, one at a time
                                m = __fadd_rn(m, exact_quotient(__fsub_rn(v[k + j], m), n0 + (float)j, s_rcp[k + j]));
                        }
                    };
                    load16(va, ya, 0);
                    for (int k = 0; k < cnt; k += 32) {
                        load16(vb, yb, k + 16);
                        chain16(va, ya, k);
                        if (k + 16 < cnt) {
                            load16(va, ya, k + 32);
                            chain16(vb, yb, k + 16);
                        }
                    }
                } else {
                    for (int k = 0; k < cnt; ++k) m = __fadd_rn(m, __fdiv_rn(__fsub_rn(v[k], m), (float)(N + k + 1)));
                }
            }
            N += cnt;
            __syncthreads();
        }
        if (lane < 3) centers[3 * c + lane] = m;
    }
}

__global__ void k_member_tail(int* __restrict__ member_start, const int* __restrict__ total_assigned, int n,
                              const int* __restrict__ n_clusters) {
    // scan wrote member_start[0..n-1]; entries >= C all equal total; write the [n] sentinel
    if (threadIdx.x == 0 && blockIdx.x == 0) member_start[n] = *total_assigned;
    (void)n_clusters;
}

struct Workspace {
    int *seg_off, *seg_of_pt, *slot_of_pt, *hcount, *hstart, *hcursor, *sseg, *sslot, *cell_rep, *parent, *lab, *root, *semseed, *size,
        *keep, *newid, *lab2, *clt_seg, *last_assigned, *fsize, *noise_flag, *noise_pos, *noise_list, *scan_tmp,
        *scalars, *mstart_tmp;
    unsigned long long* scan_state[4];
    unsigned long long* hkeys;
    float4 *spt, *cand;
    unsigned hcap;
    size_t zero_end, ff_end, big_end;  // ends of the three fill blocks (byte offsets in the workspace)
};

unsigned hash_capacity(int n) {
    unsigned c = 1024;
    while (c < 2u * (unsigned)n) c <<= 1;
    return c;
}

size_t carve(Carver& cv, Workspace& w, int n, int n_seg, int general) {
    const size_t N = (size_t)(n > 0 ? n : 1);
    w.hcap = hash_capacity(n);
    // zero-filled block (one memset): scalars | hcount | hcursor | size | fsize
    w.scalars = cv.take<int>(64);
    for (int j = 0; j < 4; ++j)          // states of the four chained scans (zeroed with the block)
        w.scan_state[j] = cv.take<unsigned long long>(scan_chained_state_words((long long)(w.hcap > N ? w.hcap : N)));
    w.hcount = cv.take<int>(w.hcap);
    w.hcursor = cv.take<int>(w.hcap);
    w.size = cv.take<int>(N);
    w.fsize = cv.take<int>(N);
    w.zero_end = cv.off;
    // 0xff-filled block (one memset): hkeys (EMPTY) | last_assigned (-1)
    w.hkeys = cv.take<unsigned long long>(w.hcap);
    w.last_assigned = cv.take<int>((size_t)n_seg + 1);
    w.ff_end = cv.off;
    // 0x7f-filled block (one memset): cell_rep (BIG) | semseed (BIG)
    w.cell_rep = cv.take<int>(w.hcap);
    w.semseed = cv.take<int>(general ? 18 * N : 1);
    w.big_end = cv.off;
    w.hstart = cv.take<int>(w.hcap);
    w.spt = cv.take<float4>(N);
    w.cand = cv.take<float4>(N);
    w.seg_off = cv.take<int>((size_t)n_seg + 1);
    w.seg_of_pt = cv.take<int>(N);
    w.slot_of_pt = cv.take<int>(N);
    w.sseg = cv.take<int>(N);
    w.sslot = cv.take<int>(N);
    w.parent = cv.take<int>(N);
    w.lab = cv.take<int>(N);
    w.root = cv.take<int>(N);
    w.keep = cv.take<int>(N);
    w.newid = cv.take<int>(N);
    w.lab2 = cv.take<int>(N);
    w.clt_seg = cv.take<int>(N);
    w.noise_flag = cv.take<int>(N);
    w.noise_pos = cv.take<int>(N);
    w.noise_list = cv.take<int>(N);
    w.mstart_tmp = cv.take<int>(N + 1);
    w.scan_tmp = cv.take<int>(scan_tmp_ints((long long)(w.hcap > N ? w.hcap : N)));
    return align_up(cv.off, 256);
}

}  // namespace
}  // namespace pbn

using namespace pbn;

extern "C" size_t pbn_cluster_workspace_bytes(int n_points, int n_segments, int general_sem) {
    if (n_points < 0 || n_segments < 0) return 0;
    Carver cv(nullptr, 0);
    Workspace w;
    return carve(cv, w, n_points, n_segments, general_sem);
}

extern "C" int pbn_binary_cluster(const float* off_xyz, const float* org_xyz, const int32_t* sem,
                                  const int32_t* seg_len, int n, int n_seg, float radius, int min_pts, float para_f,
                                  int nv_flag, int flags, int32_t* cluster_id, int32_t* cluster_num, int32_t* den,
                                  float* centers, int32_t* clt_sem, int32_t* n_clusters, int32_t* member_start,
                                  int32_t* member_idx, void* workspace, size_t workspace_bytes, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int general = flags & 1;
    const int capacity = (flags >> 1) & 1;   // n is a capacity; the points that exist are the first sum(seg_len) rows
    if (n < 0 || n_seg < 0 || n_seg > 65535 || !(radius > 0.0f) || !n_clusters || (n_seg > 0 && !cluster_num))
        return PBN_ERR_ARG;
    if ((member_start == nullptr) != (member_idx == nullptr)) return PBN_ERR_ARG;
    if (n > 0 && (!off_xyz || !org_xyz || !sem || !seg_len || !cluster_id || !den || !centers || !clt_sem || !workspace))
        return PBN_ERR_ARG;
    if (n == 0 || n_seg == 0) {
        if (n_seg > 0) { const int frc_ = fill_bytes(cluster_num, 0, sizeof(int) * (size_t)n_seg, stream); if (frc_ != PBN_OK) return frc_; }
        { const int frc_ = fill_bytes(n_clusters, 0, sizeof(int), stream); if (frc_ != PBN_OK) return frc_; }
        if (member_start) { const int frc_ = fill_bytes(member_start, 0, sizeof(int) * ((size_t)n + 1), stream); if (frc_ != PBN_OK) return frc_; }
        return PBN_OK;
    }
    Carver cv(workspace, workspace_bytes);
    Workspace w;
    carve(cv, w, n, n_seg, general);
    if (!cv.ok) return PBN_ERR_WORKSPACE;

    const float cell = radius * 0.5625f;  // diagonal 0.974 r: same cell => within r; r-neighbours within +-2 cells
    const float inv_cell = 1.0f / cell;
    const float r2 = radius * radius;      // binary_cuda_functions.cu:85 (fp32 product)
    const unsigned hmask = w.hcap - 1;
    const int nb = cdiv(n, TPB);
    const NRef nr{n, capacity ? w.seg_off + n_seg : nullptr};
    int* status = w.scalars + 0;
    int* total_kept = w.scalars + 1;
    int* n_noise = w.scalars + 2;
    int* total_assigned = w.scalars + 3;

    {   // three fills cover every array that needs an initial value (see carve)
        char* base = (char*)workspace;
        char* z0 = (char*)w.scalars;
        char* f0 = (char*)w.hkeys;
        char* b0 = (char*)w.cell_rep;
        // ... and the caller's counters, all in one launch (unaligned ranges fall back to hipMemsetAsync inside)
        FillRange fr[7] = {{cluster_num, sizeof(int) * (size_t)n_seg, 0}, {n_clusters, sizeof(int), 0},
                           {z0, (size_t)(base + w.zero_end - z0), 0}, {f0, (size_t)(base + w.ff_end - f0), 0xff},
                           {b0, (size_t)(base + w.big_end - b0), 0x7f}, {nullptr, 0, 0}, {nullptr, 0, 0}};
        if (capacity) {   // the scans below run over the whole capacity: flags of rows that do not exist must read 0
            fr[5] = FillRange{w.keep, sizeof(int) * (size_t)n, 0};
            fr[6] = FillRange{w.noise_flag, sizeof(int) * (size_t)n, 0};
        }
        const int frc = fill_ranges(fr, 7, stream);
        if (frc != PBN_OK) return frc;
    }

    hipLaunchKernelGGL(k_seg_offsets, dim3(1), dim3(64), 0, stream, seg_len, n_seg, n, capacity, w.seg_off, status);
    hipLaunchKernelGGL(k_cell_insert, dim3(nb), dim3(TPB), 0, stream, off_xyz, sem, w.seg_off, n_seg, nr, inv_cell,
                       w.hkeys, w.hcount, hmask, w.slot_of_pt, w.seg_of_pt, status);
    int rc = scan_exclusive_i32_chained(w.hcount, w.hstart, (int)w.hcap, w.scan_state[0], nullptr, status, stream);
    if (rc != PBN_OK) return rc;
    hipLaunchKernelGGL(k_cell_scatter, dim3(nb), dim3(TPB), 0, stream, off_xyz, nr, w.slot_of_pt, w.hstart, w.hcursor,
                       w.seg_of_pt, w.spt, w.sseg, w.sslot);
    hipLaunchKernelGGL(k_count, dim3(cdiv((long long)n * NB_Q, TPB)), dim3(TPB), 0, stream, w.spt, w.sseg, nr, inv_cell, r2, w.hkeys, hmask,
                       w.hstart, w.hcount, den);
    hipLaunchKernelGGL(k_tag_hp, dim3(nb), dim3(TPB), 0, stream, w.spt, nr, den, min_pts, w.parent, w.lab, w.sslot, w.cell_rep);
    const dim3 nbq(cdiv((long long)n * NB_Q, TPB));
    hipLaunchKernelGGL(k_union, nbq, dim3(TPB), 0, stream, w.spt, w.sseg, nr, inv_cell, r2, w.hkeys, hmask,
                       w.hstart, w.hcount, w.parent, w.cell_rep, w.sslot, 0);
    hipLaunchKernelGGL(k_compress, dim3(nb), dim3(TPB), 0, stream, w.spt, nr, w.parent);
    hipLaunchKernelGGL(k_union, nbq, dim3(TPB), 0, stream, w.spt, w.sseg, nr, inv_cell, r2, w.hkeys, hmask,
                       w.hstart, w.hcount, w.parent, w.cell_rep, w.sslot, 1);
    hipLaunchKernelGGL(k_flatten, dim3(nb), dim3(TPB), 0, stream, w.spt, nr, w.parent, sem, general, w.semseed, w.lab);
    hipLaunchKernelGGL(k_copy_i32, dim3(nb), dim3(TPB), 0, stream, w.lab, w.root, nr);
    if (general)
        hipLaunchKernelGGL(k_hp_seed_general, dim3(nb), dim3(TPB), 0, stream, w.spt, nr, sem, w.semseed, w.lab);
    hipLaunchKernelGGL(k_border, nbq, dim3(TPB), 0, stream, w.spt, w.sseg, nr, inv_cell, r2, w.hkeys, hmask,
                       w.hstart, w.hcount, sem, general, w.semseed, w.root, w.lab, w.cell_rep);
    hipLaunchKernelGGL(k_sizes, dim3(nb), dim3(TPB), 0, stream, w.lab, nr, w.size);
    hipLaunchKernelGGL(k_keep, dim3(nb), dim3(TPB), 0, stream, w.lab, w.size, sem, nr, para_f, w.keep);
    rc = scan_exclusive_i32_chained(w.keep, w.newid, n, w.scan_state[1], total_kept, status, stream);
    if (rc != PBN_OK) return rc;
    hipLaunchKernelGGL(k_relabel, dim3(nb), dim3(TPB), 0, stream, w.lab, w.keep, w.newid, sem, w.seg_of_pt, org_xyz, nr,
                       w.lab2, cluster_id, clt_sem, w.clt_seg, w.last_assigned, w.fsize, w.noise_flag, w.cand, w.size);
    hipLaunchKernelGGL(k_cluster_num, dim3(cdiv(n_seg, 64)), dim3(64), 0, stream, w.newid, w.seg_off, n_seg, nr,
                       total_kept, cluster_num, status, n_clusters);
    if (nv_flag) {
        rc = scan_exclusive_i32_chained(w.noise_flag, w.noise_pos, n, w.scan_state[2], n_noise, status, stream);
        if (rc != PBN_OK) return rc;
        hipLaunchKernelGGL(k_compact_noise, dim3(nb), dim3(TPB), 0, stream, w.noise_flag, w.noise_pos, nr, w.noise_list);
        hipLaunchKernelGGL(k_noise_nn, dim3(cdiv((long long)n * 64, TPB)), dim3(TPB), 0, stream, w.noise_list, n_noise, w.cand, sem, w.seg_of_pt,
                           w.seg_off, w.lab2, w.last_assigned, cluster_id, w.fsize);
    }
    int* mstart = member_start ? member_start : w.mstart_tmp;
    rc = scan_exclusive_i32_chained(w.fsize, mstart, n, w.scan_state[3], total_assigned, status, stream);
    if (rc != PBN_OK) return rc;
    hipLaunchKernelGGL(k_member_tail, dim3(1), dim3(64), 0, stream, mstart, total_assigned, n, total_kept);
    const int center_blocks = 2048;  // one wave per cluster, persistent over clusters
    hipLaunchKernelGGL(k_centers, dim3(center_blocks), dim3(CTR_TPB), 0, stream, cluster_id, off_xyz, w.clt_seg, w.seg_off,
                       total_kept, mstart, member_idx, centers);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}
