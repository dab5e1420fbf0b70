// coords_dev.h -- device helpers shared by the coordinate kernels (coords.hip: hash-table pipeline in any row order;
// pyramid.hip: the sorted, hash-free pyramid of pbn_coords_prepare).
#pragma once
#include "pbn_common.h"

namespace pbn {
namespace {

constexpr int TPB = 256;
constexpr unsigned long long EMPTY_KEY = ~0ULL;

__device__ __forceinline__ bool in_range(int b, int x, int y, int z) {
    return b >= 0 && b < 65535 && x >= -32768 && x <= 32767 && y >= -32768 && y <= 32767 && z >= -32768 && z <= 32767;
}

__device__ __forceinline__ unsigned long long pack4(int b, int x, int y, int z) {
    return ((unsigned long long)(unsigned)b << 48) | ((unsigned long long)(unsigned)((x + 32768) & 0xffff) << 32) |
           ((unsigned long long)(unsigned)((y + 32768) & 0xffff) << 16) | (unsigned long long)(unsigned)((z + 32768) & 0xffff);
}

__device__ __forceinline__ int floor_div(int a, int s) {  // s > 0
    int q = a / s;
    return (a % s != 0 && a < 0) ? q - 1 : q;
}

// Every open-addressing probe is bounded by the table capacity: a table that was not cleared, is over-full or was
// corrupted ends the probe with the PBN_TABLE_FULL bit raised in `status` (the pipeline then reports a count of -1 and the
// caller gets an error) instead of spinning for ever.
constexpr int PBN_STATUS_RANGE = 1;        // a coordinate outside the packable range
constexpr int PBN_STATUS_TABLE_FULL = 2;   // a probe sequence visited every slot

__device__ __forceinline__ int table_insert_min(unsigned long long* __restrict__ keys, int* __restrict__ vals,
                                                unsigned mask, unsigned long long key, int row, int* status) {
    unsigned h = hash64(key) & mask;
    for (unsigned probes = 0; probes <= mask; ++probes) {
        unsigned long long prev = atomicCAS(&keys[h], EMPTY_KEY, key);
        if (prev == EMPTY_KEY || prev == key) {
            atomicMin(&vals[h], row);
            return (int)h;
        }
        h = (h + 1) & mask;
    }
    if (status) atomicOr(status, PBN_STATUS_TABLE_FULL);
    return 0;    // a valid slot index: the result is garbage, flagged
}

__device__ __forceinline__ int table_find(const unsigned long long* __restrict__ keys, const int* __restrict__ vals,
                                          unsigned mask, unsigned long long key) {
    unsigned h = hash64(key) & mask;
    for (unsigned probes = 0; probes <= mask; ++probes) {
        const unsigned long long k = keys[h];
        if (k == key) return vals[h];
        if (k == EMPTY_KEY) return -1;
        h = (h + 1) & mask;
    }
    return -1;   // a full table without the key: not found
}

__device__ __forceinline__ int real_n(const int* n_dev, int n_max) {
    if (!n_dev) return n_max;
    const int v = *n_dev;
    return v < n_max ? v : n_max;
}


__device__ __forceinline__ unsigned long long spread3(unsigned v) {  // 16 bits -> every third bit of 48
    unsigned long long x = v & 0xffffu;
    x = (x | (x << 32)) & 0x00ff00000000ffffULL;
    x = (x | (x << 16)) & 0x00ff0000ff0000ffULL;
    x = (x | (x << 8)) & 0xf00f00f00f00f00fULL;
    x = (x | (x << 4)) & 0x30c30c30c30c30c3ULL;
    x = (x | (x << 2)) & 0x9249249249249249ULL;
    return x;
}

__device__ __forceinline__ int wave_incl_scan_i(int v) {
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

}  // namespace
}  // namespace pbn
