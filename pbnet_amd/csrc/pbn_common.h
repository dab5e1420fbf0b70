// pbn_common.h -- shared helpers for the gfx950 kernels of libpbnet_hip.so (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pbnet_hip.h"

namespace pbn {

constexpr int WAVE = 64;

extern thread_local int g_last_hip_error;

#define PBN_HIP_CHECK(expr)                                  \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) {                              \
            pbn::g_last_hip_error = (int)_e;                 \
            return PBN_ERR_HIP;                              \
        }                                                    \
    } while (0)

#define PBN_LAUNCH_CHECK() PBN_HIP_CHECK(hipGetLastError())

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over a caller-provided workspace (256-byte aligned carves).
struct Carver {
    char* base;
    size_t off;
    size_t cap;
    bool ok;
    Carver(void* p, size_t bytes) : base((char*)p), off(0), cap(bytes), ok(true) {}
    template <typename T>
    T* take(size_t n) {
        size_t start = align_up(off, 256);
        size_t end = start + n * sizeof(T);
        if (base != nullptr && end > cap) ok = false;
        off = end;
        return base ? (T*)(base + start) : (T*)nullptr;
    }
};

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ int wave_reduce_add(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 64-bit finalizer (splitmix64) -> table slot
__device__ __forceinline__ uint32_t hash64(uint64_t k) {
    k ^= k >> 30;
    k *= 0xbf58476d1ce4e5b9ULL;
    k ^= k >> 27;
    k *= 0x94d049bb133111ebULL;
    k ^= k >> 31;
    return (uint32_t)k;
}

// ---- device-wide exclusive scan (int32), three launches; tmp must hold cdiv(n, SCAN_TILE)+1 ints -------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048 elements per block

static inline size_t scan_tmp_ints(long long n) { return (size_t)cdiv(n, SCAN_TILE) + 2; }

// out[i] = sum_{j<i} in[j] for i in [0,n); if total != nullptr, *total = sum of all.  in may alias out.
int scan_exclusive_i32(const int* in, int* out, int n, int* tmp, int* total, hipStream_t stream);
// the same in ONE launch (chained scan over the workgroups); state: scan_chained_state_words(n) 64-bit words zeroed by the caller
size_t scan_chained_state_words(long long n);
int scan_exclusive_i32_chained(const int* in, int* out, int n, unsigned long long* state_zeroed, int* total, int* status,
                               hipStream_t stream);

// coords.hip: every cube map of a pyramid in one launch (see the definition)
struct MapJobs {
    const int* coords[6]; const int* n_dev[6]; const unsigned long long* keys[6]; const int* vals[6]; int* nbr[6];
    unsigned mask[6]; int ksize[6], stride[6];
    int n_jobs, n_max, x_fastest;
};
int kernel_maps_multi(const MapJobs& jobs, long long total_max, hipStream_t stream);

// Several byte fills in ONE kernel launch (a stage's zero / 0xff / 0x7f groups), any alignment and size; replaces
// hipMemsetAsync on every path that may be captured in a HIP graph (see common.hip for why).
struct FillRange { void* p; size_t bytes; unsigned char value; };
int fill_ranges(const FillRange* ranges, int n, hipStream_t stream);
static inline int fill_bytes(void* p, unsigned char value, size_t bytes, hipStream_t stream) {
    const FillRange r{p, bytes, value};
    return fill_ranges(&r, 1, stream);
}

// coords.hip internals shared with executor.hip (see the definitions for the `clear` contract)
int coords_unique_impl(const int32_t* coords, const int32_t* n_dev, int n_max, uint64_t* table_keys, int32_t* table_vals,
                       int capacity, int32_t* unique_index, int32_t* inverse, int32_t* unique_coords, int32_t* n_unique,
                       void* workspace, size_t workspace_bytes, int32_t* status, bool clear, hipStream_t stream);
int coords_stride_impl(const int32_t* fine_coords, const int32_t* n_fine_dev, int n_fine_max, int stride_out,
                       uint64_t* table_keys, int32_t* table_vals, int capacity, int32_t* coarse_coords,
                       int32_t* parent_row, int32_t* child_k, int32_t* nbr_down, int32_t* nbr_up, int32_t* n_coarse,
                       void* workspace, size_t workspace_bytes, int32_t* status, bool clear, hipStream_t stream);

int coords_insert_identity(const int32_t* coords, const int32_t* n_dev, int n_max, uint64_t* keys, int32_t* vals,
                           int capacity, int32_t* out_coords, int32_t* n_out, int32_t* status, hipStream_t stream);
int coords_morton_iota(const int32_t* coords, const int32_t* n_dev, int n_max, uint64_t* keys, int32_t* iota,
                       hipStream_t stream);
int coords_apply_perm(const int32_t* ucoords, const int32_t* perm32, const int32_t* uidx32, const int32_t* inv32, int n_max,
                      int32_t* sorted_coords, int64_t* perm64, int64_t* inv_perm64, int64_t* uidx64, int64_t* inv64,
                      hipStream_t stream);
// executor.hip: the levels above level 0 and every map, given a finished level 0 (coords, table, count)
int coords_build_upper(int n, int want_k5, int x_fastest, void* arena, const pbn_coords_layout* L, hipStream_t stream);
int coords_number_first(const int32_t* slot_of_row, const int32_t* table_vals, const int32_t* n_dev, int n_max, int32_t* scan_tmp,
                        int32_t* newid, int32_t* first_row, int32_t* n_out, const int32_t* status, hipStream_t stream);
// pyramid.hip: the sorted, hash-free lineage of pbn_coords_prepare (scratch lives in the arena's sort_temp block)
size_t pyramid_scratch_bytes(int n);
int coords_prepare_sorted(const int32_t* coords, const int32_t* n_dev, int n, int want_k5, int x_fastest, void* arena,
                          const pbn_prepare_layout* P, hipStream_t stream);
// prepare.hip: device radix sort of (key, value) pairs; temp from sort_pairs_temp_bytes(n)
size_t sort_pairs_temp_bytes(int n);
int sort_pairs_u64_i32(const uint64_t* keys_in, uint64_t* keys_out, const int32_t* vals_in, int32_t* vals_out, int n,
                       void* temp, size_t temp_bytes, hipStream_t stream);

}  // namespace pbn
