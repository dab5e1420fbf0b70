// coords.hip -- COO coordinate hashing for the sparse-voxel backbone on gfx950.
//
// Re-creates (does not port) the coordinate-manager behaviour the reference obtains from MinkowskiEngine
// (/root/reference/network/PBNet.py:117,240-247,265-271; network/Mink.py:221-288; dataset_preprocess.py:269-272):
//   * voxelisation / de-duplication with first-occurrence winner and ascending survivor order + inverse map,
//   * strided coordinate sets floor(c/s)*s with child->parent map and the 8-way child table of k=2,s=2 convs,
//   * kernel maps ("rulebooks") as OUTPUT-STATIONARY neighbour tables nbr[row][k] = input row or -1, which is what
//     the gather-GEMM kernels in spconv.hip consume (no scatter, no atomics on features, deterministic sums).
//
// Keys: (batch:16 | x+2^15:16 | y+2^15:16 | z+2^15:16) in one 64-bit word; open addressing, linear probing,
// capacity = pow2 >= 2n.  Values: smallest inserting row (atomicMin) => deterministic "first occurrence".
// Row counts that depend on the data stay on the DEVICE (n_out pointers); kernels take an upper bound for the grid
// and read the real count, so a whole coordinate pyramid is built without a host round trip.
#include "coords_dev.h"

namespace pbn {
namespace {

// ---- unique ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_insert_rows(const int* __restrict__ coords, const int* n_dev, int n_max,
                                                    unsigned long long* __restrict__ keys, int* __restrict__ vals,
                                                    unsigned mask, int* __restrict__ slot_of_row, int* __restrict__ status) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= real_n(n_dev, n_max)) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    if (!in_range(c.x, c.y, c.z, c.w)) atomicOr(status, PBN_STATUS_RANGE);
    slot_of_row[i] = table_insert_min(keys, vals, mask, pack4(c.x, c.y, c.z, c.w), i, status);
}

// "row i is the first occurrence of its key": the slot's value is the smallest inserting row
__device__ __forceinline__ int first_flag(const int* __restrict__ slot_of_row, const int* __restrict__ vals, int i, int n) {
    return (i < n && vals[slot_of_row[i]] == i) ? 1 : 0;
}

// Survivor numbering = exclusive scan of the first-occurrence flags, in two launches instead of flags + 3-launch scan:
//   k_flag_block_sums : per 2048-row block, the number of first rows;
//   k_flag_number     : every block sums the block totals in front of it (a few hundred ints, L2-resident), scans its
//                       own flags, writes newid[i] and first_row[i] (the slot's value BEFORE any renumbering, so that
//                       the write kernels can renumber the table in place without racing their own readers);
//                       the last block publishes the survivor count (or -1 on a range error).
__global__ __launch_bounds__(SCAN_THREADS) void k_flag_block_sums(const int* __restrict__ slot_of_row,
                                                                  const int* __restrict__ vals, const int* n_dev,
                                                                  int n_max, int* __restrict__ sums) {
    __shared__ int wtot[SCAN_THREADS / 64];
    const int n = real_n(n_dev, n_max);
    const int base = blockIdx.x * SCAN_TILE;
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) s += first_flag(slot_of_row, vals, base + k * SCAN_THREADS + threadIdx.x, n);
    s = wave_reduce_add(s);
    if (lane_id() == 0) wtot[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < SCAN_THREADS / 64; ++w) t += wtot[w];
        sums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_flag_number(const int* __restrict__ slot_of_row,
                                                              const int* __restrict__ vals, const int* n_dev, int n_max,
                                                              const int* __restrict__ sums, int* __restrict__ newid,
                                                              int* __restrict__ first_row, int* __restrict__ n_out,
                                                              const int* __restrict__ status) {
    __shared__ int wtot[SCAN_THREADS / 64];
    __shared__ int s_base;
    const int n = real_n(n_dev, n_max);
    // blocks in front of this one
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_THREADS) part += sums[b];
    part = wave_reduce_add(part);
    if (lane_id() == 0) wtot[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < SCAN_THREADS / 64; ++w) t += wtot[w];
        s_base = t;
    }
    __syncthreads();
    const int block_base = s_base;
    // thread t owns SCAN_ITEMS consecutive rows
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int f[SCAN_ITEMS], fr[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int i = base + k;
        fr[k] = (i < n) ? vals[slot_of_row[i]] : -1;
        f[k] = (i < n && fr[k] == i) ? 1 : 0;
        s += f[k];
    }
    const int incl = wave_incl_scan_i(s);
    __syncthreads();
    if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        const int t = wtot[w];
        if (w < (int)(threadIdx.x >> 6)) woff += t;
        tot += t;
    }
    int ex = block_base + woff + incl - s;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int i = base + k;
        if (i < n_max) { newid[i] = ex; first_row[i] = fr[k]; }
        ex += f[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_out = (status && *status != 0) ? -1 : block_base + tot;
}

// unique_index[new] = first row; inverse[i] = new id of i's survivor; survivor coordinates; table value -> new id
__global__ __launch_bounds__(TPB) void k_unique_write(const int* __restrict__ coords, const int* __restrict__ slot_of_row,
                                                     const int* __restrict__ first_row, const int* __restrict__ newid,
                                                     const int* n_dev, int n_max, int* __restrict__ unique_index,
                                                     int* __restrict__ inverse, int* __restrict__ out_coords,
                                                     int* __restrict__ vals) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= real_n(n_dev, n_max)) return;
    const int first = first_row[i];
    if (first == i) {
        const int id = newid[i];
        unique_index[id] = i;
        if (out_coords) reinterpret_cast<int4*>(out_coords)[id] = reinterpret_cast<const int4*>(coords)[i];
        vals[slot_of_row[i]] = id;
    }
    if (inverse) inverse[i] = newid[first];
}

// ---- stride: parent coordinate of every fine row ------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_insert_parents(const int* __restrict__ coords, const int* n_dev, int n_max,
                                                       int stride_out, unsigned long long* __restrict__ keys,
                                                       int* __restrict__ vals, unsigned mask,
                                                       int* __restrict__ slot_of_row, int* __restrict__ status) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= real_n(n_dev, n_max)) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const int px = floor_div(c.y, stride_out) * stride_out, py = floor_div(c.z, stride_out) * stride_out,
              pz = floor_div(c.w, stride_out) * stride_out;
    slot_of_row[i] = table_insert_min(keys, vals, mask, pack4(c.x, px, py, pz), i, status);
}

// coarse coords (first-occurrence order), child -> (parent row, k), 8-way child table of the k=2,s=2 convolution,
// optionally the transposed table row (nbr_up[i][k] = parent at k = child_k, -1 elsewhere), table value -> coarse row
__global__ __launch_bounds__(TPB) void k_stride_write(const int* __restrict__ coords, const int* n_dev, int n_max,
                                                     int stride_out, const int* __restrict__ slot_of_row,
                                                     const int* __restrict__ first_row, const int* __restrict__ newid,
                                                     int* __restrict__ vals, int* __restrict__ coarse_coords,
                                                     int* __restrict__ parent_row, int* __restrict__ child_k,
                                                     int* __restrict__ nbr_down, int* __restrict__ nbr_up) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= real_n(n_dev, n_max)) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const int s_in = stride_out >> 1;
    const int px = floor_div(c.y, stride_out) * stride_out, py = floor_div(c.z, stride_out) * stride_out,
              pz = floor_div(c.w, stride_out) * stride_out;
    const int first = first_row[i];
    const int prow = newid[first];
    if (first == i) {
        reinterpret_cast<int4*>(coarse_coords)[prow] = make_int4(c.x, px, py, pz);
        vals[slot_of_row[i]] = prow;
    }
    // even kernel (K=2): offsets 0..1 per axis, x fastest (ME convention C1/C2)
    const int k = ((c.y - px) / s_in) + 2 * ((c.z - py) / s_in) + 4 * ((c.w - pz) / s_in);
    parent_row[i] = prow;
    child_k[i] = k;
    nbr_down[(size_t)prow * 8 + k] = i;
    if (nbr_up) {
        int4 lo = make_int4(-1, -1, -1, -1), hi = lo;
        int* v = (k < 4) ? &lo.x : &hi.x;
        v[k & 3] = prow;
        reinterpret_cast<int4*>(nbr_up)[2 * (size_t)i + 0] = lo;
        reinterpret_cast<int4*>(nbr_up)[2 * (size_t)i + 1] = hi;
    }
}

// ---- kernel map: nbr[row][k] = row of (coords[row] + offsets[k]) in the table, or -1 -----------------------------
__global__ __launch_bounds__(TPB) void k_kernel_map(const int* __restrict__ out_coords, const int* n_dev, int n_max,
                                                   const int* __restrict__ offsets, int K,
                                                   const unsigned long long* __restrict__ keys, const int* __restrict__ vals,
                                                   unsigned mask, int* __restrict__ nbr) {
    const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
    const int n = real_n(n_dev, n_max);
    if (e >= (long long)n * K) return;
    const int row = (int)(e / K), k = (int)(e % K);
    const int4 c = reinterpret_cast<const int4*>(out_coords)[row];
    const int x = c.y + offsets[3 * k + 0], y = c.z + offsets[3 * k + 1], z = c.w + offsets[3 * k + 2];
    int r = -1;
    if (in_range(c.x, x, y, z)) r = table_find(keys, vals, mask, pack4(c.x, x, y, z));
    nbr[e] = r;
}

// up-conv (transposed k=2,s=2) table: fine row -> nbr_up[row][k] = parent row at k = child_k[row], -1 elsewhere
__global__ __launch_bounds__(TPB) void k_up_table(const int* __restrict__ parent_row, const int* __restrict__ child_k,
                                                 const int* n_dev, int n_max, int* __restrict__ nbr_up) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= real_n(n_dev, n_max)) return;
    const int k = child_k[i], p = parent_row[i];
    int4 lo = make_int4(-1, -1, -1, -1), hi = lo;
    int* v = (k < 4) ? &lo.x : &hi.x;
    v[k & 3] = p;
    reinterpret_cast<int4*>(nbr_up)[2 * (size_t)i + 0] = lo;
    reinterpret_cast<int4*>(nbr_up)[2 * (size_t)i + 1] = hi;
}

}  // namespace
}  // namespace pbn

using namespace pbn;

extern "C" int pbn_hash_capacity(int n) {
    unsigned c = 1024;
    while (c < 2u * (unsigned)(n > 0 ? n : 0)) c <<= 1;
    return (int)c;
}

extern "C" size_t pbn_coords_workspace_bytes(int n_max) {
    const size_t N = (size_t)(n_max > 0 ? n_max : 1);
    return 3 * align_up(N * sizeof(int), 256) + align_up(scan_tmp_ints((long long)N) * sizeof(int), 256) + 512;
}

namespace pbn {
namespace {
struct CoordWs {
    int *slot_of_row, *first_row, *newid, *scan_tmp, *status;
};
bool carve_ws(void* ws, size_t bytes, int n_max, CoordWs& w) {
    Carver cv(ws, bytes);
    const size_t N = (size_t)(n_max > 0 ? n_max : 1);
    w.slot_of_row = cv.take<int>(N);
    w.first_row = cv.take<int>(N);
    w.newid = cv.take<int>(N);
    w.scan_tmp = cv.take<int>(scan_tmp_ints((long long)N));
    w.status = cv.take<int>(4);
    return cv.ok;
}
}  // namespace

// clear = false: the caller has already filled the table (keys 0xff, values 0x7f), nbr_down (0xff) and zeroed the
// count / status words in bulk (pbn_coords_build does that for a whole pyramid with three memsets).
int coords_unique_impl(const int32_t* coords, const int32_t* n_dev, int n_max, uint64_t* table_keys, int32_t* table_vals,
                       int capacity, int32_t* unique_index, int32_t* inverse, int32_t* unique_coords, int32_t* n_unique,
                       void* workspace, size_t workspace_bytes, int32_t* status, bool clear, hipStream_t stream) {
    if (n_max < 0 || capacity < 1024 || (capacity & (capacity - 1)) || (long long)capacity < 2LL * n_max || !n_unique ||
        !table_keys || !table_vals)
        return PBN_ERR_ARG;
    if (clear) {
        { const int frc_ = fill_bytes(table_keys, 0xff, sizeof(uint64_t) * (size_t)capacity, stream); if (frc_ != PBN_OK) return frc_; }
        { const int frc_ = fill_bytes(table_vals, 0x7f, sizeof(int) * (size_t)capacity, stream); if (frc_ != PBN_OK) return frc_; }
        { const int frc_ = fill_bytes(n_unique, 0, sizeof(int), stream); if (frc_ != PBN_OK) return frc_; }
    }
    if (n_max == 0) return PBN_OK;
    if (!coords || !unique_index || !workspace) return PBN_ERR_ARG;
    CoordWs w;
    if (!carve_ws(workspace, workspace_bytes, n_max, w)) return PBN_ERR_WORKSPACE;
    if (!status) {
        status = w.status;
        { const int frc_ = fill_bytes(status, 0, sizeof(int) * 4, stream); if (frc_ != PBN_OK) return frc_; }
    }
    const int nb = cdiv(n_max, TPB), nsb = cdiv(n_max, SCAN_TILE);
    const unsigned mask = (unsigned)capacity - 1;
    hipLaunchKernelGGL(k_insert_rows, dim3(nb), dim3(TPB), 0, stream, coords, n_dev, n_max,
                       (unsigned long long*)table_keys, table_vals, mask, w.slot_of_row, status);
    hipLaunchKernelGGL(k_flag_block_sums, dim3(nsb), dim3(SCAN_THREADS), 0, stream, w.slot_of_row, table_vals, n_dev,
                       n_max, w.scan_tmp);
    hipLaunchKernelGGL(k_flag_number, dim3(nsb), dim3(SCAN_THREADS), 0, stream, w.slot_of_row, table_vals, n_dev, n_max,
                       w.scan_tmp, w.newid, w.first_row, n_unique, status);
    hipLaunchKernelGGL(k_unique_write, dim3(nb), dim3(TPB), 0, stream, coords, w.slot_of_row, w.first_row, w.newid, n_dev,
                       n_max, unique_index, inverse, unique_coords, table_vals);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

// survivor numbering over first-occurrence flags (the two launches of coords_unique_impl), for pyramid.hip
int coords_number_first(const int32_t* slot_of_row, const int32_t* table_vals, const int32_t* n_dev, int n_max, int32_t* scan_tmp,
                        int32_t* newid, int32_t* first_row, int32_t* n_out, const int32_t* status, hipStream_t stream) {
    if (n_max <= 0) return PBN_OK;
    const int nsb = cdiv(n_max, SCAN_TILE);
    hipLaunchKernelGGL(k_flag_block_sums, dim3(nsb), dim3(SCAN_THREADS), 0, stream, slot_of_row, table_vals, n_dev, n_max, scan_tmp);
    hipLaunchKernelGGL(k_flag_number, dim3(nsb), dim3(SCAN_THREADS), 0, stream, slot_of_row, table_vals, n_dev, n_max, scan_tmp,
                       newid, first_row, n_out, status);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

int coords_stride_impl(const int32_t* fine_coords, const int32_t* n_fine_dev, int n_fine_max, int stride_out,
                       uint64_t* table_keys, int32_t* table_vals, int capacity, int32_t* coarse_coords,
                       int32_t* parent_row, int32_t* child_k, int32_t* nbr_down, int32_t* nbr_up, int32_t* n_coarse,
                       void* workspace, size_t workspace_bytes, int32_t* status, bool clear, hipStream_t stream) {
    if (n_fine_max < 0 || stride_out < 2 || (stride_out & 1) || capacity < 1024 || (capacity & (capacity - 1)) ||
        (long long)capacity < 2LL * n_fine_max || !n_coarse || !table_keys || !table_vals)
        return PBN_ERR_ARG;
    if (clear) {
        { const int frc_ = fill_bytes(table_keys, 0xff, sizeof(uint64_t) * (size_t)capacity, stream); if (frc_ != PBN_OK) return frc_; }
        { const int frc_ = fill_bytes(table_vals, 0x7f, sizeof(int) * (size_t)capacity, stream); if (frc_ != PBN_OK) return frc_; }
        { const int frc_ = fill_bytes(n_coarse, 0, sizeof(int), stream); if (frc_ != PBN_OK) return frc_; }
    }
    if (n_fine_max == 0) return PBN_OK;
    if (!fine_coords || !coarse_coords || !parent_row || !child_k || !nbr_down || !workspace) return PBN_ERR_ARG;
    CoordWs w;
    if (!carve_ws(workspace, workspace_bytes, n_fine_max, w)) return PBN_ERR_WORKSPACE;
    if (clear) { const int frc_ = fill_bytes(nbr_down, 0xff, sizeof(int) * 8 * (size_t)n_fine_max, stream); if (frc_ != PBN_OK) return frc_; }
    if (!status) {
        status = w.status;
        { const int frc_ = fill_bytes(status, 0, sizeof(int) * 4, stream); if (frc_ != PBN_OK) return frc_; }
    }
    const int nb = cdiv(n_fine_max, TPB), nsb = cdiv(n_fine_max, SCAN_TILE);
    const unsigned mask = (unsigned)capacity - 1;
    hipLaunchKernelGGL(k_insert_parents, dim3(nb), dim3(TPB), 0, stream, fine_coords, n_fine_dev, n_fine_max, stride_out,
                       (unsigned long long*)table_keys, table_vals, mask, w.slot_of_row, status);
    hipLaunchKernelGGL(k_flag_block_sums, dim3(nsb), dim3(SCAN_THREADS), 0, stream, w.slot_of_row, table_vals, n_fine_dev,
                       n_fine_max, w.scan_tmp);
    hipLaunchKernelGGL(k_flag_number, dim3(nsb), dim3(SCAN_THREADS), 0, stream, w.slot_of_row, table_vals, n_fine_dev,
                       n_fine_max, w.scan_tmp, w.newid, w.first_row, n_coarse, (const int*)status);
    hipLaunchKernelGGL(k_stride_write, dim3(nb), dim3(TPB), 0, stream, fine_coords, n_fine_dev, n_fine_max, stride_out,
                       w.slot_of_row, w.first_row, w.newid, table_vals, coarse_coords, parent_row, child_k, nbr_down,
                       nbr_up);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}
}  // namespace pbn

extern "C" int pbn_coords_unique(const int32_t* coords, const int32_t* n_dev, int n_max, uint64_t* table_keys,
                                 int32_t* table_vals, int capacity, int32_t* unique_index, int32_t* inverse,
                                 int32_t* unique_coords, int32_t* n_unique, void* workspace, size_t workspace_bytes,
                                 pbn_stream_t stream_) {
    return coords_unique_impl(coords, n_dev, n_max, table_keys, table_vals, capacity, unique_index, inverse, unique_coords,
                              n_unique, workspace, workspace_bytes, nullptr, true, (hipStream_t)stream_);
}

extern "C" int pbn_coords_stride(const int32_t* fine_coords, const int32_t* n_fine_dev, int n_fine_max, int stride_out,
                                 uint64_t* table_keys, int32_t* table_vals, int capacity, int32_t* coarse_coords,
                                 int32_t* parent_row, int32_t* child_k, int32_t* nbr_down, int32_t* n_coarse,
                                 void* workspace, size_t workspace_bytes, pbn_stream_t stream_) {
    return coords_stride_impl(fine_coords, n_fine_dev, n_fine_max, stride_out, table_keys, table_vals, capacity,
                              coarse_coords, parent_row, child_k, nbr_down, nullptr, n_coarse, workspace, workspace_bytes,
                              nullptr, true, (hipStream_t)stream_);
}

extern "C" int pbn_kernel_map(const int32_t* out_coords, const int32_t* n_out_dev, int n_out_max, const int32_t* offsets,
                              int n_offsets, const uint64_t* table_keys, const int32_t* table_vals, int capacity,
                              int32_t* nbr, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_out_max < 0 || n_offsets < 1 || capacity < 1024 || (capacity & (capacity - 1))) return PBN_ERR_ARG;
    if (n_out_max == 0) return PBN_OK;
    if (!out_coords || !offsets || !table_keys || !table_vals || !nbr) return PBN_ERR_ARG;
    const long long total = (long long)n_out_max * n_offsets;
    hipLaunchKernelGGL(k_kernel_map, dim3(cdiv(total, TPB)), dim3(TPB), 0, stream, out_coords, n_out_dev, n_out_max,
                       offsets, n_offsets, (const unsigned long long*)table_keys, table_vals, (unsigned)capacity - 1, nbr);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_up_table(const int32_t* parent_row, const int32_t* child_k, const int32_t* n_fine_dev, int n_fine_max,
                            int32_t* nbr_up, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_fine_max < 0) return PBN_ERR_ARG;
    if (n_fine_max == 0) return PBN_OK;
    if (!parent_row || !child_k || !nbr_up) return PBN_ERR_ARG;
    hipLaunchKernelGGL(k_up_table, dim3(cdiv(n_fine_max, TPB)), dim3(TPB), 0, stream, parent_row, child_k, n_fine_dev,
                       n_fine_max, nbr_up);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

// ---- kernel map of a K^3 hyper-cube generated on the fly (no offsets array): ME conventions C1/C2 ---------------------
namespace pbn {
namespace {
__global__ __launch_bounds__(TPB) void k_kernel_map_cube(const int* __restrict__ out_coords, const int* n_dev, int n_max,
                                                        int ksize, int stride, int x_fastest,
                                                        const unsigned long long* __restrict__ keys,
                                                        const int* __restrict__ vals, unsigned mask, int* __restrict__ nbr) {
    // grid-stride over the REAL row count: the launch only knows an upper bound (the input size), and the coarse levels
    // hold a few hundred rows of it
    const int K = ksize * ksize * ksize;
    const int n = real_n(n_dev, n_max);
    const long long total = (long long)n * K;
    const int c0 = (ksize & 1) ? ksize / 2 : 0;  // odd kernels are centred, even ones are not
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const int row = (int)(e / K), k = (int)(e % K);
        int a = k % ksize - c0, b = (k / ksize) % ksize - c0, c = k / (ksize * ksize) - c0;
        const int dx = (x_fastest ? a : c) * stride, dy = b * stride, dz = (x_fastest ? c : a) * stride;
        const int4 cc = reinterpret_cast<const int4*>(out_coords)[row];
        const int x = cc.y + dx, y = cc.z + dy, z = cc.w + dz;
        int r = -1;
        if (in_range(cc.x, x, y, z)) r = table_find(keys, vals, mask, pack4(cc.x, x, y, z));
        nbr[e] = r;
    }
}
}  // namespace

// All cube maps of a pyramid (the k=3 map of every level, the k=5 map of level 0) in ONE launch: job j covers
// n_j * K_j table entries, the grid strides over their concatenation (row counts are read on the device).
namespace {
__global__ __launch_bounds__(TPB) void k_kernel_maps_multi(const MapJobs jb) {
    long long first[7];
    first[0] = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        long long cnt = 0;
        if (j < jb.n_jobs) cnt = (long long)real_n(jb.n_dev[j], jb.n_max) * (jb.ksize[j] * jb.ksize[j] * jb.ksize[j]);
        first[j + 1] = first[j] + cnt;
    }
    for (long long g = (long long)blockIdx.x * TPB + threadIdx.x; g < first[6]; g += (long long)gridDim.x * TPB) {
        int j = 0;
#pragma unroll
        for (int t = 1; t < 6; ++t) j += (t < jb.n_jobs && g >= first[t]) ? 1 : 0;
        const long long e = g - first[j];
        const int ksize = jb.ksize[j], K = ksize * ksize * ksize, stride = jb.stride[j];
        const int c0 = (ksize & 1) ? ksize / 2 : 0;
        const int row = (int)(e / K), k = (int)(e % K);
        const int a = k % ksize - c0, b = (k / ksize) % ksize - c0, c = k / (ksize * ksize) - c0;
        const int dx = (jb.x_fastest ? a : c) * stride, dy = b * stride, dz = (jb.x_fastest ? c : a) * stride;
        const int4 cc = reinterpret_cast<const int4*>(jb.coords[j])[row];
        const int x = cc.y + dx, y = cc.z + dy, z = cc.w + dz;
        int r = -1;
        if (in_range(cc.x, x, y, z)) r = table_find(jb.keys[j], jb.vals[j], jb.mask[j], pack4(cc.x, x, y, z));
        jb.nbr[j][e] = r;
    }
}
}  // namespace

int kernel_maps_multi(const MapJobs& jb, long long total_max, hipStream_t stream) {
    if (jb.n_jobs < 1 || jb.n_jobs > 6) return PBN_ERR_ARG;
    if (total_max <= 0) return PBN_OK;
    long long blocks = cdiv(total_max, TPB);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_kernel_maps_multi, dim3((unsigned)blocks), dim3(TPB), 0, stream, jb);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}
}  // namespace pbn

extern "C" int pbn_kernel_map_cube(const int32_t* out_coords, const int32_t* n_out_dev, int n_out_max, int kernel_size,
                                   int tensor_stride, int x_fastest, const uint64_t* table_keys,
                                   const int32_t* table_vals, int capacity, int32_t* nbr, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_out_max < 0 || kernel_size < 1 || tensor_stride < 1 || capacity < 1024 || (capacity & (capacity - 1)))
        return PBN_ERR_ARG;
    if (n_out_max == 0) return PBN_OK;
    if (!out_coords || !table_keys || !table_vals || !nbr) return PBN_ERR_ARG;
    const long long total = (long long)n_out_max * kernel_size * kernel_size * kernel_size;
    const long long blocks = cdiv(total, TPB);
    hipLaunchKernelGGL(k_kernel_map_cube, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(TPB), 0, stream, out_coords, n_out_dev, n_out_max,
                       kernel_size, tensor_stride, x_fastest, (const unsigned long long*)table_keys, table_vals,
                       (unsigned)capacity - 1, nbr);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}


// ---- Morton (Z-order) keys: batch-major, then bit-interleaved (x, y, z): rows sorted by this key make every run of
// consecutive rows a compact spatial block, which is what the convolution tiles and their L2 locality want -------------
namespace pbn {
namespace {
__global__ __launch_bounds__(TPB) void k_morton_keys(const int* __restrict__ coords, const int* n_dev, int n_max,
                                                    long long* __restrict__ keys) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n_max) return;
    if (i >= real_n(n_dev, n_max)) { keys[i] = 0x7fffffffffffffffLL; return; }  // padding rows sort to the end
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const unsigned long long m = spread3((unsigned)(c.y + 32768)) | (spread3((unsigned)(c.z + 32768)) << 1) |
                                 (spread3((unsigned)(c.w + 32768)) << 2);
    keys[i] = (long long)((((unsigned long long)(unsigned)c.x & 0x7fffULL) << 48) | (m & 0xffffffffffffULL));
}
}  // namespace
}  // namespace pbn

extern "C" int pbn_morton_keys(const int32_t* coords, const int32_t* n_dev, int n_max, int64_t* keys, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_max < 0) return PBN_ERR_ARG;
    if (n_max == 0) return PBN_OK;
    if (!coords || !keys) return PBN_ERR_ARG;
    hipLaunchKernelGGL(k_morton_keys, dim3(cdiv(n_max, TPB)), dim3(TPB), 0, stream, coords, n_dev, n_max, (long long*)keys);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

// ---- internals for the one-call prepare path (prepare.hip) -----------------------------------------------------------
namespace pbn {
namespace {
// level-0 table of rows that are already unique: value = row id, coordinates copied, no numbering pass
__global__ __launch_bounds__(TPB) void k_insert_identity(const int* __restrict__ coords, const int* n_dev, int n_max,
                                                        unsigned long long* __restrict__ keys, int* __restrict__ vals,
                                                        unsigned mask, int* __restrict__ out_coords,
                                                        int* __restrict__ n_out, int* __restrict__ status) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    const int n = real_n(n_dev, n_max);
    if (i == 0) *n_out = n_dev ? *n_dev : n_max;      // keeps a -1 (range error) visible
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    reinterpret_cast<int4*>(out_coords)[i] = c;
    unsigned h = hash64(pack4(c.x, c.y, c.z, c.w)) & mask;
    const unsigned long long key = pack4(c.x, c.y, c.z, c.w);
    for (unsigned probes = 0; probes <= mask; ++probes) {
        if (atomicCAS(&keys[h], EMPTY_KEY, key) == EMPTY_KEY) { vals[h] = i; return; }
        h = (h + 1) & mask;
    }
    if (status) atomicOr(status, PBN_STATUS_TABLE_FULL);
}

__global__ __launch_bounds__(TPB) void k_morton_iota(const int* __restrict__ coords, const int* n_dev, int n_max,
                                                    unsigned long long* __restrict__ keys, int* __restrict__ iota) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n_max) return;
    iota[i] = i;
    if (i >= real_n(n_dev, n_max)) { keys[i] = 0x7fffffffffffffffULL; return; }  // padding rows sort to the end
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const unsigned long long m = spread3((unsigned)(c.y + 32768)) | (spread3((unsigned)(c.z + 32768)) << 1) |
                                 (spread3((unsigned)(c.w + 32768)) << 2);
    keys[i] = (((unsigned long long)(unsigned)c.x & 0x7fffULL) << 48) | (m & 0xffffffffffffULL);
}

// sorted coordinates + the permutation pair, and the 64-bit copies of the de-duplication maps (torch indexes with int64)
__global__ __launch_bounds__(TPB) void k_apply_perm(const int* __restrict__ ucoords, const int* __restrict__ perm32,
                                                   const int* __restrict__ uidx32, const int* __restrict__ inv32, int n_max,
                                                   int* __restrict__ sorted_coords, long long* __restrict__ perm64,
                                                   long long* __restrict__ inv_perm64, long long* __restrict__ uidx64,
                                                   long long* __restrict__ inv64) {
    const int j = blockIdx.x * TPB + threadIdx.x;
    if (j >= n_max) return;
    const int r = perm32[j];
    reinterpret_cast<int4*>(sorted_coords)[j] = reinterpret_cast<const int4*>(ucoords)[r];
    perm64[j] = r;
    inv_perm64[r] = j;
    uidx64[j] = uidx32[j];
    inv64[j] = inv32[j];
}
}  // namespace

int coords_insert_identity(const int32_t* coords, const int32_t* n_dev, int n_max, uint64_t* keys, int32_t* vals,
                           int capacity, int32_t* out_coords, int32_t* n_out, int32_t* status, hipStream_t stream) {
    if (n_max <= 0) return PBN_OK;
    hipLaunchKernelGGL(k_insert_identity, dim3(cdiv(n_max, TPB)), dim3(TPB), 0, stream, coords, n_dev, n_max,
                       (unsigned long long*)keys, vals, (unsigned)capacity - 1, out_coords, n_out, status);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}
int coords_morton_iota(const int32_t* coords, const int32_t* n_dev, int n_max, uint64_t* keys, int32_t* iota,
                       hipStream_t stream) {
    if (n_max <= 0) return PBN_OK;
    hipLaunchKernelGGL(k_morton_iota, dim3(cdiv(n_max, TPB)), dim3(TPB), 0, stream, coords, n_dev, n_max,
                       (unsigned long long*)keys, iota);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}
int coords_apply_perm(const int32_t* ucoords, const int32_t* perm32, const int32_t* uidx32, const int32_t* inv32, int n_max,
                      int32_t* sorted_coords, int64_t* perm64, int64_t* inv_perm64, int64_t* uidx64, int64_t* inv64,
                      hipStream_t stream) {
    if (n_max <= 0) return PBN_OK;
    hipLaunchKernelGGL(k_apply_perm, dim3(cdiv(n_max, TPB)), dim3(TPB), 0, stream, ucoords, perm32, uidx32, inv32, n_max,
                       sorted_coords, (long long*)perm64, (long long*)inv_perm64, (long long*)uidx64, (long long*)inv64);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}
}  // namespace pbn
