// prepare.hip -- device radix sort used by pbn_coords_prepare (Z-ordering of the de-duplicated voxels).  Kept in its own
// translation unit: it is the only place that pulls in the hipCUB / rocPRIM headers.
#include <hipcub/hipcub.hpp>
#include "pbn_common.h"

namespace pbn {

size_t sort_pairs_temp_bytes(int n) {
    if (n <= 0) return 256;
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int32_t*)nullptr,
                                       (int32_t*)nullptr, n, 0, 63, (hipStream_t)0);
    return bytes ? bytes : 256;
}

// stable ascending sort of 63-bit keys (the Z-order key never sets bit 63) with their row ids
int sort_pairs_u64_i32(const uint64_t* keys_in, uint64_t* keys_out, const int32_t* vals_in, int32_t* vals_out, int n,
                       void* temp, size_t temp_bytes, hipStream_t stream) {
    if (n <= 0) return PBN_OK;
    size_t need = temp_bytes;
    PBN_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(temp, need, keys_in, keys_out, vals_in, vals_out, n, 0, 63, stream));
    return PBN_OK;
}

}  // namespace pbn
