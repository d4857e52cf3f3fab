// sort_general.hip — descending radix sort of 64-bit sortkeys for the large-k / collect-all path
// (k beyond the fused top-k tiers, or k >= nrows: the reference's collect-all branch,
// crates/frankensearch-index/src/search.rs:449-473).  Uses rocPRIM's device radix sort (AMD's own
// primitive library, header-only in /opt/rocm/include); the hot fused path never comes here.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "kernels.hpp"

namespace fsgpu {

hipError_t sort_keys_desc_temp_bytes(size_t n, size_t* temp_bytes) {
    *temp_bytes = 0;
    return rocprim::radix_sort_keys_desc(nullptr, *temp_bytes, (const u64*)nullptr, (u64*)nullptr, n, 0, 64, 0);
}

hipError_t sort_keys_desc(void* temp, size_t temp_bytes, const u64* keys_in, u64* keys_out, size_t n,
                          hipStream_t stream) {
    return rocprim::radix_sort_keys_desc(temp, temp_bytes, keys_in, keys_out, n, 0, 64, stream);
}

}  // namespace fsgpu
