// sort_scan.hip -- thin wrappers around the rocPRIM/hipCUB 64-bit key sort and unique of the read-out
// (the distinct neighbour pairs of a chunk, once per chunk boundary, when the pair hash table is not
// used).
//
// Nothing on the merge path comes from the library any more: its stable (key, value) radix sort is
// radix_sort.hip, scans, run detection and compactions are scan_device.h, the bucket sort of the edge
// slots is edge_sort.hip.
#include <hipcub/hipcub.hpp>

#include "device_graph.h"

namespace vsg {

size_t SortPairsU32LibTempBytes(int n) {
  size_t bytes = 0;
  VSG_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr,
                                             (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                             (uint32_t*)nullptr, n, 0, 32));
  return bytes;
}

void SortPairsU32Lib(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                     const uint32_t* vals_in, uint32_t* vals_out, int n, int end_bit, hipStream_t s) {
  VSG_HIP(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in,
                                             vals_out, n, 0, end_bit, s));
}

size_t SortKeysU64TempBytes(int n) {
  size_t bytes = 0;
  VSG_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, (const unsigned long long*)nullptr,
                                            (unsigned long long*)nullptr, n, 0, 64));
  return bytes;
}

void SortKeysU64(void* temp, size_t temp_bytes, const unsigned long long* in, unsigned long long* out, int n,
                 hipStream_t s) {
  VSG_HIP(hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, in, out, n, 0, 64, s));
}

size_t UniqueU64TempBytes(int n) {
  size_t bytes = 0;
  VSG_HIP(hipcub::DeviceSelect::Unique(nullptr, bytes, (const unsigned long long*)nullptr,
                                       (unsigned long long*)nullptr, (int32_t*)nullptr, n));
  return bytes;
}

void UniqueU64(void* temp, size_t temp_bytes, const unsigned long long* in, unsigned long long* out,
               int32_t* num_out, int n, hipStream_t s) {
  VSG_HIP(hipcub::DeviceSelect::Unique(temp, temp_bytes, in, out, num_out, n, s));
}

}  // namespace vsg
