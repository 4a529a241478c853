// sort_scan.hip -- thin wrappers around the rocPRIM/hipCUB device-wide radix sort (and the 64-bit
// unique of the read-out).
//
// The stable LSD radix sort is the one generic building block left from the library; scans, run
// detection and compactions are hand-written and fused into the kernels around them
// (scan_device.h), like every kernel that encodes the segmentation algorithm itself
// (build_kernels.hip / merge_*.hip / readout_kernels.hip; the bucket sort of the edge slots:
// edge_sort.hip).
#include <hipcub/hipcub.hpp>

#include "device_graph.h"

namespace vsg {

size_t SortPairsU32TempBytes(int n) {
  size_t bytes = 0;
  VSG_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr,
                                             (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                             (uint32_t*)nullptr, n, 0, 32));
  return bytes;
}

void SortPairsU32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
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
