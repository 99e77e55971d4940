// Prefix sums and flag selection over device arrays -- the two primitives the host side of the kernels asks for between launches
// (row offsets from row sizes, list numbering from flags).  Three short kernels per call: tile sums, their scan by one block, the
// tiles again with their offsets.  Written here rather than taken from hipCUB: its dispatch instantiates every kernel once per known
// architecture, and the four translation units that used it carried 3 300 kernel stubs (8.5 of the library's 9.4 MB of device
// code, symbol names mostly) that the runtime registers at start and walks through when it loads a code object -- a tenth of a
// second of a run that takes seven.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pf {

// bytes of device scratch any of the calls below needs for n elements
size_t scan_scratch_bytes(uint64_t n);

// out[i] = in[0] + ... + in[i - 1] (exclusive) or ... + in[i] (inclusive); in and out do not overlap.
hipError_t scan_exclusive_u32(const uint32_t *in, uint32_t *out, uint64_t n, void *scratch, hipStream_t st);
hipError_t scan_inclusive_u32(const uint32_t *in, uint32_t *out, uint64_t n, void *scratch, hipStream_t st);
hipError_t scan_exclusive_u32_u64(const uint32_t *in, uint64_t *out, uint64_t n, void *scratch, hipStream_t st);   // sums in 64 bits
hipError_t scan_exclusive_u64(const uint64_t *in, uint64_t *out, uint64_t n, void *scratch, hipStream_t st);

// ids[0 .. count) = the indices i < n with flags[i] != 0, ascending; the count goes to *count32 and / or *count64 (device, either
// may be null)
hipError_t select_flagged_u8(const uint8_t *flags, uint32_t *ids, uint32_t *count32, uint64_t *count64, uint64_t n, void *scratch, hipStream_t st);
hipError_t select_flagged_u32(const uint32_t *flags, uint32_t *ids, uint32_t *count32, uint64_t *count64, uint64_t n, void *scratch, hipStream_t st);

}  // namespace pf
