/* zstd_dec.hip -- placeholder until the zstd frame decoder kernel lands. */
#include "kernels.h"
namespace cryo {
size_t zstd_decompress_workspace(uint64_t, uint32_t) { return 256; }
hipError_t launch_zstd_decompress(hipStream_t, const uint8_t *, const uint64_t *, const uint32_t *,
                                  uint8_t *, uint64_t, uint32_t, uint64_t, int32_t *, void *, size_t)
{
    return hipErrorNotSupported;
}
} // namespace cryo
