// Encode side (placeholder until the kernels land): reports LZFSE_MI_UNSUPPORTED.
#include "internal.h"

namespace lzmi {

void enc_scratch_release(EncScratch &s) {
    for (int i = 0; i < 16; i++) {
        if (s.bufs[i]) (void)hipFree(s.bufs[i]);
        s.bufs[i] = nullptr;
        s.caps[i] = 0;
    }
}

int enc_batch_device(lzfse_mi_ctx *, uint32_t count, const uint8_t *, const uint64_t *, const uint64_t *, uint8_t *,
                     const uint64_t *, const uint64_t *, uint64_t *out_lens, int *statuses) {
    for (uint32_t i = 0; i < count; i++) {
        out_lens[i] = 0;
        statuses[i] = LZFSE_MI_UNSUPPORTED;
    }
    return LZFSE_MI_OK;
}

}  // namespace lzmi
