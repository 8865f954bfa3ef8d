// attention.hip - placeholder until the fused attention kernels land (next commit).
#include "common.h"
extern "C" int mm_attn_fwd(void*, float*, const void*, const uint8_t*, int, int, int, int, int, int, float, int, void*) {
    mm::set_error("mm_attn_fwd: not built yet");
    return MM_EUNSUPPORTED;
}
extern "C" int mm_attn_bwd(void*, float*, const void*, const void*, const float*, const void*, const uint8_t*, int, int, int, int,
                           int, int, float, int, void*) {
    mm::set_error("mm_attn_bwd: not built yet");
    return MM_EUNSUPPORTED;
}
