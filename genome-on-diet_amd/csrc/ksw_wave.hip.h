// placeholder until the register-resident kernel lands (next commit)
#pragma once
#include <hip/hip_runtime.h>
#include "ksw_common.h"
static inline bool gd_wave_supported(int, int, int, int) { return false; }
static inline void gd_launch_wave64(const KswTask *, const int32_t *, int, const uint8_t *, const uint8_t *, uint8_t *, int32_t *, int32_t *, KswConst, hipStream_t) {}
static inline void gd_launch_wave16(const KswTask *, const int32_t *, int, const uint8_t *, const uint8_t *, uint8_t *, int32_t *, int32_t *, KswConst, hipStream_t) {}
