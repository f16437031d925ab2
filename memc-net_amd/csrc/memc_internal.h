// memc_internal.h -- private to libmemc_hip.so and its measurement scripts; NOT part of the drop-in ABI
// (that is include/memc_warp.h).
#pragma once
#include "memc_warp.h"

#ifdef __cplusplus
extern "C" {
#endif

// A/B measurement hooks (tools/bench_ops.py).  variant < 0 restores automatic selection.
// They change which of several equivalent kernels a launcher picks -- never the results.
void memc_debug_set_fi_fwd_variant(int variant);
void memc_debug_set_projection_variant(int variant);

#ifdef __cplusplus
}
#endif
