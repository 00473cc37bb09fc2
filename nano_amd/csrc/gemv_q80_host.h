// gemv_q80_host.h -- host-side routing shared by the Q80 GEMV translation units.
#pragma once
#include "kernels.h"
#include <hip/hip_ext.h>
#include <stdlib.h>

namespace nano {

constexpr uint32_t STREAM_MIN_ROWS = 16384;     // taller matrices go to the stream kernel
// workgroups of a STREAM launch: 1024 = two resident rounds of 512 (2 per CU at the kernel's register footprint)
static inline uint32_t stream_wgs() { return 1024u; }
#define STREAM_WGS (::nano::stream_wgs())

static inline uint32_t total_rows(const GemvArgs &a) {
    if (a.epi == GEMV_EPI_SWIGLU) return a.seg[0].rows;
    uint32_t r = 0;
    for (uint32_t s = 0; s < a.nseg; s++) r += a.seg[s].rows;
    return r;
}
static inline bool use_stream(const GemvArgs &a) {
    return a.nseg == 1 && a.epi == GEMV_EPI_STORE && a.seg[0].rows >= STREAM_MIN_ROWS && a.seg[0].out_pstride == 0 && !a.attn_part && a.nb <= 8;
}


// measurement hook: when both are set, the next STREAM (classifier) launch is issued with hipExtLaunchKernelGGL so that the
// two events carry the kernel's own start / stop timestamps (what rocprofv3 reports), then the hook clears itself
extern hipEvent_t g_q80_probe_start, g_q80_probe_stop;

}  // namespace nano
