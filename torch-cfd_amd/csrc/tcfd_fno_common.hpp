// tcfd_fno_common.hpp -- host-side helpers shared by the translation units of the FNO kernels
// (tcfd_fno.hip: plans, pruned transforms, contraction; tcfd_fno_pw.hip: pointwise block forward / backward, LayerNorm
//  statistics, lifting operator, small reductions; tcfd_fno_tiles.hip: the tiled all-MFMA pointwise backward).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/tcfd.h"

extern "C" const char* tcfd_last_error(void);
int tcfd_set_error(int code, const char* fmt, ...);  // defined in tcfd_ns2d.hip
#define FAIL(...) tcfd_set_error(__VA_ARGS__)
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return FAIL(TCFD_EHIP, "%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

template <typename K>
static int set_lds_attr(K kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)bytes));
    return 0;
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

// ------------------------------------------------------------------ per-launch event timing of the FNO kernels (measurement aid)
// Between tcfd_fno_profile_begin and tcfd_fno_profile_end every kernel launched through a FnoProfScope is bracketed by a pair of
// HIP events on ITS launch stream (the solver's tcfd_ns2d_profile_begin / _end, process-wide here: the pointwise entry points
// have no plan).  Off: one load of a global flag per launch.  kinds:
enum { FNO_K_FWD_TY = 0, FNO_K_FWD_X = 1, FNO_K_CONTRACT = 2, FNO_K_INV_X = 3, FNO_K_INV_TY = 4, FNO_K_POINTWISE = 5,
       FNO_K_POINTWISE_BWD = 6, FNO_K_POINTWISE_1 = 7 /* single-layer forms */, FNO_K_CONTRACT_WGRAD = 8, FNO_K_OTHER = 9, FNO_K_POINTWISE_BWD_1 = 10 /* backward of the single-layer forms */ };
extern bool tcfd_fno_prof_on;                       // tcfd_fno.hip
int tcfd_fno_prof_open(int kind, hipStream_t st);   // -> record index or -1
void tcfd_fno_prof_close(int idx, hipStream_t st);
struct FnoProfScope {
    int idx;
    hipStream_t st;
    FnoProfScope(int kind, hipStream_t s) : idx(tcfd_fno_prof_on ? tcfd_fno_prof_open(kind, s) : -1), st(s) {}
    ~FnoProfScope() { if (idx >= 0) tcfd_fno_prof_close(idx, st); }
};
