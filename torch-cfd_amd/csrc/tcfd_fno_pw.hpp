// tcfd_fno_pw.hpp -- what the backward kernels of the fused pointwise block share across translation units
// (tcfd_fno.hip: the LDS-staged kernels and the register-resident all-MFMA kernel of widths <= 14;
//  tcfd_fno_bwd.hip: the tiled all-MFMA kernel of every width up to 32).
#pragma once
#include <hip/hip_runtime.h>

struct PwBwdArgs {
    const float* pe;     // (CI, P) or null.  Not null (k_pointwise_bwd only): x is ONE channel (b, 1, P) and the block input is
                         // x + pe[c] -- the lifting operator's input + positional encoding, never materialised
    const float* x;      // (b, CI, P)
    const float* s;      // (b, CI, P) skip input (skip_mode 1) or null
    const float* dout;   // (b, CO, P)
    const float* out;    // (b, CO, P) the block's forward OUTPUT, or null (k_pointwise_bwd_mfma with both activations ReLU: the
                         // mask of the output activation is read from it instead of recomputing z2 -- 13 of 93 MFMAs per 16 points)
    float* dx;           // (b, CI, P)
    float* ds;           // (b, CI, P) or null
    const float* w1;     // (CM, CI) or null
    const float* b1;
    const float* w2t;    // (CM, CO)
    const float* b2;
    const float* wst;    // (CI, CO)
    const float* bs;
    float* partials;     // (waves, PW_FLOATS) padded tiles, see pw_bwd_layout
    long P;
    long chunks_per_batch, total_chunks;
    int act1, act2, skip_mode;
    int T, sT;           // skip_mode 2: s is (b, CO, P / T * sT), its last time slice is added; ds receives dL/dz2 (b, CO, P)
    int per_sample;      // 1: wave w only visits batch element w % batch, so its partial sums belong to ONE sample
    int batch;
};

template <int CI, int CM, int CO, bool HAS_L1>
struct PwBwdGeom {
    static constexpr int COP = (CO + 15) / 16 * 16;
    static constexpr int CIP = (CI + 1 + 15) / 16 * 16;                   // [x, 1]
    static constexpr int CB = ((HAS_L1 ? CM : CI) + 1 + CI + 15) / 16 * 16; // [h, 1, s]   (single layer: [x, 1, s])
    static constexpr int CM1 = HAS_L1 ? (CM + 15) / 16 * 16 : 0;           // g1 rows (in place over h)
    static constexpr int PITCH = 66;
    static constexpr int R0 = COP > CIP ? COP : CIP;                       // rows of the first operand slot: g2, later [x, 1]
    static constexpr int ROWS = R0 + CB;
    static constexpr int N_A = COP * CB;                                   // g2 (x) [h, 1, s]
    static constexpr int N_B = CM1 * CIP;                                  // g1 (x) [x, 1]
    static constexpr int TOTAL = N_A + N_B;
    static constexpr int WAVES = 2;                                        // per workgroup (21 KB of LDS per wave at width 10)
};

template <int ACT>
__device__ __forceinline__ void pw_act_pair(float z, float& h, float& d) {
    if constexpr (ACT == 1) { h = z > 0.f ? z : 0.f; d = z > 0.f ? 1.f : 0.f; }
    else if constexpr (ACT == 2) {
        // GELU and its derivative from ONE exponential: erf(x) = 1 - (a1 t + ... + a5 t^5) exp(-x^2), t = 1 / (1 + p x), x >= 0
        // (Abramowitz & Stegun 7.1.26, |error| < 1.5e-7), and exp(-x^2) with x = |z| / sqrt(2) is the Gaussian of the
        // derivative's second term.  (erff + expf per element made the GELU backward 1.6 x the ReLU one.)
        const float ax = fabsf(z) * 0.70710678118654752f;
        const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
        const float e = __expf(-ax * ax);
        const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
        const float erf_abs = fmaf(-poly, e, 1.f);
        const float cdf = 0.5f * (1.f + (z < 0.f ? -erf_abs : erf_abs));
        h = z * cdf;
        d = fmaf(z * 0.3989422804014327f, e, cdf);
    } else if constexpr (ACT == 3) { const float sg = 1.f / (1.f + __expf(-z)); h = z * sg; d = sg * (1.f + z * (1.f - sg)); }
    else if constexpr (ACT == 4) { const float t = tanhf(z); h = t; d = 1.f - t * t; }
    else { h = z; d = 1.f; }
}

// tcfd_fno_bwd.hip: the tiled all-MFMA backward (two-layer form, P % 4 == 0).  Returns 0 and sets *handled = 1 when it took the
// call (or answered the layout query), leaves *handled = 0 for combinations it does not cover.
int tcfd_pwb_tiles_dispatch(const PwBwdArgs& a, int batch, int ci, int cm, int co, int max_rows, int* dims, hipStream_t st,
                            int* handled);
