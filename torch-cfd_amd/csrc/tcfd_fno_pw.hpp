// tcfd_fno_pw.hpp -- what the backward kernels of the fused pointwise block share across translation units
// (tcfd_fno.hip: the LDS-staged kernels and the register-resident all-MFMA kernel of widths <= 14;
//  tcfd_fno_tiles.hip: the tiled all-MFMA kernel of every width up to 32).
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
    int ds_tsum;         // skip_mode 2 only: ds is (b, CO, P / T) and receives the SUM over t of dL/dz2 (the tiled kernel adds the T steps
                         // of a row itself; `tsum_groups` consecutive groups of 16 points hold whole rows)
    int tsum_groups;
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

// ------------------------------------------------------------------ fused pointwise block of the SFNO layer
//   out = act2( W2 . act1( W1 . x + b1 ) + b2  [+ Ws . s + bs | + s[..., -1:]] )
// i.e. PointwiseFFN (two 1x1x1 convolutions, fno/base.py:86-111) + the 1x1x1 skip convolution + sum + activation
// of one SFNO layer (fno/sfno.py:607-614), or the lifting operator's tail act(v[..., -1:] + mlp(.)) (:258-259),
// or a single 1x1x1 convolution (W1 absent).  One lane per point, channels in registers, weights through the
// scalar unit (they are lane uniform): the (b, C, P) activations are read once and written once, where the
// reference-style op stream makes ~6 passes and materialises the 4x wider hidden tensor.
struct PwArgs {
    const float* pe;    // (CI, P) or null.  Not null: x is ONE channel (b, 1, P) and the block input is x + pe[c]
                        // (the lifting operator's v + positional encoding, fno/sfno.py:109-113, never materialised)
    const float* x;     // (b, CI, P)
    const float* s;     // skip input or null: mode 1 (b, CI, P); mode 2 (b, CO, P / T * sT), last time slice is added
    float* out;         // (b, CO, P)
    const float* w1;    // (CM, CI) or null (then CM == CI and the hidden vector is x itself)
    const float* b1;    // (CM) or null
    const float* w2t;   // (CM, CO)  = W2 transposed
    const float* b2;    // (CO) or null
    const float* wst;   // (CI, CO)  = Ws transposed (mode 1)
    const float* bs;    // (CO) or null
    long P;
    long w2_bstride, b2_bstride;  // per-batch-element offsets of w2t / b2 (0: shared) -- lets a per-sample
                                  // affine map (e.g. a folded LayerNorm) ride in the single-layer form
    int T, sT, act1, act2, skip_mode;
    int cm;             // hidden width when the kernel is instantiated with CM = 0 (any channel expansion)
    float* pre;         // not null: the pre-activation z2 (b, CO, P) is stored as well -- what the backward of a block whose output
                        // activation is not ReLU needs (its derivative is a function of z2, not of the output); training only
    const float* frame; // not null: the output is (b, CO, P / T * (T + 1)) -- every (x, y) row of T steps is written behind ONE
    int fT;             // extra leading step that holds frame[b][xy][fT - 1] (frame (b, P / T, fT): the last input frame the output
                        // operator prepends to the latent steps, fno/sfno.py:314-315) -- its torch.cat never runs
};

// max(v, 0) as ONE v_max_f32 (fmaxf / a select add a canonicalising v_max_f32 v, v, v in front of it)
__device__ __forceinline__ float relu_f(float v) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
}
// GELU (exact form, torch default) of one value: the branch-free evaluation of gelu_pk below, see there.
__device__ __forceinline__ float gelu_f(float v) {
    const float u = fabsf(v);
    float p = fmaf(-1.690403337e-06f, u, 2.508333091e-05f);
    p = fmaf(p, u, -1.144628186e-04f);
    p = fmaf(p, u, -3.233417228e-04f);
    p = fmaf(p, u, 7.333383430e-03f);
    p = fmaf(p, u, -5.271419883e-02f);
    p = fmaf(p, u, -4.591154456e-01f);
    p = fmaf(p, u, -1.151123285e+00f);
    p = fmaf(p, u, -9.999988675e-01f);
    return fmaf(-u, __builtin_amdgcn_exp2f(p), relu_f(v));
}
__device__ __forceinline__ float pw_act(float v, int act) {
    switch (act) {
        case 1: return relu_f(v);                                          // ReLU
        case 2: return gelu_f(v);                                          // GELU (exact, torch default)
        case 3: return v / (1.f + __expf(-v));                             // SiLU
        case 4: return tanhf(v);
        default: return v;
    }
}

// V = 2: every lane carries two neighbouring points as a packed pair, so each weight (lane uniform, read through
// the scalar unit) feeds one v_pk_fma_f32 = two FMAs.  The block is VALU bound with one point per lane
// (900 FMAs per point at width 10: 0.73 ms against ~0.5 ms of HBM time), packed math is the fp32 vector peak.
typedef float v2f __attribute__((ext_vector_type(2)));
template <int V> struct PwVec { typedef float type; };
template <> struct PwVec<2> { typedef v2f type; };
// GELU of a packed pair without erff.  The library erff is ~40 instructions per element (two data-dependent branches and
// a full-range expf), four times the 2 x 10 packed FMAs of the hidden unit it follows -- the block was bound by it, not by its
// 900 FMAs per point.  Here  gelu(v) = v Phi(v) = max(v, 0) - |v| Phi(-|v|)  with  Phi(-u) = 2^-s(u):  s(u) = -log2 Phi(-u)
// is smooth (~ u^2 / 2 ln 2), one degree-8 polynomial covers every u (fitted with weight u Phi(-u), the sensitivity of the
// result; its leading coefficient is positive, so 2^-s underflows to 0 beyond the fitted range [0, 9]), and the hardware's
// v_exp_f32 IS 2^x.  Eight v_pk_fma_f32 + two v_exp_f32 per pair, no branch; error <= 8.4e-8 max(|gelu|, 1) for every
// finite v, i.e. tighter than the float32 formula 0.5 v (1 + erf(v / sqrt 2)) itself (its 1 + erf cancels for v < 0).
// When every lane of the wave has |v| < 2 the exponential is skipped too:  gelu(v) = v (1/2 + v P(v^2))  with a degree-6
// P (absolute error <= 2.7e-7); the test is wave uniform, so no lane diverges.  TCFD_GELU_SMALL 0 compiles that path out.
#ifndef TCFD_GELU_SMALL
#define TCFD_GELU_SMALL 1
#endif
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f gelu_pk(v2f v) {
#if TCFD_GELU_SMALL
    const bool big = !(fabsf(v.x) < 2.f) || !(fabsf(v.y) < 2.f);
    if (__builtin_amdgcn_ballot_w64(big) == 0) {
        const v2f s = v * v;
        v2f p = pk_fma(v2f{2.765524414e-07f, 2.765524414e-07f}, s, v2f{-7.518318853e-06f, -7.518318853e-06f});
        p = pk_fma(p, s, v2f{1.101917369e-04f, 1.101917369e-04f});
        p = pk_fma(p, s, v2f{-1.179484301e-03f, -1.179484301e-03f});
        p = pk_fma(p, s, v2f{9.967512451e-03f, 9.967512451e-03f});
        p = pk_fma(p, s, v2f{-6.648835540e-02f, -6.648835540e-02f});
        p = pk_fma(p, s, v2f{3.989420831e-01f, 3.989420831e-01f});
        return v * pk_fma(v, p, v2f{0.5f, 0.5f});
    }
#endif
    const v2f u = v2f{fabsf(v.x), fabsf(v.y)};
    v2f p = pk_fma(v2f{-1.690403337e-06f, -1.690403337e-06f}, u, v2f{2.508333091e-05f, 2.508333091e-05f});
    p = pk_fma(p, u, v2f{-1.144628186e-04f, -1.144628186e-04f});
    p = pk_fma(p, u, v2f{-3.233417228e-04f, -3.233417228e-04f});
    p = pk_fma(p, u, v2f{7.333383430e-03f, 7.333383430e-03f});
    p = pk_fma(p, u, v2f{-5.271419883e-02f, -5.271419883e-02f});
    p = pk_fma(p, u, v2f{-4.591154456e-01f, -4.591154456e-01f});
    p = pk_fma(p, u, v2f{-1.151123285e+00f, -1.151123285e+00f});
    p = pk_fma(p, u, v2f{-9.999988675e-01f, -9.999988675e-01f});
    const v2f e = v2f{__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
    return pk_fma(-u, e, v2f{relu_f(v.x), relu_f(v.y)});
}
// GELU and its derivative of a packed PAIR (the backward kernels' gate): the formula of pw_act_pair<2> -- Abramowitz & Stegun
// 7.1.26, one exponential shared by erf and the Gaussian of the derivative -- as 16 vector instructions (9 of them packed) + 2
// reciprocals + 2 exponentials per pair, against 2 x 17 + 4 for two scalar evaluations: the GELU backward spends a third of its
// issue slots here (50 evaluations per point at width 10).
__device__ __forceinline__ void pw_gelu_pair_pk(v2f z, v2f& h, v2f& d) {
    constexpr float C = 0.3275911f * 0.70710678118654752f;
    const v2f t = v2f{__builtin_amdgcn_rcpf(fmaf(fabsf(z.x), C, 1.f)), __builtin_amdgcn_rcpf(fmaf(fabsf(z.y), C, 1.f))};
    const v2f q = (z * z) * v2f{-0.72134752044448170f, -0.72134752044448170f};            // -(z^2 / 2) log2 e
    const v2f e = v2f{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};          // exp(-z^2 / 2)
    v2f p = pk_fma(t, v2f{1.061405429f, 1.061405429f}, v2f{-1.453152027f, -1.453152027f});
    p = pk_fma(p, t, v2f{1.421413741f, 1.421413741f});
    p = pk_fma(p, t, v2f{-0.284496736f, -0.284496736f});
    p = pk_fma(p, t, v2f{0.254829592f, 0.254829592f});
    p = p * t;
    const v2f erf_abs = pk_fma(-p, e, v2f{1.f, 1.f});
    const v2f es = v2f{__builtin_copysignf(erf_abs.x, z.x), __builtin_copysignf(erf_abs.y, z.y)};
    const v2f cdf = pk_fma(es, v2f{0.5f, 0.5f}, v2f{0.5f, 0.5f});
    h = z * cdf;
    d = pk_fma(z * e, v2f{0.3989422804014327f, 0.3989422804014327f}, cdf);
}
__device__ __forceinline__ v2f pw_act(v2f v, int act) {
    if (act == 2) return gelu_pk(v);
    return v2f{pw_act(v.x, act), pw_act(v.y, act)};
}
__device__ __forceinline__ float pw_fma(float w, float x, float acc) { return fmaf(w, x, acc); }
__device__ __forceinline__ v2f pw_fma(float w, v2f x, v2f acc) { return __builtin_elementwise_fma(v2f{w, w}, x, acc); }

// o = b2 + W2 . act1(W1 . x + b1)   (HAS_L1)   |   o = b2 + W2 . x   -- the block without its skip term and final activation
template <int CI, int CM, int CO, bool HAS_L1, typename vf, int ACT = -1>
__device__ __forceinline__ void pw_core(const PwArgs& a, int b, const vf (&x)[CI], vf (&o)[CO]) {
    const int act1 = ACT >= 0 ? ACT : a.act1;
    const float* w2t_b = a.w2t + (size_t)b * a.w2_bstride;
    const float* b2_b = a.b2 ? a.b2 + (size_t)b * a.b2_bstride : nullptr;
#pragma unroll
    for (int c = 0; c < CO; ++c) o[c] = (vf)(b2_b ? b2_b[c] : 0.f);
    if constexpr (HAS_L1) {
        const int cm = CM > 0 ? CM : a.cm;   // CM = 0: hidden width at run time (it is only a trip count)
#pragma unroll 4
        for (int m = 0; m < cm; ++m) {
            vf h = (vf)(a.b1 ? a.b1[m] : 0.f);
            const float* w1 = a.w1 + m * CI;
#pragma unroll
            for (int i = 0; i < CI; ++i) h = pw_fma(w1[i], x[i], h);
            h = pw_act(h, act1);
            const float* w2 = w2t_b + m * CO;
#pragma unroll
            for (int c = 0; c < CO; ++c) o[c] = pw_fma(w2[c], h, o[c]);
        }
    } else {
#pragma unroll
        for (int m = 0; m < CI; ++m) {
            const float* w2 = w2t_b + m * CO;
#pragma unroll
            for (int c = 0; c < CO; ++c) o[c] = pw_fma(w2[c], x[m], o[c]);
        }
    }
}
// o += Ws . s + bs   (the 1x1x1 skip convolution)
template <int CI, int CO, typename vf>
__device__ __forceinline__ void pw_skip_conv(const PwArgs& a, const vf (&sv)[CI], vf (&o)[CO]) {
#pragma unroll
    for (int i = 0; i < CI; ++i) {
        const float* ws = a.wst + i * CO;
#pragma unroll
        for (int c = 0; c < CO; ++c) o[c] = pw_fma(ws[c], sv[i], o[c]);
    }
    if (a.bs) {
#pragma unroll
        for (int c = 0; c < CO; ++c) o[c] += (vf)a.bs[c];
    }
}

// ACT >= 0: both activations are that code at compile time (the reference's ReLU / ReLU and GELU / GELU layers): the
// run-time switch inside the hidden-unit loop costs ~25 scalar instructions and several taken branches per unit.
// activations are read once and outlive every cache: TCFD_PW_NT_LOADS=1 at build time marks the reads non-temporal as well
#ifndef TCFD_PW_NT_LOADS
#define TCFD_PW_NT_LOADS 1
#endif
#if TCFD_PW_NT_LOADS
#define PW_LOAD(p_) __builtin_nontemporal_load(p_)
#else
#define PW_LOAD(p_) (*(p_))
#endif

// tcfd_fno_tiles.hip: the tiled all-MFMA backward (two-layer form, P % 4 == 0).  Returns 0 and sets *handled = 1 when it took the
// call (or answered the layout query), leaves *handled = 0 for combinations it does not cover.
int tcfd_pwb_tiles_dispatch(const PwBwdArgs& a, int batch, int ci, int cm, int co, int max_rows, int* dims, hipStream_t st,
                            int* handled);
// tcfd_fno_tiles.hip: the forward block of the wide layers on the matrix pipe (same contract: *handled = 0 -> not covered)
int tcfd_pwf_tiles_dispatch(const PwArgs& a, int batch, int ci, int cm, int co, hipStream_t st, int* handled);
