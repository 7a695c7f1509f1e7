// tcfd_fno_pw.hip -- MI355X (gfx950) kernels + C ABI for everything of an SFNO layer that is NOT a transform:
// the fused pointwise block (PointwiseFFN + skip convolution + activation, fno/base.py:86-111, fno/sfno.py:607-614) forward
// (fp32 / fp64) and backward, the LayerNormnd statistics and the folded lifting projection (fno/sfno.py:252-259), and the
// small reductions of the training step.  The transforms and the contraction are tcfd_fno.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/tcfd.h"
#include "tcfd_fft.hpp"
#include "tcfd_fno_common.hpp"
#include "tcfd_fno_pw.hpp"

using namespace tcfd;
typedef cx<float> cf;

// (PwArgs, the activations and pw_core / pw_skip_conv -- the per-point arithmetic of the block -- live in tcfd_fno_pw.hpp: the fused
//  pointwise + forward-transform kernel of tcfd_fno.hip runs the same code)
template <int CI, int CM, int CO, bool HAS_L1, int V, int ACT = -1>
__global__ __launch_bounds__(256) void k_pointwise(PwArgs a) {
    typedef typename PwVec<V>::type vf;
    const int act2 = ACT >= 0 ? ACT : a.act2;
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * V;
    const int b = blockIdx.y;
    if (p >= a.P) return;
    vf x[CI], o[CO];
    if (a.pe) {
        const vf v1 = *reinterpret_cast<const vf*>(a.x + (size_t)b * a.P + p);
#pragma unroll
        for (int i = 0; i < CI; ++i) x[i] = v1 + *reinterpret_cast<const vf*>(a.pe + (size_t)i * a.P + p);
    } else {
        const float* xb = a.x + (size_t)b * CI * a.P + p;
#pragma unroll
        for (int i = 0; i < CI; ++i) x[i] = PW_LOAD(reinterpret_cast<const vf*>(xb + (size_t)i * a.P));
    }
    pw_core<CI, CM, CO, HAS_L1, vf, ACT>(a, b, x, o);
    if (a.skip_mode == 1) {
        const float* sb = a.s + (size_t)b * CI * a.P + p;
        vf sv[CI];
#pragma unroll
        for (int i = 0; i < CI; ++i) sv[i] = PW_LOAD(reinterpret_cast<const vf*>(sb + (size_t)i * a.P));
        pw_skip_conv<CI, CO, vf>(a, sv, o);
    } else if (a.skip_mode == 2) {
        const long xy = p / a.T;   // V = 2 needs an even T: both points of a lane share (x, y)
        const long sP = (a.P / a.T) * a.sT;
        const float* sb = a.s + (size_t)b * CO * sP + xy * a.sT + (a.sT - 1);
#pragma unroll
        for (int c = 0; c < CO; ++c) o[c] += (vf)sb[(size_t)c * sP];
    }
    if constexpr (!HAS_L1 && CO == 1) if (a.frame) {     // (the channel reduction in front of the output operator only)
        const long xy = p / a.T;             // V = 2 needs an even T (checked by the host): both points of a lane share (x, y)
        const int t = (int)(p - xy * a.T);
        const long oP = (a.P / a.T) * (a.T + 1);
        float* ob = a.out + (size_t)b * CO * oP + xy * (a.T + 1) + t + 1;
        const float fr = a.frame[((size_t)b * (a.P / a.T) + xy) * a.fT + (a.fT - 1)];
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const vf r = pw_act(o[c], act2);
            float* oc = ob + (size_t)c * oP;
            if constexpr (V == 2) { oc[0] = r.x; oc[1] = r.y; } else { oc[0] = r; }
            if (t == 0) oc[-1] = fr;
        }
        return;
    }
    float* ob = a.out + (size_t)b * CO * a.P + p;
    if (a.pre) {
        float* zb = a.pre + (size_t)b * CO * a.P + p;
#pragma unroll
        for (int c = 0; c < CO; ++c) __builtin_nontemporal_store(o[c], reinterpret_cast<vf*>(zb + (size_t)c * a.P));
    }
#pragma unroll
    for (int c = 0; c < CO; ++c)   // streamed out (non-temporal): an 0.8 GB activation tensor outlives every cache; SFNO forward 5.77 -> 5.42 ms
        __builtin_nontemporal_store(pw_act(o[c], act2), reinterpret_cast<vf*>(ob + (size_t)c * a.P));
}

template <int CI, int CM, int CO, bool HAS_L1>
static int launch_pw(const PwArgs& a, int batch, hipStream_t st) {
    FnoProfScope prof(HAS_L1 ? FNO_K_POINTWISE : FNO_K_POINTWISE_1, st);
    // packed pairs need 8-byte aligned rows: even P (every channel row starts on a pair), even T for the
    // broadcast skip, 8-byte aligned base pointers
    // (width 32 keeps one point per lane: two need > 128 VGPRs and lose more in occupancy than they gain; 20: +5 %.
    //  FOUR points per lane -- 16-byte loads / stores, two packed FMAs per weight -- at width 10: SFNO forward 5.41 -> 5.51 ms,
    //  measured late in round 3 and not kept)
    const bool pairs = (CI <= env_int("TCFD_PW_PAIR_MAXC", 20)) && (a.P % 2 == 0) && (a.skip_mode != 2 || a.T % 2 == 0) &&
                       (((uintptr_t)a.x | (uintptr_t)a.out | (uintptr_t)a.pe | (uintptr_t)(a.skip_mode == 1 ? a.s : nullptr)) % 8 == 0);
    if constexpr (CI <= 20) {      // (the packed form is not even compiled for wider layers)
        if (pairs) {
            dim3 grid((unsigned)((a.P / 2 + 255) / 256), (unsigned)batch);
            if constexpr (HAS_L1 && CM > 0) {      // the 4 x width layers: activations known at compile time
                if (a.act1 == a.act2 && (a.act1 == 1 || a.act1 == 2) && env_int("TCFD_PW_ACT_T", 1)) {
                    if (a.act1 == 1) hipLaunchKernelGGL((k_pointwise<CI, CM, CO, HAS_L1, 2, 1>), grid, dim3(256), 0, st, a);
                    else hipLaunchKernelGGL((k_pointwise<CI, CM, CO, HAS_L1, 2, 2>), grid, dim3(256), 0, st, a);
                    HIP_TRY(hipGetLastError());
                    return 0;
                }
            }
            hipLaunchKernelGGL((k_pointwise<CI, CM, CO, HAS_L1, 2>), grid, dim3(256), 0, st, a);
            HIP_TRY(hipGetLastError());
            return 0;
        }
    }
    dim3 grid((unsigned)((a.P + 255) / 256), (unsigned)batch);
    hipLaunchKernelGGL((k_pointwise<CI, CM, CO, HAS_L1, 1>), grid, dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int pw_dispatch(PwArgs a, int batch, int ci, int cm, int co, hipStream_t st);
extern "C" int tcfd_fno_pointwise(const void* x, const void* skip, void* out, const void* w1, const void* b1,
                                  const void* w2t, const void* b2, const void* wst, const void* bs, int batch, int ci,
                                  int cm, int co, long P, int T, int skip_T, int act1, int act2, int skip_mode,
                                  long w2_bstride, long b2_bstride, const void* pe, void* stream) {
    return tcfd_fno_pointwise_pre(x, skip, out, nullptr, w1, b1, w2t, b2, wst, bs, batch, ci, cm, co, P, T, skip_T, act1, act2,
                                  skip_mode, w2_bstride, b2_bstride, pe, stream);
}
// The same, also storing the pre-activation z2 (batch, co, P) into `pre` (may be NULL): the training forward of a block whose
// output activation is not ReLU / identity -- its backward (tcfd_fno_pointwise_bwd_out) reads act2'(z2) from it.
extern "C" int tcfd_fno_pointwise_pre(const void* x, const void* skip, void* out, void* pre, const void* w1, const void* b1,
                                      const void* w2t, const void* b2, const void* wst, const void* bs, int batch, int ci,
                                      int cm, int co, long P, int T, int skip_T, int act1, int act2, int skip_mode,
                                      long w2_bstride, long b2_bstride, const void* pe, void* stream) {
    if (!x || !out || !w2t || batch <= 0 || P <= 0) return FAIL(TCFD_EINVAL, "fno_pointwise: bad argument");
    if (skip_mode && !skip) return FAIL(TCFD_EINVAL, "fno_pointwise: skip input missing");
    if (skip_mode == 2 && (T <= 0 || skip_T <= 0 || P % T != 0)) return FAIL(TCFD_EINVAL, "fno_pointwise: bad T");
    PwArgs a;
    a.x = (const float*)x; a.s = (const float*)skip; a.out = (float*)out;
    a.w1 = (const float*)w1; a.b1 = (const float*)b1; a.w2t = (const float*)w2t; a.b2 = (const float*)b2;
    a.wst = (const float*)wst; a.bs = (const float*)bs; a.pe = (const float*)pe;
    a.P = P; a.T = T; a.sT = skip_T; a.act1 = act1; a.act2 = act2; a.skip_mode = skip_mode;
    a.w2_bstride = w2_bstride; a.b2_bstride = b2_bstride;
    a.cm = cm;
    a.pre = (float*)pre;
    a.frame = nullptr; a.fT = 0;
    return pw_dispatch(a, batch, ci, cm, co, (hipStream_t)stream);
}

// The channel reduction in front of the output operator (fno/sfno.py:618 `reduction`, then :314-315): out (b, 1, P / T * (T + 1))
// = [last frame of `frame` (b, P / T, frame_T) | conv1x1(x) (b, ci, P) -> 1 channel] along t -- the reference's torch.cat of
// the last input frame and the T latent steps, written by the reduction itself.  T even (two points per lane share a row).
extern "C" int tcfd_fno_reduce_frames(const void* x, void* out, const void* w2t, const void* b2, const void* frame, int frame_T,
                                      int batch, int ci, long P, int T, void* stream) {
    if (!x || !out || !w2t || !frame || batch <= 0 || P <= 0 || T <= 0 || frame_T <= 0 || P % T != 0 || (T & 1))
        return FAIL(TCFD_EINVAL, "fno_reduce_frames: bad argument (T must be even)");
    PwArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const float*)x; a.out = (float*)out; a.w2t = (const float*)w2t; a.b2 = (const float*)b2;
    a.P = P; a.T = T; a.cm = ci;
    a.frame = (const float*)frame; a.fT = frame_T;
    return pw_dispatch(a, batch, ci, ci, 1, (hipStream_t)stream);
}

static int pw_dispatch(PwArgs a, int batch, int ci, int cm, int co, hipStream_t st) {
    const bool l1 = a.w1 != nullptr;
    if (l1) {     // wide layers: the block as matrix instructions (tcfd_fno_tiles.hip)
        int handled = 0;
        const int rc = tcfd_pwf_tiles_dispatch(a, batch, ci, cm, co, st, &handled);
        if (handled) return rc;
    }
#define PW_CASE(CI_, CM_, CO_)                                                             \
    if (ci == CI_ && cm == CM_ && co == CO_)                                                \
        return l1 ? launch_pw<CI_, CM_, CO_, true>(a, batch, st) : launch_pw<CI_, CM_, CO_, false>(a, batch, st);
    // the reference's default expansion (4 x width) with a compile-time trip count, then ANY hidden width for every
    // width up to 32 -- odd ones too: fno/sfno.py:607-614 and PointwiseFFN accept any, only SpaceTimePositionalEncoding
    // wants an even width > 3 -- and 36 / 40 / 48 / 64 (one point per lane, the channels still fit the register file):
    // the channel counts index register arrays and stay template parameters, the hidden width is a loop bound
#define PW_ANY(W_)                                                                         \
    if (ci == W_ && co == W_ && l1) return launch_pw<W_, 0, W_, true>(a, batch, st);
    if (l1) {
        PW_CASE(4, 16, 4) PW_CASE(8, 32, 8) PW_CASE(10, 40, 10) PW_CASE(16, 64, 16) PW_CASE(20, 80, 20) PW_CASE(32, 128, 32)
        PW_ANY(4) PW_ANY(6) PW_ANY(8) PW_ANY(10) PW_ANY(12) PW_ANY(14) PW_ANY(16) PW_ANY(18) PW_ANY(20) PW_ANY(24) PW_ANY(28)
        PW_ANY(32)
        PW_ANY(3) PW_ANY(5) PW_ANY(7) PW_ANY(9) PW_ANY(11) PW_ANY(13) PW_ANY(15) PW_ANY(17) PW_ANY(19) PW_ANY(21) PW_ANY(22)
        PW_ANY(23) PW_ANY(25) PW_ANY(26) PW_ANY(27) PW_ANY(29) PW_ANY(30) PW_ANY(31) PW_ANY(36) PW_ANY(40) PW_ANY(48) PW_ANY(64)
    } else {
        if (cm != ci) return FAIL(TCFD_EINVAL, "fno_pointwise: single layer needs cm == ci");
        PW_CASE(4, 4, 4) PW_CASE(4, 4, 1) PW_CASE(8, 8, 8) PW_CASE(8, 8, 1) PW_CASE(10, 10, 10) PW_CASE(10, 10, 1)
        PW_CASE(16, 16, 16) PW_CASE(16, 16, 1) PW_CASE(20, 20, 20) PW_CASE(20, 20, 1) PW_CASE(32, 32, 32) PW_CASE(32, 32, 1)
        PW_CASE(6, 6, 6) PW_CASE(6, 6, 1) PW_CASE(12, 12, 12) PW_CASE(12, 12, 1) PW_CASE(14, 14, 14) PW_CASE(14, 14, 1)
        PW_CASE(18, 18, 18) PW_CASE(18, 18, 1) PW_CASE(24, 24, 24) PW_CASE(24, 24, 1) PW_CASE(28, 28, 28) PW_CASE(28, 28, 1)
#define PW_ONE(W_) PW_CASE(W_, W_, W_) PW_CASE(W_, W_, 1)
        PW_ONE(3) PW_ONE(5) PW_ONE(7) PW_ONE(9) PW_ONE(11) PW_ONE(13) PW_ONE(15) PW_ONE(17) PW_ONE(19) PW_ONE(21) PW_ONE(22)
        PW_ONE(23) PW_ONE(25) PW_ONE(26) PW_ONE(27) PW_ONE(29) PW_ONE(30) PW_ONE(31) PW_ONE(36) PW_ONE(40) PW_ONE(48) PW_ONE(64)
#undef PW_ONE
    }
#undef PW_CASE
#undef PW_ANY
    return FAIL(TCFD_EINVAL, "fno_pointwise: channels (%d -> %d -> %d) not instantiated", ci, cm, co);
}


// ------------------------------------------------------------------ the same block in float64 (FNOBase.double())
// One point per lane, channels in registers, weights lane-uniform through the scalar unit; fp64 VALU has no packed form
// and half the rate, so the block is compute bound here (width 10: 1800 DFMA per point) -- it exists so that an SFNO
// converted with .double() (fno/base.py:342-349) stays on hand-written kernels end to end, not for speed.
struct PwArgsD {
    const double *x, *s, *w1, *b1, *w2t, *b2, *wst, *bs;
    double* out;
    long P;
    long w2_bstride, b2_bstride;   // per-batch-element offsets of w2t / b2 (0: shared): a folded LayerNorm rides in the single-layer form
    int T, sT, act1, act2, skip_mode, cm;
};
__device__ __forceinline__ double pw_act(double v, int act) {
    switch (act) {
        case 1: return v > 0.0 ? v : 0.0;
        case 2: return 0.5 * v * (1.0 + erf(v * 0.70710678118654752440));
        case 3: return v / (1.0 + exp(-v));
        case 4: return tanh(v);
        default: return v;
    }
}
template <int CI, int CO, bool HAS_L1>
__global__ __launch_bounds__(256) void k_pointwise_f64(PwArgsD a) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    const long b = blockIdx.y;
    if (p >= a.P) return;
    double x[CI], o[CO];
    const double* xb = a.x + (size_t)b * CI * a.P + p;
#pragma unroll
    for (int i = 0; i < CI; ++i) x[i] = xb[(size_t)i * a.P];
    const double* w2t_b = a.w2t + (size_t)b * a.w2_bstride;
    const double* b2_b = a.b2 ? a.b2 + (size_t)b * a.b2_bstride : nullptr;
#pragma unroll
    for (int c = 0; c < CO; ++c) o[c] = b2_b ? b2_b[c] : 0.0;
    if constexpr (HAS_L1) {
        for (int m = 0; m < a.cm; ++m) {
            double z = a.b1 ? a.b1[m] : 0.0;
            const double* w1 = a.w1 + (size_t)m * CI;
#pragma unroll
            for (int i = 0; i < CI; ++i) z = fma(w1[i], x[i], z);
            const double h = pw_act(z, a.act1);
            const double* w2 = w2t_b + (size_t)m * CO;
#pragma unroll
            for (int c = 0; c < CO; ++c) o[c] = fma(w2[c], h, o[c]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < CI; ++i) {
            const double* w2 = w2t_b + (size_t)i * CO;
#pragma unroll
            for (int c = 0; c < CO; ++c) o[c] = fma(w2[c], x[i], o[c]);
        }
    }
    if (a.skip_mode == 1) {
        const double* sb = a.s + (size_t)b * CI * a.P + p;
#pragma unroll
        for (int i = 0; i < CI; ++i) {
            const double sv = sb[(size_t)i * a.P];
            const double* ws = a.wst + (size_t)i * CO;
#pragma unroll
            for (int c = 0; c < CO; ++c) o[c] = fma(ws[c], sv, o[c]);
        }
        if (a.bs) {
#pragma unroll
            for (int c = 0; c < CO; ++c) o[c] += a.bs[c];
        }
    } else if (a.skip_mode == 2) {
        const long sP = (a.P / a.T) * a.sT;
        const double* sb = a.s + (size_t)b * CO * sP + (p / a.T) * a.sT + (a.sT - 1);
#pragma unroll
        for (int c = 0; c < CO; ++c) o[c] += sb[(size_t)c * sP];
    }
    double* ob = a.out + (size_t)b * CO * a.P + p;
#pragma unroll
    for (int c = 0; c < CO; ++c) ob[(size_t)c * a.P] = pw_act(o[c], a.act2);
}

// float64 form of tcfd_fno_pointwise (no positional-encoding input, shared weights).  Instantiated for the widths
// 4, 6, 8, 10, 12, 16, 20, 24, 32 (any hidden width) and their single-convolution forms W -> W, W -> 1.
extern "C" int tcfd_fno_pointwise_f64(const void* x, const void* skip, void* out, const void* w1, const void* b1,
                                      const void* w2t, const void* b2, const void* wst, const void* bs, int batch,
                                      int ci, int cm, int co, long P, int T, int skip_T, int act1, int act2,
                                      int skip_mode, long w2_bstride, long b2_bstride, void* stream) {
    if (!x || !out || !w2t || batch <= 0 || P <= 0) return FAIL(TCFD_EINVAL, "fno_pointwise_f64: bad argument");
    if (skip_mode && !skip) return FAIL(TCFD_EINVAL, "fno_pointwise_f64: skip input missing");
    if (skip_mode == 1 && !wst) return FAIL(TCFD_EINVAL, "fno_pointwise_f64: skip weights missing");
    if (skip_mode == 2 && (T <= 0 || skip_T <= 0 || P % T != 0)) return FAIL(TCFD_EINVAL, "fno_pointwise_f64: bad T");
    PwArgsD a;
    a.x = (const double*)x; a.s = (const double*)skip; a.out = (double*)out; a.w1 = (const double*)w1;
    a.b1 = (const double*)b1; a.w2t = (const double*)w2t; a.b2 = (const double*)b2; a.wst = (const double*)wst;
    a.bs = (const double*)bs; a.P = P; a.T = T; a.sT = skip_T; a.act1 = act1; a.act2 = act2; a.skip_mode = skip_mode; a.cm = cm;
    a.w2_bstride = w2_bstride; a.b2_bstride = b2_bstride;
    const bool l1 = w1 != nullptr;
    if (!l1 && cm != ci) return FAIL(TCFD_EINVAL, "fno_pointwise_f64: single layer needs cm == ci");
    const dim3 grid((unsigned)((P + 255) / 256), (unsigned)batch);
    hipStream_t st = (hipStream_t)stream;
#define PWD(CI_, CO_)                                                                                      \
    if (ci == CI_ && co == CO_) {                                                                           \
        if (l1) hipLaunchKernelGGL((k_pointwise_f64<CI_, CO_, true>), grid, dim3(256), 0, st, a);           \
        else hipLaunchKernelGGL((k_pointwise_f64<CI_, CO_, false>), grid, dim3(256), 0, st, a);             \
        HIP_TRY(hipGetLastError());                                                                         \
        return 0;                                                                                           \
    }
#define PWD_W(W_) PWD(W_, W_) PWD(W_, 1)
    PWD_W(4) PWD_W(6) PWD_W(8) PWD_W(10) PWD_W(12) PWD_W(16) PWD_W(20) PWD_W(24) PWD_W(32)
#undef PWD_W
#undef PWD
    return FAIL(TCFD_EINVAL, "fno_pointwise_f64: channels (%d -> %d -> %d) not instantiated", ci, cm, co);
}


// ------------------------------------------------------------------ backward of the fused pointwise block
// Given dL/dout, ONE pass recomputes the block per point (hidden vector in registers, as the forward does) and
// produces dL/dx, dL/dskip and the weight / bias gradients.  The weight gradients are sums over ALL points of
// outer products (dW2 = sum g2 (x) h, dW1 = sum g1 (x) x, dWs = sum g2 (x) s): a GEMM whose K axis is the points
// of a wave.  Each wave stages its 64 points channel-major in its own LDS slice ([channel][point], row pitch 66
// floats: lane-consecutive conflict-free stores, conflict-free operand fetches) and accumulates on
// v_mfma_f32_16x16x4_f32 (A lane l -> [row l&15][k l>>4], B -> [k l>>4][col l&15], D -> [row 4(l>>4)+r][col l&15]);
// a constant-1 channel appended to h / x makes the bias gradients fall out of the same products.  The accumulators
// (28 registers at width 10) live across the wave's whole grid-stride loop; every wave writes its partial sums
// once, the caller adds the partials (deterministic, no atomics).
// (PwBwdArgs, PwBwdGeom, pw_act_pair: tcfd_fno_pw.hpp, shared with tcfd_fno_tiles.hip)

__device__ __forceinline__ float pw_dact(float z, int act) {   // d act / dz
    switch (act) {
        case 1: return z > 0.f ? 1.f : 0.f;
        case 2: {
            const float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752f));
            return cdf + z * 0.3989422804014327f * __expf(-0.5f * z * z);
        }
        case 3: { const float sg = 1.f / (1.f + __expf(-z)); return sg * (1.f + z * (1.f - sg)); }
        case 4: { const float t = tanhf(z); return 1.f - t * t; }
        default: return 1.f;
    }
}


// weights of the block, staged once per workgroup in LDS (rows padded to 4 floats): with ~1.5 waves per SIMD the
// scalar-cache latency of per-row s_loads is exposed (measured 8x slower); uniform-address ds_reads pipeline.
template <int CI, int CM, int CO, bool HAS_L1>
struct PwBwdW {
    static constexpr int RI = (CI + 3) & ~3, RO = (CO + 3) & ~3;
    static constexpr int W1 = 0;                                   // (CM, RI)
    static constexpr int B1 = W1 + (HAS_L1 ? CM * RI : 0);         // (CM)
    static constexpr int W2 = B1 + (HAS_L1 ? ((CM + 3) & ~3) : 0); // (CH, RO)   CH = CM or CI
    static constexpr int B2 = W2 + (HAS_L1 ? CM : CI) * RO;        // (RO)  b2 + bs
    static constexpr int WS = B2 + RO;                             // (CI, RO)
    static constexpr int TOTAL = WS + CI * RO;
};

template <int CI, int CM, int CO, bool HAS_L1>
__global__ __launch_bounds__(128) void k_pointwise_bwd(PwBwdArgs a) {
    using Gm = PwBwdGeom<CI, CM, CO, HAS_L1>;
    using Wm = PwBwdW<CI, CM, CO, HAS_L1>;
    constexpr int PITCH = Gm::PITCH, CH = HAS_L1 ? CM : CI;   // channels of the second operand's first block
    constexpr int TO = Gm::COP / 16, TB = Gm::CB / 16, TI = Gm::CIP / 16, TM = Gm::CM1 / 16;
    constexpr int RI = Wm::RI, RO = Wm::RO;
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;   // wave uniform, and hipcc knows it
    float* Wl = reinterpret_cast<float*>(smem_raw);
    float* L0 = Wl + ((Wm::TOTAL + 3) & ~3) + (size_t)wave * Gm::ROWS * PITCH;   // g2, later [x, 1]
    float* L1 = L0 + Gm::R0 * PITCH;                                              // [h, 1, s], later g1 over h
    for (int i = threadIdx.x; i < Wm::TOTAL; i += blockDim.x) Wl[i] = 0.f;
    __syncthreads();
    if constexpr (HAS_L1) {
        for (int i = threadIdx.x; i < CM * CI; i += blockDim.x) Wl[Wm::W1 + (i / CI) * RI + i % CI] = a.w1[i];
        if (a.b1) for (int i = threadIdx.x; i < CM; i += blockDim.x) Wl[Wm::B1 + i] = a.b1[i];
    }
    for (int i = threadIdx.x; i < CH * CO; i += blockDim.x) Wl[Wm::W2 + (i / CO) * RO + i % CO] = a.w2t[i];
    for (int i = threadIdx.x; i < CO; i += blockDim.x)
        Wl[Wm::B2 + i] = (a.b2 ? a.b2[i] : 0.f) + ((a.skip_mode == 1 && a.bs) ? a.bs[i] : 0.f);
    if (a.skip_mode == 1)
        for (int i = threadIdx.x; i < CI * CO; i += blockDim.x) Wl[Wm::WS + (i / CO) * RO + i % CO] = a.wst[i];
    __syncthreads();
    const int kq = lane >> 4, kc = lane & 15;
    f4 accA[TO * TB];
    f4 accB[HAS_L1 ? TM * TI : 1];
#pragma unroll
    for (auto& v : accA) v = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (auto& v : accB) v = f4{0.f, 0.f, 0.f, 0.f};
    const long gw = (long)blockIdx.x * Gm::WAVES + wave, nw = (long)gridDim.x * Gm::WAVES;
    // default: waves stride over all (sample, 64-point chunk) pairs.  per_sample: wave gw owns sample gw % batch and
    // strides over that sample's chunks with the nw / batch waves that share it (nw is a multiple of batch).
    const long c_first = a.per_sample ? (gw % a.batch) * a.chunks_per_batch + gw / a.batch : gw;
    const long c_stride = a.per_sample ? nw / a.batch : nw;
    const long c_end = a.per_sample ? (gw % a.batch + 1) * a.chunks_per_batch : a.total_chunks;
    for (long chunk = c_first; chunk < c_end; chunk += c_stride) {
        const long b = chunk / a.chunks_per_batch;
        const long p = (chunk - b * a.chunks_per_batch) * 64 + lane;
        const bool live = p < a.P;
        const long pc = live ? p : a.P - 1;
        float x[CI], g2[CO], z2[CO], dx[CI];
        const float* db = a.dout + (size_t)b * CO * a.P + pc;
        if (a.pe) {
            const float v1 = a.x[(size_t)b * a.P + pc];
#pragma unroll
            for (int i = 0; i < CI; ++i) x[i] = v1 + a.pe[(size_t)i * a.P + pc];
        } else {
            const float* xb = a.x + (size_t)b * CI * a.P + pc;
#pragma unroll
            for (int i = 0; i < CI; ++i) x[i] = xb[(size_t)i * a.P];
        }
#pragma unroll
        for (int c = 0; c < CO; ++c) g2[c] = live ? db[(size_t)c * a.P] : 0.f;
#pragma unroll
        for (int c = 0; c < CO; ++c) z2[c] = Wl[Wm::B2 + c];
#pragma unroll
        for (int i = 0; i < CI; ++i) dx[i] = 0.f;
        if (a.skip_mode == 2) {
            const long xy = pc / a.T;
            const long sP = (a.P / a.T) * a.sT;
            const float* sb = a.s + (size_t)b * CO * sP + xy * a.sT + (a.sT - 1);
#pragma unroll
            for (int c = 0; c < CO; ++c) z2[c] += sb[(size_t)c * sP];
        }
        // second operand rows [CH] = 1, [CH+1, CH+1+CI) = skip input (zero without a skip convolution)
        L1[CH * PITCH + lane] = live ? 1.f : 0.f;
        if (a.skip_mode == 1) {
            const float* sb = a.s + (size_t)b * CI * a.P + pc;
#pragma unroll
            for (int i = 0; i < CI; ++i) {
                const float sv = live ? sb[(size_t)i * a.P] : 0.f;
                L1[(CH + 1 + i) * PITCH + lane] = sv;
                const float* ws = Wl + Wm::WS + i * RO;
#pragma unroll
                for (int c = 0; c < CO; ++c) z2[c] = fmaf(ws[c], sv, z2[c]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < CI; ++i) L1[(CH + 1 + i) * PITCH + lane] = 0.f;
        }
        if constexpr (HAS_L1) {
#pragma unroll 4
            for (int m = 0; m < CM; ++m) {   // hidden vector: kept in this lane's LDS column, not in registers
                float z = Wl[Wm::B1 + m];
                const float* w1 = Wl + Wm::W1 + m * RI;
#pragma unroll
                for (int i = 0; i < CI; ++i) z = fmaf(w1[i], x[i], z);
                const float h = live ? pw_act(z, a.act1) : 0.f;
                L1[m * PITCH + lane] = h;
                const float* w2 = Wl + Wm::W2 + m * RO;
#pragma unroll
                for (int c = 0; c < CO; ++c) z2[c] = fmaf(w2[c], h, z2[c]);
            }
        } else {
#pragma unroll
            for (int m = 0; m < CI; ++m) {
                L1[m * PITCH + lane] = live ? x[m] : 0.f;
                const float* w2 = Wl + Wm::W2 + m * RO;
#pragma unroll
                for (int c = 0; c < CO; ++c) z2[c] = fmaf(w2[c], x[m], z2[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            g2[c] *= pw_dact(z2[c], a.act2);
            L0[c * PITCH + lane] = g2[c];
        }
        group_sync<false>();
#pragma unroll 2
        for (int q = 0; q < 16; ++q) {   // [dW2 | db2 | dWs][o][.] += g2[o] [h, 1, s][.] over the 4 points of the k-step
            float av[TO];
#pragma unroll
            for (int to = 0; to < TO; ++to) av[to] = L0[(16 * to + kc) * PITCH + 4 * q + kq];
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                const float bv = L1[(16 * tb + kc) * PITCH + 4 * q + kq];
#pragma unroll
                for (int to = 0; to < TO; ++to)
                    accA[to * TB + tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[to], bv, accA[to * TB + tb], 0, 0, 0);
            }
        }
        group_sync<false>();
        if constexpr (HAS_L1) {
            // g1 = (W2^T g2) act1'(z1) written over h;  dx = W1^T g1
            const bool from_h = a.act1 == 0 || a.act1 == 1 || a.act1 == 4;   // act1' is a function of h itself
#pragma unroll 4
            for (int m = 0; m < CM; ++m) {
                const float* w1 = Wl + Wm::W1 + m * RI;
                const float h = L1[m * PITCH + lane];
                float d1;
                if (from_h) {
                    d1 = a.act1 == 1 ? (h > 0.f ? 1.f : 0.f) : (a.act1 == 4 ? 1.f - h * h : 1.f);
                } else {
                    float z = Wl[Wm::B1 + m];
#pragma unroll
                    for (int i = 0; i < CI; ++i) z = fmaf(w1[i], x[i], z);
                    d1 = pw_dact(z, a.act1);
                }
                const float* w2 = Wl + Wm::W2 + m * RO;
                float dh = 0.f;
#pragma unroll
                for (int c = 0; c < CO; ++c) dh = fmaf(w2[c], g2[c], dh);
                const float g1 = dh * d1;      // g2 = 0 on dead lanes, so g1 is too
                L1[m * PITCH + lane] = g1;
#pragma unroll
                for (int i = 0; i < CI; ++i) dx[i] = fmaf(w1[i], g1, dx[i]);
            }
#pragma unroll
            for (int i = 0; i < CI; ++i) L0[i * PITCH + lane] = live ? x[i] : 0.f;
            L0[CI * PITCH + lane] = live ? 1.f : 0.f;
            group_sync<false>();
#pragma unroll 2
            for (int q = 0; q < 16; ++q) {   // [dW1 | db1][m][.] += g1[m] [x, 1][.]
                float bv[TI];
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) bv[ti] = L0[(16 * ti + kc) * PITCH + 4 * q + kq];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const float av = L1[(16 * tm + kc) * PITCH + 4 * q + kq];
#pragma unroll
                    for (int ti = 0; ti < TI; ++ti)
                        accB[tm * TI + ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[ti], accB[tm * TI + ti], 0, 0, 0);
                }
            }
            group_sync<false>();
        } else {
#pragma unroll
            for (int m = 0; m < CI; ++m) {
                const float* w2 = Wl + Wm::W2 + m * RO;
#pragma unroll
                for (int c = 0; c < CO; ++c) dx[m] = fmaf(w2[c], g2[c], dx[m]);
            }
        }
        if (live && a.dx) {
            float* dxb = a.dx + (size_t)b * CI * a.P + p;
#pragma unroll
            for (int i = 0; i < CI; ++i) dxb[(size_t)i * a.P] = dx[i];
        }
        if (live) {
            if (a.skip_mode == 1 && a.ds) {
                float* dsb = a.ds + (size_t)b * CI * a.P + p;
#pragma unroll
                for (int i = 0; i < CI; ++i) {
                    const float* ws = Wl + Wm::WS + i * RO;
                    float v = 0.f;
#pragma unroll
                    for (int c = 0; c < CO; ++c) v = fmaf(ws[c], g2[c], v);
                    dsb[(size_t)i * a.P] = v;
                }
            } else if (a.skip_mode == 2 && a.ds) {   // dL/dz2: the caller sums it over t into the skip's last slice
                float* dsb = a.ds + (size_t)b * CO * a.P + p;
#pragma unroll
                for (int c = 0; c < CO; ++c) dsb[(size_t)c * a.P] = g2[c];
            }
        }
    }
    // this wave's partial sums as row-major padded tiles:  A (COP x CB) | B (CM1 x CIP)
    float* out = a.partials + ((size_t)blockIdx.x * Gm::WAVES + wave) * Gm::TOTAL;
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * to + 4 * kq + r) * Gm::CB + 16 * tb + kc] = accA[to * TB + tb][r];
    if constexpr (HAS_L1) {
        float* o1 = out + Gm::N_A;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) o1[(16 * tm + 4 * kq + r) * Gm::CIP + 16 * ti + kc] = accB[tm * TI + ti][r];
    }
}

template <int CI, int CM, int CO, bool HAS_L1>
static int launch_pw_bwd(PwBwdArgs a, int batch, int max_waves, int* dims, hipStream_t st) {
    FnoProfScope prof(HAS_L1 ? FNO_K_POINTWISE_BWD : FNO_K_POINTWISE_BWD_1, st);
    using Gm = PwBwdGeom<CI, CM, CO, HAS_L1>;
    dims[0] = Gm::COP; dims[1] = Gm::CB; dims[2] = Gm::CM1; dims[3] = Gm::CIP; dims[4] = Gm::TOTAL; dims[5] = 0;
    if (!a.x) return 0;   // layout query
    a.chunks_per_batch = (a.P + 63) / 64;
    a.total_chunks = a.chunks_per_batch * batch;
    using Wm = PwBwdW<CI, CM, CO, HAS_L1>;
    const size_t lds = ((size_t)((Wm::TOTAL + 3) & ~3) + (size_t)Gm::WAVES * Gm::ROWS * Gm::PITCH) * sizeof(float);
    auto kern = k_pointwise_bwd<CI, CM, CO, HAS_L1>;
    int rc = set_lds_attr(kern, lds);
    if (rc) return rc;
    // persistent grid: exactly the resident workgroups (a second, partial round would double the run time)
    int per_cu = 0, dev = 0, cus = 256;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 64 * Gm::WAVES, lds));
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    long resident = (long)std::max(per_cu, 1) * cus;
    int blocks = (int)std::min<long>({(a.total_chunks + Gm::WAVES - 1) / Gm::WAVES, (long)(max_waves / Gm::WAVES), resident});
    if (blocks < 1) blocks = 1;
    a.batch = batch;
    if (a.per_sample) {   // the wave count must be a multiple of the batch size
        long waves = (long)blocks * Gm::WAVES / batch * batch;
        if (waves < batch) waves = batch;
        while (waves % Gm::WAVES) waves += batch;
        if (waves > max_waves) return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: %ld waves needed for per-sample partials, %d rows given", waves, max_waves);
        blocks = (int)(waves / Gm::WAVES);
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * Gm::WAVES), lds, st, a);
    HIP_TRY(hipGetLastError());
    dims[5] = blocks * Gm::WAVES;
    return 0;
}

// Single layer with ONE output channel (the channel reduction in front of the output operator, fno/sfno.py:313): dx[c] = w[c] g,
// dW[c] = sum g x[c], db = sum g with g = dout act'(z) -- a streaming job (read x and dout, write dx: 1.76 GB at config 5) that the
// general one-wave kernel above ran at 3.1 TB/s through its LDS staging.  Here a lane owns four consecutive points as 16-byte
// lanes, the 2 CI + 1 accumulators of a wave meet in its row of partial sums (the layout of k_pointwise_bwd: A[0][c], A[0][CI]).
template <int CI>
__global__ __launch_bounds__(256) void k_pwb_reduce1(PwBwdArgs a, long chunks_per_batch, long total_chunks) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    using Gm = PwBwdGeom<CI, CI, 1, false>;
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    float w[CI];
#pragma unroll
    for (int c = 0; c < CI; ++c) w[c] = a.w2t[c];
    const float bias = a.b2 ? a.b2[0] : 0.f;
    float dw[CI], db = 0.f;
#pragma unroll
    for (int c = 0; c < CI; ++c) dw[c] = 0.f;
    for (long ch = wid; ch < total_chunks; ch += nw) {
        const long b = ch / chunks_per_batch;
        const long p = (ch - b * chunks_per_batch) * 256 + 4 * lane;
        const bool live = p < a.P;                              // P % 4 == 0: a lane's four points are all inside or all outside
        const long pc = live ? p : 0;
        f4 xv[CI];
#pragma unroll
        for (int c = 0; c < CI; ++c) xv[c] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.x + ((size_t)b * CI + c) * a.P + pc));
        f4 g = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.dout + (size_t)b * a.P + pc));
        if (a.act2 != 0) {                                      // the gate of the output activation from the recomputed pre-activation
            f4 z = {bias, bias, bias, bias};
#pragma unroll
            for (int c = 0; c < CI; ++c) z += w[c] * xv[c];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float h, d;
                switch (a.act2) {
                    case 1: pw_act_pair<1>(z[r], h, d); break;
                    case 2: pw_act_pair<2>(z[r], h, d); break;
                    case 3: pw_act_pair<3>(z[r], h, d); break;
                    default: pw_act_pair<4>(z[r], h, d); break;
                }
                g[r] *= d;
            }
        }
        if (!live) g = f4{0.f, 0.f, 0.f, 0.f};
        db += (g[0] + g[1]) + (g[2] + g[3]);
#pragma unroll
        for (int c = 0; c < CI; ++c) {
            const f4 t = g * xv[c];
            dw[c] += (t[0] + t[1]) + (t[2] + t[3]);
            if (live && a.dx) __builtin_nontemporal_store(w[c] * g, reinterpret_cast<f4*>(a.dx + ((size_t)b * CI + c) * a.P + pc));
        }
    }
    // the wave's sums (butterflies: deterministic), its row of partials written by lane 0
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        db += __shfl_xor(db, off, 64);
#pragma unroll
        for (int c = 0; c < CI; ++c) dw[c] += __shfl_xor(dw[c], off, 64);
    }
    if (lane == 0) {
        float* out = a.partials + (size_t)wid * Gm::TOTAL;
#pragma unroll
        for (int c = 0; c < CI; ++c) out[c] = dw[c];
        out[CI] = db;
    }
}
template <int CI>
static int launch_pwb_reduce1(PwBwdArgs a, int batch, int max_waves, int* dims, hipStream_t st) {
    FnoProfScope prof(FNO_K_POINTWISE_BWD_1, st);
    using Gm = PwBwdGeom<CI, CI, 1, false>;
    dims[0] = Gm::COP; dims[1] = Gm::CB; dims[2] = Gm::CM1; dims[3] = Gm::CIP; dims[4] = Gm::TOTAL;
    const long cpb = (a.P + 255) / 256, total = cpb * batch;
    int blocks = (int)std::min<long>({(total + 3) / 4, (long)(max_waves / 4), 2048L});     // <= 8 waves per SIMD's worth of rows
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_pwb_reduce1<CI>, dim3(blocks), dim3(256), 0, st, a, cpb, total);
    HIP_TRY(hipGetLastError());
    dims[5] = blocks * 4;
    return 0;
}

// Backward of tcfd_fno_pointwise (shared weights; skip_mode 2 writes dL/dz2 (b, co, P) into dskip).  `partials` holds `max_waves` rows of
// dims[4] floats; on return dims = {COP, CB, CM1, CIP, floats per row, rows written}: row-major padded tiles
//   A (COP x CB):  A[o][0:ch] = dW2[o][.] (ch = cm, single layer: ci),  A[o][ch] = db2[o] (= dbs),  A[o][ch+1 : ch+1+ci] = dWs[o][.]
//   B (CM1 x CIP): B[m][0:ci] = dW1[m][.],  B[m][ci] = db1[m]            (two-layer form only)
// The caller sums the rows.  Passing x == NULL only fills dims (layout query).
static int pointwise_bwd_impl(const void* pe, const void* x, const void* skip, const void* dout, void* dx, void* dskip,
                              const void* w1, const void* b1, const void* w2t, const void* b2, const void* wst,
                              const void* bs, void* partials, int max_waves, int* dims, int batch, int ci,
                              int cm, int co, long P, int T, int skip_T, int act1, int act2, int skip_mode,
                              int per_sample, void* stream, const void* out = nullptr);
extern "C" int tcfd_fno_pointwise_bwd(const void* x, const void* skip, const void* dout, void* dx, void* dskip,
                                      const void* w1, const void* b1, const void* w2t, const void* b2, const void* wst,
                                      const void* bs, void* partials, int max_waves, int* dims, int batch, int ci,
                                      int cm, int co, long P, int T, int skip_T, int act1, int act2, int skip_mode,
                                      int per_sample, void* stream) {
    return pointwise_bwd_impl(nullptr, x, skip, dout, dx, dskip, w1, b1, w2t, b2, wst, bs, partials, max_waves, dims, batch, ci, cm,
                              co, P, T, skip_T, act1, act2, skip_mode, per_sample, stream);
}
// The same with the block's forward output `out` (batch, co, P) handed over (may be NULL = the call above): kernels that can
// read the mask of a ReLU output activation from it do so instead of recomputing the pre-activation.
extern "C" int tcfd_fno_pointwise_bwd_out(const void* x, const void* skip, const void* dout, const void* out, void* dx, void* dskip,
                                          const void* w1, const void* b1, const void* w2t, const void* b2, const void* wst,
                                          const void* bs, void* partials, int max_waves, int* dims, int batch, int ci,
                                          int cm, int co, long P, int T, int skip_T, int act1, int act2, int skip_mode,
                                          int per_sample, void* stream) {
    return pointwise_bwd_impl(nullptr, x, skip, dout, dx, dskip, w1, b1, w2t, b2, wst, bs, partials, max_waves, dims, batch, ci, cm,
                              co, P, T, skip_T, act1, act2, skip_mode, per_sample, stream, out);
}
// The single-layer form whose input is x1 (batch, 1, P) + pe (ci, P) (the `pe` mode of tcfd_fno_pointwise): weight-gradient
// partial sums (and dx, if asked for) without the (batch, ci, P) input ever being materialised.
extern "C" int tcfd_fno_pointwise_bwd_pe(const void* x1, const void* pe, const void* dout, void* dx, const void* w2t,
                                         const void* b2, void* partials, int max_waves, int* dims, int batch, int ci, int co,
                                         long P, int per_sample, void* stream) {
    if (x1 && !pe) return FAIL(TCFD_EINVAL, "fno_pointwise_bwd_pe: null table");
    return pointwise_bwd_impl(pe, x1, nullptr, dout, dx, nullptr, nullptr, nullptr, w2t, b2, nullptr, nullptr, partials, max_waves,
                              dims, batch, ci, ci, co, P, 0, 0, 0, 0, 0, per_sample, stream);
}
// The two-layer block runs the tiled all-MFMA kernel of tcfd_fno_tiles.hip (every even width 4 ... 16, 20, 24, 32 with cm = 4 ci,
// P % 4 == 0): ReLU from the saved output, every other activation from the saved pre-activation.  TCFD_PW_BWD_TILES=0 keeps it
// out -- the LDS-staged one-wave kernel below (widths 4 / 8 / 10, recomputes everything) then serves as the cross-check.
static bool pwb_tiles_selected() { return env_int("TCFD_PW_BWD_TILES", 1) != 0; }
// What the backward of the two-layer block (ci -> cm -> co at P points per sample) wants the forward to keep, to be handed to
// tcfd_fno_pointwise_bwd_out as `out`: 0 nothing, 1 the block's output, 2 its pre-activation (tcfd_fno_pointwise_pre).
extern "C" int tcfd_fno_pointwise_bwd_saved(int ci, int cm, int co, long P, int act1, int act2) {
    (void)act1;
    if (act2 == 0 || !pwb_tiles_selected()) return 0;
    PwBwdArgs q;
    memset(&q, 0, sizeof(q));
    q.P = P;
    int dims[6], handled = 0;
    if (tcfd_pwb_tiles_dispatch(q, 1, ci, cm, co, 0, dims, nullptr, &handled) == 0 && handled) return act2 == 1 ? 1 : 2;
    return 0;
}
static int pointwise_bwd_impl(const void* pe, const void* x, const void* skip, const void* dout, void* dx, void* dskip,
                              const void* w1, const void* b1, const void* w2t, const void* b2, const void* wst,
                              const void* bs, void* partials, int max_waves, int* dims, int batch, int ci,
                              int cm, int co, long P, int T, int skip_T, int act1, int act2, int skip_mode,
                              int per_sample, void* stream, const void* out) {
    if (!dims) return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: null dims");
    if (x && (!dout || !w2t || !partials || batch <= 0 || P <= 0 || max_waves < 2))
        return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: bad argument");
    if (skip_mode < 0 || skip_mode > 3) return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: skip_mode %d not supported", skip_mode);
    const bool tsum = skip_mode == 3;       // mode 2 whose skip gradient leaves the kernel summed over t: dskip is (b, co, P / T)
    if (tsum) skip_mode = 2;
    if (x && skip_mode == 1 && (!skip || !wst)) return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: skip input missing");
    if (x && skip_mode == 2 && (!skip || T <= 0 || skip_T <= 0 || P % T != 0)) return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: bad T");
    PwBwdArgs a;
    a.pe = (const float*)pe;
    a.x = (const float*)x; a.s = (const float*)skip; a.dout = (const float*)dout; a.dx = (float*)dx; a.ds = (float*)dskip;
    a.out = (const float*)out;
    a.w1 = (const float*)w1; a.b1 = (const float*)b1; a.w2t = (const float*)w2t; a.b2 = (const float*)b2;
    a.wst = (const float*)wst; a.bs = (const float*)bs; a.partials = (float*)partials;
    a.P = P; a.act1 = act1; a.act2 = act2; a.skip_mode = skip_mode; a.T = T; a.sT = skip_T;
    a.per_sample = per_sample; a.batch = batch;
    a.ds_tsum = 0; a.tsum_groups = 0;
    a.chunks_per_batch = a.total_chunks = 0;
    hipStream_t st = (hipStream_t)stream;
    const bool l1 = cm != ci || w1 != nullptr;
#define PWB_CASE(CI_, CM_, CO_, L1_) \
    if (ci == CI_ && cm == CM_ && co == CO_ && l1 == L1_) return launch_pw_bwd<CI_, CM_, CO_, L1_>(a, batch, max_waves, dims, st);
    if (tsum) {   // whole rows of T steps inside `tsum_groups` consecutive groups of 16 points: T | 16 (one group) or T | 80 (five)
        a.ds_tsum = 1;
        a.tsum_groups = (T > 0 && 16 % T == 0) ? 1 : ((T > 0 && 80 % T == 0) ? 5 : 0);
        if (!l1 || !pwb_tiles_selected() || !a.tsum_groups || P % (16 * a.tsum_groups) != 0)
            return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: skip_mode 3 (t-summed skip gradient) not instantiated for this shape");
    }
    if (l1 && pwb_tiles_selected()) {
        int handled = 0;
        const int rc = tcfd_pwb_tiles_dispatch(a, batch, ci, cm, co, max_waves, dims, st, &handled);
        if (handled) return rc;
    }
    if (tsum) return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: skip_mode 3 (t-summed skip gradient) not instantiated for this shape");
    // one output channel, no skip: the streaming kernel (TCFD_PWB_REDUCE1=0: the general kernel below, its cross-check)
    if (x && !l1 && co == 1 && skip_mode == 0 && !pe && !per_sample && P % 4 == 0 && max_waves >= 4 && env_int("TCFD_PWB_REDUCE1", 1)) {
#define PWB_R1(CI_) if (ci == CI_) return launch_pwb_reduce1<CI_>(a, batch, max_waves, dims, st);
        PWB_R1(4) PWB_R1(6) PWB_R1(8) PWB_R1(10) PWB_R1(12) PWB_R1(14) PWB_R1(16) PWB_R1(20) PWB_R1(24) PWB_R1(32)
#undef PWB_R1
    }
    // the LDS-staged one-wave kernel: the single-layer forms, P % 4 != 0, and the cross-check of the tiled kernel
    PWB_CASE(4, 16, 4, true) PWB_CASE(8, 32, 8, true) PWB_CASE(10, 40, 10, true)
    PWB_CASE(4, 4, 4, false) PWB_CASE(4, 4, 1, false) PWB_CASE(8, 8, 8, false) PWB_CASE(8, 8, 1, false)
    PWB_CASE(10, 10, 10, false) PWB_CASE(10, 10, 1, false)
    // the single-layer forms (lifting projection with its per-sample sums, channel reduction) of every width whose two-layer
    // block has a backward kernel: a model of that width then trains without any einsum recompute
    PWB_CASE(6, 6, 6, false) PWB_CASE(6, 6, 1, false) PWB_CASE(12, 12, 12, false) PWB_CASE(12, 12, 1, false)
    PWB_CASE(14, 14, 14, false) PWB_CASE(14, 14, 1, false) PWB_CASE(16, 16, 16, false) PWB_CASE(16, 16, 1, false)
    PWB_CASE(20, 20, 20, false) PWB_CASE(20, 20, 1, false) PWB_CASE(24, 24, 24, false) PWB_CASE(24, 24, 1, false)
    PWB_CASE(32, 32, 32, false) PWB_CASE(32, 32, 1, false)
#undef PWB_CASE
    return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: channels (%d -> %d -> %d) not instantiated", ci, cm, co);
}


// ------------------------------------------------------------------ LayerNormnd statistics
// sum and sum of squares of every row of a (rows, L) fp32 matrix (one row = one sample's (C, X, Y, T) block),
// accumulated in double.  torch's GroupNorm moments kernel runs ONE workgroup per row (6 ms for 32 rows of
// 6.5 M elements on MI355X); here every row is cut into chunks reduced by different workgroups.
__global__ __launch_bounds__(256) void k_row_moments(const float* __restrict__ x, double* __restrict__ stats, long L,
                                                     int chunks) {
    __shared__ double sh[2][4];
    const int row = blockIdx.y, chunk = blockIdx.x;
    const long per = ((L + chunks - 1) / chunks + 3) & ~3L;
    const long lo = (long)chunk * per, hi = lo + per < L ? lo + per : L;
    const float* r = x + (size_t)row * L;
    double s1 = 0.0, s2 = 0.0;
    float a1 = 0.f, a2 = 0.f;
    int cnt = 0;
    const bool vec = ((L & 3) == 0);
    if (vec) {
        for (long i = lo + (long)threadIdx.x * 4; i < hi; i += 256 * 4) {
            const float4 v = *reinterpret_cast<const float4*>(r + i);
            a1 += (v.x + v.y) + (v.z + v.w);
            a2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            if (++cnt == 16) { s1 += a1; s2 += a2; a1 = a2 = 0.f; cnt = 0; }  // short fp32 runs, double totals
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += 256) {
            const float v = r[i];
            a1 += v;
            a2 += v * v;
            if (++cnt == 64) { s1 += a1; s2 += a2; a1 = a2 = 0.f; cnt = 0; }
        }
    }
    s1 += a1;
    s2 += a2;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off);
        s2 += __shfl_down(s2, off);
    }
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    if (lane == 0) { sh[0][wave] = s1; sh[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&stats[2 * row], sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
        atomicAdd(&stats[2 * row + 1], sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    }
}

// the same for float64 rows (plain double accumulation)
__global__ __launch_bounds__(256) void k_row_moments_f64(const double* __restrict__ x, double* __restrict__ stats, long L,
                                                         int chunks) {
    __shared__ double sh[2][4];
    const int row = blockIdx.y, chunk = blockIdx.x;
    const long per = ((L + chunks - 1) / chunks + 1) & ~1L;
    const long lo = (long)chunk * per, hi = lo + per < L ? lo + per : L;
    const double* r = x + (size_t)row * L;
    double s1 = 0.0, s2 = 0.0;
    if ((L & 1) == 0) {
        for (long i = lo + (long)threadIdx.x * 2; i < hi; i += 256 * 2) {
            const double2 v = *reinterpret_cast<const double2*>(r + i);
            s1 += v.x + v.y;
            s2 += v.x * v.x + v.y * v.y;
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += 256) { const double v = r[i]; s1 += v; s2 += v * v; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off);
        s2 += __shfl_down(s2, off);
    }
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    if (lane == 0) { sh[0][wave] = s1; sh[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&stats[2 * row], sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
        atomicAdd(&stats[2 * row + 1], sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    }
}

extern "C" int tcfd_row_moments_f64(const void* x, void* stats, int rows, long L, void* stream) {
    if (!x || !stats || rows <= 0 || L <= 0) return FAIL(TCFD_EINVAL, "row_moments_f64: bad argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(stats, 0, (size_t)rows * 2 * sizeof(double), st));
    int chunks = (int)std::min<long>(std::max<long>(L / (256 * 2 * 8), 1), 2048 / std::max(rows, 1) + 1);
    hipLaunchKernelGGL(k_row_moments_f64, dim3((unsigned)chunks, (unsigned)rows), dim3(256), 0, st, (const double*)x,
                       (double*)stats, L, chunks);
    HIP_TRY(hipGetLastError());
    return 0;
}

// stats (rows, 2) double, zeroed by this call (memset node on the stream) before the accumulation.
extern "C" int tcfd_row_moments(const void* x, void* stats, int rows, long L, void* stream) {
    if (!x || !stats || rows <= 0 || L <= 0) return FAIL(TCFD_EINVAL, "row_moments: bad argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(stats, 0, (size_t)rows * 2 * sizeof(double), st));
    int chunks = (int)std::min<long>(std::max<long>(L / (256 * 4 * 8), 1), 2048 / std::max(rows, 1) + 1);
    hipLaunchKernelGGL(k_row_moments, dim3((unsigned)chunks, (unsigned)rows), dim3(256), 0, st, (const float*)x,
                       (double*)stats, L, chunks);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ lifting operator: LayerNorm folded into the projection
// proj(LayerNormnd(v + q)) of the lifting operator (fno/sfno.py:252-254, fno/base.py:61-83) with v ONE channel (b, P) and q the
// (C, P) positional table: the statistics of a sample's (C, P) block follow from three sums over v -- sum, sum of squares and
// the dot product with qs[p] = sum_c q[c][p] -- and two constants of the table (sq = sum q, sq2 = sum q^2):
//     s1 = C sum(v) + sq ,   s2 = C sum(v^2) + 2 <v, qs> + sq2 ,   mu = s1 / (C P) ,   rstd = 1 / sqrt(s2 / (C P) - mu^2 + eps)
// and normalisation + affine + projection collapse into per-sample weights for the pointwise kernel (its `pe` mode):
//     w2t[b][c][o] = W[o][c] gamma[c] rstd_b ,   fb[b][o] = sum_c (beta[c] - gamma[c] mu_b rstd_b) W[o][c] + bias[o].
// Two launches replace ~25 tensor-op launches (a GEMV, a dozen 0-dim double ops, broadcasts) per forward.
__global__ __launch_bounds__(256) void k_row_moments_dot(const float* __restrict__ x, const float* __restrict__ qs,
                                                         double* __restrict__ stats, long L, int chunks) {
    __shared__ double sh[3][4];
    const int row = blockIdx.y, chunk = blockIdx.x;
    const long per = ((L + chunks - 1) / chunks + 3) & ~3L;
    const long lo = (long)chunk * per, hi = lo + per < L ? lo + per : L;
    const float* r = x + (size_t)row * L;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int cnt = 0;
    if ((L & 3) == 0) {
        for (long i = lo + (long)threadIdx.x * 4; i < hi; i += 256 * 4) {
            const float4 v = *reinterpret_cast<const float4*>(r + i);
            const float4 q = *reinterpret_cast<const float4*>(qs + i);
            a1 += (v.x + v.y) + (v.z + v.w);
            a2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            a3 += (v.x * q.x + v.y * q.y) + (v.z * q.z + v.w * q.w);
            if (++cnt == 16) { s1 += a1; s2 += a2; s3 += a3; a1 = a2 = a3 = 0.f; cnt = 0; }  // short fp32 runs, double totals
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += 256) {
            const float v = r[i];
            a1 += v; a2 += v * v; a3 += v * qs[i];
            if (++cnt == 64) { s1 += a1; s2 += a2; s3 += a3; a1 = a2 = a3 = 0.f; cnt = 0; }
        }
    }
    s1 += a1; s2 += a2; s3 += a3;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off);
        s2 += __shfl_down(s2, off);
        s3 += __shfl_down(s3, off);
    }
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    if (lane == 0) { sh[0][wave] = s1; sh[1][wave] = s2; sh[2][wave] = s3; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        atomicAdd(&stats[3 * row + k], sh[k][0] + sh[k][1] + sh[k][2] + sh[k][3]);
    }
}

__global__ __launch_bounds__(256) void k_lift_fold(const double* __restrict__ stats, const double* __restrict__ sq,
                                                   const double* __restrict__ sq2, const float* __restrict__ W,
                                                   const float* __restrict__ bias, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, double eps, float* __restrict__ w2t,
                                                   float* __restrict__ fb, double* __restrict__ moments, int C, int co, long P) {
    const int b = blockIdx.x;
    const double L = (double)C * (double)P;
    const double s1 = C * stats[3 * b] + sq[0];
    const double s2 = C * stats[3 * b + 1] + 2.0 * stats[3 * b + 2] + sq2[0];
    const double mu = s1 / L;
    double var = s2 / L - mu * mu;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + eps);
    if (threadIdx.x == 0 && moments) { moments[2 * b] = s1; moments[2 * b + 1] = s2; }
    for (int i = threadIdx.x; i < C * co; i += 256) {
        const int c = i / co, o = i - c * co;
        const double g = gamma ? (double)gamma[c] : 1.0;
        w2t[((size_t)b * C + c) * co + o] = (float)((double)W[(size_t)o * C + c] * (g * rstd));
    }
    for (int o = threadIdx.x; o < co; o += 256) {
        double acc = bias ? (double)bias[o] : 0.0;
        double dot = 0.0;
        for (int c = 0; c < C; ++c) {
            const double g = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
            dot += (be - g * (mu * rstd)) * (double)W[(size_t)o * C + c];
        }
        fb[(size_t)b * co + o] = (float)(dot + acc);
    }
}

// v (b, P) fp32; qs (P) fp32; sq, sq2: ONE double each on the device (constants of the table); W (co, C), bias (co) / gamma (C) /
// beta (C) fp32 or NULL; outputs w2t (b, C, co), fb (b, co) fp32, moments (b, 2) double or NULL; scratch (b, 3) double.
extern "C" int tcfd_fno_lift_fold(const void* v, const void* qs, const void* sq, const void* sq2, const void* W, const void* bias,
                                  const void* gamma, const void* beta, double eps, void* w2t, void* fb, void* moments,
                                  void* scratch, int batch, int C, int co, long P, void* stream) {
    if (!v || !qs || !sq || !sq2 || !W || !w2t || !fb || !scratch || batch <= 0 || C <= 0 || co <= 0 || P <= 0)
        return FAIL(TCFD_EINVAL, "lift_fold: bad argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(scratch, 0, (size_t)batch * 3 * sizeof(double), st));
    int chunks = (int)std::min<long>(std::max<long>(P / (256 * 4 * 8), 1), 2048 / std::max(batch, 1) + 1);
    hipLaunchKernelGGL(k_row_moments_dot, dim3((unsigned)chunks, (unsigned)batch), dim3(256), 0, st, (const float*)v,
                       (const float*)qs, (double*)scratch, P, chunks);
    hipLaunchKernelGGL(k_lift_fold, dim3((unsigned)batch), dim3(256), 0, st, (const double*)scratch, (const double*)sq,
                       (const double*)sq2, (const float*)W, (const float*)bias, (const float*)gamma, (const float*)beta, eps,
                       (float*)w2t, (float*)fb, (double*)moments, C, co, P);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ lifting operator: the spectrum of its projection
// The lifting operator projects ONE input channel to `co` channels through a per-sample affine map of (v + table):
//     v0[b, o](p) = sum_c w2t[b, c, o] (v[b](p) + q_c(p)) + fb[b, o]                  (tcfd_fno_lift_fold, fno/sfno.py:252-254)
// and the first thing that happens to v0 is a truncated transform (SpectralConvT, :256).  The transform is linear, so
//     V0^[b, o] = sum_c w2t[b, c, o] (V^[b] + Q^_c) + fb[b, o] 1^
// with V^ the kept modes of the ONE-channel input, Q^_c those of the table channels and 1^ those of the constant field
// (left zero padding in t included) -- the last two do not depend on the input and are formed once.  One transform of one
// channel per sample instead of `co`, and v0 (an activation-sized tensor) is neither written nor read.
// vh (b, K), table (C + 1, K) = [Q^_0 .. Q^_{C-1}, 1^] complex64; w2t (b, C, co), fb (b, co) fp32; out (b, co, K) complex64.
#define LIFT_MAXC 32
__global__ __launch_bounds__(256) void k_lift_spectrum(const cf* __restrict__ vh, const cf* __restrict__ table,
                                                       const float* __restrict__ w2t, const float* __restrict__ fb,
                                                       cf* __restrict__ out, int C, int co, long K) {
    const long k = blockIdx.x * 256L + threadIdx.x;
    const int b = blockIdx.y;
    if (k >= K) return;
    const cf v = vh[(size_t)b * K + k];
    cf e[LIFT_MAXC];
#pragma unroll
    for (int c = 0; c < LIFT_MAXC; ++c)
        if (c < C) {
            const cf q = table[(size_t)c * K + k];
            e[c] = mk<float>(v.x + q.x, v.y + q.y);
        }
    const cf one = table[(size_t)C * K + k];
    const float* wb = w2t + (size_t)b * C * co;
    for (int o = 0; o < co; ++o) {
        const float f = fb[(size_t)b * co + o];
        float re = f * one.x, im = f * one.y;
#pragma unroll
        for (int c = 0; c < LIFT_MAXC; ++c)
            if (c < C) {
                const float w = wb[(size_t)c * co + o];       // wave uniform: scalar loads
                re = fmaf(w, e[c].x, re);
                im = fmaf(w, e[c].y, im);
            }
        out[((size_t)b * co + o) * K + k] = mk<float>(re, im);
    }
}
extern "C" int tcfd_fno_lift_spectrum(const void* vh, const void* table, const void* w2t, const void* fb, void* out, int batch,
                                      int C, int co, long K, void* stream) {
    if (!vh || !table || !w2t || !fb || !out || batch <= 0 || C <= 0 || co <= 0 || K <= 0)
        return FAIL(TCFD_EINVAL, "lift_spectrum: bad argument");
    if (C > LIFT_MAXC) return FAIL(TCFD_EINVAL, "lift_spectrum: %d table channels > %d", C, LIFT_MAXC);
    if (batch > 65535) return FAIL(TCFD_EINVAL, "lift_spectrum: batch %d exceeds the grid's y range", batch);
    hipLaunchKernelGGL(k_lift_spectrum, dim3((unsigned)((K + 255) / 256), (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                       (const cf*)vh, (const cf*)table, (const float*)w2t, (const float*)fb, (cf*)out, C, co, K);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ small reductions of the training step
// Column sums of a (rows, cols) fp32 matrix in double: the per-wave rows of partial weight-gradient sums of the pointwise
// backward (2048 x ~1800 values).  torch's sum(dim=0) runs this shape at ~80 GB/s (0.19 ms per layer); here lanes run along
// the columns, `slices` row ranges go to blockIdx.y, a second tiny launch adds the slices: deterministic, ~10 us.
__global__ __launch_bounds__(256) void k_sum_rows_stage(const float* __restrict__ in, double* __restrict__ scratch, long rows,
                                                        long cols, int slices) {
    const long c = blockIdx.x * 256L + threadIdx.x;
    if (c >= cols) return;
    const long r0 = rows * blockIdx.y / slices, r1 = rows * (blockIdx.y + 1) / slices;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    long r = r0;
    for (; r + 4 <= r1; r += 4) {
        a0 += (double)in[r * cols + c];
        a1 += (double)in[(r + 1) * cols + c];
        a2 += (double)in[(r + 2) * cols + c];
        a3 += (double)in[(r + 3) * cols + c];
    }
    for (; r < r1; ++r) a0 += (double)in[r * cols + c];
    scratch[(long)blockIdx.y * cols + c] = (a0 + a1) + (a2 + a3);
}
__global__ __launch_bounds__(256) void k_sum_rows_final(const double* __restrict__ scratch, double* __restrict__ out, long cols,
                                                        int slices) {
    const long c = blockIdx.x * 256L + threadIdx.x;
    if (c >= cols) return;
    double a = 0;
    for (int s = 0; s < slices; ++s) a += scratch[(long)s * cols + c];
    out[c] = a;
}
// out (cols) double; scratch: tcfd_sum_rows_slices(rows) * cols doubles
extern "C" int tcfd_sum_rows_slices(long rows) { return (int)std::max<long>(1, std::min<long>(64, rows / 32)); }
extern "C" int tcfd_sum_rows(const void* in, void* out, void* scratch, long rows, long cols, void* stream) {
    if (!in || !out || !scratch || rows <= 0 || cols <= 0) return FAIL(TCFD_EINVAL, "sum_rows: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int slices = tcfd_sum_rows_slices(rows);
    const unsigned bx = (unsigned)((cols + 255) / 256);
    hipLaunchKernelGGL(k_sum_rows_stage, dim3(bx, (unsigned)slices), dim3(256), 0, st, (const float*)in, (double*)scratch, rows, cols,
                       slices);
    hipLaunchKernelGGL(k_sum_rows_final, dim3(bx), dim3(256), 0, st, (const double*)scratch, (double*)out, cols, slices);
    HIP_TRY(hipGetLastError());
    return 0;
}

// The same sums delivered where they belong: up to 8 segments (src_off, nrows, ncols, pitch) of the summed row -- sub-matrices
// of the [dW2 | db2 | dWs] / [dW1 | db1] blocks a backward kernel accumulates -- written as fp32 into their own dense tensors by
// the final pass (the parameter gradients of a block were one conversion and six strided copies after tcfd_sum_rows).
struct SumSegs {
    long src_off[8], pitch[8];
    int nrows[8], ncols[8];
    float* dst[8];
    int n;
};
__global__ __launch_bounds__(256) void k_sum_rows_final_scatter(const double* __restrict__ scratch, long cols, int slices, SumSegs sg) {
    const long c = blockIdx.x * 256L + threadIdx.x;
    if (c >= cols) return;
    // eight loads in flight per lane (the launch is a handful of workgroups: one dependent chain of `slices` loads took 19 us)
    double p8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int s = 0;
    for (; s + 8 <= slices; s += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) p8[u] += scratch[(long)(s + u) * cols + c];
    }
    for (; s < slices; ++s) p8[0] += scratch[(long)s * cols + c];
    const double a = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
    for (int k = 0; k < sg.n; ++k) {
        const long rel = c - sg.src_off[k];
        if (rel < 0) continue;
        const long r = rel / sg.pitch[k], col = rel - r * sg.pitch[k];
        if (r < sg.nrows[k] && col < sg.ncols[k]) sg.dst[k][r * sg.ncols[k] + col] = (float)a;
    }
}
// segs: nseg x 4 longs {src_off, nrows, ncols, pitch} (host), dst: nseg device pointers (host array)
extern "C" int tcfd_sum_rows_scatter(const void* in, void* scratch, long rows, long cols, int nseg, const long* segs,
                                     void* const* dst, void* stream) {
    if (!in || !scratch || rows <= 0 || cols <= 0 || nseg < 1 || nseg > 8 || !segs || !dst)
        return FAIL(TCFD_EINVAL, "sum_rows_scatter: bad argument");
    SumSegs sg;
    sg.n = nseg;
    for (int k = 0; k < nseg; ++k) {
        sg.src_off[k] = segs[4 * k]; sg.nrows[k] = (int)segs[4 * k + 1]; sg.ncols[k] = (int)segs[4 * k + 2]; sg.pitch[k] = segs[4 * k + 3];
        sg.dst[k] = (float*)dst[k];
        if (!dst[k] || sg.pitch[k] < 1 || sg.src_off[k] < 0 || sg.nrows[k] < 1 || sg.ncols[k] < 1 || sg.ncols[k] > sg.pitch[k] ||
            sg.src_off[k] + (long)(sg.nrows[k] - 1) * sg.pitch[k] + sg.ncols[k] > cols)
            return FAIL(TCFD_EINVAL, "sum_rows_scatter: segment %d out of the row", k);
    }
    hipStream_t st = (hipStream_t)stream;
    const int slices = tcfd_sum_rows_slices(rows);
    const unsigned bx = (unsigned)((cols + 255) / 256);
    hipLaunchKernelGGL(k_sum_rows_stage, dim3(bx, (unsigned)slices), dim3(256), 0, st, (const float*)in, (double*)scratch, rows, cols,
                       slices);
    hipLaunchKernelGGL(k_sum_rows_final_scatter, dim3(bx), dim3(256), 0, st, (const double*)scratch, cols, slices, sg);
    HIP_TRY(hipGetLastError());
    return 0;
}

// g (rows, sT) = 0 except g[r][sT - 1] = sum_t d[r][t], d (rows, T): the gradient of a skip input of which only the LAST time
// slice was used, broadcast over the T output steps (lifting operator, fno/sfno.py:258-259), from the full dL/dz2 in one pass
// (zeros_like + sum(dim=-1) + strided copy before: 1.1 ms at config 5).
__global__ __launch_bounds__(256) void k_sum_t_into_last(const float* __restrict__ d, float* __restrict__ g, long rows, int T, int sT) {
    const long r = blockIdx.x * 256L + threadIdx.x;
    if (r >= rows) return;
    const float* p = d + r * T;
    float a = 0.f;
    if ((T & 1) == 0) {
        for (int t = 0; t < T; t += 2) {
            const float2 v = *reinterpret_cast<const float2*>(p + t);
            a += v.x + v.y;
        }
    } else {
        for (int t = 0; t < T; ++t) a += p[t];
    }
    float* q = g + r * sT;
    for (int t = 0; t < sT - 1; ++t) q[t] = 0.f;
    q[sT - 1] = a;
}
extern "C" int tcfd_sum_t_into_last(const void* d, void* g, long rows, int T, int sT, void* stream) {
    if (!d || !g || rows <= 0 || T <= 0 || sT <= 0) return FAIL(TCFD_EINVAL, "sum_t_into_last: bad argument");
    hipLaunchKernelGGL(k_sum_t_into_last, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)d,
                       (float*)g, rows, T, sT);
    HIP_TRY(hipGetLastError());
    return 0;
}

// M_b[o][c] = sum_p dy[b][o][p] xin[b][c][p]  for c < C,  M_b[o][C] = sum_p dy[b][o][p]      (per sample b)
// -- everything the backward of proj(LayerNorm(xin)) needs from the data (fno.py::_hip_norm_proj_backward), with
// xin = x (b, C, P) or x1 (b, P) + pe (C, P).  One wave takes 16 points at a time: lane (q, c) loads the 16-byte run
// dy[c][4q .. 4q+3] and xin[c][4q .. 4q+3]; register r of the two runs IS the A resp. B fragment of the k-step over the points
// {4q + r}, so the 16 x 16 tile of sums grows by four v_mfma_f32_16x16x4_f32 per group and nothing else: the kernel runs at
// the rate its two loads arrive (the LDS-staged k_pointwise_bwd<10,10,10,false> spent 0.59 ms on the same sums at config 5).
// partials: (waves_per_sample, batch, 256) floats, row-major 16 x 16 tiles [o][c]; added up by tcfd_sum_rows.
// TR x TC tiles of 16 x 16 (co <= 16 TR rows, C + 1 <= 16 TC columns): wide layers run the same kernel with more tiles -- before,
// C > 15 fell to the LDS-staged k_pointwise_bwd<C, C, C, false> with per-sample rows: 36 ms at width 32, a third of its training step.
// partials: (waves_per_sample, batch, 16 TR, 16 TC) floats, row-major.
template <int TR, int TC>
__global__ __launch_bounds__(256) void k_sample_outer_mfma(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ pe, float* __restrict__ partials, long P,
                                                           int C, int CO, int waves_per_sample, int batch) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
    const int w = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), b = blockIdx.y;
    const long groups = P / 16;
    const float* dyr[TR];
    const float* xr[TC];
    const float* per[TC];
#pragma unroll
    for (int tr = 0; tr < TR; ++tr) { const int o = 16 * tr + c; dyr[tr] = dy + ((size_t)b * CO + (o < CO ? o : 0)) * P + 4 * q; }
#pragma unroll
    for (int tc = 0; tc < TC; ++tc) {
        const int ch = 16 * tc + c, chc = ch < C ? ch : 0;
        xr[tc] = pe ? x + (size_t)b * P + 4 * q : x + ((size_t)b * C + chc) * P + 4 * q;
        per[tc] = pe ? pe + (size_t)chc * P + 4 * q : nullptr;
    }
    f4 acc[TR][TC];
#pragma unroll
    for (int tr = 0; tr < TR; ++tr)
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) acc[tr][tc] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (long g = w; g < groups; g += waves_per_sample) {
        f4 a[TR], v[TC];
#pragma unroll
        for (int tr = 0; tr < TR; ++tr) {
            a[tr] = *reinterpret_cast<const f4*>(dyr[tr] + g * 16);
            if (16 * tr + c >= CO) a[tr] = f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) {
            const int ch = 16 * tc + c;
            if (16 * tc < C) {                                    // (a tile past the channels holds the column of ones only: no load)
                v[tc] = *reinterpret_cast<const f4*>(xr[tc] + g * 16);
                if (per[tc]) v[tc] += *reinterpret_cast<const f4*>(per[tc] + g * 16);
            }
            if (ch >= C) { const float ones = ch == C ? 1.f : 0.f; v[tc] = f4{ones, ones, ones, ones}; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int tr = 0; tr < TR; ++tr)
#pragma unroll
                for (int tc = 0; tc < TC; ++tc) acc[tr][tc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tr][r], v[tc][r], acc[tr][tc], 0, 0, 0);
    }
    float* out = partials + ((size_t)w * batch + b) * (256 * TR * TC);
#pragma unroll
    for (int tr = 0; tr < TR; ++tr)
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * tr + 4 * q + r) * (16 * TC) + 16 * tc + c] = acc[tr][tc][r];
}
// co <= 32, c <= 47; the caller's partials hold (waves_per_sample, batch, 16 ceil(co / 16), 16 ceil((c + 1) / 16)) floats
extern "C" int tcfd_fno_sample_outer_sums(const void* dy, const void* x, const void* pe, void* partials, int batch, int c, int co,
                                          long P, int waves_per_sample, void* stream) {
    if (!dy || !x || !partials || batch <= 0 || c < 1 || c > 47 || co < 1 || co > 32 || P <= 0 || P % 16 != 0 ||
        waves_per_sample < 4 || waves_per_sample % 4 != 0)
        return FAIL(TCFD_EINVAL, "fno_sample_outer_sums: bad argument (needs c <= 47, co <= 32, P %% 16 == 0, whole workgroups)");
    const int tr = (co + 15) / 16, tc = (c + 16) / 16;
    const dim3 grid((unsigned)(waves_per_sample / 4), (unsigned)batch);
#define TCFD_OUTER(TR_, TC_)                                                                                                         \
    if (tr == TR_ && tc == TC_)                                                                                                      \
        hipLaunchKernelGGL((k_sample_outer_mfma<TR_, TC_>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy, (const float*)x, \
                           (const float*)pe, (float*)partials, P, c, co, waves_per_sample, batch);
    TCFD_OUTER(1, 1) TCFD_OUTER(1, 2) TCFD_OUTER(1, 3) TCFD_OUTER(2, 1) TCFD_OUTER(2, 2) TCFD_OUTER(2, 3)
#undef TCFD_OUTER
    HIP_TRY(hipGetLastError());
    return 0;
}
