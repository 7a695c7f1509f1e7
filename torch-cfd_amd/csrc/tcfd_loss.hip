// tcfd_loss.hip -- MI355X (gfx950) kernels + C ABI for the Fourier-domain Sobolev loss of config 5 ("forward + loss")
// reference: fno/losses.py:263-315 (SobolevLoss.forward): fftn over dims (1, 2) of the time-last tensors x and y, the
// multiplier sqrt(alpha + 4 pi^2 |k|^2)^(order/2), per-time Frobenius norms, time-l2, relative / mesh-weighted /
// time-averaged / batch-mean.
//
// The reference (and the round-3 path here) forms x - y, permutes both to time-first copies, transforms them with two
// full 2-D transforms each (spectrum written and re-read) and reduces: >= 15 A_1 of traffic for a 4 A_1 job
// (A_1 = bytes of x).  Here three launches, and the time-last layout is read in place:
//
//   k_loss_rows   per (b, x-row) slab [Y][T] of x and of y (10 KB contiguous each at config 5): d = x - y formed while
//                 staging, ONE complex Y-point FFT per time step of z_t = d_t + i y_t (two real sequences per transform),
//                 Hermitian separation -> half-spectrum rows D1 / Y1 (f, b, t, x, ky <= Y/2)
//   k_loss_cols   X-point FFT down 16-column (128-byte) tiles of D1 / Y1, |.|^2 times the weight table (Hermitian
//                 multiplicity and fft-norm folded in), accumulated in double: no spectrum is written
//   k_loss_finish the (b, t) sums -> the scalar: sqrt, relative, mesh weighting, time average, batch mean
//
// Algorithmic bytes: read x, y (2 A_1), write + read the two half-spectrum planes (4 A_1) = 6 A_1 (0.5 GB at config 5).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/tcfd.h"
#include "tcfd_fft.hpp"

using namespace tcfd;

extern "C" const char* tcfd_last_error(void);
int tcfd_set_error(int code, const char* fmt, ...);  // defined in tcfd_ns2d.hip
#define FAIL(...) tcfd_set_error(__VA_ARGS__)
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return FAIL(TCFD_EHIP, "%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

struct tcfd_loss_plan {
    int n;       // square grid: 2^k in [16, 1024], 3 * 2^k in [96, 768] or 5 * 2^k in [80, 640]
    int dtype;   // TCFD_C64: float data, TCFD_C128: double data
    void* tw;    // [n] exp(-2 pi i k / n) in the plan's precision
};

typedef unsigned int b128 __attribute__((ext_vector_type(4)));

// elements per lane of the in-wave row transforms (a transform's n / EPT lanes must fit one wave) and of the column tiles
template <typename T, int N>
struct LossCfg {
    static constexpr int BASE = sizeof(T) == 8 ? 8 : 16;
    static constexpr int MIX = N % 3 == 0 ? 12 : (N % 5 == 0 ? 20 : 0);    // 3 * 2^k / 5 * 2^k: radix-12 / radix-20 first pass (tcfd_fft.hpp)
    static constexpr int ROW_EPT0 = N >= 256 ? BASE : (N >= 64 ? 8 : 4);
    static constexpr int ROW_EPT = MIX ? MIX : (N / ROW_EPT0 > 64 ? N / 64 : ROW_EPT0);
    static constexpr int COLS = sizeof(T) == 8 ? 8 : 16;                    // 128-byte tile rows
    static constexpr int COL_EPT0 = N >= 256 ? BASE : (N >= 64 ? 8 : 4);
    static constexpr int COL_EPT = MIX ? MIX : (COLS * (N / COL_EPT0) > 1024 ? COLS * N / 1024 : COL_EPT0);
};

// ------------------------------------------------------------------ pass 1: rows
// A workgroup owns NS consecutive slabs (b, x); thread = (transform tr = s * P + p, lane j of its G-lane group).  The F * nt
// real sequences of a slab (sequence q = t * F + f; f = 0: x - y, f = 1: y) ride two per complex transform.
template <typename T, int Y, int EPT>
__global__ __launch_bounds__(1024) void k_loss_rows(const T* __restrict__ x, const T* __restrict__ y, cx<T>* __restrict__ out,
                                                    const cx<T>* __restrict__ tw, int nt, int F, int P, int NS, long slabs,
                                                    int X, long batch, int ldk, unsigned per) {
    typedef cx<T> cf;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int G = Y / EPT;
    constexpr int NV = 16 / (int)sizeof(T);
    const int tr = threadIdx.x / G, j = threadIdx.x % G;
    const int s = tr / P, p = tr - s * P;
    const long base = (long)blockIdx.x * NS;
    const int count = (int)(slabs - base < NS ? slabs - base : NS);
    const size_t slab_elems = (size_t)Y * nt;
    {   // stage d = x - y (and y) of the workgroup's slabs, 16 bytes per lane
        const int n4 = (int)(slab_elems / NV);
        const b128* x4 = reinterpret_cast<const b128*>(x + (size_t)base * slab_elems);
        const b128* y4 = y ? reinterpret_cast<const b128*>(y + (size_t)base * slab_elems) : nullptr;
        // all loads of a trip are issued before the first is used (a plain load -> store loop makes one memory round trip
        // per 16 bytes and lane: hipcc waits for each load before issuing the next)
        const int total4 = count * n4;
        constexpr int UN = 4;
        for (int i0 = threadIdx.x; i0 < total4; i0 += UN * blockDim.x) {
            union U { b128 v; T e[NV]; };
            U a[UN], b[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int idx = i0 + u * blockDim.x;
                if (idx < total4) {
                    a[u].v = __builtin_nontemporal_load(x4 + idx);
                    if (y4) b[u].v = __builtin_nontemporal_load(y4 + idx);
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int idx = i0 + u * blockDim.x;
                if (idx < total4) {
                    const int q = idx / n4, i = idx - q * n4;
                    b128* d4 = reinterpret_cast<b128*>(smem_raw + (size_t)q * per);
                    if (y4) {
#pragma unroll
                        for (int w = 0; w < NV; ++w) a[u].e[w] -= b[u].e[w];
                        if (F == 2) d4[n4 + i] = b[u].v;
                    }
                    d4[i] = a[u].v;
                }
            }
        }
    }
    __syncthreads();
    const int nseq = F * nt;
    const int q0 = 2 * p, q1 = 2 * p + 1;
    const bool live = s < count && q0 < nseq;
    const bool two = q1 < nseq;
    const int t0 = q0 / F, f0 = q0 - t0 * F, t1 = two ? q1 / F : 0, f1 = two ? q1 - t1 * F : 0;
    cf z[EPT];
    {
        const T* reg = reinterpret_cast<const T*>(smem_raw + (size_t)s * per);
        const T* a0 = reg + (size_t)f0 * slab_elems + t0;
        const T* a1 = reg + (size_t)f1 * slab_elems + t1;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const size_t o = (size_t)(j + e * G) * nt;
            z[e] = live ? mk<T>(a0[o], two ? a1[o] : (T)0) : mk<T>((T)0, (T)0);
        }
    }
    __syncthreads();   // every transform of the workgroup holds its input: the slab bytes become the exchange buffers
    cf* lds = reinterpret_cast<cf*>(smem_raw + (size_t)s * per) + (size_t)p * Y;
    tile_fft<T, Y, EPT, -1, 1, true, false>(z, lds, tw, j, 0);
#pragma unroll
    for (int e = 0; e < EPT; ++e) lds[j + e * G] = z[e];   // Z in natural order: the separation below pairs k with Y - k
    group_sync<0>();
    if (!live) return;
    // D[k] = (Z[k] + conj Z[-k]) / 2 ,  E[k] = (Z[k] - conj Z[-k]) / 2i   for the two real sequences of this transform
    const long slab = base + s;
    const long b = slab / X, i = slab - b * X;
    cf* o0 = out + ((((size_t)f0 * batch + b) * nt + t0) * X + i) * ldk;
    cf* o1 = out + ((((size_t)f1 * batch + b) * nt + t1) * X + i) * ldk;
    const T h = (T)0.5;
    for (int k = j; k <= Y / 2; k += G) {
        const cf za = lds[k], zb = lds[k ? Y - k : 0];
        o0[k] = mk<T>((za.x + zb.x) * h, (za.y - zb.y) * h);
        if (two) o1[k] = mk<T>((za.y + zb.y) * h, (zb.x - za.x) * h);
    }
}

// ------------------------------------------------------------------ pass 2: columns + weighted |.|^2
template <typename T, int X, int EPT, int C>
__global__ __launch_bounds__(C*(X / EPT)) void k_loss_cols(const cx<T>* __restrict__ in, const T* __restrict__ w2,
                                                           double* __restrict__ partial, const cx<T>* __restrict__ tw, int m,
                                                           int ldk, int ntiles) {
    typedef cx<T> cf;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    constexpr int G = X / EPT;
    constexpr int THREADS = C * G;
    const int c = threadIdx.x % C, j = threadIdx.x / C;
    const int tile = blockIdx.x % ntiles;
    const size_t img = blockIdx.x / ntiles;
    const int q = tile * C + c;
    const bool valid = q < m;
    cf z[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        z[e] = valid ? in[(img * X + j + e * G) * (size_t)ldk + q] : mk<T>((T)0, (T)0);
    T wv[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) wv[e] = valid ? w2[(size_t)(j + e * G) * m + q] : (T)0;   // in flight across the transform
    tile_fft<T, X, EPT, -1, C, false, true>(z, lds, tw, j, c);
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        acc += (double)wv[e] * ((double)z[e].x * (double)z[e].x + (double)z[e].y * (double)z[e].y);
    // deterministic block sum: wave butterflies, then one value per wave through LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __syncthreads();                                   // the exchange buffer is free again
    double* red = reinterpret_cast<double*>(smem_raw);
    constexpr int NW = (THREADS + 63) / 64;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < NW; ++w) t += red[w];
        partial[blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------ pass 3: the scalar
// partial (F, batch, nt, ntiles).  loss_b = sqrt(sum_t ||w (x - y)^_t||^2) / yn_b, yn_b = sqrt(sum_t ||w y^_t||^2) (relative) or 1,
// yn_b / n when mesh weighted; / sqrt(nt) when time averaged; mean or sum over b; / n when mesh weighted (losses.py:297-314).
template <typename T>
__global__ __launch_bounds__(256) void k_loss_finish(const double* __restrict__ partial, T* __restrict__ out,
                                                     double* __restrict__ sums, long batch, int nt, int ntiles, int F, int n,
                                                     int relative, int mesh_weighted, int time_average, int reduction) {
    // one block.  Step 1: every (field, sample, time) triple adds its tiles (all lanes busy, the loads of a lane independent:
    // one lane per SAMPLE walking F * nt * ntiles values one after the other took 33 us of latency at config 5) and leaves
    // the sum in `sums` (the caller's array or a corner of the workspace).  Step 2: one lane per sample adds its nt terms in
    // a fixed order (deterministic), block sum.
    __shared__ double red[256];
    const long triples = (long)F * batch * nt;
    for (long i = threadIdx.x; i < triples; i += 256) {
        const double* pp = partial + (size_t)i * ntiles;
        double v = 0.0;
        for (int k = 0; k < ntiles; ++k) v += pp[k];
        sums[i] = v;
    }
    __syncthreads();      // the sums of this (only) workgroup are visible to it
    double mine = 0.0;
    for (long b = threadIdx.x; b < batch; b += 256) {
        double t0 = 0.0, t1 = 0.0;
        const volatile double* s0 = sums + (size_t)b * nt;
        const volatile double* s1 = sums + ((size_t)batch + b) * nt;
        for (int t = 0; t < nt; ++t) t0 += s0[t];
        if (F == 2)
            for (int t = 0; t < nt; ++t) t1 += s1[t];
        double loss = sqrt(t0);
        double yn = (relative && F == 2) ? sqrt(t1) : 1.0;
        if (mesh_weighted) {
            // mesh_weighted == 2: the unit norm of a non-relative loss is float32 in the reference (torch.ones under a float32
            // default dtype, fno/losses.py:297), so 1 / n is rounded to float32 before it divides the loss
            if (mesh_weighted == 2 && !(relative && F == 2)) yn = (double)(1.0f / (float)n);
            else yn /= (double)n;
        }
        loss /= yn;
        if (time_average) loss /= sqrt((double)nt);
        mine += loss;
    }
    red[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double v = red[0];
        if (reduction) v /= (double)batch;
        if (mesh_weighted) v /= (double)n;
        out[0] = (T)v;
    }
}

// ------------------------------------------------------------------ the gradient with respect to x
// loss = red * sum_b sqrt(S0_b) / yn_b (times the time average), S0_b = sum_{t, k} wf_k |D^_{b,t,k}|^2 over the FULL spectrum of
// d = x - y, so   dloss/dx_{b,t} = gout * red * ta / (yn_b sqrt(S0_b)) * c2r( wf (.) rfft2(d_{b,t}) )   with wf = w2 without the
// Hermitian multiplicity (c2r sums both halves itself).  Three launches, the time-last layout written in place:
//   k_loss_rows (F = 1)   d = x - y -> half-spectrum rows, as in the forward pass (recomputed: no 94 MB kept across the step)
//   k_loss_cols_bwd       X-point FFT down a column tile, times wf and the sample's coefficient (from the per-time sums the
//                         forward pass left), inverse X-point FFT, back in place
//   k_loss_rows_bwd       per slab (b, x): the half-spectrum rows of two time steps ride one complex inverse transform
//                         (Z = A + i B, Hermitian extension), real parts / imaginary parts = the two gradient rows, the slab
//                         [Y][T] leaves LDS as 16-byte lanes
// Composed from torch ops (round 4) the same step was two permuted copies, an rfft2, five elementwise kernels over the
// spectrum, its adjoint and a subtraction: 0.9 ms at config 5; this is 0.2 ms.
template <typename T, int X, int EPT, int C>
__global__ __launch_bounds__(C*(X / EPT)) void k_loss_cols_bwd(cx<T>* __restrict__ planes, const T* __restrict__ wf,
                                                               const double* __restrict__ sums, const T* __restrict__ gout,
                                                               const cx<T>* __restrict__ tw, int m, int ldk, int ntiles, long batch,
                                                               int nt, int F, int relative, int mesh_weighted, int time_average,
                                                               int reduction) {
    typedef cx<T> cf;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    constexpr int G = X / EPT;
    const int c = threadIdx.x % C, j = threadIdx.x / C;
    const int tile = blockIdx.x % ntiles;
    const size_t img = blockIdx.x / ntiles;                   // (b, t)
    const int q = tile * C + c;
    const bool valid = q < m;
    cf z[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        z[e] = valid ? planes[(img * X + j + e * G) * (size_t)ldk + q] : mk<T>((T)0, (T)0);
    T wv[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) wv[e] = valid ? wf[(size_t)(j + e * G) * m + q] : (T)0;
    // the sample's coefficient, by the rules of k_loss_finish
    const long b = (long)(img / nt);
    double s0 = 0.0, s1 = 0.0;
    for (int t = 0; t < nt; ++t) s0 += sums[(size_t)b * nt + t];
    if (F == 2)
        for (int t = 0; t < nt; ++t) s1 += sums[((size_t)batch + b) * nt + t];
    double yn = (relative && F == 2) ? sqrt(s1) : 1.0;
    if (mesh_weighted) {
        if (mesh_weighted == 2 && !(relative && F == 2)) yn = (double)(1.0f / (float)X);
        else yn /= (double)X;
    }
    double coef = (double)gout[0] / (yn * sqrt(s0));
    if (time_average) coef /= sqrt((double)nt);
    if (reduction) coef /= (double)batch;
    if (mesh_weighted) coef /= (double)X;
    tile_fft<T, X, EPT, -1, C, false, true>(z, lds, tw, j, c);
#pragma unroll
    for (int e = 0; e < EPT; ++e) z[e] = cscale(z[e], (T)((double)wv[e] * coef));
    tile_fft<T, X, EPT, +1, C, false, true>(z, lds, tw, j, c);
    if (valid) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) planes[(img * X + j + e * G) * (size_t)ldk + q] = z[e];
    }
}

template <typename T, int Y, int EPT>
__global__ __launch_bounds__(1024) void k_loss_rows_bwd(const cx<T>* __restrict__ in, T* __restrict__ grad,
                                                        const cx<T>* __restrict__ tw, int nt, int P, int NS, long slabs, int X,
                                                        int ldk, unsigned per) {
    typedef cx<T> cf;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int G = Y / EPT;
    constexpr int NV = 16 / (int)sizeof(T);
    const int tr = threadIdx.x / G, j = threadIdx.x % G;
    const int s = tr / P, p = tr - s * P;
    const long base = (long)blockIdx.x * NS;
    const int count = (int)(slabs - base < NS ? slabs - base : NS);
    const size_t slab_elems = (size_t)Y * nt;
    const int t0 = 2 * p, t1 = 2 * p + 1;
    const bool live = s < count && t0 < nt;
    const bool two = t1 < nt;
    cf* lds = reinterpret_cast<cf*>(smem_raw + (size_t)s * per) + (size_t)p * Y;
    if (live) {
        const long slab = base + s;
        const long b = slab / X, i = slab - b * X;
        const cf* r0 = in + (((size_t)b * nt + t0) * X + i) * ldk;
        const cf* r1 = in + (((size_t)b * nt + t1) * X + i) * ldk;
        for (int k = j; k <= Y / 2; k += G) {
            cf A = r0[k], B = two ? r1[k] : mk<T>((T)0, (T)0);
            if (k == 0 || k == Y / 2) A.y = B.y = (T)0;       // a c2r transform ignores them
            lds[k] = mk<T>(A.x - B.y, A.y + B.x);             // Z[k] = A[k] + i B[k]
            if (k > 0 && k < Y / 2) lds[Y - k] = mk<T>(A.x + B.y, B.x - A.y);   // Z[-k] = conj A[k] + i conj B[k]
        }
    }
    group_sync<0>();
    cf z[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) z[e] = live ? lds[j + e * G] : mk<T>((T)0, (T)0);
    group_sync<0>();
    tile_fft<T, Y, EPT, +1, 1, true, false>(z, lds, tw, j, 0);
    __syncthreads();      // every transform of the workgroup is done: the exchange buffers become the slabs [Y][nt]
    if (live) {
        T* reg = reinterpret_cast<T*>(smem_raw + (size_t)s * per);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const size_t o = (size_t)(j + e * G) * nt;
            reg[o + t0] = z[e].x;
            if (two) reg[o + t1] = z[e].y;
        }
    }
    __syncthreads();
    const int n4 = (int)(slab_elems / NV);
    b128* g4 = reinterpret_cast<b128*>(grad + (size_t)base * slab_elems);
    for (int idx = threadIdx.x; idx < count * n4; idx += blockDim.x) {
        const int qs = idx / n4, i4 = idx - qs * n4;
        g4[idx] = reinterpret_cast<const b128*>(smem_raw + (size_t)qs * per)[i4];
    }
}

// ------------------------------------------------------------------ host side
static bool loss_n_ok(int n) {
    if (n >= 16 && n <= 1024 && (n & (n - 1)) == 0) return true;
    return n == 96 || n == 192 || n == 384 || n == 768 || n == 80 || n == 160 || n == 320 || n == 640;
}
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static int loss_ldk(const tcfd_loss_plan* p) {   // row pitch of the half-spectrum planes: n/2 + 1 rounded up to 128 bytes
    const int per_line = p->dtype == TCFD_C128 ? 8 : 16;
    return ((p->n / 2 + 1) + per_line - 1) / per_line * per_line;
}
static int loss_ntiles(const tcfd_loss_plan* p) {
    const int C = p->dtype == TCFD_C128 ? 8 : 16;
    return (p->n / 2 + 1 + C - 1) / C;
}

extern "C" int tcfd_loss_plan_create(tcfd_loss_plan** out, int n, int dtype) {
    if (!out) return FAIL(TCFD_EINVAL, "loss_plan_create: null argument");
    if (dtype != TCFD_C64 && dtype != TCFD_C128) return FAIL(TCFD_EINVAL, "loss_plan_create: bad dtype %d", dtype);
    if (!loss_n_ok(n)) return FAIL(TCFD_EINVAL, "loss_plan_create: n = %d is not a grid of the fused kernels (2^k in [16, 1024], 3 * 2^k in [96, 768], 5 * 2^k in [80, 640])", n);
    tcfd_loss_plan* p = new tcfd_loss_plan();
    p->n = n; p->dtype = dtype; p->tw = nullptr;
    const long double PI2 = 2.0L * 3.141592653589793238462643383279502884L;
    hipError_t e;
    if (dtype == TCFD_C128) {
        std::vector<cx<double>> w(n);
        for (int t = 0; t < n; ++t) { w[t].x = (double)cosl(-PI2 * t / n); w[t].y = (double)sinl(-PI2 * t / n); }
        e = hipMalloc(&p->tw, n * sizeof(cx<double>));
        if (e == hipSuccess) e = hipMemcpy(p->tw, w.data(), n * sizeof(cx<double>), hipMemcpyHostToDevice);
    } else {
        std::vector<cx<float>> w(n);
        for (int t = 0; t < n; ++t) { w[t].x = (float)cosl(-PI2 * t / n); w[t].y = (float)sinl(-PI2 * t / n); }
        e = hipMalloc(&p->tw, n * sizeof(cx<float>));
        if (e == hipSuccess) e = hipMemcpy(p->tw, w.data(), n * sizeof(cx<float>), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (p->tw) (void)hipFree(p->tw);
        delete p;
        return FAIL(TCFD_EHIP, "loss_plan_create: %s", hipGetErrorString(e));
    }
    *out = p;
    return 0;
}

extern "C" void tcfd_loss_plan_destroy(tcfd_loss_plan* p) {
    if (!p) return;
    if (p->tw) (void)hipFree(p->tw);
    delete p;
}

extern "C" size_t tcfd_loss_workspace_bytes(const tcfd_loss_plan* p, long batch, int nt, int nfields) {
    if (!p || batch <= 0 || nt <= 0 || nfields < 1 || nfields > 2) return 0;
    const size_t cs = p->dtype == TCFD_C128 ? 16 : 8;
    const size_t planes = al256((size_t)nfields * batch * nt * p->n * loss_ldk(p) * cs);
    const size_t partial = al256((size_t)nfields * batch * nt * (loss_ntiles(p) + 1) * sizeof(double));   // tiles + the per-time sums
    return planes + partial;
}

// slabs per workgroup / bytes per slab of pass 1; 0 when the shape does not fit one workgroup (caller: other path)
template <typename T, int N>
static bool rows_geometry(int nt, int F, int* P_, int* NS_, unsigned* per_, size_t* lds_) {
    constexpr int EPT = LossCfg<T, N>::ROW_EPT, G = N / EPT;
    const int P = (F * nt + 1) / 2;
    if ((long)P * G > 1024) return false;
    size_t per = std::max((size_t)F * N * nt * sizeof(T), (size_t)P * N * sizeof(cx<T>));
    per = (per + 15) & ~(size_t)15;
    if (per > 160 * 1024) return false;
    int NS = std::min<long>(std::min<long>(4, 1024 / (P * G)), (long)((80 * 1024) / per));   // two workgroups per CU when possible
    if (NS < 1) NS = 1;
    *P_ = P; *NS_ = NS; *per_ = (unsigned)per; *lds_ = per * NS;
    return true;
}

// the dynamic-LDS attribute of a kernel is set once per (kernel instantiation, device)
template <typename K>
static int raise_lds(K kernel, size_t bytes) {
    static std::atomic<unsigned long long> done{0};     // one per instantiation of this template = per kernel
    if (bytes <= 48 * 1024) return 0;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

template <typename T, int N>
static int loss_impl(const tcfd_loss_plan* p, const void* x, const void* y, const void* w2, long batch, int nt, int F,
                     int relative, int mesh_weighted, int time_average, int reduction, void* out, void* sums, void* ws,
                     hipStream_t st) {
    typedef cx<T> cf;
    constexpr int REPT = LossCfg<T, N>::ROW_EPT, CEPT = LossCfg<T, N>::COL_EPT, C = LossCfg<T, N>::COLS;
    const int m = N / 2 + 1, ldk = loss_ldk(p), ntiles = loss_ntiles(p);
    int P, NS;
    unsigned per;
    size_t lds1;
    if (!rows_geometry<T, N>(nt, F, &P, &NS, &per, &lds1))
        return FAIL(TCFD_EINVAL, "sobolev_loss: %d time steps of a %d-point row do not fit one workgroup", nt, N);
    cf* planes = (cf*)ws;
    double* partial = (double*)((unsigned char*)ws + al256((size_t)F * batch * nt * N * ldk * sizeof(cf)));
    const long slabs = batch * N;
    int rc;
    {
        auto kern = k_loss_rows<T, N, REPT>;
        if ((rc = raise_lds(kern, lds1))) return rc;
        const int threads = NS * P * (N / REPT);
        hipLaunchKernelGGL(kern, dim3((unsigned)((slabs + NS - 1) / NS)), dim3(threads), lds1, st, (const T*)x, (const T*)y, planes,
                           (const cf*)p->tw, nt, F, P, NS, slabs, N, batch, ldk, per);
        HIP_TRY(hipGetLastError());
    }
    {
        auto kern = k_loss_cols<T, N, CEPT, C>;
        const size_t lds2 = std::max((size_t)N * C * sizeof(cf), (size_t)64 * sizeof(double));
        if ((rc = raise_lds(kern, lds2))) return rc;
        const long blocks = (long)F * batch * nt * ntiles;
        if (blocks >= 2147483647L) return FAIL(TCFD_EINVAL, "sobolev_loss: too many column tiles");
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C * (N / CEPT)), lds2, st, (const cf*)planes, (const T*)w2, partial,
                           (const cf*)p->tw, m, ldk, ntiles);
        HIP_TRY(hipGetLastError());
    }
    double* tsum = sums ? (double*)sums : partial + (size_t)F * batch * nt * ntiles;     // (F, batch, nt) per-time sums
    hipLaunchKernelGGL(k_loss_finish<T>, dim3(1), dim3(256), 0, st, (const double*)partial, (T*)out, tsum, batch, nt,
                       ntiles, F, N, relative, mesh_weighted, time_average, reduction);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T>
static int loss_dispatch(const tcfd_loss_plan* p, const void* x, const void* y, const void* w2, long batch, int nt, int F,
                         int relative, int mesh_weighted, int time_average, int reduction, void* out, void* sums, void* ws,
                         hipStream_t st) {
#define TCFD_LOSS_CASE(N_)                                                                                                  \
    case N_:                                                                                                                \
        return loss_impl<T, N_>(p, x, y, w2, batch, nt, F, relative, mesh_weighted, time_average, reduction, out, sums, ws, st);
    switch (p->n) {
        TCFD_LOSS_CASE(16) TCFD_LOSS_CASE(32) TCFD_LOSS_CASE(64) TCFD_LOSS_CASE(128) TCFD_LOSS_CASE(256) TCFD_LOSS_CASE(512)
        TCFD_LOSS_CASE(1024) TCFD_LOSS_CASE(96) TCFD_LOSS_CASE(192) TCFD_LOSS_CASE(384) TCFD_LOSS_CASE(768) TCFD_LOSS_CASE(80)
        TCFD_LOSS_CASE(160) TCFD_LOSS_CASE(320) TCFD_LOSS_CASE(640)
    }
#undef TCFD_LOSS_CASE
    return FAIL(TCFD_EINVAL, "sobolev_loss: unsupported n = %d", p->n);
}

template <typename T, int N>
static int loss_bwd_impl(const tcfd_loss_plan* p, const void* x, const void* y, const void* wf, const void* sums, const void* gout,
                         long batch, int nt, int F, int relative, int mesh_weighted, int time_average, int reduction, void* grad,
                         void* ws, hipStream_t st) {
    typedef cx<T> cf;
    constexpr int REPT = LossCfg<T, N>::ROW_EPT, CEPT = LossCfg<T, N>::COL_EPT, C = LossCfg<T, N>::COLS;
    const int m = N / 2 + 1, ldk = loss_ldk(p), ntiles = loss_ntiles(p);
    int P, NS;
    unsigned per;
    size_t lds1;
    if (!rows_geometry<T, N>(nt, 1, &P, &NS, &per, &lds1))
        return FAIL(TCFD_EINVAL, "sobolev_loss_backward: %d time steps of a %d-point row do not fit one workgroup", nt, N);
    cf* planes = (cf*)ws;
    const long slabs = batch * N;
    const int threads = NS * P * (N / REPT);
    const unsigned blocks1 = (unsigned)((slabs + NS - 1) / NS);
    int rc;
    {
        auto kern = k_loss_rows<T, N, REPT>;
        if ((rc = raise_lds(kern, lds1))) return rc;
        hipLaunchKernelGGL(kern, dim3(blocks1), dim3(threads), lds1, st, (const T*)x, (const T*)y, planes, (const cf*)p->tw, nt, 1, P,
                           NS, slabs, N, batch, ldk, per);
        HIP_TRY(hipGetLastError());
    }
    {
        auto kern = k_loss_cols_bwd<T, N, CEPT, C>;
        const size_t lds2 = (size_t)N * C * sizeof(cf);
        if ((rc = raise_lds(kern, lds2))) return rc;
        const long blocks = batch * nt * ntiles;
        if (blocks >= 2147483647L) return FAIL(TCFD_EINVAL, "sobolev_loss_backward: too many column tiles");
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C * (N / CEPT)), lds2, st, planes, (const T*)wf, (const double*)sums,
                           (const T*)gout, (const cf*)p->tw, m, ldk, ntiles, batch, nt, F, relative, mesh_weighted, time_average,
                           reduction);
        HIP_TRY(hipGetLastError());
    }
    {
        auto kern = k_loss_rows_bwd<T, N, REPT>;
        if ((rc = raise_lds(kern, lds1))) return rc;
        hipLaunchKernelGGL(kern, dim3(blocks1), dim3(threads), lds1, st, (const cf*)planes, (T*)grad, (const cf*)p->tw, nt, P, NS,
                           slabs, N, ldk, per);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

template <typename T>
static int loss_bwd_dispatch(const tcfd_loss_plan* p, const void* x, const void* y, const void* wf, const void* sums,
                             const void* gout, long batch, int nt, int F, int relative, int mesh_weighted, int time_average,
                             int reduction, void* grad, void* ws, hipStream_t st) {
#define TCFD_LOSS_CASE(N_)                                                                                                  \
    case N_:                                                                                                                \
        return loss_bwd_impl<T, N_>(p, x, y, wf, sums, gout, batch, nt, F, relative, mesh_weighted, time_average, reduction, grad, ws, st);
    switch (p->n) {
        TCFD_LOSS_CASE(16) TCFD_LOSS_CASE(32) TCFD_LOSS_CASE(64) TCFD_LOSS_CASE(128) TCFD_LOSS_CASE(256) TCFD_LOSS_CASE(512)
        TCFD_LOSS_CASE(1024) TCFD_LOSS_CASE(96) TCFD_LOSS_CASE(192) TCFD_LOSS_CASE(384) TCFD_LOSS_CASE(768) TCFD_LOSS_CASE(80)
        TCFD_LOSS_CASE(160) TCFD_LOSS_CASE(320) TCFD_LOSS_CASE(640)
    }
#undef TCFD_LOSS_CASE
    return FAIL(TCFD_EINVAL, "sobolev_loss_backward: unsupported n = %d", p->n);
}

extern "C" int tcfd_sobolev_loss_backward(const tcfd_loss_plan* p, const void* x, const void* y, const void* wf, const void* sums,
                                          const void* gout, long batch, int nt, int nfields, int relative, int mesh_weighted,
                                          int time_average, int reduction, void* grad, void* ws, size_t ws_bytes, void* stream) {
    if (!p || !x || !wf || !sums || !gout || !grad || !ws) return FAIL(TCFD_EINVAL, "sobolev_loss_backward: null argument");
    if (batch <= 0 || nt <= 0 || nfields < 1 || nfields > 2) return FAIL(TCFD_EINVAL, "sobolev_loss_backward: bad sizes");
    if ((p->n * nt * (p->dtype == TCFD_C128 ? 8 : 4)) % 16)
        return FAIL(TCFD_EINVAL, "sobolev_loss_backward: a slab must be a multiple of 16 bytes");
    const size_t need = tcfd_loss_workspace_bytes(p, batch, nt, 1);
    if (ws_bytes < need) return FAIL(TCFD_EWORKSPACE, "workspace %zu B < required %zu B", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    if (p->dtype == TCFD_C128)
        return loss_bwd_dispatch<double>(p, x, y, wf, sums, gout, batch, nt, nfields, relative, mesh_weighted, time_average,
                                         reduction, grad, ws, st);
    return loss_bwd_dispatch<float>(p, x, y, wf, sums, gout, batch, nt, nfields, relative, mesh_weighted, time_average, reduction,
                                    grad, ws, st);
}

extern "C" int tcfd_sobolev_loss_supported(const tcfd_loss_plan* p, int nt, int nfields) {
    if (!p || nt < 1 || nfields < 1 || nfields > 2) return 0;
    int P, NS;
    unsigned per;
    size_t lds;
    const int n = p->n;
    bool ok = false;
#define TCFD_LOSS_GEO(N_)                                                                                     \
    case N_:                                                                                                  \
        ok = p->dtype == TCFD_C128 ? rows_geometry<double, N_>(nt, nfields, &P, &NS, &per, &lds)              \
                                   : rows_geometry<float, N_>(nt, nfields, &P, &NS, &per, &lds);              \
        break;
    switch (n) {
        TCFD_LOSS_GEO(16) TCFD_LOSS_GEO(32) TCFD_LOSS_GEO(64) TCFD_LOSS_GEO(128) TCFD_LOSS_GEO(256) TCFD_LOSS_GEO(512)
        TCFD_LOSS_GEO(1024) TCFD_LOSS_GEO(96) TCFD_LOSS_GEO(192) TCFD_LOSS_GEO(384) TCFD_LOSS_GEO(768) TCFD_LOSS_GEO(80)
        TCFD_LOSS_GEO(160) TCFD_LOSS_GEO(320) TCFD_LOSS_GEO(640)
    }
#undef TCFD_LOSS_GEO
    return ok ? 1 : 0;
}

extern "C" int tcfd_sobolev_loss(const tcfd_loss_plan* p, const void* x, const void* y, const void* w2, long batch, int nt,
                                 int nfields, int relative, int mesh_weighted, int time_average, int reduction, void* out,
                                 void* sums, void* ws, size_t ws_bytes, void* stream) {
    if (!p || !x || !w2 || !out || !ws) return FAIL(TCFD_EINVAL, "sobolev_loss: null argument");
    if (batch <= 0 || nt <= 0 || nfields < 1 || nfields > 2) return FAIL(TCFD_EINVAL, "sobolev_loss: bad sizes");
    if (nfields == 2 && !y) return FAIL(TCFD_EINVAL, "sobolev_loss: two fields need y");
    if ((p->n * nt * (p->dtype == TCFD_C128 ? 8 : 4)) % 16) return FAIL(TCFD_EINVAL, "sobolev_loss: a slab must be a multiple of 16 bytes");
    const size_t need = tcfd_loss_workspace_bytes(p, batch, nt, nfields);
    if (ws_bytes < need) return FAIL(TCFD_EWORKSPACE, "workspace %zu B < required %zu B", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    if (p->dtype == TCFD_C128)
        return loss_dispatch<double>(p, x, y, w2, batch, nt, nfields, relative, mesh_weighted, time_average, reduction, out, sums, ws, st);
    return loss_dispatch<float>(p, x, y, w2, batch, nt, nfields, relative, mesh_weighted, time_average, reduction, out, sums, ws, st);
}
