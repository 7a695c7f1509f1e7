// tcfd_fno.hip -- MI355X (gfx950) kernels + C ABI for the FNO / SFNO spectral convolution
//   y = irfftn( W (.) rfftn(v)[kept modes] , s = out size )
// reference: fno/base.py:229-237 (forward), fno/sfno.py:364-391 (4-corner contraction + bias),
// fno/sfno.py:433-457 (time padding / resampling variant), fno/fno3d.py:86-116 (SpectralConv3d).
//
// The reference transforms the FULL (b, C, X, Y, T/2+1) spectrum both ways and zero-fills a full
// output spectrum although only 2mx x 2my x mt modes are ever read or written.  Here the transforms
// are pruned: five kernels, the big activation tensors are read / written exactly once,
//
//   k_fwd_ty2  per (b,c,x) slab [Y][T]: Y-point FFTs on pairs of time samples, then the short real DFT in t on
//              the 2my kept ky only                                -> W1 (b,C,X,Q)   Q = 2my*mt
//   k_fwd_x    X-point FFT down the columns of W1, store only the 2mx kept kx -> V (b,C,2mx,Q)
//   k_contract per-mode (b x Ci)(Ci x Co) complex products on MFMA (f32 16x16x4), + delta*bias
//   k_inv_x    zero-padded X-point inverse FFT                     -> W2 (b,C,X,Q)
//   k_inv_ty2  c2r step in t on the kept ky, then zero-padded Y-point inverse FFTs (two output steps per transform)
//                                                                  -> (b,C,X,Y,T_keep)
//
// fp32 and fp64 (the kernels are templates on the real type; SpectralConv3d itself is cfloat-only in the reference,
// SURVEY a16, FNOBase.double() makes SpectralConvS / T / SFNO run in float64, fno/base.py:342-349).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/tcfd.h"
#include "tcfd_fft.hpp"

using namespace tcfd;
typedef cx<float> cf;

extern "C" const char* tcfd_last_error(void);
int tcfd_set_error(int code, const char* fmt, ...);  // defined in tcfd_ns2d.hip
#define FAIL(...) tcfd_set_error(__VA_ARGS__)
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return FAIL(TCFD_EHIP, "%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

// ------------------------------------------------------------------ plan
struct tcfd_fno_plan {
    int X, Y, T_in, t_pad, T_out, mx, my, mt;
    int Xs, Ys;    // grid the truncated spectrum was TAKEN from (inverse transforms only; = X, Y unless the layer resamples):
                   // torch's irfftn(s = other size) pads / trims the spectrum ARRAY at its end, so the high block keeps
                   // its array indices [Xs - mx, Xs) / [Ys - my, Ys) in a transform of length X / Y (fno/base.py:229-237)
    int Tp;        // padded input length  T_in + t_pad (the rfft length in t)
    int dtype;     // TCFD_C64: fp32 data / complex64 spectra; TCFD_C128: fp64 / complex128 (FNOBase.double(), fno/base.py:342-349)
    void* tw_x;    // [X]   exp(-2 pi i k / X)            (complex of the plan's precision, as all four tables)
    void* tw_y;    // [Y]
    void* tw_tf;   // [mt][Tp]   forward:  exp(-2 pi i kt t / Tp)
    void* tw_ti;   // [T_out][mt] inverse: c_kt * exp(+2 pi i kt t / T_out), c = 1 (kt = 0 or Nyquist) else 2
};
static size_t csize(const tcfd_fno_plan* p) { return p->dtype == TCFD_C128 ? 16 : 8; }

// lengths of the FFT kernels: 2^k in [8, 1024], 3 * 2^k in [96, 768], 5 * 2^k in [80, 640]; every other length in [4, 1024] runs the
// pruned direct-DFT kernels (tcfd_fno_dft.hpp)
static bool fft_len(int n) {
    if (n >= 8 && n <= 1024 && (n & (n - 1)) == 0) return true;
    return n == 96 || n == 192 || n == 384 || n == 768 || n == 80 || n == 160 || n == 320 || n == 640;
}

template <typename V>
static int upload_vec(void** dst, const std::vector<V>& h) {
    HIP_TRY(hipMalloc(dst, h.size() * sizeof(V)));
    HIP_TRY(hipMemcpy(*dst, h.data(), h.size() * sizeof(V), hipMemcpyHostToDevice));
    return 0;
}

template <typename T>
static std::vector<cx<T>> unit_roots(int n) {
    std::vector<cx<T>> w(n);
    for (int t = 0; t < n; ++t) {
        const long double a = -2.0L * 3.141592653589793238462643383279502884L * t / n;
        w[t].x = (T)cosl(a);
        w[t].y = (T)sinl(a);
    }
    return w;
}

extern "C" void tcfd_fno_plan_destroy(tcfd_fno_plan* p) {
    if (!p) return;
    void* ptrs[] = {p->tw_x, p->tw_y, p->tw_tf, p->tw_ti};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    delete p;
}

extern "C" int tcfd_fno_plan_create(tcfd_fno_plan** out, int X, int Y, int T_in, int t_pad, int T_out, int mx, int my,
                                    int mt) {
    return tcfd_fno_plan_create_resample(out, X, Y, T_in, t_pad, T_out, mx, my, mt, X, Y);
}

template <typename T>
static int fill_tables(tcfd_fno_plan* p) {
    const int Tp = p->Tp, mt = p->mt, T_out = p->T_out;
    std::vector<cx<T>> tf((size_t)mt * Tp), ti((size_t)T_out * mt);
    const long double PI2 = 2.0L * 3.141592653589793238462643383279502884L;
    for (int k = 0; k < mt; ++k)
        for (int t = 0; t < Tp; ++t) {
            const long double a = -PI2 * (long double)((long)k * t % Tp) / Tp;
            tf[(size_t)k * Tp + t].x = (T)cosl(a);
            tf[(size_t)k * Tp + t].y = (T)sinl(a);
        }
    for (int t = 0; t < T_out; ++t)
        for (int k = 0; k < mt; ++k) {
            // c2r: x[t] = sum_k c_k Re( X_k e^{+2 pi i k t / T} ), Im(X_0) and Im(X_Nyquist) ignored
            const bool edge = (k == 0) || (2 * k == T_out);
            const long double a = PI2 * (long double)((long)k * t % T_out) / T_out;
            const T c = edge ? (T)1 : (T)2;
            ti[(size_t)t * mt + k].x = c * (T)cosl(a);
            ti[(size_t)t * mt + k].y = edge ? (T)0 : c * (T)sinl(a);
        }
    int rc;
    if ((rc = upload_vec(&p->tw_x, unit_roots<T>(p->X))) || (rc = upload_vec(&p->tw_y, unit_roots<T>(p->Y))) ||
        (rc = upload_vec(&p->tw_tf, tf)) || (rc = upload_vec(&p->tw_ti, ti)))
        return rc;
    return 0;
}

extern "C" int tcfd_fno_plan_create_resample(tcfd_fno_plan** out, int X, int Y, int T_in, int t_pad, int T_out, int mx,
                                             int my, int mt, int Xs, int Ys) {
    return tcfd_fno_plan_create_dtype(out, X, Y, T_in, t_pad, T_out, mx, my, mt, Xs, Ys, TCFD_C64);
}

extern "C" int tcfd_fno_plan_create_dtype(tcfd_fno_plan** out, int X, int Y, int T_in, int t_pad, int T_out, int mx,
                                          int my, int mt, int Xs, int Ys, int dtype) {
    if (!out) return FAIL(TCFD_EINVAL, "fno_plan_create: null argument");
    if (dtype != TCFD_C64 && dtype != TCFD_C128) return FAIL(TCFD_EINVAL, "fno_plan_create: bad dtype %d", dtype);
    if (X < 4 || Y < 4 || X > 1024 || Y > 1024) return FAIL(TCFD_EINVAL, "fno_plan_create: X=%d, Y=%d must lie in [4, 1024]", X, Y);
    if (T_in < 1 || t_pad < 0 || T_out < 1 || mx < 1 || my < 1 || mt < 1)
        return FAIL(TCFD_EINVAL, "fno_plan_create: bad sizes");
    const int Tp = T_in + t_pad;
    const bool resample = (Xs != X || Ys != Y);
    if (2 * mx > Xs || 2 * my > Ys) return FAIL(TCFD_EINVAL, "fno_plan_create: 2*modes exceed the grid (%d,%d vs %d,%d)", mx, my, Xs, Ys);
    if (resample && (Xs < 1 || Ys < 1)) return FAIL(TCFD_EINVAL, "fno_plan_create: bad source grid");
    if (mt > Tp / 2 + 1 || mt > T_out / 2 + 1 || mt > 16)
        return FAIL(TCFD_EINVAL, "fno_plan_create: modes_t=%d exceeds the half spectrum of T=%d / T_out=%d (or 16)", mt, Tp, T_out);
    tcfd_fno_plan* p = new tcfd_fno_plan();
    memset(p, 0, sizeof(*p));
    p->X = X; p->Y = Y; p->T_in = T_in; p->t_pad = t_pad; p->T_out = T_out;
    p->mx = mx; p->my = my; p->mt = mt; p->Tp = Tp;
    p->Xs = Xs; p->Ys = Ys;
    p->dtype = dtype;
    const int rc = dtype == TCFD_C128 ? fill_tables<double>(p) : fill_tables<float>(p);
    if (rc) {
        tcfd_fno_plan_destroy(p);
        return rc;
    }
    *out = p;
    return 0;
}

static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
__host__ __device__ constexpr size_t al16c(size_t x) { return (x + 15) & ~(size_t)15; }

extern "C" size_t tcfd_fno_workspace_bytes(const tcfd_fno_plan* p, int batch, int cin, int cout) {
    if (!p) return 0;
    const size_t Q = (size_t)2 * p->my * p->mt;
    const int cmax = std::max(cin, cout);
    const size_t w = al256((size_t)batch * cmax * p->X * Q * csize(p));        // W1 / W2 (shared)
    const size_t v = al256((size_t)batch * cin * 2 * p->mx * Q * csize(p));    // truncated input spectrum
    const size_t o = al256((size_t)batch * cout * 2 * p->mx * Q * csize(p));   // truncated output spectrum
    return w + v + o;
}

// ------------------------------------------------------------------ t/y transforms, packed form
// The activations are REAL, so the Y-point transforms can run first, two time samples per complex transform
// (z[y] = v[y][2p] + i v[y][2p+1]); the short real DFT in t then only touches the 2my kept rows instead of all Y.
// Against the kernels above this removes the mt-fold redundant LDS sweep of the slab (forward) and the
// Y x t_keep x mt LDS dot products (inverse): both were LDS-bound, not HBM-bound.
//   forward :  Z_p = FFT_y(z_p);  v^[ky][2p] = (Z_p[ky] + conj Z_p[-ky]) / 2,  v^[ky][2p+1] = (Z_p[ky] - conj Z_p[-ky]) / 2i
//              out[ky][kt] = sum_t v^[ky][t] w[kt][t]
//   inverse :  out[y][t] = sum_kt c_k Re(spec[y][kt] E[t][kt])  (c2r semantics, spec = IFFT_y W)
//                        = IFFT_y G[.][t],  G[ky][t] = 1/2 sum_kt (W[ky][kt] E[t][kt] + conj(W[-ky][kt] E[t][kt]))
//              and two output steps ride through one complex IFFT:  H_p = G[.][t0+2p] + i G[.][t0+2p+1].
// A workgroup owns NS consecutive slabs; thread = (transform tr = s*P + p, lane j of its G-lane group), G <= 64 so
// every transform lives inside one wave and the exchange needs no workgroup barrier.
template <int Y, typename T = float>
struct TyCfg2 {
    static constexpr int EPT0 = Y >= 256 ? 16 : (Y >= 64 ? 8 : (Y >= 16 ? 4 : 2));
    // Y = 3 * 2^k / 5 * 2^k (96 ... 768, 80 ... 640): the solver's radix-12 / radix-20 first pass (tcfd_fft.hpp), 8 ... 64 lanes
    static constexpr int EPT = Y % 3 == 0 ? 12 : (Y % 5 == 0 ? 20 :
                               ((sizeof(T) == 8 && EPT0 > 8) ? 8 : EPT0));   // fp64: 16 complex doubles per lane are 64 VGPRs of data alone
    static constexpr int G = Y / EPT;
};
typedef unsigned int b128 __attribute__((ext_vector_type(4)));     // 16 bytes of anything (slab copies)

template <typename T, int Y, int EPT>
__global__ __launch_bounds__(1024) void k_fwd_ty2(const T* __restrict__ v, cx<T>* __restrict__ w1,
                                                  const cx<T>* __restrict__ tw_y, const cx<T>* __restrict__ tw_tf, int T_in,
                                                  int t_pad, int mt, int my, T scale, int P, int NS, long slabs,
                                                  unsigned mt_magic) {
    typedef cx<T> cf;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int G = Y / EPT;
    const int Tp = T_in + t_pad, Q = 2 * my * mt;
    const size_t per = (size_t)Y * T_in * sizeof(T) > (size_t)P * Y * sizeof(cf) ? (size_t)Y * T_in * sizeof(T) : (size_t)P * Y * sizeof(cf);
    unsigned char* slabs_b = smem_raw;                                            // [NS] slab, later [P][Y] spectra
    cf* twt = reinterpret_cast<cf*>(smem_raw + (size_t)NS * per);                  // [mt][Tp]
    const int tr = threadIdx.x / G, j = threadIdx.x % G;
    const int s = tr / P, p = tr - s * P;
    const long base = (long)blockIdx.x * NS;
    const int count = (int)(slabs - base < NS ? slabs - base : NS);
    const size_t slab_elems = (size_t)Y * T_in;
    {
        // Stage the workgroup's slabs (one contiguous range of global memory).  EIGHT 16-byte loads per lane are issued before
        // the first one is used: written as a plain load -> LDS-store loop, hipcc waits for every load before it issues the
        // next (`global_load_dwordx4; s_waitcnt vmcnt(0); ds_write_b128` per trip: one KB in flight per wave, eight memory round
        // trips per workgroup -- the kernel read at 3.9 TB/s where the box's read probe reaches 6.3).
        const b128* s4 = reinterpret_cast<const b128*>(v + (size_t)base * slab_elems);
        const int n4 = (int)(slab_elems * sizeof(T) / 16);
        const int total4 = count * n4;
        const bool contig = per == (size_t)n4 * 16;      // the exchange buffers are no larger than the slab: LDS is one range too
        constexpr int UN = 8;
        for (int i0 = threadIdx.x; i0 < total4; i0 += UN * blockDim.x) {
            b128 r[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int idx = i0 + u * blockDim.x;
                if (idx < total4) r[u] = __builtin_nontemporal_load(s4 + idx);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int idx = i0 + u * blockDim.x;
                if (idx < total4) {
                    if (contig) {
                        reinterpret_cast<b128*>(slabs_b)[idx] = r[u];
                    } else {
                        const int q = idx / n4;
                        reinterpret_cast<b128*>(slabs_b + (size_t)q * per)[idx - q * n4] = r[u];
                    }
                }
            }
        }
        for (int i = threadIdx.x; i < mt * Tp; i += blockDim.x) twt[i] = tw_tf[i];
    }
    __syncthreads();
    cf x[EPT];
    {   // slabs past the end of the batch (last workgroup only) transform stale LDS; nothing of theirs is stored
        const T* sl = reinterpret_cast<const T*>(slabs_b + (size_t)s * per) + (size_t)j * T_in + 2 * p;
        if ((T_in & 1) == 0) {
#pragma unroll
            for (int t = 0; t < EPT; ++t) x[t] = *reinterpret_cast<const cf*>(sl + (size_t)t * G * T_in);
        } else {
            const bool pair = 2 * p + 1 < T_in;
#pragma unroll
            for (int t = 0; t < EPT; ++t) {
                const T* r = sl + (size_t)t * G * T_in;
                x[t] = mk<T>(r[0], pair ? r[1] : (T)0);
            }
        }
    }
    __syncthreads();  // every transform of the slab has its input: the slab bytes become the exchange buffers
    cf* lds = reinterpret_cast<cf*>(slabs_b + (size_t)s * per) + (size_t)p * Y;
    tile_fft<T, Y, EPT, -1, 1, true, false>(x, lds, tw_y, j, 0);
#pragma unroll
    for (int t = 0; t < EPT; ++t) lds[j + t * G] = x[t];   // Z_p in natural order, read by the whole slab below
    __syncthreads();
    // v^[-ky][t] = conj v^[ky][t] (real input): one task (ky in [0, my], kt) produces out[ky][kt] and out[-ky][kt]
    // from four real sums.  The slab's P*G lanes stride over the (my+1)*mt tasks.
    if (s < count) {
        const T hsc = (T)0.5 * scale;
        cf* dst = w1 + (size_t)(base + s) * Q;
        const cf* zbase = reinterpret_cast<const cf*>(slabs_b + (size_t)s * per);
        const int ntask = (my + 1) * mt;
        for (int task = p * G + j; task < ntask; task += P * G) {
            const int ky = mt == 1 ? task : (int)__umulhi((unsigned)task, mt_magic);
            const int kt = task - ky * mt;
            const int kyn = ky ? Y - ky : 0;
            const cf* w = twt + (size_t)kt * Tp + t_pad;
            T sce = 0, sdf = 0, scf = 0, sde = 0;
            for (int pp = 0; pp < P; ++pp) {
                const cf za = zbase[(size_t)pp * Y + ky], zb = zbase[(size_t)pp * Y + kyn];
                // 2 v^[ky][2pp] = za + conj zb ;  2 v^[ky][2pp+1] = -i (za - conj zb)
                const T c0 = za.x + zb.x, d0 = za.y - zb.y;
                const T c1 = za.y + zb.y, d1 = zb.x - za.x;
                const cf w0 = w[2 * pp];
                sce += c0 * w0.x; sdf += d0 * w0.y; scf += c0 * w0.y; sde += d0 * w0.x;
                if (2 * pp + 1 < T_in) {
                    const cf w1v = w[2 * pp + 1];
                    sce += c1 * w1v.x; sdf += d1 * w1v.y; scf += c1 * w1v.y; sde += d1 * w1v.x;
                }
            }
            if (ky < my) dst[(size_t)ky * mt + kt] = mk<T>((sce - sdf) * hsc, (scf + sde) * hsc);
            if (ky >= 1) dst[(size_t)(2 * my - ky) * mt + kt] = mk<T>((sce + sdf) * hsc, (scf - sde) * hsc);
        }
    }
}

template <typename T, int Y, int EPT>
__global__ __launch_bounds__(1024) void k_inv_ty2(const cx<T>* __restrict__ w2, T* out /* may be == acc: no restrict */,
                                                  const cx<T>* __restrict__ tw_y, const cx<T>* __restrict__ tw_ti, int T_out,
                                                  int t_keep, int mt, int my, T scale, int P, int NS, long slabs, int Ys,
                                                  const T* acc, const T* __restrict__ accb, int accT) {
    typedef cx<T> cf;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int G = Y / EPT;
    const int Q = 2 * my * mt, t0 = T_out - t_keep;
    cf* ex = reinterpret_cast<cf*>(smem_raw);                  // [NS*P][Y] input / exchange, later [NS][Y][t_keep] floats
    cf* win = ex + (size_t)NS * P * Y;                         // [NS][Q]
    cf* twt = win + (size_t)NS * Q;                            // [t_keep][mt]
    const int tr = threadIdx.x / G, j = threadIdx.x % G;
    const int s = tr / P, p = tr - s * P;
    const long base = (long)blockIdx.x * NS;
    const int count = (int)(slabs - base < NS ? slabs - base : NS);
    cf* lds = ex + (size_t)tr * Y;
    {
        const cf* src = w2 + (size_t)base * Q;
        for (int i = threadIdx.x; i < count * Q; i += blockDim.x) win[i] = src[i];
        for (int i = threadIdx.x; i < t_keep * mt; i += blockDim.x) twt[i] = tw_ti[(size_t)t0 * mt + i];
#pragma unroll
        for (int t = 0; t < EPT; ++t) lds[j + t * G] = mk<T>((T)0, (T)0);   // zero this transform's spectrum (the padding)
    }
    __syncthreads();
    // G[-ky][t] = conj G[ky][t]: the transform's own lanes stride over ky in [0, my] and fill H_p[ky], H_p[-ky].
    // With a = W[ky][k], b = W[-ky][k], u = a + b, d = a - b:  a E + conj(b E) = (E.x u.x - E.y u.y) + i (E.y d.x + E.x d.y)
    if (s < count) {
        const T hsc = (T)0.5 * scale;
        const bool pair = 2 * p + 1 < t_keep;
        const cf* e0 = twt + (size_t)(2 * p) * mt;
        const cf* e1 = twt + (size_t)(pair ? 2 * p + 1 : 2 * p) * mt;
        const cf* wq = win + (size_t)s * Q;
        // kept array indices of the length-Y spectrum: [0, my) and [Ys - my, Ys) (Ys = Y unless the layer resamples, see
        // tcfd_fno_plan); slot() = row of the truncated spectrum, or -1.  Without resampling only ky in [0, my] and
        // their mirrors carry data; with it any pair (ky, Y - ky) may.
        auto slot = [&](int k) { return k < my ? k : ((k >= Ys - my && k < Ys) ? k - (Ys - 2 * my) : -1); };
        const int kmax = (Ys == Y) ? my : Y / 2;
        for (int ky = j; ky <= kmax; ky += G) {
            const int kyn = ky ? Y - ky : 0;
            const int sa = slot(ky), sb = slot(kyn);
            const bool ha = sa >= 0, hb = sb >= 0;
            if (!ha && !hb) continue;      // the buffer is pre-zeroed
            const cf* wa = wq + (size_t)(ha ? sa : 0) * mt;
            const cf* wb = wq + (size_t)(hb ? sb : 0) * mt;
            T g0x = 0, g0y = 0, g1x = 0, g1y = 0;
            for (int k = 0; k < mt; ++k) {
                cf a = wa[k];
                if (!ha) a = mk<T>((T)0, (T)0);
                cf b = wb[k];
                if (!hb) b = mk<T>((T)0, (T)0);
                const T ux = a.x + b.x, uy = a.y + b.y, dx = a.x - b.x, dy = a.y - b.y;
                const cf E0 = e0[k], E1 = e1[k];
                g0x += E0.x * ux - E0.y * uy;  g0y += E0.y * dx + E0.x * dy;
                g1x += E1.x * ux - E1.y * uy;  g1y += E1.y * dx + E1.x * dy;
            }
            if (!pair) { g1x = 0; g1y = 0; }
            // H[ky] = G0 + i G1 ;  H[-ky] = conj G0 + i conj G1
            lds[ky] = mk<T>((g0x - g1y) * hsc, (g0y + g1x) * hsc);
            if (kyn != ky) lds[kyn] = mk<T>((g0x + g1y) * hsc, (g1x - g0y) * hsc);
        }
    }
    group_sync<false>();
    cf x[EPT];
#pragma unroll
    for (int t = 0; t < EPT; ++t) x[t] = lds[j + t * G];
    group_sync<false>();
    tile_fft<T, Y, EPT, +1, 1, true, false>(x, lds, tw_y, j, 0);
    __syncthreads();  // all exchanges done: the buffers become the output slabs [y][t_keep]
    T* oslab = reinterpret_cast<T*>(ex + (size_t)s * P * Y) + (size_t)j * t_keep + 2 * p;
    if ((t_keep & 1) == 0) {
#pragma unroll
        for (int t = 0; t < EPT; ++t) *reinterpret_cast<cf*>(oslab + (size_t)t * G * t_keep) = x[t];
    } else {
        const bool pair = 2 * p + 1 < t_keep;
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            T* o = oslab + (size_t)t * G * t_keep;
            o[0] = x[t].x;
            if (pair) o[1] = x[t].y;
        }
    }
    __syncthreads();
    const int n4 = (int)((size_t)Y * t_keep * sizeof(T) / 16);
    b128* d4 = reinterpret_cast<b128*>(out + (size_t)base * Y * t_keep);
    if (acc) {   // out = acc + transform (acc may BE out): a gradient that joins another one, e.g. the skip path's (training)
        const b128* a4 = reinterpret_cast<const b128*>(acc + (size_t)base * Y * t_keep);
        constexpr int NV = 16 / (int)sizeof(T);
        for (int q = 0; q < count; ++q) {
            const b128* s4 = reinterpret_cast<const b128*>(ex + (size_t)q * P * Y);
            for (int i = threadIdx.x; i < n4; i += blockDim.x) {
                union { b128 v; T e[NV]; } s_, a_;
                s_.v = s4[i];
                a_.v = __builtin_nontemporal_load(a4 + (size_t)q * n4 + i);
#pragma unroll
                for (int u = 0; u < NV; ++u) s_.e[u] += a_.e[u];
                __builtin_nontemporal_store(s_.v, d4 + (size_t)q * n4 + i);
            }
        }
        return;
    }
    if (accb) {   // out[slab][y][t] = transform + accb[slab][y][accT - 1]: the residual frame of the output operator
                  // (v_res[..., -1:] + conv(...), fno/sfno.py:327) added by the store loop instead of a pass of its own
        constexpr int NV = 16 / (int)sizeof(T);
        for (int q = 0; q < count; ++q) {
            const b128* s4 = reinterpret_cast<const b128*>(ex + (size_t)q * P * Y);
            if (accT < 0) {   // accb (slab, y) joins the LAST kept step only: the gradient of a skip input whose last time slice
                              // alone was used (lifting operator, fno/sfno.py:258-259) meets the transform's here, compact
                const T* rl = accb + (size_t)(base + q) * Y;
                for (int i = threadIdx.x; i < n4; i += blockDim.x) {
                    union { b128 v; T e[NV]; } s_;
                    s_.v = s4[i];
#pragma unroll
                    for (int u = 0; u < NV; ++u) {
                        const int idx = i * NV + u, row = idx / t_keep;
                        if (idx - row * t_keep == t_keep - 1) s_.e[u] += rl[row];
                    }
                    __builtin_nontemporal_store(s_.v, d4 + (size_t)q * n4 + i);
                }
                continue;
            }
            const T* rb = accb + (size_t)(base + q) * Y * accT + (accT - 1);
            for (int i = threadIdx.x; i < n4; i += blockDim.x) {
                union { b128 v; T e[NV]; } s_;
                s_.v = s4[i];
#pragma unroll
                for (int u = 0; u < NV; ++u) s_.e[u] += rb[(size_t)((i * NV + u) / t_keep) * accT];
                __builtin_nontemporal_store(s_.v, d4 + (size_t)q * n4 + i);
            }
        }
        return;
    }
    for (int q = 0; q < count; ++q) {   // streamed out: nothing on this GPU reads it before it has left the caches
        const b128* s4 = reinterpret_cast<const b128*>(ex + (size_t)q * P * Y);
        for (int i = threadIdx.x; i < n4; i += blockDim.x) __builtin_nontemporal_store(s4[i], d4 + (size_t)q * n4 + i);
    }
}

// ------------------------------------------------------------------ x transforms on (X, Q) column tiles
// FWD: in (b*c, X, Q) -> out (b*c, 2mx, Q) kept rows;  INV: in (b*c, 2mx, Q) -> out (b*c, X, Q)
template <typename T, int X, int EPT, int C, bool FWD>
__global__ __launch_bounds__(C*(X / EPT)) void k_x(const cx<T>* __restrict__ in, cx<T>* __restrict__ out,
                                                   const cx<T>* __restrict__ tw_x, int Q, int mx, int ntiles, int Xs) {
    typedef cx<T> cf;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    constexpr int G = X / EPT;
    const int c = threadIdx.x % C, j = threadIdx.x / C;
    const int tile = blockIdx.x % ntiles;
    const size_t bc = blockIdx.x / ntiles;
    const int q = tile * C + c;
    const bool valid = q < Q;
    cf x[EPT];
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        const int kx = j + t * G;
        if constexpr (FWD) {
            x[t] = valid ? in[(bc * X + kx) * Q + q] : mk<T>((T)0, (T)0);
        } else {
            int kxi = -1;   // array index kx of a length-X spectrum cut out of / padded from one of length Xs
            if (kx < mx) kxi = kx;
            else if (kx >= Xs - mx && kx < Xs) kxi = kx - (Xs - 2 * mx);
            x[t] = (valid && kxi >= 0) ? in[(bc * 2 * mx + kxi) * Q + q] : mk<T>((T)0, (T)0);
        }
    }
    tile_fft<T, X, EPT, FWD ? -1 : +1, C, false, true>(x, lds, tw_x, j, c);
    if (valid) {
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            const int kx = j + t * G;
            if constexpr (FWD) {
                int kxi = -1;
                if (kx < mx) kxi = kx;
                else if (kx >= X - mx) kxi = kx - (X - 2 * mx);
                if (kxi >= 0) out[(bc * 2 * mx + kxi) * Q + q] = x[t];
            } else {
                out[(bc * X + kx) * Q + q] = x[t];
            }
        }
    }
}

#include "tcfd_fno_dft.hpp"   // k_fwd_ty_dft / k_x_dft / k_inv_ty_dft: the same pipeline for sizes off the FFT kernels

// ------------------------------------------------------------------ contraction
template <typename T>
struct ContractArgsT {
    const cx<T>* vin;     // (b, ci, 2mx, 2my, mt)
    cx<T>* vout;          // (b, co, 2mx, 2my, mt)
    const cx<T>* w[4];    // (ci, co, mx, my, mt)  block index ix + 2*iy
    const cx<T>* bias[4]; // (mx, my, mt) or null
    T delta;
    int b, ci, co, mx, my, mt;
    int adjoint;          // 1: w holds the blocks of the FORWARD contraction, (co, ci, mx, my, mt), and is applied as its
                          // conjugate transpose (the gradient w.r.t. the spectrum): no transposed copy of the weights is made
};
typedef ContractArgsT<float> ContractArgs;

// Plain VALU form: one thread per (batch, out channel, mode); lanes run along the modes so both the
// spectrum and the weight reads are contiguous.  Used for shapes the MFMA kernel does not cover and
// as its cross-check.
template <typename T>
__global__ void k_contract_valu(ContractArgsT<T> a) {
    typedef cx<T> cf;
    const int M = 4 * a.mx * a.my * a.mt;  // kept modes per (b, channel)
    const long total = (long)a.b * a.co * M;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int mode = (int)(idx % M);
        const int o = (int)((idx / M) % a.co);
        const int bb = (int)(idx / ((long)M * a.co));
        const int kt = mode % a.mt;
        const int kyi = (mode / a.mt) % (2 * a.my);
        const int kxi = mode / (a.mt * 2 * a.my);
        const int ix = kxi >= a.mx, iy = kyi >= a.my;
        const int blk = ix + 2 * iy;
        const int wm = ((kxi - ix * a.mx) * a.my + (kyi - iy * a.my)) * a.mt + kt;  // mode inside the block
        const int MB = a.mx * a.my * a.mt;
        const cf* w = a.w[blk];
        T re = 0, im = 0;
        for (int i = 0; i < a.ci; ++i) {
            const cf xv = a.vin[((long)bb * a.ci + i) * M + mode];
            cf wv = a.adjoint ? w[((long)o * a.ci + i) * MB + wm] : w[((long)i * a.co + o) * MB + wm];
            if (a.adjoint) wv.y = -wv.y;
            re += xv.x * wv.x - xv.y * wv.y;
            im += xv.x * wv.y + xv.y * wv.x;
        }
        if (a.bias[blk]) {
            const cf bv = a.bias[blk][wm];
            re += a.delta * bv.x;
            im += a.delta * bv.y;
        }
        a.vout[idx] = mk<T>(re, im);
    }
}

// MFMA form (v_mfma_f32_16x16x4_f32: exact fp32 FMA chain at the fp32 vector rate).
// One wave owns one mode at a time: C[16 b x 16 o] += A[16 b x 4 i] * B[4 i x 16 o], complex product as
// four real MFMA chains (rr, ii, ri, ir).  A workgroup stages, for NM consecutive modes of one
// (block, kx, ky-run), the spectrum slice [b][ci][NM] and the weight slice [ci][co][NM] in LDS with
// coalesced loads (lanes along the contiguous mode axis), then its waves sweep the modes.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
// one K-step of four: fp32 on v_mfma_f32_16x16x4_f32, fp64 on v_mfma_f64_16x16x4_f64 (same operand layout; the result
// rows are interleaved differently, see the store below)
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f64x4 mfma4(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
template <typename T> struct Acc4 { typedef f32x4 type; };
template <> struct Acc4<double> { typedef f64x4 type; };

// mode stride of a staged slice: lanes of the staging loops run along the NM modes first, so the stride (in complex
// elements) is padded to 64 / NM modulo 32 -- the NM x (64 / NM) elements a wave stores at once then fall on distinct
// 8-byte slots of the 64 banks (unpadded, 32 x 12 and 12 x 16 are multiples of 32: NM-way conflicts on every store)
#define CONTRACT_MAXW 8
template <int NM>
__host__ __device__ inline int contract_stride(int elems) { return elems + ((64 / NM) - elems % 32 + 32) % 32; }

template <typename T, int NM>
__global__ __launch_bounds__(256) void k_contract_mfma(ContractArgsT<T> a) {
    typedef cx<T> cf;
    typedef typename Acc4<T>::type acc4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int M = 4 * a.mx * a.my * a.mt;
    const int MB = a.mx * a.my * a.mt;
    const int cip = (a.ci + 3) & ~3;        // K padded to a multiple of 4
    const int bp = (a.b + 15) & ~15;        // M padded to a multiple of 16
    const int cop = (a.co + 15) & ~15;      // N padded to a multiple of 16
    // LDS: A[mode][bp][cip] complex, B[mode][cip][cop] complex (zero padded), padded mode strides SA / SB; the results
    // go back through the same memory as O[b][co][NM + 1] so that the stores to the spectrum run along the modes too
    const int SA = contract_stride<NM>(bp * cip), SB = contract_stride<NM>(cip * cop);
    cf* As = reinterpret_cast<cf*>(smem_raw);
    cf* Bs = As + (size_t)NM * SA;
    // this block's run of NM modes: runs never straddle a corner block (MB % NM == 0 is checked by the host)
    const int run = blockIdx.x;
    const int runs_per_blk = MB / NM;
    const int blk = run / runs_per_blk;
    const int wm0 = (run % runs_per_blk) * NM;               // first mode inside the weight block
    const int ix = blk & 1, iy = blk >> 1;
    const cf* w = a.w[blk];
    // the mode this lane stages and stores: blockDim.x is a multiple of NM, so it is the same in every trip
    const int mm_l = threadIdx.x % NM, rest_l = threadIdx.x / NM, rest_step = blockDim.x / NM;
    long mode_l;
    {
        const int wm = wm0 + mm_l;
        const int kt = wm % a.mt, ky = (wm / a.mt) % a.my, kx = wm / (a.mt * a.my);
        mode_l = ((long)(kx + ix * a.mx) * 2 * a.my + (ky + iy * a.my)) * a.mt + kt;
    }
    // zero fill (padding) then stage
    for (int i = threadIdx.x; i < NM * (SA + SB); i += blockDim.x) As[i] = mk<T>((T)0, (T)0);
    __syncthreads();
    for (int rest = rest_l; rest < a.b * a.ci; rest += rest_step) {
        const int ic = rest % a.ci, bb = rest / a.ci;
        As[(size_t)mm_l * SA + bb * cip + ic] = a.vin[(long)rest * M + mode_l];
    }
    for (int rest = rest_l; rest < a.ci * a.co; rest += rest_step) {
        const int o = rest % a.co, ic = rest / a.co;
        cf wv = a.adjoint ? w[((long)o * a.ci + ic) * MB + wm0 + mm_l] : w[(long)rest * MB + wm0 + mm_l];
        if (a.adjoint) wv.y = -wv.y;
        Bs[(size_t)mm_l * SB + ic * cop + o] = wv;
    }
    __syncthreads();
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const int nwaves = blockDim.x / 64;
    const int mt_tiles = bp / 16, nt_tiles = cop / 16;
    const int nwork = NM * mt_tiles * nt_tiles;
    // every wave keeps the results of its tiles in registers until the operands are no longer needed: O aliases A / B
    // when one round covers the work (the host sizes the allocation by the same rule), else it lies behind them
    constexpr int MAXW = CONTRACT_MAXW;   // tiles per wave held at once; more (large b x co) go round the outer loop again
    cf* Os = nwork <= nwaves * MAXW ? As : Bs + (size_t)NM * SB;
    for (int base = 0; base < nwork; base += nwaves * MAXW) {
        acc4 re[MAXW], im[MAXW];
#pragma unroll
        for (int u = 0; u < MAXW; ++u) {
            const int work = base + u * nwaves + wave;
            acc4 rr = {0, 0, 0, 0}, ii = {0, 0, 0, 0}, ri = {0, 0, 0, 0}, ir = {0, 0, 0, 0};
            if (work < nwork) {
                const int mm = work / (mt_tiles * nt_tiles);
                const int mtile = (work / nt_tiles) % mt_tiles, ntile = work % nt_tiles;
                const cf* Am = As + (size_t)mm * SA;
                const cf* Bm = Bs + (size_t)mm * SB;
                for (int k0 = 0; k0 < cip; k0 += 4) {
                    // A operand: lane l holds A[m = l & 15][k = l >> 4];  B operand: B[k = l >> 4][n = l & 15]
                    const cf av = Am[(size_t)(mtile * 16 + (lane & 15)) * cip + k0 + (lane >> 4)];
                    const cf bv = Bm[(size_t)(k0 + (lane >> 4)) * cop + ntile * 16 + (lane & 15)];
                    rr = mfma4(av.x, bv.x, rr);
                    ii = mfma4(av.y, bv.y, ii);
                    ri = mfma4(av.x, bv.y, ri);
                    ir = mfma4(av.y, bv.x, ir);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                re[u][r] = rr[r] - ii[r];
                im[u][r] = ri[r] + ir[r];
            }
        }
        __syncthreads();   // all operands of this round consumed: O may overwrite them
#pragma unroll
        for (int u = 0; u < MAXW; ++u) {
            const int work = base + u * nwaves + wave;
            if (work >= nwork) continue;
            const int mm = work / (mt_tiles * nt_tiles);
            const int mtile = (work / nt_tiles) % mt_tiles, ntile = work % nt_tiles;
            // C/D layout: col n = lane & 15, row m = (lane >> 4) * 4 + r
            const int o = ntile * 16 + (lane & 15);
            cf bias = mk<T>((T)0, (T)0);
            if (a.bias[blk]) bias = cscale(a.bias[blk][wm0 + mm], a.delta);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // result rows: f32 16x16x4 keeps rows 4 (l >> 4) + r in register r, f64 16x16x4 rows 4 r + (l >> 4)
                const int bb = mtile * 16 + (sizeof(T) == 8 ? 4 * r + (lane >> 4) : (lane >> 4) * 4 + r);
                if (bb < a.b && o < a.co) Os[((size_t)bb * a.co + o) * (NM + 1) + mm] = mk<T>(re[u][r] + bias.x, im[u][r] + bias.y);
            }
        }
        __syncthreads();
        // which (b, o) rows this round produced: all of them when one round covers the work (the usual case); otherwise
        // the tiles of this round only -- rows are complete per round because a round covers whole modes x tiles in order
        if (nwork <= nwaves * MAXW) {
            for (int rest = rest_l; rest < a.b * a.co; rest += rest_step)
                a.vout[(long)rest * M + mode_l] = Os[(size_t)rest * (NM + 1) + mm_l];
        } else {
            const int w_lo = base, w_hi = min(nwork, base + nwaves * MAXW);
            for (int rest = rest_l; rest < a.b * a.co; rest += rest_step) {
                const int bb = rest / a.co, o = rest % a.co;
                const int work = (mm_l * mt_tiles + bb / 16) * nt_tiles + o / 16;
                if (work >= w_lo && work < w_hi) a.vout[(long)rest * M + mode_l] = Os[(size_t)rest * (NM + 1) + mm_l];
            }
        }
        if (base + nwaves * MAXW < nwork) __syncthreads();
    }
}

// ------------------------------------------------------------------ host side
template <typename K>
static int set_lds_attr(K kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)bytes));
    return 0;
}

// ---- launchers of the any-size kernels
template <typename T>
static int launch_fwd_ty_dft(const tcfd_fno_plan* p, const T* v, cx<T>* w1, long slabs, T scale, hipStream_t st) {
    typedef cx<T> ct;
    if (p->mt > 16) return FAIL(TCFD_EINVAL, "fno: modes_t = %d > 16", p->mt);
    const size_t fixed = ((size_t)p->Y + (size_t)p->mt * p->Tp) * sizeof(ct), per = (size_t)p->Y * p->mt * sizeof(ct);
    int NS = (int)std::max<long>(1, std::min<long>(256 / (p->my + 1), (long)((64 * 1024 - (long)fixed) / (long)per)));
    if (fixed + per > 150 * 1024) return FAIL(TCFD_EINVAL, "fno: a slab of Y = %d does not fit LDS", p->Y);
    const size_t lds = fixed + (size_t)NS * per;
    const unsigned blocks = (unsigned)((slabs + NS - 1) / NS);
#define TCFD_MT_CASE(MT_)                                                                                                       \
    if (p->mt <= MT_) {                                                                                                         \
        auto kern = k_fwd_ty_dft<T, MT_>;                                                                                       \
        if (int rc = set_lds_attr(kern, lds)) return rc;                                                                        \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, v, w1, (const ct*)p->tw_y, (const ct*)p->tw_tf, p->Y, p->T_in, \
                           p->t_pad, p->mt, p->my, scale, NS, slabs);                                                          \
        HIP_TRY(hipGetLastError());                                                                                             \
        return 0;                                                                                                               \
    }
    TCFD_MT_CASE(1) TCFD_MT_CASE(2) TCFD_MT_CASE(3) TCFD_MT_CASE(4) TCFD_MT_CASE(5) TCFD_MT_CASE(6) TCFD_MT_CASE(8) TCFD_MT_CASE(12)
    TCFD_MT_CASE(16)
#undef TCFD_MT_CASE
    return FAIL(TCFD_EINVAL, "fno: modes_t = %d > 16", p->mt);
}
template <typename T, bool FWD>
static int launch_x_dft(const tcfd_fno_plan* p, const cx<T>* in, cx<T>* out, long bc, hipStream_t st) {
    typedef cx<T> ct;
    const int Q = 2 * p->my * p->mt;
    const int n_out = FWD ? 2 * p->mx : p->X;
    const size_t lds = (size_t)p->X * sizeof(ct);
    auto kern = k_x_dft<T, FWD>;
    int rc = set_lds_attr(kern, lds);
    if (rc) return rc;
    if (bc > 65535) return FAIL(TCFD_EINVAL, "fno: batch x channels = %ld exceeds the grid's z range", bc);
    hipLaunchKernelGGL(kern, dim3((unsigned)((Q + 255) / 256), (unsigned)((n_out + 15) / 16), (unsigned)bc), dim3(256), lds, st, in,
                       out, (const ct*)p->tw_x, p->X, FWD ? p->X : p->Xs, p->mx, Q);
    HIP_TRY(hipGetLastError());
    return 0;
}
template <typename T>
static int launch_inv_ty_dft(const tcfd_fno_plan* p, const cx<T>* w2, T* out, long slabs, int t_keep, T scale, hipStream_t st,
                             const T* acc, const T* accb, int accT) {
    typedef cx<T> ct;
    if (p->mt > 16) return FAIL(TCFD_EINVAL, "fno: modes_t = %d > 16", p->mt);
    const size_t Q = (size_t)2 * p->my * p->mt;
    const size_t fixed = ((size_t)p->Y + (size_t)t_keep * p->mt) * sizeof(ct);
    const size_t per = (Q + 2 * p->mt + 2 * (size_t)(p->my + 1) * p->mt) * sizeof(ct) + (size_t)p->Y * t_keep * sizeof(T);
    if (fixed + per > 150 * 1024) return FAIL(TCFD_EINVAL, "fno: a slab of Y = %d x %d steps does not fit LDS", p->Y, t_keep);
    // one lane per (slab, y) row: NS slabs per workgroup so that the rows fill ~256 lanes, the block a whole number of waves
    int NS = (int)std::max<long>(1, std::min<long>(std::max(1, 256 / p->Y), (long)((64 * 1024 - (long)fixed) / (long)per)));
    const size_t lds = fixed + (size_t)NS * per;
    const unsigned threads = (unsigned)std::min(1024, ((NS * p->Y + 63) / 64) * 64);
    const unsigned blocks = (unsigned)((slabs + NS - 1) / NS);
#define TCFD_MT_CASE(MT_)                                                                                                        \
    if (p->mt <= MT_) {                                                                                                          \
        auto kern = k_inv_ty_dft<T, MT_>;                                                                                        \
        if (int rc = set_lds_attr(kern, lds)) return rc;                                                                         \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, st, w2, out, (const ct*)p->tw_y, (const ct*)p->tw_ti, p->Y,   \
                           p->Ys, p->T_out, t_keep, p->mt, p->my, scale, NS, slabs, acc, accb, accT);                            \
        HIP_TRY(hipGetLastError());                                                                                              \
        return 0;                                                                                                                \
    }
    TCFD_MT_CASE(1) TCFD_MT_CASE(2) TCFD_MT_CASE(3) TCFD_MT_CASE(4) TCFD_MT_CASE(5) TCFD_MT_CASE(6) TCFD_MT_CASE(8) TCFD_MT_CASE(12)
    TCFD_MT_CASE(16)
#undef TCFD_MT_CASE
    return FAIL(TCFD_EINVAL, "fno: modes_t = %d > 16", p->mt);
}

template <typename T, int X, bool FWD>
static int launch_x(const tcfd_fno_plan* p, const cx<T>* in, cx<T>* out, long bc, hipStream_t st) {
    typedef cx<T> ct;
    constexpr int EPT = X % 3 == 0 ? 12 : (X % 5 == 0 ? 20 : (X >= 512 ? 16 : (X >= 64 ? 8 : 4)));
    constexpr int C = 128 / (int)sizeof(ct);  // 16 complex64 / 8 complex128 = one 128-byte line
    const int Q = 2 * p->my * p->mt;
    const int ntiles = (Q + C - 1) / C;
    constexpr size_t lds = (size_t)lds_elems<X, EPT, C, false>() * sizeof(ct);
    auto kern = k_x<T, X, EPT, C, FWD>;
    int rc = set_lds_attr(kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)(bc * ntiles)), dim3(C * (X / EPT)), lds, st, in, out, (const ct*)p->tw_x, Q,
                       p->mx, ntiles, FWD ? X : p->Xs);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
template <typename T, int Y>
static int ty2_geometry(int P, int* NS) {
    constexpr int G = TyCfg2<Y, T>::G;
    if (P * G > 1024) return FAIL(TCFD_EINVAL, "fno: %d packed time pairs x %d lanes exceed a workgroup", P, G);
    int ns = env_int("TCFD_FNO_NS", 0);
    if (ns <= 0) ns = std::max(1, std::min(8, 256 / (P * G)));
    while (ns > 1 && ns * P * G > 1024) --ns;
    *NS = ns;
    return 0;
}

template <typename T, int Y>
static int launch_fwd_ty2(const tcfd_fno_plan* p, const T* v, cx<T>* w1, long slabs, T scale, hipStream_t st) {
    typedef cx<T> ct;
    constexpr int EPT = TyCfg2<Y, T>::EPT, G = TyCfg2<Y, T>::G;
    const int P = (p->T_in + 1) / 2;
    int NS, rc;
    if ((rc = ty2_geometry<T, Y>(P, &NS))) return rc;
    if (((size_t)Y * p->T_in * sizeof(T)) % 16 != 0) return FAIL(TCFD_EINVAL, "fno: slab of %d x %d values is not a multiple of 16 bytes", Y, p->T_in);
    const size_t per = std::max((size_t)Y * p->T_in * sizeof(T), (size_t)P * Y * sizeof(ct));
    size_t lds;
    for (;; --NS) {
        lds = (size_t)NS * per + (size_t)p->mt * p->Tp * sizeof(ct);
        if (lds <= 160 * 1024 || NS == 1) break;
    }
    if (lds > 160 * 1024) return FAIL(TCFD_EINVAL, "fno: slab does not fit LDS (Y=%d, T=%d)", Y, p->T_in);
    auto kern = k_fwd_ty2<T, Y, EPT>;
    if ((rc = set_lds_attr(kern, lds))) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)((slabs + NS - 1) / NS)), dim3(NS * P * G), lds, st, v, w1, (const ct*)p->tw_y,
                       (const ct*)p->tw_tf, p->T_in, p->t_pad, p->mt, p->my, scale, P, NS, slabs,
                       p->mt > 1 ? (unsigned)(((1ull << 32) + p->mt - 1) / p->mt) : 0u);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T, int Y>
static int launch_inv_ty2(const tcfd_fno_plan* p, const cx<T>* w2, T* out, long slabs, int t_keep, T scale,
                          hipStream_t st, const T* acc, const T* accb = nullptr, int accT = 0) {
    typedef cx<T> ct;
    constexpr int EPT = TyCfg2<Y, T>::EPT, G = TyCfg2<Y, T>::G;
    const int P = (t_keep + 1) / 2;
    int NS, rc;
    if ((rc = ty2_geometry<T, Y>(P, &NS))) return rc;
    if (((size_t)Y * t_keep * sizeof(T)) % 16 != 0) return FAIL(TCFD_EINVAL, "fno: slab of %d x %d values is not a multiple of 16 bytes", Y, t_keep);
    size_t lds;
    for (;; --NS) {
        lds = ((size_t)NS * P * Y + (size_t)NS * 2 * p->my * p->mt + (size_t)t_keep * p->mt) * sizeof(ct);
        if (lds <= 160 * 1024 || NS == 1) break;
    }
    if (lds > 160 * 1024) return FAIL(TCFD_EINVAL, "fno: slab does not fit LDS (Y=%d, T_out=%d)", Y, p->T_out);
    auto kern = k_inv_ty2<T, Y, EPT>;
    if ((rc = set_lds_attr(kern, lds))) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)((slabs + NS - 1) / NS)), dim3(NS * P * G), lds, st, w2, out, (const ct*)p->tw_y,
                       (const ct*)p->tw_ti, p->T_out, t_keep, p->mt, p->my, scale, P, NS, slabs, p->Ys, acc, accb, accT);
    HIP_TRY(hipGetLastError());
    return 0;
}

#define DISPATCH_FFT(n, CALL)                                              \
    switch (n) {                                                            \
        case 8: { constexpr int N_ = 8; return CALL; }                      \
        case 16: { constexpr int N_ = 16; return CALL; }                    \
        case 32: { constexpr int N_ = 32; return CALL; }                    \
        case 64: { constexpr int N_ = 64; return CALL; }                    \
        case 128: { constexpr int N_ = 128; return CALL; }                  \
        case 256: { constexpr int N_ = 256; return CALL; }                  \
        case 512: { constexpr int N_ = 512; return CALL; }                  \
        case 1024: { constexpr int N_ = 1024; return CALL; }                \
        case 96: { constexpr int N_ = 96; return CALL; }                    \
        case 192: { constexpr int N_ = 192; return CALL; }                  \
        case 384: { constexpr int N_ = 384; return CALL; }                  \
        case 768: { constexpr int N_ = 768; return CALL; }                  \
        case 80: { constexpr int N_ = 80; return CALL; }                    \
        case 160: { constexpr int N_ = 160; return CALL; }                  \
        case 320: { constexpr int N_ = 320; return CALL; }                  \
        case 640: { constexpr int N_ = 640; return CALL; }                  \
        default: return FAIL(TCFD_EINVAL, "unsupported transform length %d", n); \
    }

static int force_dft() { return env_int("TCFD_FNO_DFT", 0); }   // 1: any-size kernels for every size (cross-check; read per call)
template <typename T>
static int do_fwd_ty(const tcfd_fno_plan* p, const T* v, cx<T>* w1, long slabs, T s, hipStream_t st) {
    if (!fft_len(p->Y) || force_dft()) return launch_fwd_ty_dft<T>(p, v, w1, slabs, s, st);
    DISPATCH_FFT(p->Y, (launch_fwd_ty2<T, N_>(p, v, w1, slabs, s, st)));
}
template <typename T>
static int do_inv_ty(const tcfd_fno_plan* p, const cx<T>* w2, T* out, long slabs, int t_keep, T s, hipStream_t st,
                     const T* acc = nullptr, const T* accb = nullptr, int accT = 0) {
    if (!fft_len(p->Y) || force_dft()) return launch_inv_ty_dft<T>(p, w2, out, slabs, t_keep, s, st, acc, accb, accT);
    DISPATCH_FFT(p->Y, (launch_inv_ty2<T, N_>(p, w2, out, slabs, t_keep, s, st, acc, accb, accT)));
}
template <typename T>
static int do_fwd_x(const tcfd_fno_plan* p, const cx<T>* in, cx<T>* out, long bc, hipStream_t st) {
    if (!fft_len(p->X) || force_dft()) return launch_x_dft<T, true>(p, in, out, bc, st);
    DISPATCH_FFT(p->X, (launch_x<T, N_, true>(p, in, out, bc, st)));
}
template <typename T>
static int do_inv_x(const tcfd_fno_plan* p, const cx<T>* in, cx<T>* out, long bc, hipStream_t st) {
    if (!fft_len(p->X) || force_dft()) return launch_x_dft<T, false>(p, in, out, bc, st);
    DISPATCH_FFT(p->X, (launch_x<T, N_, false>(p, in, out, bc, st)));
}

template <typename T, int NM>
static size_t contract_lds(const ContractArgsT<T>& a) {
    const int cip = (a.ci + 3) & ~3, bp = (a.b + 15) & ~15, cop = (a.co + 15) & ~15;
    const size_t stage = (size_t)NM * (contract_stride<NM>(bp * cip) + contract_stride<NM>(cip * cop));
    const size_t outs = (size_t)a.b * a.co * (NM + 1);
    const bool alias = NM * (bp / 16) * (cop / 16) <= 4 * CONTRACT_MAXW;   // 256 lanes = 4 waves
    return (alias ? std::max(stage, outs) : stage + outs) * sizeof(cx<T>);
}
template <typename T, int NM>
static int launch_contract_mfma(const ContractArgsT<T>& a, size_t lds, hipStream_t st) {
    auto kern = k_contract_mfma<T, NM>;
    int rc = set_lds_attr(kern, lds);
    if (rc) return rc;
    const int MB = a.mx * a.my * a.mt;
    hipLaunchKernelGGL(kern, dim3((unsigned)(4 * MB / NM)), dim3(256), lds, st, a);
    return 0;
}
template <typename T>
static int do_contract(ContractArgsT<T> a, int use_mfma, hipStream_t st) {
    const int MB = a.mx * a.my * a.mt;
    // 8 modes per workgroup (64-byte runs of fp32 spectrum and weights); TCFD_CONTRACT_NM=16 (read per call) selects
    // whole 128-byte lines with half the workgroups: measured 40.5 against 39.2 us at the config-5 shape -- the launch is
    // ~1.4 rounds of workgroups moving in step (load burst, MFMAs, store burst), not a bandwidth or LDS problem
    const int force = env_int("TCFD_CONTRACT_NM", 0);
    const size_t lds16 = contract_lds<T, 16>(a), lds8 = contract_lds<T, 8>(a);
    int rc = -1;
    if (use_mfma && force == 16 && MB % 16 == 0 && lds16 <= 150 * 1024)
        rc = launch_contract_mfma<T, 16>(a, lds16, st);
    else if (use_mfma && MB % 8 == 0 && lds8 <= 150 * 1024)
        rc = launch_contract_mfma<T, 8>(a, lds8, st);
    else {
        const long total = (long)a.b * a.co * 4 * MB;
        const unsigned blocks = (unsigned)std::min<long>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(k_contract_valu<T>, dim3(blocks), dim3(256), 0, st, a);
        rc = 0;
    }
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T>
static ContractArgsT<T> contract_args(const void* vin, void* vout, const void* const* weights, const void* const* bias,
                                      double delta, int batch, int cin, int cout, int mx, int my, int mt) {
    ContractArgsT<T> a;
    a.vin = (const cx<T>*)vin; a.vout = (cx<T>*)vout;
    for (int k = 0; k < 4; ++k) {
        a.w[k] = (const cx<T>*)weights[k];
        a.bias[k] = bias ? (const cx<T>*)bias[k] : nullptr;
    }
    a.delta = (T)delta; a.b = batch; a.ci = cin; a.co = cout; a.mx = mx; a.my = my; a.mt = mt;
    a.adjoint = 0;
    return a;
}

template <typename T>
static int spectral_conv_impl(const tcfd_fno_plan* p, const void* v, const void* const* weights, const void* const* bias,
                              double delta, void* out, int batch, int cin, int cout, int t_keep, double fwd_scale,
                              double inv_scale, int use_mfma, void* ws, hipStream_t st) {
    typedef cx<T> ct;
    const size_t Q = (size_t)2 * p->my * p->mt;
    const int cmax = std::max(cin, cout);
    unsigned char* base = (unsigned char*)ws;
    ct* W = (ct*)base;
    ct* V = (ct*)(base + al256((size_t)batch * cmax * p->X * Q * sizeof(ct)));
    ct* O = (ct*)((unsigned char*)V + al256((size_t)batch * cin * 2 * p->mx * Q * sizeof(ct)));
    int rc;
    if ((rc = do_fwd_ty<T>(p, (const T*)v, W, (long)batch * cin * p->X, (T)fwd_scale, st))) return rc;
    if ((rc = do_fwd_x<T>(p, W, V, (long)batch * cin, st))) return rc;
    if ((rc = do_contract<T>(contract_args<T>(V, O, weights, bias, delta, batch, cin, cout, p->mx, p->my, p->mt), use_mfma, st)))
        return rc;
    if ((rc = do_inv_x<T>(p, O, W, (long)batch * cout, st))) return rc;
    return do_inv_ty<T>(p, W, (T*)out, (long)batch * cout * p->X, t_keep, (T)inv_scale, st);
}

// Full spectral convolution.  v (b, ci, X, Y, T_in) real -> out (b, co, X, Y, t_keep) real (the last t_keep of the T_out
// reconstructed steps) in the plan's precision.  weights[k] (ci, co, mx, my, mt, 2), bias[k] (mx, my, mt, 2) or NULL.
// fwd_scale / inv_scale: normalisation of rfftn / irfftn ("backward": 1 and 1/(X*Y*T_out)).
extern "C" int tcfd_fno_spectral_conv(const tcfd_fno_plan* p, const void* v, const void* const* weights,
                                      const void* const* bias, double delta, void* out, int batch, int cin, int cout,
                                      int t_keep, double fwd_scale, double inv_scale, int use_mfma, void* ws,
                                      size_t ws_bytes, void* stream) {
    if (!p || !v || !weights || !out) return FAIL(TCFD_EINVAL, "fno_spectral_conv: null argument");
    if (batch <= 0 || cin <= 0 || cout <= 0 || t_keep <= 0 || t_keep > p->T_out)
        return FAIL(TCFD_EINVAL, "fno_spectral_conv: bad sizes");
    const size_t need = tcfd_fno_workspace_bytes(p, batch, cin, cout);
    if (!ws || ws_bytes < need) return FAIL(TCFD_EWORKSPACE, "workspace %zu B < required %zu B", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    return p->dtype == TCFD_C128
               ? spectral_conv_impl<double>(p, v, weights, bias, delta, out, batch, cin, cout, t_keep, fwd_scale, inv_scale, use_mfma, ws, st)
               : spectral_conv_impl<float>(p, v, weights, bias, delta, out, batch, cin, cout, t_keep, fwd_scale, inv_scale, use_mfma, ws, st);
}

// The two halves of the spectral convolution on their own, for layers that post-process the spectrum between
// the contraction and the inverse transform (SpectralConvT(postprocess=HelmholtzProjection), fno/sfno.py:449):
//   forward_trunc : v (batch, c, X, Y, T_in) real  -> vh (batch, c, 2mx, 2my, mt) complex (kept modes only)
//   inverse_trunc : vh (batch, c, 2mx, 2my, mt)    -> out (batch, c, X, Y, t_keep) real
// Workspace: tcfd_fno_workspace_bytes(plan, batch, c, c).
extern "C" int tcfd_fno_forward_trunc(const tcfd_fno_plan* p, const void* v, void* vh, int batch, int c, double fwd_scale,
                                      void* ws, size_t ws_bytes, void* stream) {
    if (!p || !v || !vh || batch <= 0 || c <= 0) return FAIL(TCFD_EINVAL, "fno_forward_trunc: bad argument");
    if (!ws || ws_bytes < tcfd_fno_workspace_bytes(p, batch, c, c)) return FAIL(TCFD_EWORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (p->dtype == TCFD_C128) {
        if ((rc = do_fwd_ty<double>(p, (const double*)v, (cx<double>*)ws, (long)batch * c * p->X, fwd_scale, st))) return rc;
        return do_fwd_x<double>(p, (const cx<double>*)ws, (cx<double>*)vh, (long)batch * c, st);
    }
    if ((rc = do_fwd_ty<float>(p, (const float*)v, (cf*)ws, (long)batch * c * p->X, (float)fwd_scale, st))) return rc;
    return do_fwd_x<float>(p, (const cf*)ws, (cf*)vh, (long)batch * c, st);
}

// out = [acc +] inverse transform; acc (same shape as out, may be out itself) or NULL
extern "C" int tcfd_fno_inverse_trunc_acc(const tcfd_fno_plan* p, const void* vh, void* out, const void* acc, int batch, int c,
                                          int t_keep, double inv_scale, void* ws, size_t ws_bytes, void* stream) {
    if (!p || !vh || !out || batch <= 0 || c <= 0 || t_keep <= 0 || t_keep > p->T_out)
        return FAIL(TCFD_EINVAL, "fno_inverse_trunc: bad argument");
    if (!ws || ws_bytes < tcfd_fno_workspace_bytes(p, batch, c, c)) return FAIL(TCFD_EWORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (p->dtype == TCFD_C128) {
        if ((rc = do_inv_x<double>(p, (const cx<double>*)vh, (cx<double>*)ws, (long)batch * c, st))) return rc;
        return do_inv_ty<double>(p, (const cx<double>*)ws, (double*)out, (long)batch * c * p->X, t_keep, inv_scale, st,
                                 (const double*)acc);
    }
    if ((rc = do_inv_x<float>(p, (const cf*)vh, (cf*)ws, (long)batch * c, st))) return rc;
    return do_inv_ty<float>(p, (const cf*)ws, (float*)out, (long)batch * c * p->X, t_keep, (float)inv_scale, st, (const float*)acc);
}
// out = transform + res[..., -1:] broadcast over the kept steps: res (batch * c, X, Y, res_T) real, its LAST time slice is the
// residual frame the output operator adds to the convolution (fno/sfno.py:327)
extern "C" int tcfd_fno_inverse_trunc_residual(const tcfd_fno_plan* p, const void* vh, void* out, const void* res, int res_T,
                                               int batch, int c, int t_keep, double inv_scale, void* ws, size_t ws_bytes,
                                               void* stream) {
    if (!p || !vh || !out || !res || res_T <= 0 || batch <= 0 || c <= 0 || t_keep <= 0 || t_keep > p->T_out)
        return FAIL(TCFD_EINVAL, "fno_inverse_trunc_residual: bad argument");
    if (!ws || ws_bytes < tcfd_fno_workspace_bytes(p, batch, c, c)) return FAIL(TCFD_EWORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (p->dtype == TCFD_C128) {
        if ((rc = do_inv_x<double>(p, (const cx<double>*)vh, (cx<double>*)ws, (long)batch * c, st))) return rc;
        return do_inv_ty<double>(p, (const cx<double>*)ws, (double*)out, (long)batch * c * p->X, t_keep, inv_scale, st, nullptr,
                                 (const double*)res, res_T);
    }
    if ((rc = do_inv_x<float>(p, (const cf*)vh, (cf*)ws, (long)batch * c, st))) return rc;
    return do_inv_ty<float>(p, (const cf*)ws, (float*)out, (long)batch * c * p->X, t_keep, (float)inv_scale, st, nullptr,
                            (const float*)res, res_T);
}
// out = transform, plus last (batch * c, X, Y) at the LAST kept step only: the adjoint of "use the last time slice of the input"
// joining the adjoint of the forward transform in its store loop (training: the lifting operator's skip, fno/sfno.py:258-259)
extern "C" int tcfd_fno_inverse_trunc_last(const tcfd_fno_plan* p, const void* vh, void* out, const void* last, int batch, int c,
                                           int t_keep, double inv_scale, void* ws, size_t ws_bytes, void* stream) {
    if (!p || !vh || !out || !last || batch <= 0 || c <= 0 || t_keep <= 0 || t_keep > p->T_out)
        return FAIL(TCFD_EINVAL, "fno_inverse_trunc_last: bad argument");
    if (!ws || ws_bytes < tcfd_fno_workspace_bytes(p, batch, c, c)) return FAIL(TCFD_EWORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (p->dtype == TCFD_C128) {
        if ((rc = do_inv_x<double>(p, (const cx<double>*)vh, (cx<double>*)ws, (long)batch * c, st))) return rc;
        return do_inv_ty<double>(p, (const cx<double>*)ws, (double*)out, (long)batch * c * p->X, t_keep, inv_scale, st, nullptr,
                                 (const double*)last, -1);
    }
    if ((rc = do_inv_x<float>(p, (const cf*)vh, (cf*)ws, (long)batch * c, st))) return rc;
    return do_inv_ty<float>(p, (const cf*)ws, (float*)out, (long)batch * c * p->X, t_keep, (float)inv_scale, st, nullptr,
                            (const float*)last, -1);
}
extern "C" int tcfd_fno_inverse_trunc(const tcfd_fno_plan* p, const void* vh, void* out, int batch, int c, int t_keep,
                                      double inv_scale, void* ws, size_t ws_bytes, void* stream) {
    return tcfd_fno_inverse_trunc_acc(p, vh, out, nullptr, batch, c, t_keep, inv_scale, ws, ws_bytes, stream);
}

// Contraction alone on caller-provided truncated spectra (tests, MFMA vs VALU cross-check); dtype TCFD_C64 / TCFD_C128.
extern "C" int tcfd_fno_contract(const void* vin, const void* const* weights, const void* const* bias, double delta,
                                 void* vout, int batch, int cin, int cout, int mx, int my, int mt, int use_mfma,
                                 int dtype, void* stream) {
    if (!vin || !weights || !vout) return FAIL(TCFD_EINVAL, "fno_contract: null argument");
    if (dtype == TCFD_C128)
        return do_contract<double>(contract_args<double>(vin, vout, weights, bias, delta, batch, cin, cout, mx, my, mt), use_mfma,
                                   (hipStream_t)stream);
    if (dtype != TCFD_C64) return FAIL(TCFD_EINVAL, "fno_contract: bad dtype %d", dtype);
    return do_contract<float>(contract_args<float>(vin, vout, weights, bias, delta, batch, cin, cout, mx, my, mt), use_mfma,
                              (hipStream_t)stream);
}

// gv[b][i] = sum_o conj(W[i][o]) gh[b][o]: the same kernels reading the forward blocks (cin_fwd = cout here) transposed
extern "C" int tcfd_fno_contract_adjoint(const void* gh, const void* const* weights, void* gv, int batch, int cout_fwd,
                                         int cin_fwd, int mx, int my, int mt, int use_mfma, int dtype, void* stream) {
    if (!gh || !weights || !gv) return FAIL(TCFD_EINVAL, "fno_contract_adjoint: null argument");
    if (dtype == TCFD_C128) {
        auto a = contract_args<double>(gh, gv, weights, nullptr, 0.0, batch, cout_fwd, cin_fwd, mx, my, mt);
        a.adjoint = 1;
        return do_contract<double>(a, use_mfma, (hipStream_t)stream);
    }
    if (dtype != TCFD_C64) return FAIL(TCFD_EINVAL, "fno_contract_adjoint: bad dtype %d", dtype);
    auto a = contract_args<float>(gh, gv, weights, nullptr, 0.0, batch, cout_fwd, cin_fwd, mx, my, mt);
    a.adjoint = 1;
    return do_contract<float>(a, use_mfma, (hipStream_t)stream);
}

// ------------------------------------------------------------------ weight / bias gradient of the contraction
//   gw_k[i][o][x'][y'][t] = sum_b conj(vh[b][i][x][y][t]) gh[b][o][x][y][t]        (k = corner of (x, y), torch's convention
//   gb_k[x'][y'][t]       = delta sum_{b, o} gh[b][o][x][y][t]                      dL/dRe + i dL/dIm for a complex leaf)
// One lane per (mode, group of IC input channels), IC x OC accumulators in registers, the batch as the loop: lanes run along
// the contiguous mode axis, so spectrum reads and gradient writes are coalesced.  Replaces four einsum("bixyt,boxyt->ioxyt") on strided corner views (conj + copies + bmm: ~0.44 ms per
// layer at config 5) with one launch that reads both spectra once (2 x 29.5 MB) and writes the 9.2 MB gradient.
template <typename T>
struct WgradArgsT {
    const cx<T>* vh;      // (b, ci, 2mx, 2my, mt)
    const cx<T>* gh;      // (b, co, 2mx, 2my, mt)
    cx<T>* gw[4];         // (ci, co, mx, my, mt) or null
    cx<T>* gb[4];         // (mx, my, mt) or null
    T delta;
    int b, ci, co, mx, my, mt;
};

template <typename T, int IC, int OC>
__global__ __launch_bounds__(256) void k_contract_wgrad(WgradArgsT<T> a) {
    const int M = 4 * a.mx * a.my * a.mt;
    const int mode = blockIdx.x * 256 + threadIdx.x;
    if (mode >= M) return;
    const int i0 = blockIdx.y * IC, o0 = blockIdx.z * OC;
    const int kt = mode % a.mt;
    const int kyi = (mode / a.mt) % (2 * a.my);
    const int kxi = mode / (a.mt * 2 * a.my);
    const int ix = kxi >= a.mx, iy = kyi >= a.my;
    const int blk = ix + 2 * iy;
    const long MB = (long)a.mx * a.my * a.mt;
    const long wm = ((long)(kxi - ix * a.mx) * a.my + (kyi - iy * a.my)) * a.mt + kt;
    if (a.gw[blk]) {
        // IC input x OC output channels of ONE mode in registers: every spectrum value is read co / OC (vh) resp. ci / IC (gh)
        // times in all (one lane per (mode, input channel) read gh ci times: 0.09 ms, bound by the L2s)
        T re[IC][OC], im[IC][OC];
#pragma unroll
        for (int v = 0; v < IC; ++v)
#pragma unroll
            for (int u = 0; u < OC; ++u) re[v][u] = im[v][u] = 0;
        for (int bb = 0; bb < a.b; ++bb) {
            cx<T> vv[IC], gg[OC];
#pragma unroll
            for (int v = 0; v < IC; ++v) vv[v] = i0 + v < a.ci ? a.vh[((long)bb * a.ci + i0 + v) * M + mode] : mk<T>((T)0, (T)0);
#pragma unroll
            for (int u = 0; u < OC; ++u) gg[u] = o0 + u < a.co ? a.gh[((long)bb * a.co + o0 + u) * M + mode] : mk<T>((T)0, (T)0);
#pragma unroll
            for (int v = 0; v < IC; ++v)
#pragma unroll
                for (int u = 0; u < OC; ++u) {
                    re[v][u] += vv[v].x * gg[u].x + vv[v].y * gg[u].y;          // conj(v) g
                    im[v][u] += vv[v].x * gg[u].y - vv[v].y * gg[u].x;
                }
        }
#pragma unroll
        for (int v = 0; v < IC; ++v)
#pragma unroll
            for (int u = 0; u < OC; ++u)
                if (i0 + v < a.ci && o0 + u < a.co) a.gw[blk][((long)(i0 + v) * a.co + o0 + u) * MB + wm] = mk<T>(re[v][u], im[v][u]);
    }
    if (a.gb[blk] && blockIdx.y == 0 && blockIdx.z == 0) {
        T re = 0, im = 0;
        for (long r = 0; r < (long)a.b * a.co; ++r) {
            const cx<T> g = a.gh[r * M + mode];
            re += g.x;
            im += g.y;
        }
        a.gb[blk][wm] = mk<T>(a.delta * re, a.delta * im);
    }
}

template <typename T>
static int do_contract_wgrad(const void* vh, const void* gh, void* const* gw, void* const* gb, double delta, int batch, int cin,
                             int cout, int mx, int my, int mt, hipStream_t st) {
    WgradArgsT<T> a;
    a.vh = (const cx<T>*)vh; a.gh = (const cx<T>*)gh;
    bool any = false;
    for (int k = 0; k < 4; ++k) {
        a.gw[k] = gw ? (cx<T>*)gw[k] : nullptr;
        a.gb[k] = gb ? (cx<T>*)gb[k] : nullptr;
        any = any || a.gw[k] || a.gb[k];
    }
    if (!any) return 0;
    a.delta = (T)delta; a.b = batch; a.ci = cin; a.co = cout; a.mx = mx; a.my = my; a.mt = mt;
    const long M = 4L * mx * my * mt;
    if (M <= 0 || M > (1L << 30) || cin < 1 || cout < 1 || batch < 1) return FAIL(TCFD_EINVAL, "fno_contract_wgrad: bad shape");
    constexpr int OC = 10, IC = sizeof(T) == 8 ? 2 : 5;      // 100 resp. 80 accumulator registers
    hipLaunchKernelGGL((k_contract_wgrad<T, IC, OC>), dim3((unsigned)((M + 255) / 256), (cin + IC - 1) / IC, (cout + OC - 1) / OC),
                       dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int tcfd_fno_contract_wgrad(const void* vh, const void* gh, void* const* gw, void* const* gb, double delta,
                                       int batch, int cin, int cout, int mx, int my, int mt, int dtype, void* stream) {
    if (!vh || !gh) return FAIL(TCFD_EINVAL, "fno_contract_wgrad: null argument");
    if (dtype == TCFD_C128) return do_contract_wgrad<double>(vh, gh, gw, gb, delta, batch, cin, cout, mx, my, mt, (hipStream_t)stream);
    if (dtype != TCFD_C64) return FAIL(TCFD_EINVAL, "fno_contract_wgrad: bad dtype %d", dtype);
    return do_contract_wgrad<float>(vh, gh, gw, gb, delta, batch, cin, cout, mx, my, mt, (hipStream_t)stream);
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
#pragma unroll
    for (int c = 0; c < CO; ++c)   // streamed out (non-temporal): an 0.8 GB activation tensor outlives every cache; SFNO forward 5.77 -> 5.42 ms
        __builtin_nontemporal_store(pw_act(o[c], act2), reinterpret_cast<vf*>(ob + (size_t)c * a.P));
}

// ------------------------------------------------------------------ inverse t/y transform + pointwise block in ONE kernel
// The tail of an SFNO layer, v <- act(FFN(SpectralConv(v)) + W v) (fno/sfno.py:607-614) or the lifting operator's
// act(v[..., -1:] + FFN(SpectralConvT(v))) (:258-259): k_inv_ty2 writes the convolution's output (b, C, X, Y, T) to HBM
// and k_pointwise reads it straight back, together 2 of the layer's 5 passes over the activations.  Here ONE workgroup
// owns the row (b, x) of ALL CI channels: it runs their inverse t/y transforms exactly as k_inv_ty2 does (slab s =
// channel s), leaves the CI output slabs [y][t] in LDS, and feeds the pointwise block from there -- same arithmetic in
// the same order, so the result is bit-identical to the two-kernel path.  The skip operand of a lane's first points is
// requested before the transforms start, so that its HBM latency hides behind them.
// MEASURED SLOWER and therefore opt-in (TCFD_FNO_FUSE_TAIL=1): at config 5 (width 10, 256 x 256 x 10) a row of 10
// channels is 100 KB of LDS -- one 800-lane workgroup per CU, 3 waves per SIMD, its phases (spectrum in, transforms,
// two trips of the pointwise block with their scalar weight loads) strictly one after the other: 1.29 ms per layer
// against 0.29 + 0.59 ms for the two kernels, which run 4 resp. 8 workgroups per CU and hide exactly those latencies.
// Saving 2 of 5 passes over the activations does not pay for losing the occupancy.
template <int Y, int EPT, int CI, int CM, int CO>
__global__ __launch_bounds__(1024) void k_inv_ty_pw(const cf* __restrict__ w2, PwArgs a, const float* __restrict__ pw_w1,
                                                    const float* __restrict__ pw_b1, const float* __restrict__ pw_w2t,
                                                    const float* __restrict__ pw_b2, const float* __restrict__ pw_wst,
                                                    const float* __restrict__ pw_bs, const cf* __restrict__ tw_y,
                                                    const cf* __restrict__ tw_ti, int T_out, int t_keep, int mt, int my,
                                                    float scale, int P, int X, int Ys) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typedef v2f vf;
    constexpr int G = Y / EPT;
    constexpr int NS = CI;
    const int Q = 2 * my * mt, t0 = T_out - t_keep;
    cf* ex = reinterpret_cast<cf*>(smem_raw);                  // [NS*P][Y] input / exchange, later [NS][Y][t_keep] floats
    cf* win = ex + (size_t)NS * P * Y;                         // [NS][Q]
    cf* twt = win + (size_t)NS * Q;                            // [t_keep][mt]
    const int tr = threadIdx.x / G, j = threadIdx.x % G;
    const int s = tr / P, p = tr - s * P;
    const int b = blockIdx.x / X, xrow = blockIdx.x - b * X;
    cf* lds = ex + (size_t)tr * Y;
    const int slab_pts = Y * t_keep;                           // points of one (b, c, x) slab
    const int n_items = slab_pts / 2;                          // packed pairs (t_keep is even)
    const int nthr = NS * P * G;
    // ---- skip operand of this lane's first pair
    vf sv0[CI];
    const bool has0 = (int)threadIdx.x < n_items;
    if (a.skip_mode == 1 && has0) {
        const float* sb = a.s + ((size_t)b * CI * X + xrow) * slab_pts + 2 * threadIdx.x;
#pragma unroll
        for (int i = 0; i < CI; ++i) sv0[i] = *reinterpret_cast<const vf*>(sb + (size_t)i * X * slab_pts);
    }
    {
        for (int q = 0; q < NS; ++q) {
            const cf* src = w2 + ((size_t)(b * CI + q) * X + xrow) * Q;
            for (int i = threadIdx.x; i < Q; i += nthr) win[(size_t)q * Q + i] = src[i];
        }
        for (int i = threadIdx.x; i < t_keep * mt; i += nthr) twt[i] = tw_ti[(size_t)t0 * mt + i];
        float4* z4 = reinterpret_cast<float4*>(lds);           // zero this transform's spectrum (the padding)
#pragma unroll
        for (int t = 0; t < EPT / 2; ++t) z4[j + t * G] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    {   // as k_inv_ty2
        const float hsc = 0.5f * scale;
        const bool pair = 2 * p + 1 < t_keep;
        const cf* e0 = twt + (size_t)(2 * p) * mt;
        const cf* e1 = twt + (size_t)(pair ? 2 * p + 1 : 2 * p) * mt;
        const cf* wq = win + (size_t)s * Q;
        auto slot = [&](int k) { return k < my ? k : ((k >= Ys - my && k < Ys) ? k - (Ys - 2 * my) : -1); };
        const int kmax = (Ys == Y) ? my : Y / 2;
        for (int ky = j; ky <= kmax; ky += G) {
            const int kyn = ky ? Y - ky : 0;
            const int sa = slot(ky), sb = slot(kyn);
            const bool ha = sa >= 0, hb = sb >= 0;
            if (!ha && !hb) continue;      // the buffer is pre-zeroed
            const cf* wa = wq + (size_t)(ha ? sa : 0) * mt;
            const cf* wb = wq + (size_t)(hb ? sb : 0) * mt;
            float g0x = 0.f, g0y = 0.f, g1x = 0.f, g1y = 0.f;
            for (int k = 0; k < mt; ++k) {
                cf av = wa[k];
                if (!ha) av = mk<float>(0.f, 0.f);
                cf bv = wb[k];
                if (!hb) bv = mk<float>(0.f, 0.f);
                const float ux = av.x + bv.x, uy = av.y + bv.y, dx = av.x - bv.x, dy = av.y - bv.y;
                const cf E0 = e0[k], E1 = e1[k];
                g0x += E0.x * ux - E0.y * uy;  g0y += E0.y * dx + E0.x * dy;
                g1x += E1.x * ux - E1.y * uy;  g1y += E1.y * dx + E1.x * dy;
            }
            if (!pair) { g1x = 0.f; g1y = 0.f; }
            lds[ky] = mk<float>((g0x - g1y) * hsc, (g0y + g1x) * hsc);
            if (kyn != ky) lds[kyn] = mk<float>((g0x + g1y) * hsc, (g1x - g0y) * hsc);
        }
    }
    group_sync<false>();
    cf x[EPT];
#pragma unroll
    for (int t = 0; t < EPT; ++t) x[t] = lds[j + t * G];
    group_sync<false>();
    tile_fft<float, Y, EPT, +1, 1, true, false>(x, lds, tw_y, j, 0);
    __syncthreads();  // all exchanges done: the buffers become the output slabs [y][t_keep]
    {
        float* oslab = reinterpret_cast<float*>(ex + (size_t)s * P * Y) + (size_t)j * t_keep + 2 * p;
#pragma unroll
        for (int t = 0; t < EPT; ++t)
            *reinterpret_cast<float2*>(oslab + (size_t)t * G * t_keep) = make_float2(x[t].x, x[t].y);
    }
    __syncthreads();
    // ---- pointwise block on the row's Y * t_keep points, two neighbouring points per lane
    const size_t chan_in = (size_t)P * Y * sizeof(cf) / sizeof(float);   // floats between the LDS slabs of two channels
    const float* tile = reinterpret_cast<const float*>(ex);
    for (int item = threadIdx.x; item < n_items; item += nthr) {
        const int pt = 2 * item;
        // The weights reach the FMAs through the scalar unit, as in k_pointwise.  That needs (i) the compiler's proof
        // that the stores of an earlier trip cannot have changed them -- the tables are separate __restrict__ kernel
        // arguments, not fields of `a` -- and (ii) the ~CI (2 CM + CO) loop-invariant scalar loads NOT hoisted out of
        // this one- or two-trip loop (they would be spilled): an opaque zero per trip is added to the pointers.
        int zero = 0;
        asm volatile("" : "+s"(zero));
        a.w1 = pw_w1 + zero;
        a.b1 = pw_b1 ? pw_b1 + zero : nullptr;
        a.w2t = pw_w2t + zero;
        a.b2 = pw_b2 ? pw_b2 + zero : nullptr;
        a.wst = pw_wst ? pw_wst + zero : nullptr;
        a.bs = pw_bs ? pw_bs + zero : nullptr;
        vf xin[CI], o[CO];
#pragma unroll
        for (int i = 0; i < CI; ++i) xin[i] = *reinterpret_cast<const vf*>(tile + (size_t)i * chan_in + pt);
        pw_core<CI, CM, CO, true, vf>(a, b, xin, o);
        if (a.skip_mode == 1) {
            vf sv[CI];
            if (item == (int)threadIdx.x) {
#pragma unroll
                for (int i = 0; i < CI; ++i) sv[i] = sv0[i];
            } else {
                const float* sb = a.s + ((size_t)b * CI * X + xrow) * slab_pts + pt;
#pragma unroll
                for (int i = 0; i < CI; ++i) sv[i] = *reinterpret_cast<const vf*>(sb + (size_t)i * X * slab_pts);
            }
            pw_skip_conv<CI, CO, vf>(a, sv, o);
        } else if (a.skip_mode == 2) {
            const int y = pt / t_keep;
            const size_t sP = (size_t)X * Y * a.sT;
            const float* sb = a.s + (size_t)b * CO * sP + ((size_t)xrow * Y + y) * a.sT + (a.sT - 1);
#pragma unroll
            for (int c = 0; c < CO; ++c) o[c] += (vf)sb[(size_t)c * sP];
        }
        float* ob = a.out + ((size_t)b * CO * X + xrow) * slab_pts + pt;
#pragma unroll
        for (int c = 0; c < CO; ++c) *reinterpret_cast<vf*>(ob + (size_t)c * X * slab_pts) = pw_act(o[c], a.act2);
    }
}

template <int CI, int CM, int CO, bool HAS_L1>
static int launch_pw(const PwArgs& a, int batch, hipStream_t st) {
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

template <int Y, int CI, int CM, int CO>
static int launch_inv_ty_pw(const tcfd_fno_plan* p, const cf* w2, const PwArgs& a, int batch, int t_keep, float scale,
                            bool probe, hipStream_t st) {
    constexpr int EPT = TyCfg2<Y>::EPT, G = TyCfg2<Y>::G;
    const int P = t_keep / 2;
    const size_t lds = ((size_t)CI * P * Y + (size_t)CI * 2 * p->my * p->mt + (size_t)t_keep * p->mt) * sizeof(cf);
    if ((t_keep & 1) || CI * P * G > 1024 || lds > 160 * 1024)
        return FAIL(TCFD_EINVAL, "fno: fused layer tail not instantiated (row of %d channels does not fit a workgroup)", CI);
    if (probe) return 0;
    auto kern = k_inv_ty_pw<Y, EPT, CI, CM, CO>;
    int rc;
    if ((rc = set_lds_attr(kern, lds))) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)batch * p->X)), dim3(CI * P * G), lds, st, w2, a, a.w1, a.b1, a.w2t, a.b2,
                       a.wst, a.bs, (const cf*)p->tw_y,
                       (const cf*)p->tw_ti, p->T_out, t_keep, p->mt, p->my, scale, P, p->X, p->Ys);
    HIP_TRY(hipGetLastError());
    return 0;
}

// probe = true: only answer whether the combination is instantiated and fits (nothing is launched)
static int do_inv_ty_pw(const tcfd_fno_plan* p, const cf* w2, const PwArgs& a, int batch, int ci, int cm, int co, int t_keep,
                        float scale, bool probe, hipStream_t st) {
#define TYPW_CASE(Y_, CI_, CM_, CO_)                                                                       \
    if (p->Y == Y_ && ci == CI_ && co == CO_ && (CM_ == 0 || cm == CM_))                                    \
        return launch_inv_ty_pw<Y_, CI_, CM_, CO_>(p, w2, a, batch, t_keep, scale, probe, st);
    // widths above 10 spill under the 128-register cap of the row-sized workgroup: not instantiated
#define TYPW_WIDTHS(Y_) TYPW_CASE(Y_, 10, 40, 10) TYPW_CASE(Y_, 8, 0, 8)
    TYPW_WIDTHS(64) TYPW_WIDTHS(256)
#undef TYPW_WIDTHS
#undef TYPW_CASE
    return FAIL(TCFD_EINVAL, "fno: fused layer tail not instantiated (Y = %d, channels %d -> %d -> %d)", p->Y, ci, cm, co);
}

// Spectral convolution + the layer's pointwise block:  out = act2( W2 . act1(W1 . conv(v) + b1) + b2 + skip term ), the
// convolution's output never leaves the chip (k_inv_ty_pw).  Arguments as tcfd_fno_spectral_conv followed by those of
// tcfd_fno_pointwise (its `x` is the convolution, ci = cout); out is (batch, co_pw, X, Y, t_keep).  Returns TCFD_EINVAL
// with "not instantiated" in the message -- before anything is launched -- when the combination is not covered; the
// caller then makes the two calls.
extern "C" int tcfd_fno_spectral_conv_pointwise(const tcfd_fno_plan* p, const void* v, const void* const* weights,
                                                const void* const* bias, float delta, void* out, int batch, int cin,
                                                int cout, int t_keep, float fwd_scale, float inv_scale, int use_mfma,
                                                void* ws, size_t ws_bytes, const void* skip, const void* w1,
                                                const void* b1, const void* w2t, const void* b2, const void* wst,
                                                const void* bs, int cm, int co_pw, int act1, int act2, int skip_mode,
                                                int skip_T, void* stream) {
    if (!p || !v || !weights || !out || !w1 || !w2t) return FAIL(TCFD_EINVAL, "fno_spectral_conv_pointwise: null argument");
    if (p->dtype != TCFD_C64) return FAIL(TCFD_EINVAL, "fno: fused layer tail not instantiated (fp32 plans only)");
    if (batch <= 0 || cin <= 0 || cout <= 0 || t_keep <= 0 || t_keep > p->T_out)
        return FAIL(TCFD_EINVAL, "fno_spectral_conv_pointwise: bad sizes");
    if (skip_mode && !skip) return FAIL(TCFD_EINVAL, "fno_spectral_conv_pointwise: skip input missing");
    if (skip_mode == 2 && skip_T <= 0) return FAIL(TCFD_EINVAL, "fno_spectral_conv_pointwise: bad skip_T");
    if (((uintptr_t)out | (uintptr_t)skip) % 8 != 0) return FAIL(TCFD_EINVAL, "fno: fused layer tail not instantiated (unaligned)");
    PwArgs a;
    a.pe = nullptr; a.x = nullptr; a.s = (const float*)skip; a.out = (float*)out;
    a.w1 = (const float*)w1; a.b1 = (const float*)b1; a.w2t = (const float*)w2t; a.b2 = (const float*)b2;
    a.wst = (const float*)wst; a.bs = (const float*)bs;
    a.P = (long)p->X * p->Y * t_keep; a.T = t_keep; a.sT = skip_T; a.act1 = act1; a.act2 = act2; a.skip_mode = skip_mode;
    a.w2_bstride = 0; a.b2_bstride = 0; a.cm = cm;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = do_inv_ty_pw(p, nullptr, a, batch, cout, cm, co_pw, t_keep, inv_scale, true, st))) return rc;
    const size_t need = tcfd_fno_workspace_bytes(p, batch, cin, cout);
    if (!ws || ws_bytes < need) return FAIL(TCFD_EWORKSPACE, "workspace %zu B < required %zu B", ws_bytes, need);
    const size_t Q = (size_t)2 * p->my * p->mt;
    const int cmax = std::max(cin, cout);
    unsigned char* base = (unsigned char*)ws;
    cf* W = (cf*)base;
    cf* V = (cf*)(base + al256((size_t)batch * cmax * p->X * Q * sizeof(cf)));
    cf* O = (cf*)((unsigned char*)V + al256((size_t)batch * cin * 2 * p->mx * Q * sizeof(cf)));
    if ((rc = do_fwd_ty<float>(p, (const float*)v, W, (long)batch * cin * p->X, fwd_scale, st))) return rc;
    if ((rc = do_fwd_x<float>(p, W, V, (long)batch * cin, st))) return rc;
    if ((rc = do_contract<float>(contract_args<float>(V, O, weights, bias, delta, batch, cin, cout, p->mx, p->my, p->mt), use_mfma, st)))
        return rc;
    if ((rc = do_inv_x<float>(p, O, W, (long)batch * cout, st))) return rc;
    return do_inv_ty_pw(p, W, a, batch, cout, cm, co_pw, t_keep, inv_scale, false, st);
}

// Returns TCFD_EINVAL (with a message) for channel combinations that are not instantiated; the caller then
// uses its own pointwise modules.
static int pw_dispatch(PwArgs a, int batch, int ci, int cm, int co, hipStream_t st);
extern "C" int tcfd_fno_pointwise(const void* x, const void* skip, void* out, const void* w1, const void* b1,
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

// Two-layer form, FOUR waves on the same 64 points.  The one-wave kernel above runs at 1.5 waves per SIMD (21 KB of
// staging per wave).  Here a 256-thread workgroup owns 64 points: every wave holds the points' inputs, takes a quarter of
// the hidden units (z1, h, its share of z2, g1 and its share of dx -- the partial sums meet in LDS), one tile of each
// weight-gradient product on MFMA, and a quarter of the output channels.  One staging region per workgroup instead of per
// wave: ~4 waves per SIMD, a quarter of the weight reads per wave.  Same partial-sum layout (one row per WORKGROUP).
// NW waves share the 64 points of a chunk: 4 (each wave a quarter of the hidden units, one MFMA tile per wave) or 2
// (half the hidden units, two tiles per wave: half the barrier partners, twice the staging per wave)
template <int CI, int CM, int CO, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_pointwise_bwd4(PwBwdArgs a) {
    using Gm = PwBwdGeom<CI, CM, CO, true>;
    using Wm = PwBwdW<CI, CM, CO, true>;
    constexpr int PITCH = Gm::PITCH;
    constexpr int TO = Gm::COP / 16, TB = Gm::CB / 16, TI = Gm::CIP / 16, TM = Gm::CM1 / 16;
    constexpr int RI = Wm::RI, RO = Wm::RO;
    constexpr int MQ = CM / NW;                      // hidden units per wave
    constexpr int RED = CO > CI ? CO : CI;           // rows per wave of the cross-wave reduction scratch
    static_assert((NW == 2 || NW == 4) && CM % NW == 0 && TO == 1 && TB <= 4 && TM <= 4 && TI == 1, "k_pointwise_bwd4 geometry");
    constexpr int TA_W = (TB + NW - 1) / NW, TB_W = (TM + NW - 1) / NW;   // MFMA tiles per wave: tile t belongs to wave t % NW
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float* Wl = reinterpret_cast<float*>(smem_raw);
    float* L0 = Wl + ((Wm::TOTAL + 3) & ~3);         // g2, later [x, 1]
    float* L1 = L0 + Gm::R0 * PITCH;                 // [h, 1, s], later g1 over h
    float* RD = L1 + Gm::CB * PITCH;                 // [NW][RED][PITCH] partial sums of z2, later of dx
    for (int i = threadIdx.x; i < Wm::TOTAL; i += blockDim.x) Wl[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < CM * CI; i += blockDim.x) Wl[Wm::W1 + (i / CI) * RI + i % CI] = a.w1[i];
    if (a.b1) for (int i = threadIdx.x; i < CM; i += blockDim.x) Wl[Wm::B1 + i] = a.b1[i];
    for (int i = threadIdx.x; i < CM * CO; i += blockDim.x) Wl[Wm::W2 + (i / CO) * RO + i % CO] = a.w2t[i];
    for (int i = threadIdx.x; i < CO; i += blockDim.x)
        Wl[Wm::B2 + i] = (a.b2 ? a.b2[i] : 0.f) + ((a.skip_mode == 1 && a.bs) ? a.bs[i] : 0.f);
    if (a.skip_mode == 1)
        for (int i = threadIdx.x; i < CI * CO; i += blockDim.x) Wl[Wm::WS + (i / CO) * RO + i % CO] = a.wst[i];
    __syncthreads();
    // weights come from the LDS copy (uniform-address ds_reads): read from global memory inside the persistent loop the
    // compiler cannot prove them invariant against the kernel's own stores and emits VECTOR loads (50.9 vs 34.9 ms per
    // training step)
    auto W1row = [&](int m) -> const float* { return Wl + Wm::W1 + m * RI; };
    auto W2row = [&](int m) -> const float* { return Wl + Wm::W2 + m * RO; };
    auto WSrow = [&](int i) -> const float* { return Wl + Wm::WS + i * RO; };
    auto B1at = [&](int m) -> float { return Wl[Wm::B1 + m]; };
    const int kq = lane >> 4, kc = lane & 15;
    f4 accA[TA_W], accB[TB_W];
#pragma unroll
    for (auto& v : accA) v = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (auto& v : accB) v = f4{0.f, 0.f, 0.f, 0.f};
    const long c_first = a.per_sample ? ((long)blockIdx.x % a.batch) * a.chunks_per_batch + blockIdx.x / a.batch : blockIdx.x;
    const long c_stride = a.per_sample ? gridDim.x / a.batch : gridDim.x;
    const long c_end = a.per_sample ? ((long)blockIdx.x % a.batch + 1) * a.chunks_per_batch : a.total_chunks;
    const bool from_h = a.act1 == 0 || a.act1 == 1 || a.act1 == 4;
    for (long chunk = c_first; chunk < c_end; chunk += c_stride) {
        const long b = chunk / a.chunks_per_batch;
        const long p = (chunk - b * a.chunks_per_batch) * 64 + lane;
        const bool live = p < a.P;
        const long pc = live ? p : a.P - 1;
        float x[CI], g2[CO], z2[CO];
        const float* xb = a.x + (size_t)b * CI * a.P + pc;
        const float* db = a.dout + (size_t)b * CO * a.P + pc;
#pragma unroll
        for (int i = 0; i < CI; ++i) x[i] = xb[(size_t)i * a.P];
#pragma unroll
        for (int c = 0; c < CO; ++c) g2[c] = live ? db[(size_t)c * a.P] : 0.f;
#pragma unroll
        for (int c = 0; c < CO; ++c) z2[c] = 0.f;
        // this wave's hidden units
#pragma unroll 2
        for (int mm = 0; mm < MQ; ++mm) {
            const int m = wave * MQ + mm;
            float z = B1at(m);
            const float* w1 = W1row(m);
#pragma unroll
            for (int i = 0; i < CI; ++i) z = fmaf(w1[i], x[i], z);
            const float h = live ? pw_act(z, a.act1) : 0.f;
            L1[m * PITCH + lane] = h;
            const float* w2 = W2row(m);
#pragma unroll
            for (int c = 0; c < CO; ++c) z2[c] = fmaf(w2[c], h, z2[c]);
        }
#pragma unroll
        for (int c = 0; c < CO; ++c) RD[(wave * RED + c) * PITCH + lane] = z2[c];
        // the constant-1 channel and the skip input rows of the second operand (wave 3 and wave 2: spread the stores)
        if (wave == NW - 1) L1[CM * PITCH + lane] = live ? 1.f : 0.f;
        float sv[CI];
        if (a.skip_mode == 1) {
            const float* sb = a.s + (size_t)b * CI * a.P + pc;
#pragma unroll
            for (int i = 0; i < CI; ++i) sv[i] = live ? sb[(size_t)i * a.P] : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < CI; ++i) sv[i] = 0.f;
        }
        if (wave == (NW > 2 ? 2 : 0)) {
#pragma unroll
            for (int i = 0; i < CI; ++i) L1[(CM + 1 + i) * PITCH + lane] = sv[i];
        }
        __syncthreads();
        // z2 = bias + skip part + the four partial sums; g2 (every wave, the same values)
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            float v = Wl[Wm::B2 + c];
#pragma unroll
            for (int w = 0; w < NW; ++w) v += RD[(w * RED + c) * PITCH + lane];
            z2[c] = v;
        }
        if (a.skip_mode == 1) {
#pragma unroll
            for (int i = 0; i < CI; ++i) {
                const float* ws = WSrow(i);
#pragma unroll
                for (int c = 0; c < CO; ++c) z2[c] = fmaf(ws[c], sv[i], z2[c]);
            }
        } else if (a.skip_mode == 2) {
            const long xy = pc / a.T;
            const long sP = (a.P / a.T) * a.sT;
            const float* sb = a.s + (size_t)b * CO * sP + xy * a.sT + (a.sT - 1);
#pragma unroll
            for (int c = 0; c < CO; ++c) z2[c] += sb[(size_t)c * sP];
        }
#pragma unroll
        for (int c = 0; c < CO; ++c) g2[c] *= pw_dact(z2[c], a.act2);
        if (wave == 0) {
#pragma unroll
            for (int c = 0; c < CO; ++c) L0[c * PITCH + lane] = g2[c];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TA_W; ++u) {   // [dW2 | db2 | dWs] tiles wave, wave + NW, ...
            const int tile = wave + u * NW;
            if (tile < TB) {
#pragma unroll 4
                for (int q = 0; q < 16; ++q) {
                    const float av = L0[kc * PITCH + 4 * q + kq];
                    const float bv = L1[(16 * tile + kc) * PITCH + 4 * q + kq];
                    accA[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accA[u], 0, 0, 0);
                }
            }
        }
        __syncthreads();   // the products have read h and g2: h rows become g1, the g2 rows become [x, 1]
        float dx[CI];
#pragma unroll
        for (int i = 0; i < CI; ++i) dx[i] = 0.f;
#pragma unroll 2
        for (int mm = 0; mm < MQ; ++mm) {
            const int m = wave * MQ + mm;
            const float* w1 = W1row(m);
            const float h = L1[m * PITCH + lane];
            float d1;
            if (from_h) {
                d1 = a.act1 == 1 ? (h > 0.f ? 1.f : 0.f) : (a.act1 == 4 ? 1.f - h * h : 1.f);
            } else {
                float z = B1at(m);
#pragma unroll
                for (int i = 0; i < CI; ++i) z = fmaf(w1[i], x[i], z);
                d1 = pw_dact(z, a.act1);
            }
            const float* w2 = W2row(m);
            float dh = 0.f;
#pragma unroll
            for (int c = 0; c < CO; ++c) dh = fmaf(w2[c], g2[c], dh);
            const float g1 = dh * d1;
            L1[m * PITCH + lane] = g1;
#pragma unroll
            for (int i = 0; i < CI; ++i) dx[i] = fmaf(w1[i], g1, dx[i]);
        }
#pragma unroll
        for (int i = 0; i < CI; ++i) RD[(wave * RED + i) * PITCH + lane] = dx[i];
        if (wave == 1) {
#pragma unroll
            for (int i = 0; i < CI; ++i) L0[i * PITCH + lane] = live ? x[i] : 0.f;
            L0[CI * PITCH + lane] = live ? 1.f : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TB_W; ++u) {   // [dW1 | db1] tiles wave, wave + NW, ...
            const int tile = wave + u * NW;
            if (tile < TM) {
#pragma unroll 4
                for (int q = 0; q < 16; ++q) {
                    const float bv = L0[kc * PITCH + 4 * q + kq];
                    const float av = L1[(16 * tile + kc) * PITCH + 4 * q + kq];
                    accB[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accB[u], 0, 0, 0);
                }
            }
        }
        // outputs: channel i is summed and stored by wave i % NW
        if (live) {
            if (a.dx) {
                float* dxb = a.dx + (size_t)b * CI * a.P + p;
#pragma unroll
                for (int i = 0; i < CI; ++i)
                    if ((i % NW) == wave) {
                        float v = 0.f;
#pragma unroll
                        for (int w = 0; w < NW; ++w) v += RD[(w * RED + i) * PITCH + lane];
                        dxb[(size_t)i * a.P] = v;
                    }
            }
            if (a.skip_mode == 1 && a.ds) {
                float* dsb = a.ds + (size_t)b * CI * a.P + p;
#pragma unroll
                for (int i = 0; i < CI; ++i)
                    if ((i % NW) == wave) {
                        const float* ws = WSrow(i);
                        float v = 0.f;
#pragma unroll
                        for (int c = 0; c < CO; ++c) v = fmaf(ws[c], g2[c], v);
                        dsb[(size_t)i * a.P] = v;
                    }
            } else if (a.skip_mode == 2 && a.ds) {
                float* dsb = a.ds + (size_t)b * CO * a.P + p;
#pragma unroll
                for (int c = 0; c < CO; ++c)
                    if ((c % NW) == wave) dsb[(size_t)c * a.P] = g2[c];
            }
        }
        __syncthreads();   // the next chunk overwrites the staging rows and the reduction scratch
    }
    // partial sums of this workgroup: A (COP x CB) | B (CM1 x CIP), tile `wave` of each from wave `wave`
    float* out = a.partials + (size_t)blockIdx.x * Gm::TOTAL;
#pragma unroll
    for (int u = 0; u < TA_W; ++u) {
        const int tile = wave + u * NW;
        if (tile < TB) {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(4 * kq + r) * Gm::CB + 16 * tile + kc] = accA[u][r];
        }
    }
    float* o1 = out + Gm::N_A;
#pragma unroll
    for (int u = 0; u < TB_W; ++u) {
        const int tile = wave + u * NW;
        if (tile < TM) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o1[(16 * tile + 4 * kq + r) * Gm::CIP + kc] = accB[u][r];
        }
    }
}

// activation and its derivative of a set of D tiles, the switch on the (run-time) activation OUTSIDE the element loops:
// one uniform branch per call instead of one per element
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
template <int N, typename V4>
__device__ __forceinline__ void pw_act_tiles(const V4 (&z)[N], V4 (&h)[N], V4 (&d)[N], int act) {
#define PW_ACT_ALL(ACT_)                                                   \
    _Pragma("unroll") for (int t = 0; t < N; ++t)                          \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                    \
            float hv, dv;                                                  \
            pw_act_pair<ACT_>(z[t][r], hv, dv);                            \
            h[t][r] = hv;                                                  \
            d[t][r] = dv;                                                  \
        }
    switch (act) {
        case 1: PW_ACT_ALL(1) break;
        case 2: PW_ACT_ALL(2) break;
        case 3: PW_ACT_ALL(3) break;
        case 4: PW_ACT_ALL(4) break;
        default: PW_ACT_ALL(0) break;
    }
#undef PW_ACT_ALL
}

// ------------------------------------------------------------------ two-layer backward, everything on MFMA
// k_pointwise_bwd4 spends its time in the LDS pipe (weights, channel-major staging of the points for the weight-gradient
// products, cross-wave reductions: ~1500 DS instructions and 5 workgroup barriers per 64 points).  Here a wave owns 16
// points at a time and NOTHING goes through LDS: the weights live in registers as MFMA operand fragments, and every
// product of the block -- forward recompute, input gradients, weight gradients -- is a chain of v_mfma_f32_16x16x4_f32
// whose D registers are fed straight back as operands.
//
// Lane l = (q, c) = (l >> 4, l & 15).  The instruction takes A[row c][k q] and B[k q][col c] from lane (q, c) and leaves
// D[row 4q + r][col c] in register r.  Hence a D tile of a matrix M (rows R, columns C) IS, register r by register r,
//   * the B operand of  X . M   (k-step r contracts over rows {4q + r})                 -> D tile of X M
//   * the A operand of  M^T . Y (same k-steps)                                          -> D tile of M^T Y
// provided the other operand's fragment lists its k index in the same order (weights: laid out that way once, at kernel
// start).  Both contract over M's ROW index.  The block needs contractions over channels (the chain z1 -> z2 -> g2 -> dh
// -> dx) and over points (weight gradients = sums over points of outer products), so every intermediate is produced in
// two orientations:
//   "O1"  rows = channels, cols = the 16 points     z1 = W1' [x;1],  z2 = W2 h + Ws' [s;1],  dh = W2^T g2
//   "OT"  rows = the 16 points, cols = channels     z1^T = [x;1]^T W1'^T (same fragments, operands swapped),
//                                                   g2^T = g2^T I (identity fragment), dh^T = g2^T W2,
//                                                   dx^T = g1^T W1, ds^T = g2^T Ws           (O1 tiles as A operands)
// and the weight gradients are  dW2 += (g2^T)^T h^T,  [dWs | db2] += (g2^T)^T [s;1]^T,  [dW1 | db1] += (g1^T)^T [x;1]^T
// with OT tiles as both operands.  Biases ride as a constant-1 channel.  93 MFMAs per 16 points at width 10 (45 for the
// chain, 20 for the second orientation, 28 for the weight gradients; 105 before the row map pw_tile_row): the kernel is bound by the matrix pipe (~1.8 ms for
// the (32, 10, 256, 256, 10) activations of config 5) instead of 4.3 ms of LDS issue.  Inputs are read in the two
// fragment layouts they are needed in ([ch 4j + q][pt c]: 64-byte rows; [ch c][pt 4q .. 4q + 3]: 16-byte lanes), the
// second read of a line hits the vector cache; dx / ds leave as 16-byte lanes.  One row of partial sums per wave.
// Measured at config 5 (ReLU, per launch): LDS-staged two-wave kernel 3.83 ms -> first version 4.95 (a switch on the run-time
// activation per element) -> 3.51 (one switch per tile set) -> 3.02 (buffer loads: the prefetch stays in flight) -> 2.87
// (sched_barrier behind the prefetch) -> 2.82 (issue order).  Three waves per SIMD (weights re-read from LDS, 168 registers)
// gain nothing (3.06): what is left is the ~40-cycle gap every time a product's D tile turns into the next product's operand.
// Row map of a 16-row D tile that holds only N < 16 channels: D register r of lane (q, c) is row 4q + r, and a k-step of a
// chained product contracts over the four rows {4q + r : q} of ONE register index r.  Channels are therefore dealt to the
// rows with r < RV = ceil(N / 4) only (row 4q + r <-> channel q RV + r), so that the k-steps r >= RV of a partly filled
// tile hold nothing and are never issued: width 10 / 40 / 10 has 3 of 4 steps over the output channels and 10 of 12 over the
// hidden ones, 93 instead of 105 MFMAs per 16 points.  (The order of channels inside a tile is free: it only has to be the
// same in the weight fragments, the loads and the rows of the partial sums.)
template <int N>
__device__ __forceinline__ int pw_tile_row(int rho) {
    constexpr int RV = (N + 3) / 4;
    const int i = (rho >> 2) * RV + (rho & 3);
    return ((rho & 3) < RV && i < N) ? i : -1;
}
constexpr int pw_tile_steps(int n_total, int t) { return n_total - 16 * t >= 16 ? 4 : (n_total - 16 * t + 3) / 4; }
template <int CM>
__device__ __forceinline__ int pw_hid_row(int t, int rho) {      // hidden channel of row rho of tile t, or -1
    if (CM - 16 * t >= 16) return 16 * t + rho;
    constexpr int REM = CM % 16 ? CM % 16 : 16;
    const int i = pw_tile_row<REM>(rho);
    return i < 0 ? -1 : 16 * t + i;
}

// ACT = 1: both activations are ReLU, known at compile time.  Vector instructions do not hide behind the matrix instructions
// on this part (profiles/r03_mfma_f32_vs_valu_overlap.txt), so every one of them counts: the generic path spends 5 per
// hidden element (compare, two v_max -- hipcc canonicalises the operand of the select first --, the 0 / 1 derivative, and the
// product with it later); here h = v_max_i32(0, z) is ONE instruction and the derivative is never formed -- z itself is kept and
// the cotangent passes through a compare + select: 3 per element, 56 + 28 instructions less per group of 16 points.
__device__ __forceinline__ float pw_relu(float z) {      // max(0, z) on the bit pattern: positive floats are positive integers
    const int zi = __float_as_int(z);                     // (an inline-asm v_max_f32 is invisible to hipcc's MFMA hazard handling:
    return __int_as_float(zi > 0 ? zi : 0);               //  it read the accumulators before they were written)
}
// YMASK (ReLU / ReLU only): a.out holds the block's forward output y = max(0, z2).  The backward needs z2 for nothing but the
// sign that gates the cotangent, and y > 0 <=> z2 > 0 -- so the whole z2 chain (the skip product and W2 h: 3 + 10 of the 93
// MFMAs at width 10, and the O1 copy of h that only it consumes) is replaced by four 4-byte loads per lane.  It is also the
// mask the forward kernel actually applied (its FMA order differs from the MFMA chain's: a z2 within rounding of 0 could come
// out on the other side here).
// YMASK = 2 goes on from there.  (i) g2^T is formed from a second read of dout / out in the [ch c][pt 4q .. 4q + 3] layout (16-byte
// lanes of lines the first read brought in) instead of g2 . I: 3 MFMAs less.  (ii) The O1 side of the hidden layer -- z1 (for its
// mask), dh = W2^T g2 and g1 = dh (.) mask, 9 + 9 MFMAs -- exists only to feed dx^T = g1^T W1 with an A operand; g1 in that
// orientation is the TRANSPOSE of the OT tile g1^T the weight gradient needs anyway, and a transposition is one product with
// the identity per k-step (12).  71 instead of 80 MFMAs per 16 points, and the O1 compares / selects go too.
// TCFD_PWB_NT=1 (build time): the kernel's tensors as streaming data -- buffer loads with the nt bit (aux 2), non-temporal stores
#ifndef TCFD_PWB_NT
#define TCFD_PWB_NT 0
#endif
#if TCFD_PWB_NT
#define PWB_AUX 2
#define PWB_STORE(p_, v_) __builtin_nontemporal_store((v_), (p_))
#else
#define PWB_AUX 0
#define PWB_STORE(p_, v_) (*(p_) = (v_))
#endif
template <int CI, int CM, int CO, int MODE, int ACT = -1, int YMASK = 0>
__global__ __launch_bounds__(256, 2) void k_pointwise_bwd_mfma(PwBwdArgs a) {
    static_assert(!YMASK || ACT == 1, "the output mask stands in for z2 only under ReLU");
    using Gm = PwBwdGeom<CI, CM, CO, true>;
    constexpr int KI = (CI + 1 + 3) / 4;        // k-steps over [x ; 1]
    constexpr int TM = (CM + 15) / 16;          // 16-row tiles of the hidden layer
    constexpr int RO = (CO + 3) / 4;            // k-steps over the output channels (pw_tile_row)
    static_assert(CI + 1 <= 16 && CO <= 16 && TM <= 4, "k_pointwise_bwd_mfma geometry");
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
    constexpr int mode = MODE;                  // = a.skip_mode (compile time: the loads must be straight-line code)
    // ---- weight fragments (registers, for the whole kernel)
    float W1a[TM][KI], W2a[TM][4], Wsa[KI], Idf[4], W2b[TM][4], W1b[TM][4], Wsb[4];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int hid = pw_hid_row<CM>(t, c);                      // hidden channel of tile row / column c (or -1)
#pragma unroll
        for (int j = 0; j < KI; ++j) {
            const int k = 4 * j + q;
            float v = 0.f;
            if (hid >= 0) v = k < CI ? a.w1[hid * CI + k] : ((k == CI && a.b1) ? a.b1[hid] : 0.f);
            W1a[t][j] = v;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hq = pw_hid_row<CM>(t, 4 * q + r), cq = pw_tile_row<CO>(4 * q + r), cc = pw_tile_row<CO>(c);
            W2a[t][r] = (cc >= 0 && hq >= 0) ? a.w2t[hq * CO + cc] : 0.f;        // W2[co of row c][hid of row 4q+r]
            W2b[t][r] = (cq >= 0 && hid >= 0) ? a.w2t[hid * CO + cq] : 0.f;      // W2[co of row 4q+r][hid of row c]
            W1b[t][r] = (hq >= 0 && c < CI) ? a.w1[hq * CI + c] : 0.f;           // W1[hid of row 4q+r][ci c]
        }
    }
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        const int k = 4 * j + q;
        float v = 0.f;
        const int cc = pw_tile_row<CO>(c);
        if (cc >= 0) {
            if (k < CI) v = mode == 1 ? a.wst[k * CO + cc] : 0.f;
            else if (k == CI) v = (a.b2 ? a.b2[cc] : 0.f) + ((mode == 1 && a.bs) ? a.bs[cc] : 0.f);
        }
        Wsa[j] = v;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cq = pw_tile_row<CO>(4 * q + r);
        Idf[r] = c == cq ? 1.f : 0.f;                                // g2^T comes out with its columns in natural order
        Wsb[r] = (mode == 1 && cq >= 0 && c < CI) ? a.wst[c * CO + cq] : 0.f;
    }
    float Idp[4];                                                    // identity over the 16 points, k-step r lists rows {4q + r}
#pragma unroll
    for (int r = 0; r < 4; ++r) Idp[r] = c == 4 * q + r ? 1.f : 0.f;
    f4 accW2[TM], accW1[TM], accWs = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < TM; ++t) accW2[t] = accW1[t] = f4{0.f, 0.f, 0.f, 0.f};

    const int gpb = (int)((a.P + 15) / 16);                 // groups of 16 points per batch element (the launcher checks
    const int total = gpb * a.batch;                        // that the group count fits 31 bits)
    // the wave index through readfirstlane: hipcc then KNOWS that the group index and everything derived from it (batch element,
    // first point, base offsets: an integer division and several 32 / 64-bit multiplies per group) is wave uniform and puts it on
    // the scalar unit -- as vector code those were ~40 instructions, a quarter of them quarter rate, that the matrix pipe waits for
    const int wid = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wstride = gridDim.x * 4;

    struct In {
        float xa[KI], sa[KI], dz[4], sl[4], yo[4];
        f4 xb, sb, dzb, yob;
    };
    // Loads go through buffer descriptors (one per tensor, built from the kernel arguments): a lane that has nothing to
    // read -- padding channel, point beyond P -- passes an offset beyond the buffer and gets 0 from the bounds check.  No
    // lane condition ever guards a load, so the loads of the NEXT group are straight-line code the compiler counts
    // (s_waitcnt vmcnt(N) at the first use, one iteration later).  With plain conditional loads every load sat in its
    // own exec-masked branch and the prefetch ended in s_waitcnt vmcnt(0) in front of the current group's first MFMA:
    // 3.5 ms per launch, every iteration paid a memory round trip.
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    constexpr unsigned OOB = 0x80000000u;                        // the launcher checks that every tensor is < 2 GiB
    const unsigned P4 = (unsigned)a.P * 4u;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)((size_t)a.batch * CI * a.P * 4), 0x00020000);
    const auto rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dout), 0, (int)((size_t)a.batch * CO * a.P * 4), 0x00020000);
    const size_t s_bytes = mode == 1 ? (size_t)a.batch * CI * a.P * 4 : (mode == 2 ? (size_t)a.batch * CO * (a.P / a.T) * a.sT * 4 : 0);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mode ? a.s : a.x), 0, (int)s_bytes, 0x00020000);
    const auto ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(YMASK ? a.out : a.dout), 0, (int)((size_t)a.batch * CO * a.P * 4), 0x00020000);
    // per-lane constants of the two layouts: channel row offsets (or OOB) and the constant-1 channel
    unsigned ka_off[KI];
    float ka_one[KI];
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        const int k = 4 * j + q;
        ka_off[j] = k < CI ? (unsigned)k * P4 : OOB;
        ka_one[j] = k == CI ? 1.f : 0.f;
    }
    const unsigned cb_off = c < CI ? (unsigned)c * P4 : OOB;
    const float cb_one = c == CI ? 1.f : 0.f;
    unsigned co_off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cq = pw_tile_row<CO>(4 * q + r);
        co_off[r] = cq >= 0 ? (unsigned)cq * P4 : OOB;
    }
    const unsigned sP4 = mode == 2 ? (unsigned)((a.P / a.T) * a.sT) * 4u : 0u;
    unsigned cs_off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cq = pw_tile_row<CO>(4 * q + r);
        cs_off[r] = cq >= 0 ? (unsigned)cq * sP4 : OOB;
    }
    auto load = [&](int G, In& in) {
        const int b = G / gpb;
        const unsigned p0 = (unsigned)(G - b * gpb) * 16u;
        const unsigned pa = p0 + c, pb = p0 + 4 * q;
        const bool live_a = pa < (unsigned)a.P, live_b = pb < (unsigned)a.P;   // P % 4 == 0: a 16-byte lane is all live or all dead
        const unsigned base_i = (unsigned)b * CI * P4, base_o = (unsigned)b * CO * P4;
        const unsigned oa = live_a ? base_i + pa * 4u : OOB, ob = live_b ? base_i + pb * 4u : OOB;
        const float one_b = live_b ? cb_one : 0.f;
#pragma unroll
        for (int j = 0; j < KI; ++j)
            in.xa[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, (ka_off[j] == OOB || oa == OOB) ? OOB : ka_off[j] + oa, 0, PWB_AUX)) + ka_one[j];
        {
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, (cb_off == OOB || ob == OOB) ? OOB : cb_off + ob, 0, PWB_AUX);
            in.xb = f4{__uint_as_float(v.x) + one_b, __uint_as_float(v.y) + one_b, __uint_as_float(v.z) + one_b, __uint_as_float(v.w) + one_b};
        }
        if constexpr (MODE == 1) {
            if constexpr (!YMASK) {      // (the O1 copy of the skip input feeds the z2 chain only)
#pragma unroll
                for (int j = 0; j < KI; ++j)
                    in.sa[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (ka_off[j] == OOB || oa == OOB) ? OOB : ka_off[j] + oa, 0, PWB_AUX)) + ka_one[j];
            } else {
#pragma unroll
                for (int j = 0; j < KI; ++j) in.sa[j] = 0.f;
            }
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (cb_off == OOB || ob == OOB) ? OOB : cb_off + ob, 0, PWB_AUX);
            in.sb = f4{__uint_as_float(v.x) + one_b, __uint_as_float(v.y) + one_b, __uint_as_float(v.z) + one_b, __uint_as_float(v.w) + one_b};
        } else {
#pragma unroll
            for (int j = 0; j < KI; ++j) in.sa[j] = ka_one[j];
            in.sb = f4{one_b, one_b, one_b, one_b};
        }
        if constexpr (YMASK == 2) {
            const unsigned cbo = (c < CO && live_b) ? (unsigned)c * P4 + base_o + pb * 4u : OOB;
            const u4 vd = __builtin_amdgcn_raw_buffer_load_b128(rd, cbo, 0, PWB_AUX), vy = __builtin_amdgcn_raw_buffer_load_b128(ry, cbo, 0, PWB_AUX);
            in.dzb = f4{__uint_as_float(vd.x), __uint_as_float(vd.y), __uint_as_float(vd.z), __uint_as_float(vd.w)};
            in.yob = f4{__uint_as_float(vy.x), __uint_as_float(vy.y), __uint_as_float(vy.z), __uint_as_float(vy.w)};
        }
        const unsigned od = live_a ? base_o + pa * 4u : OOB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            in.dz[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, (co_off[r] == OOB || od == OOB) ? OOB : co_off[r] + od, 0, PWB_AUX));
            in.sl[r] = 0.f;
            if constexpr (YMASK)
                in.yo[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, (co_off[r] == OOB || od == OOB) ? OOB : co_off[r] + od, 0, PWB_AUX));
        }
        if constexpr (MODE == 2 && !YMASK) {
            const unsigned pc = live_a ? pa : (unsigned)a.P - 1u;
            const unsigned os = (unsigned)b * CO * sP4 + ((pc / (unsigned)a.T) * (unsigned)a.sT + (unsigned)(a.sT - 1)) * 4u;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                in.sl[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, cs_off[r] != OOB ? cs_off[r] + os : OOB, 0, PWB_AUX));
        }
    };
#define PW_MFMA(A_, B_, C_) __builtin_amdgcn_mfma_f32_16x16x4f32((A_), (B_), (C_), 0, 0, 0)
    In cur;
    load(wid < total ? wid : total - 1, cur);
    for (int G = wid; G < total; G += wstride) {
        In nxt;
        load(G + wstride < total ? G + wstride : total - 1, nxt);   // unconditional (clamped): no branch around the prefetch
        __builtin_amdgcn_sched_barrier(0);                          // ... and issued HERE, a whole iteration ahead of their use
        const int b = G / gpb;
        const long pb = (long)(G - b * gpb) * 16 + 4 * q;
        const bool live_b = pb < a.P;
        // Issue order: independent matrix work is placed between a product and the element-wise step that consumes it,
        // so the wave has MFMAs in flight while it runs its VALU part (a wave issues in order).
        // ---- z1 in both orientations (the same two fragments, operands swapped)
        f4 h[TM], d1[TM], hT[TM], dT[TM];
        {
            f4 z[TM], zT[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                z[t] = zT[t] = f4{0.f, 0.f, 0.f, 0.f};
                if constexpr (YMASK != 2) {
#pragma unroll
                    for (int j = 0; j < KI; ++j) z[t] = PW_MFMA(W1a[t][j], cur.xa[j], z[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int j = 0; j < KI; ++j) zT[t] = PW_MFMA(cur.xa[j], W1a[t][j], zT[t]);
            if constexpr (ACT == 1) {
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    d1[t] = z[t];                                       // the pre-activation stands in for the derivative
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[t][r] = pw_relu(z[t][r]);
                }
            } else {
                pw_act_tiles<TM>(z, h, d1, a.act1);
            }
            // ---- O1: z2 = [Ws | b] [s ; 1] + W2 h  (the skip part first: it does not wait for the activation)
            f4 z2 = f4{cur.sl[0], cur.sl[1], cur.sl[2], cur.sl[3]};
            if constexpr (!YMASK) {
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int j = 0; j < KI; ++j) z2 = PW_MFMA(Wsa[j], cur.sa[j], z2);
                } else {
                    z2 = PW_MFMA(Wsa[CI / 4], cur.sa[CI / 4], z2);        // only the constant-1 channel (the bias) is there
                }
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int r = 0; r < pw_tile_steps(CM, t); ++r) z2 = PW_MFMA(W2a[t][r], h[t][r], z2);
            }
            if constexpr (ACT == 1) {
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    dT[t] = zT[t];
#pragma unroll
                    for (int r = 0; r < 4; ++r) hT[t][r] = pw_relu(zT[t][r]);
                }
            } else {
                pw_act_tiles<TM>(zT, hT, dT, a.act1);                  // under the z2 chain
            }
            if constexpr (YMASK) {
#pragma unroll
                for (int r = 0; r < 4; ++r) z2[r] = cur.yo[r] > 0.f ? cur.dz[r] : 0.f;
            } else if constexpr (ACT == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) z2[r] = z2[r] > 0.f ? cur.dz[r] : 0.f;
            } else {
                f4 zz[1] = {z2}, hh[1], dd[1];
                pw_act_tiles<1>(zz, hh, dd, a.act2);
#pragma unroll
                for (int r = 0; r < 4; ++r) z2[r] = cur.dz[r] * dd[0][r];
            }
            // g2 now lives in z2's registers
            // ---- everything that needs only g2: dh (O1), dh^T and g2^T (OT)
            f4 dh[TM], dhT[TM], g2T = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                dh[t] = dhT[t] = f4{0.f, 0.f, 0.f, 0.f};
                if constexpr (YMASK != 2) {
#pragma unroll
                    for (int r = 0; r < RO; ++r) dh[t] = PW_MFMA(W2b[t][r], z2[r], dh[t]);
                }
            }
            if constexpr (YMASK == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) g2T[r] = cur.yob[r] > 0.f ? cur.dzb[r] : 0.f;
            } else {
#pragma unroll
                for (int r = 0; r < RO; ++r) g2T = PW_MFMA(z2[r], Idf[r], g2T);
            }
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int r = 0; r < RO; ++r) dhT[t] = PW_MFMA(z2[r], W2b[t][r], dhT[t]);
            // ---- dx^T, ds^T from the O1 tiles as A operands
            f4 dxT = f4{0.f, 0.f, 0.f, 0.f};
            if constexpr (YMASK == 2) {
                // g1^T (OT) once, for the weight gradient below AND, transposed through the identity, for dx^T
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dhT[t][r] = dT[t][r] > 0.f ? dhT[t][r] : 0.f;
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    f4 g1 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 4; ++r) g1 = PW_MFMA(dhT[t][r], Idp[r], g1);
                    dh[t] = g1;
                }
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int r = 0; r < pw_tile_steps(CM, t); ++r) dxT = PW_MFMA(dh[t][r], W1b[t][r], dxT);
            } else {
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int r = 0; r < pw_tile_steps(CM, t); ++r)
                        dxT = PW_MFMA(ACT == 1 ? (d1[t][r] > 0.f ? dh[t][r] : 0.f) : dh[t][r] * d1[t][r], W1b[t][r], dxT);
            }
            if constexpr (MODE == 1) {
                if (a.ds) {
                    f4 dsT = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < RO; ++r) dsT = PW_MFMA(z2[r], Wsb[r], dsT);
                    if (c < CI && live_b) PWB_STORE(reinterpret_cast<f4*>(a.ds + ((size_t)b * CI + c) * a.P + pb), dsT);
                }
            } else if constexpr (MODE == 2) {
                if (a.ds && c < CO && live_b) PWB_STORE(reinterpret_cast<f4*>(a.ds + ((size_t)b * CO + c) * a.P + pb), g2T);
            }
            // ---- weight gradients: contractions over the 16 points, OT tiles on both sides
#pragma unroll
            for (int r = 0; r < 4; ++r) accWs = PW_MFMA(g2T[r], cur.sb[r], accWs);
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    accW2[t] = PW_MFMA(g2T[r], hT[t][r], accW2[t]);
                    accW1[t] = PW_MFMA(YMASK == 2 ? dhT[t][r] : (ACT == 1 ? (dT[t][r] > 0.f ? dhT[t][r] : 0.f) : dhT[t][r] * dT[t][r]),
                                       cur.xb[r], accW1[t]);
                }
            if (a.dx && c < CI && live_b)
                PWB_STORE(reinterpret_cast<f4*>(a.dx + ((size_t)b * CI + c) * a.P + pb), dxT);
        }
        cur = nxt;
    }
#undef PW_MFMA
    // ---- this wave's row of partial sums: A (COP x CB) = [dW2 | db2 | dWs],  B (CM1 x CIP) = [dW1 | db1]
    float* out = a.partials + (size_t)wid * Gm::TOTAL;
    float* o1 = out + Gm::N_A;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hc = pw_hid_row<CM>(t, c), hq = pw_hid_row<CM>(t, 4 * q + r);
            if (hc >= 0) out[(4 * q + r) * Gm::CB + hc] = accW2[t][r];
            if (c <= CI && hq >= 0) o1[hq * Gm::CIP + c] = accW1[t][r];
        }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (c <= CI) out[(4 * q + r) * Gm::CB + CM + (c == CI ? 0 : 1 + c)] = accWs[r];
}

template <int CI, int CM, int CO, int MODE, int ACT = -1, int YMASK = 0>
static int launch_pw_bwd_mfma_m(PwBwdArgs a, int batch, int max_rows, int* dims, hipStream_t st) {
    using Gm = PwBwdGeom<CI, CM, CO, true>;
    dims[0] = Gm::COP; dims[1] = Gm::CB; dims[2] = Gm::CM1; dims[3] = Gm::CIP; dims[4] = Gm::TOTAL; dims[5] = 0;
    if (!a.x) return 0;
    a.batch = batch;
    auto kern = k_pointwise_bwd_mfma<CI, CM, CO, MODE, ACT, YMASK>;
    int per_cu = 0, dev = 0, cus = 256;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, 0));
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const long groups = ((a.P + 15) / 16) * batch;
    if (groups >= (1L << 30)) return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: %ld groups of 16 points exceed the kernel's index range", groups);
    if ((size_t)batch * (CI > CO ? CI : CO) * (size_t)a.P * 4 >= ((size_t)1 << 31))
        return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: tensors of 2 GiB and more are beyond the kernel's 32-bit buffer offsets");
    long blocks = std::min<long>({(groups + 3) / 4, (long)max_rows / 4, (long)std::max(per_cu, 1) * cus});
    if (blocks < 1) blocks = 1;
    // every wave writes its row, also one that found no work; rows beyond the grid are not touched
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    dims[5] = (int)(blocks * 4);
    return 0;
}

template <int CI, int CM, int CO>
static int launch_pw_bwd_mfma(PwBwdArgs a, int batch, int max_rows, int* dims, hipStream_t st) {
    if constexpr (CI == 10 && CM == 40 && CO == 10) {      // the reference's default width with its default activation (ReLU)
        if (a.x && a.act1 == 1 && a.act2 == 1 && env_int("TCFD_PW_BWD_RELU", 1)) {
            const int ym = a.out ? env_int("TCFD_PW_BWD_YMASK", 2) : 0;   // the forward output was handed over: no z2 recompute
            if (ym >= 2) {                                                // ... and no O1 side of the hidden layer (default)
                if (a.skip_mode == 1) return launch_pw_bwd_mfma_m<CI, CM, CO, 1, 1, 2>(a, batch, max_rows, dims, st);
                if (a.skip_mode == 2) return launch_pw_bwd_mfma_m<CI, CM, CO, 2, 1, 2>(a, batch, max_rows, dims, st);
            } else if (ym == 1) {
                if (a.skip_mode == 1) return launch_pw_bwd_mfma_m<CI, CM, CO, 1, 1, 1>(a, batch, max_rows, dims, st);
                if (a.skip_mode == 2) return launch_pw_bwd_mfma_m<CI, CM, CO, 2, 1, 1>(a, batch, max_rows, dims, st);
            }
            if (a.skip_mode == 1) return launch_pw_bwd_mfma_m<CI, CM, CO, 1, 1>(a, batch, max_rows, dims, st);
            if (a.skip_mode == 2) return launch_pw_bwd_mfma_m<CI, CM, CO, 2, 1>(a, batch, max_rows, dims, st);
        }
    }
    if (a.skip_mode == 1) return launch_pw_bwd_mfma_m<CI, CM, CO, 1>(a, batch, max_rows, dims, st);
    if (a.skip_mode == 2) return launch_pw_bwd_mfma_m<CI, CM, CO, 2>(a, batch, max_rows, dims, st);
    return launch_pw_bwd_mfma_m<CI, CM, CO, 0>(a, batch, max_rows, dims, st);
}

template <int CI, int CM, int CO, int NW = 4>
static int launch_pw_bwd4(PwBwdArgs a, int batch, int max_rows, int* dims, hipStream_t st) {
    using Gm = PwBwdGeom<CI, CM, CO, true>;
    using Wm = PwBwdW<CI, CM, CO, true>;
    dims[0] = Gm::COP; dims[1] = Gm::CB; dims[2] = Gm::CM1; dims[3] = Gm::CIP; dims[4] = Gm::TOTAL; dims[5] = 0;
    if (!a.x) return 0;
    a.chunks_per_batch = (a.P + 63) / 64;
    a.total_chunks = a.chunks_per_batch * batch;
    a.batch = batch;
    constexpr int RED = CO > CI ? CO : CI;
    const size_t lds = ((size_t)((Wm::TOTAL + 3) & ~3) + (size_t)(Gm::ROWS + NW * RED) * Gm::PITCH) * sizeof(float);
    auto kern = k_pointwise_bwd4<CI, CM, CO, NW>;
    int rc = set_lds_attr(kern, lds);
    if (rc) return rc;
    int per_cu = 0, dev = 0, cus = 256;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 64 * NW, lds));
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    long blocks = std::min<long>({a.total_chunks, (long)max_rows, (long)std::max(per_cu, 1) * cus});
    if (blocks < 1) blocks = 1;
    if (a.per_sample) {
        blocks = blocks / batch * batch;
        if (blocks < batch) blocks = batch;
        if (blocks > max_rows) return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: %ld rows needed for per-sample partials, %d given", blocks, max_rows);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NW), lds, st, a);
    HIP_TRY(hipGetLastError());
    dims[5] = (int)blocks;
    return 0;
}

template <int CI, int CM, int CO, bool HAS_L1>
static int launch_pw_bwd(PwBwdArgs a, int batch, int max_waves, int* dims, hipStream_t st) {
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
static int pointwise_bwd_impl(const void* pe, const void* x, const void* skip, const void* dout, void* dx, void* dskip,
                              const void* w1, const void* b1, const void* w2t, const void* b2, const void* wst,
                              const void* bs, void* partials, int max_waves, int* dims, int batch, int ci,
                              int cm, int co, long P, int T, int skip_T, int act1, int act2, int skip_mode,
                              int per_sample, void* stream, const void* out) {
    if (!dims) return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: null dims");
    if (x && (!dout || !w2t || !partials || batch <= 0 || P <= 0 || max_waves < 2))
        return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: bad argument");
    if (skip_mode < 0 || skip_mode > 2) return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: skip_mode %d not supported", skip_mode);
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
    a.chunks_per_batch = a.total_chunks = 0;
    hipStream_t st = (hipStream_t)stream;
    const bool l1 = cm != ci || w1 != nullptr;
#define PWB_CASE(CI_, CM_, CO_, L1_) \
    if (ci == CI_ && cm == CM_ && co == CO_ && l1 == L1_) return launch_pw_bwd<CI_, CM_, CO_, L1_>(a, batch, max_waves, dims, st);
    // TCFD_PW_BWD: 0 = default (width 10: two waves per 64 points -- 33.7 vs 35.1 ms per SFNO training step; other widths four),
    // 2 / 4 = that many waves per 64 points, 1 = one wave per 64 points (k_pointwise_bwd)
    // 0 (default) / 5: the all-MFMA kernel (k_pointwise_bwd_mfma) wherever it applies; 1 / 2 / 4 select the older kernels
    const int bwd_mode = env_int("TCFD_PW_BWD", 0);
    if (l1 && !per_sample && (bwd_mode == 0 || bwd_mode == 5) && P % 4 == 0 && max_waves >= 4 &&
        (size_t)batch * (size_t)(ci > co ? ci : co) * (size_t)P * 4 < ((size_t)1 << 31)) {   // 32-bit buffer offsets
#define PWM_CASE(CI_, CM_, CO_) \
    if (ci == CI_ && cm == CM_ && co == CO_) return launch_pw_bwd_mfma<CI_, CM_, CO_>(a, batch, max_waves, dims, st);
        PWM_CASE(4, 16, 4) PWM_CASE(6, 24, 6) PWM_CASE(8, 32, 8) PWM_CASE(10, 40, 10) PWM_CASE(12, 48, 12) PWM_CASE(14, 56, 14)
#undef PWM_CASE
    }
    if (l1 && bwd_mode != 1) {
        if (ci == 10 && cm == 40 && co == 10 && bwd_mode != 4) return launch_pw_bwd4<10, 40, 10, 2>(a, batch, max_waves, dims, st);
        if (ci == 4 && cm == 16 && co == 4) return launch_pw_bwd4<4, 16, 4>(a, batch, max_waves, dims, st);
        if (ci == 8 && cm == 32 && co == 8) return launch_pw_bwd4<8, 32, 8>(a, batch, max_waves, dims, st);
        if (ci == 10 && cm == 40 && co == 10) return launch_pw_bwd4<10, 40, 10>(a, batch, max_waves, dims, st);
    }
    PWB_CASE(4, 16, 4, true) PWB_CASE(8, 32, 8, true) PWB_CASE(10, 40, 10, true)
    PWB_CASE(4, 4, 4, false) PWB_CASE(4, 4, 1, false) PWB_CASE(8, 8, 8, false) PWB_CASE(8, 8, 1, false)
    PWB_CASE(10, 10, 10, false) PWB_CASE(10, 10, 1, false)
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
__global__ __launch_bounds__(256) void k_sample_outer_mfma(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ pe, float* __restrict__ partials, long P,
                                                           int C, int CO, int waves_per_sample, int batch) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
    const int w = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), b = blockIdx.y;
    const long groups = P / 16;
    const float* dyr = dy + ((size_t)b * CO + (c < CO ? c : 0)) * P + 4 * q;
    const float* xr = pe ? x + (size_t)b * P + 4 * q : x + ((size_t)b * C + (c < C ? c : 0)) * P + 4 * q;
    const float* per = pe ? pe + (size_t)(c < C ? c : 0) * P + 4 * q : nullptr;
    const float ones = c == C ? 1.f : 0.f;
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (long g = w; g < groups; g += waves_per_sample) {
        f4 a = *reinterpret_cast<const f4*>(dyr + g * 16);
        f4 v = *reinterpret_cast<const f4*>(xr + g * 16);
        if (per) v += *reinterpret_cast<const f4*>(per + g * 16);
        if (c >= CO) a = f4{0.f, 0.f, 0.f, 0.f};
        if (c >= C) v = f4{ones, ones, ones, ones};
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], v[r], acc, 0, 0, 0);
    }
    float* out = partials + ((size_t)w * batch + b) * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(4 * q + r) * 16 + c] = acc[r];
}
extern "C" int tcfd_fno_sample_outer_sums(const void* dy, const void* x, const void* pe, void* partials, int batch, int c, int co,
                                          long P, int waves_per_sample, void* stream) {
    if (!dy || !x || !partials || batch <= 0 || c < 1 || c > 15 || co < 1 || co > 16 || P <= 0 || P % 16 != 0 ||
        waves_per_sample < 4 || waves_per_sample % 4 != 0)
        return FAIL(TCFD_EINVAL, "fno_sample_outer_sums: bad argument (needs c <= 15, co <= 16, P %% 16 == 0, whole workgroups)");
    hipLaunchKernelGGL(k_sample_outer_mfma, dim3((unsigned)(waves_per_sample / 4), (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, (const float*)dy, (const float*)x, (const float*)pe, (float*)partials, P, c, co,
                       waves_per_sample, batch);
    HIP_TRY(hipGetLastError());
    return 0;
}
