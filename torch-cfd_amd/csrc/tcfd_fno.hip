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
#include <mutex>
#include <vector>

#include "../../include/tcfd.h"
#include "tcfd_fft.hpp"
#include "tcfd_fno_common.hpp"

using namespace tcfd;
typedef cx<float> cf;


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
                                                  unsigned mt_magic, const T* __restrict__ kts) {
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
        // kts: one real factor per kept time mode (the c2r multiplicities of a backward pass), folded into the t-DFT table
        for (int i = threadIdx.x; i < mt * Tp; i += blockDim.x) twt[i] = kts ? cscale(tw_tf[i], kts[i / Tp]) : tw_tf[i];
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
                                                  const T* acc, const T* __restrict__ accb, int accT, const T* __restrict__ kts) {
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
        for (int i = threadIdx.x; i < t_keep * mt; i += blockDim.x)
            twt[i] = kts ? cscale(tw_ti[(size_t)t0 * mt + i], kts[i % mt]) : tw_ti[(size_t)t0 * mt + i];
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

// Lanes-along-the-modes form for NARROW layers (fp32, ci <= 12): the weights are unique per mode, so a mode is its own
// (b x ci) . (ci x co) product and at width 10 the 16 x 16 x 4 tiles of the kernel above are mostly padding, while its
// stage-through-LDS structure (64-byte runs per row, three barriers) leaves the launch latency-bound: 42 us for 82 MB at config 5.
// Here a lane owns ONE mode and COG output channels: CI x COG weights in registers, the batch as the loop (the next sample's
// spectrum values are in flight while this one is multiplied), every load and store a run of consecutive modes across the
// lanes.  A wave = 64 consecutive modes x one group of output channels x one slice of the batch; the waves that share weights
// (batch slices) and spectrum values (channel groups) of a chunk of modes are dealt to the SAME XCD, next to each other, so
// the re-reads are L2 hits.
template <int CI, int COG>
__global__ __launch_bounds__(64) void k_contract_lanes(ContractArgsT<float> a, int n_cg, int n_bg, int nb, int cpb) {
    typedef cx<float> cf;
    typedef unsigned u2v __attribute__((__vector_size__(2 * sizeof(unsigned))));
    const int L = a.my * a.mt, MB = a.mx * L, M = 4 * MB;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, combos = n_cg * n_bg;
    const int chunk = (slot / combos) * 8 + xcd;              // chunks of 64 modes never straddle a corner block
    if (chunk >= 4 * cpb) return;
    const int combo = slot % combos, cg = combo % n_cg, bg = combo / n_cg;
    const int blk = chunk / cpb, ix = blk & 1, iy = blk >> 1;
    const int g_raw = (chunk - blk * cpb) * 64 + (int)threadIdx.x;
    const bool valid = g_raw < MB;
    const int wm = valid ? g_raw : MB - 1;                     // mode inside the corner block (= index into its weights)
    // run kx of my * mt consecutive modes: contiguous in the spectrum and in the weights
    const int kx = wm / L, within = wm - kx * L;
    const int mode = ((kx + ix * a.mx) * 2 * a.my + iy * a.my) * a.mt + within;
    // every address = wave-uniform base (buffer descriptor) + wave-uniform channel offset (scalar) + 8 * the lane's mode
    const unsigned voff_w = (unsigned)wm * 8u, voff_s = (unsigned)mode * 8u;
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(a.w[blk]), 0, (int)((unsigned)(CI * a.co * MB) * 8u), 0x00020000);
    const int o0 = cg * COG;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 wv[CI][COG];                                            // (re, im) pairs: a complex multiply-add is two v_pk_fma_f32
    const int si = a.adjoint ? 1 : a.co, su = a.adjoint ? CI : 1;   // weight (i, o) sits at i * si + o * su blocks of MB modes
    const float conj_sign = a.adjoint ? -1.f : 1.f;
#pragma unroll
    for (int i = 0; i < CI; ++i)
#pragma unroll
        for (int u = 0; u < COG; ++u) {
            const int o = min(o0 + u, a.co - 1);
            const u2v v = __builtin_amdgcn_raw_buffer_load_b64(rw, voff_w, (unsigned)((i * si + o * su) * MB) * 8u, 0);
            wv[i][u] = (f2){__uint_as_float(v[0]), __uint_as_float(v[1]) * conj_sign};
        }
    f2 bias = {0.f, 0.f};
    if (a.bias[blk]) {
        const cf bv = a.bias[blk][wm];
        bias = (f2){bv.x * a.delta, bv.y * a.delta};
    }
    const int b0 = bg * nb, b1 = min(a.b, b0 + nb);
    if (b0 >= b1) return;
    const unsigned in_bytes = (unsigned)(CI * M) * 8u, out_bytes = (unsigned)(a.co * M) * 8u;
    auto load_x = [&](int bb, f2 (&x)[CI]) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(a.vin) + (size_t)bb * CI * M, 0, (int)in_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < CI; ++i) {
            const u2v v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff_s, (unsigned)(i * M) * 8u, 0);
            x[i] = (f2){__uint_as_float(v[0]), __uint_as_float(v[1])};
        }
    };
    f2 xn[CI];
    load_x(b0, xn);
    for (int bb = b0; bb < b1; ++bb) {
        f2 xc[CI];
#pragma unroll
        for (int i = 0; i < CI; ++i) xc[i] = xn[i];
        load_x(min(bb + 1, b1 - 1), xn);                       // the next sample in flight (clamped: no branch around the prefetch);
                                                               // without it 17.3 -> 32.7 us at config 5
        f2 pr[COG], pi[COG];                                   // pr = sum x.re * (w.re, w.im),  pi = sum x.im * (w.re, w.im)
#pragma unroll
        for (int u = 0; u < COG; ++u) { pr[u] = bias; pi[u] = (f2){0.f, 0.f}; }
#pragma unroll
        for (int i = 0; i < CI; ++i)
#pragma unroll
            for (int u = 0; u < COG; ++u) {
                // one half of x against both halves of w: the half is picked by op_sel, no broadcast copies, no rotated weights
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(pr[u]) : "v"(xc[i]), "v"(wv[i][u]));
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(pi[u]) : "v"(xc[i]), "v"(wv[i][u]));
            }
        f2 acc[COG];
#pragma unroll
        for (int u = 0; u < COG; ++u) acc[u] = (f2){pr[u].x - pi[u].y, pr[u].y + pi[u].x};
        if (valid) {
            const auto ro = __builtin_amdgcn_make_buffer_rsrc(a.vout + (size_t)bb * a.co * M, 0, (int)out_bytes, 0x00020000);
#pragma unroll
            for (int u = 0; u < COG; ++u)
                if (o0 + u < a.co) {
                    const u2v v = {__float_as_uint(acc[u].x), __float_as_uint(acc[u].y)};
                    __builtin_amdgcn_raw_buffer_store_b64(v, ro, voff_s, (unsigned)((o0 + u) * M) * 8u, 0);
                }
        }
    }
}

// Per-mode products for WIDE layers (fp32, 13 ... 32 channels): C[r][c] = sum_k opA(A[k][r]) . opB(B[k][c]) for every kept mode,
// which is the contraction (k = input channel, r = sample, c = output channel), its adjoint (k = output channel, B = conj W^T)
// and the weight gradient (k = sample, r = input channel, A = conj of the spectrum) -- three uses of one kernel that differ in
// strides only.  The matrix-pipe kernel above stages WHOLE operands of 4 - 8 modes in LDS (82 - 131 KB at widths 20 - 32: one
// workgroup per CU, four barriers, 1.0 TB/s); here a workgroup owns 16 consecutive modes (one 128-byte line per operand row)
// and 16 lanes share a mode: lane (m, tc, tr) accumulates rows {4 j + tr} x columns {4 j + tc} (8 x CT complex accumulators)
// while the k axis streams through LDS four k at a time, double buffered: the loads of the next four are in flight during the
// products of this four, one barrier per stage.  Interleaved rows / columns make every LDS read of a wave one contiguous
// 128-byte run per row (broadcast across tc) resp. 4 adjacent runs (columns): conflict free.
struct ModesGemmArgs {
    const cx<float>* a[4];    // operand bases per corner block (equal for spectra)
    const cx<float>* b[4];
    cx<float>* c[4];
    const cx<float>* bias[4]; // added (times delta) to every C[r][c] of the mode, or null
    cx<float>* gb[4];         // delta * sum_{k, c} B[k][c] per mode (the bias gradient of the weight-gradient use), or null
    int a_blk, b_blk, c_blk;  // 1: indexed inside a corner block (weights: stride MB, index wm); 0: a spectrum (stride M, index mode)
    int sAk, sAr, sBk, sBc, sCr;   // element strides in units of the operand's mode stride (C: column stride 1)
    float conj_a, conj_b;     // factor on the imaginary part: -1 conjugates
    float delta;
    int R, Cn, K, mx, my, mt;
    unsigned a_bytes, b_bytes, c_bytes;   // sizes of one operand tensor (bounds of the buffer descriptors)
};

template <int CT, int KC>
__global__ __launch_bounds__(256, 2) void k_modes_gemm(ModesGemmArgs a, int cpb) {
    typedef cx<float> cf;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef unsigned u2v __attribute__((__vector_size__(2 * sizeof(unsigned))));
    constexpr int RT = 8, NR = 4 * RT, NCP = CT <= 4 ? 16 : 32, NA = KC * NR / 16, NB = KC * NCP / 16;
    constexpr int STAGE = KC * (NR + NCP) * 16;                 // complex elements of one buffer
    extern __shared__ __attribute__((aligned(16))) unsigned char mg_raw[];
    f2* lds = reinterpret_cast<f2*>(mg_raw);
    const int L = a.my * a.mt, MB = a.mx * L, M = 4 * MB;
    const int blk = blockIdx.x / cpb, ix = blk & 1, iy = blk >> 1;
    const int t = threadIdx.x, m = t & 15, tc = (t >> 4) & 3, tr = t >> 6, q0 = t >> 4;
    const int g_raw = (blockIdx.x - blk * cpb) * 16 + m;
    const bool valid = g_raw < MB;
    const int wm = valid ? g_raw : MB - 1;
    const int kx = wm / L, within = wm - kx * L;
    const int mode = ((kx + ix * a.mx) * 2 * a.my + iy * a.my) * a.mt + within;
    const int r0 = blockIdx.y * NR;
    const unsigned strA = a.a_blk ? MB : M, strB = a.b_blk ? MB : M, strC = a.c_blk ? MB : M;
    const unsigned idxA = a.a_blk ? wm : mode, idxB = a.b_blk ? wm : mode, idxC = a.c_blk ? wm : mode;
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(a.a[blk]), 0, (int)a.a_bytes, 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(a.b[blk]), 0, (int)a.b_bytes, 0x00020000);
    const auto rc = __builtin_amdgcn_make_buffer_rsrc(a.c[blk], 0, (int)a.c_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;                       // beyond every descriptor: loads return 0, stores are dropped
    // staging: element q = q0 + 16 n of a stage is (k = n / 2, row or column = q & 31) resp. (k = n, column = q0) for NCP = 16
    unsigned offA[NA], offB[NB];
#pragma unroll
    for (int n = 0; n < NA; ++n) {
        const int r = r0 + ((q0 + 16 * n) & (NR - 1)), kk = (q0 + 16 * n) / NR;
        offA[n] = r < a.R ? ((unsigned)(kk * a.sAk + r * a.sAr) * strA + idxA) * 8u : OOB;
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int c = (q0 + 16 * n) & (NCP - 1), kk = (q0 + 16 * n) / NCP;
        offB[n] = c < a.Cn ? ((unsigned)(kk * a.sBk + c * a.sBc) * strB + idxB) * 8u : OOB;
    }
    const unsigned stepA = (unsigned)(KC * a.sAk) * strA * 8u, stepB = (unsigned)(KC * a.sBk) * strB * 8u;
    f2 sa[NA], sb[NB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int n = 0; n < NA; ++n) {
            const int kk = (16 * n) / NR;                         // uniform: q0 < 16 never carries into the k index
            const unsigned off = (k0 + kk < a.K && offA[n] != OOB) ? offA[n] + (unsigned)(k0 / KC) * stepA : OOB;
            const u2v v = __builtin_amdgcn_raw_buffer_load_b64(ra, off, 0, 0);
            sa[n] = (f2){__uint_as_float(v[0]), __uint_as_float(v[1]) * a.conj_a};
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int kk = (16 * n) / NCP;
            const unsigned off = (k0 + kk < a.K && offB[n] != OOB) ? offB[n] + (unsigned)(k0 / KC) * stepB : OOB;
            const u2v v = __builtin_amdgcn_raw_buffer_load_b64(rb, off, 0, 0);
            sb[n] = (f2){__uint_as_float(v[0]), __uint_as_float(v[1]) * a.conj_b};
        }
    };
    auto stash = [&](int buf) {                                   // element q of the stage, mode m  ->  slot q * 16 + m = t + 256 n
        f2* A_ = lds + buf * STAGE;
        f2* B_ = A_ + KC * NR * 16;
#pragma unroll
        for (int n = 0; n < NA; ++n) A_[t + 256 * n] = sa[n];
#pragma unroll
        for (int n = 0; n < NB; ++n) B_[t + 256 * n] = sb[n];
    };
    f2 acc[RT][CT];
    {
        f2 bias = {0.f, 0.f};
        if (a.bias[blk]) {
            const cf bv = a.bias[blk][wm];
            bias = (f2){bv.x * a.delta, bv.y * a.delta};
        }
#pragma unroll
        for (int jr = 0; jr < RT; ++jr)
#pragma unroll
            for (int jc = 0; jc < CT; ++jc) acc[jr][jc] = bias;
    }
    f2 bsum = {0.f, 0.f};
    const bool want_gb = a.gb[blk] != nullptr && tr == 0 && blockIdx.y == 0;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int k0 = 0, cur = 0; k0 < a.K; k0 += KC, cur ^= 1) {
        const bool more = k0 + KC < a.K;
        if (more) fetch(k0 + KC);
        const f2* A_ = lds + cur * STAGE;
        const f2* B_ = A_ + KC * NR * 16;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            f2 bv[CT], br[CT];
#pragma unroll
            for (int jc = 0; jc < CT; ++jc) {
                bv[jc] = B_[(kk * NCP + jc * 4 + tc) * 16 + m];
                br[jc] = (f2){-bv[jc].y, bv[jc].x};
            }
            if (want_gb) {
#pragma unroll
                for (int jc = 0; jc < CT; ++jc) bsum += bv[jc];
            }
#pragma unroll
            for (int jr = 0; jr < RT; ++jr) {
                if (r0 + jr * 4 + tr < a.R) {                     // wave-uniform
                    const f2 av = A_[(kk * NR + jr * 4 + tr) * 16 + m];
#pragma unroll
                    for (int jc = 0; jc < CT; ++jc) {             // acc += a.re * (b.re, b.im) + a.im * (-b.im, b.re)
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[jr][jc]) : "v"(av), "v"(bv[jc]));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[jr][jc]) : "v"(av), "v"(br[jc]));
                    }
                }
            }
        }
        if (more) stash(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int jr = 0; jr < RT; ++jr) {
        const int r = r0 + jr * 4 + tr;
#pragma unroll
        for (int jc = 0; jc < CT; ++jc) {
            const int c = jc * 4 + tc;
            const unsigned off = (valid && r < a.R && c < a.Cn) ? ((unsigned)(r * a.sCr + c) * strC + idxC) * 8u : OOB;
            const u2v v = {__float_as_uint(acc[jr][jc].x), __float_as_uint(acc[jr][jc].y)};
            __builtin_amdgcn_raw_buffer_store_b64(v, rc, off, 0, 0);
        }
    }
    if (a.gb[blk] != nullptr && tr == 0 && blockIdx.y == 0) {      // wave 0: the four column groups of a mode sit 16 lanes apart
        bsum.x += __shfl_xor(bsum.x, 16); bsum.y += __shfl_xor(bsum.y, 16);
        bsum.x += __shfl_xor(bsum.x, 32); bsum.y += __shfl_xor(bsum.y, 32);
        if (tc == 0 && valid) a.gb[blk][wm] = mk<float>(a.delta * bsum.x, a.delta * bsum.y);
    }
}

static int launch_modes_gemm(ModesGemmArgs& a, long a_elems, long b_elems, long c_elems, int row_tiles, hipStream_t st) {
    if (a.Cn > 32 || a.Cn < 1 || a.K < 1 || a.R < 1) return -1;
    if (a_elems * 8 >= (1L << 31) || b_elems * 8 >= (1L << 31) || c_elems * 8 >= (1L << 31)) return -1;
    a.a_bytes = (unsigned)(a_elems * 8); a.b_bytes = (unsigned)(b_elems * 8); a.c_bytes = (unsigned)(c_elems * 8);
    const int MB = a.mx * a.my * a.mt, cpb = (MB + 15) / 16;
    const int ct = a.Cn <= 16 ? 4 : a.Cn <= 20 ? 5 : a.Cn <= 24 ? 6 : 8;
    const int ncp = ct <= 4 ? 16 : 32;
    // k values per stage (LDS: 2 buffers x kc x (32 + ncp) x 16 modes).  At 17 ... 24 columns the registers allow three workgroups
    // per CU and 2 k per stage let LDS allow them too (width 20: 54.8 -> 46.4 us, weight gradient 63.1 -> 49.8); at 32 columns
    // (two workgroups either way) and at <= 16 the longer stage wins.  TCFD_GEMM_KC = 2 / 4 overrides.
    const int kc_env = env_int("TCFD_GEMM_KC", 0);
    const int kc = kc_env == 2 || kc_env == 4 ? kc_env : ((ct == 5 || ct == 6) ? 2 : 4);
    const size_t lds = (size_t)2 * kc * (32 + ncp) * 16 * sizeof(cx<float>);
#define TCFD_MG(CT_, KC_)                                                                                   \
    {                                                                                                       \
        auto kern = k_modes_gemm<CT_, KC_>;                                                                 \
        if (int rc = set_lds_attr(kern, lds)) return rc;                                                    \
        hipLaunchKernelGGL(kern, dim3((unsigned)(4 * cpb), (unsigned)row_tiles), dim3(256), lds, st, a, cpb); \
    }
    if (kc == 4) {
        if (ct == 4) TCFD_MG(4, 4) else if (ct == 5) TCFD_MG(5, 4) else if (ct == 6) TCFD_MG(6, 4) else TCFD_MG(8, 4)
    } else {
        if (ct == 4) TCFD_MG(4, 2) else if (ct == 5) TCFD_MG(5, 2) else if (ct == 6) TCFD_MG(6, 2) else TCFD_MG(8, 2)
    }
#undef TCFD_MG
    return 0;
}

// the contraction / its adjoint through the per-mode product kernel
static int launch_contract_gemm(const ContractArgsT<float>& c, hipStream_t st) {
    if (c.co > 32) return -1;
    const long MB = (long)c.mx * c.my * c.mt, M = 4 * MB;
    ModesGemmArgs g;
    for (int k = 0; k < 4; ++k) { g.a[k] = c.vin; g.b[k] = c.w[k]; g.c[k] = c.vout; g.bias[k] = c.bias[k]; g.gb[k] = nullptr; }
    g.a_blk = 0; g.b_blk = 1; g.c_blk = 0;
    g.sAk = 1; g.sAr = c.ci;                                     // A[k = input channel][r = sample] = vin[(b * ci + i)]
    if (c.adjoint) { g.sBk = 1; g.sBc = c.ci; g.conj_b = -1.f; }  // B[k = o'][c = i'] = conj(W[i'][o']),  W blocks are (co, ci) here
    else { g.sBk = c.co; g.sBc = 1; g.conj_b = 1.f; }
    g.sCr = c.co; g.conj_a = 1.f; g.delta = c.delta;
    g.R = c.b; g.Cn = c.co; g.K = c.ci; g.mx = c.mx; g.my = c.my; g.mt = c.mt;
    return launch_modes_gemm(g, (long)c.b * c.ci * M, (long)c.ci * c.co * MB, (long)c.b * c.co * M, (c.b + 31) / 32, st);
}
static int launch_contract_gemm(const ContractArgsT<double>&, hipStream_t) { return -1; }

// ------------------------------------------------------------------ host side

// Per-time-mode factors of the t / y kernels (tcfd_fno_forward_trunc_kt / tcfd_fno_inverse_trunc_kt): the entry point parks
// the device pointer here for the launchers of ITS call (a thread's calls are sequential; plans stay immutable and shared).
static thread_local const void* t_kt_scale = nullptr;
struct KtScaleScope {
    explicit KtScaleScope(const void* p) { t_kt_scale = p; }
    ~KtScaleScope() { t_kt_scale = nullptr; }
};

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
                           p->t_pad, p->mt, p->my, scale, NS, slabs, (const T*)t_kt_scale);                                    \
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
    // blockIdx.z = plane (batch x channel): at most 65535 per launch, more in several launches on shifted pointers
    const size_t in_plane = (size_t)(FWD ? p->X : 2 * p->mx) * Q, out_plane = (size_t)(FWD ? 2 * p->mx : p->X) * Q;
    for (long z0 = 0; z0 < bc; z0 += 65535) {
        const long nz = std::min<long>(65535, bc - z0);
        hipLaunchKernelGGL(kern, dim3((unsigned)((Q + 255) / 256), (unsigned)((n_out + 15) / 16), (unsigned)nz), dim3(256), lds, st,
                           in + (size_t)z0 * in_plane, out + (size_t)z0 * out_plane, (const ct*)p->tw_x, p->X, FWD ? p->X : p->Xs,
                           p->mx, Q);
        HIP_TRY(hipGetLastError());
    }
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
                           p->Ys, p->T_out, t_keep, p->mt, p->my, scale, NS, slabs, acc, accb, accT, (const T*)t_kt_scale);      \
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
                       p->mt > 1 ? (unsigned)(((1ull << 32) + p->mt - 1) / p->mt) : 0u, (const T*)t_kt_scale);
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
                       (const ct*)p->tw_ti, p->T_out, t_keep, p->mt, p->my, scale, P, NS, slabs, p->Ys, acc, accb, accT,
                       (const T*)t_kt_scale);
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

// ------------------------------------------------------------------ per-launch event timing (tcfd_fno_common.hpp)
bool tcfd_fno_prof_on = false;
namespace {
struct FnoProfRec { int kind; hipEvent_t e0, e1; };
struct FnoProfState {
    std::mutex mu;
    int max_records = 0;
    std::vector<FnoProfRec> recs;
} g_fno_prof;
}  // namespace
int tcfd_fno_prof_open(int kind, hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_fno_prof.mu);
    if (!tcfd_fno_prof_on || (int)g_fno_prof.recs.size() >= g_fno_prof.max_records) return -1;
    FnoProfRec r;
    r.kind = kind;
    if (hipEventCreate(&r.e0) != hipSuccess) return -1;
    if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipEventDestroy(r.e0); return -1; }
    (void)hipEventRecord(r.e0, st);
    g_fno_prof.recs.push_back(r);
    return (int)g_fno_prof.recs.size() - 1;
}
void tcfd_fno_prof_close(int idx, hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_fno_prof.mu);
    if (idx >= 0 && idx < (int)g_fno_prof.recs.size()) (void)hipEventRecord(g_fno_prof.recs[idx].e1, st);
}
extern "C" int tcfd_fno_profile_begin(int max_records) {
    if (max_records <= 0) return FAIL(TCFD_EINVAL, "fno_profile_begin: bad argument");
    std::lock_guard<std::mutex> lock(g_fno_prof.mu);
    for (auto& r : g_fno_prof.recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_fno_prof.recs.clear();
    g_fno_prof.recs.reserve(max_records);
    g_fno_prof.max_records = max_records;
    tcfd_fno_prof_on = true;
    return 0;
}
extern "C" int tcfd_fno_profile_end(int capacity, int* count, int* kinds, float* ms) {
    if (!count) return FAIL(TCFD_EINVAL, "fno_profile_end: null count");
    std::lock_guard<std::mutex> lock(g_fno_prof.mu);
    tcfd_fno_prof_on = false;
    int n = 0;
    for (auto& r : g_fno_prof.recs) {
        HIP_TRY(hipEventSynchronize(r.e1));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, r.e0, r.e1));
        if (n < capacity && kinds && ms) { kinds[n] = r.kind; ms[n] = t; }
        ++n;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_fno_prof.recs.clear();
    *count = n;
    return 0;
}

static int force_dft() { return env_int("TCFD_FNO_DFT", 0); }   // 1: any-size kernels for every size (cross-check; read per call)
template <typename T>
static int do_fwd_ty(const tcfd_fno_plan* p, const T* v, cx<T>* w1, long slabs, T s, hipStream_t st) {
    FnoProfScope prof(FNO_K_FWD_TY, st);
    if (!fft_len(p->Y) || force_dft()) return launch_fwd_ty_dft<T>(p, v, w1, slabs, s, st);
    DISPATCH_FFT(p->Y, (launch_fwd_ty2<T, N_>(p, v, w1, slabs, s, st)));
}
template <typename T>
static int do_inv_ty(const tcfd_fno_plan* p, const cx<T>* w2, T* out, long slabs, int t_keep, T s, hipStream_t st,
                     const T* acc = nullptr, const T* accb = nullptr, int accT = 0) {
    FnoProfScope prof(FNO_K_INV_TY, st);
    if (!fft_len(p->Y) || force_dft()) return launch_inv_ty_dft<T>(p, w2, out, slabs, t_keep, s, st, acc, accb, accT);
    DISPATCH_FFT(p->Y, (launch_inv_ty2<T, N_>(p, w2, out, slabs, t_keep, s, st, acc, accb, accT)));
}
template <typename T>
static int do_fwd_x(const tcfd_fno_plan* p, const cx<T>* in, cx<T>* out, long bc, hipStream_t st) {
    FnoProfScope prof(FNO_K_FWD_X, st);
    if (!fft_len(p->X) || force_dft()) return launch_x_dft<T, true>(p, in, out, bc, st);
    DISPATCH_FFT(p->X, (launch_x<T, N_, true>(p, in, out, bc, st)));
}
template <typename T>
static int do_inv_x(const tcfd_fno_plan* p, const cx<T>* in, cx<T>* out, long bc, hipStream_t st) {
    FnoProfScope prof(FNO_K_INV_X, st);
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
// the lanes kernel: CI as instantiated, COG = 5 when it divides co (width 10), else 4; the batch is sliced until ~1300 waves
// are in flight (TCFD_CONTRACT_BG overrides the number of slices; config 5: 4 slices 17.3 us, 8 slices 19.0, 2 slices 19.8,
// the matrix-pipe kernel 43.9 -- profiles/r05_contract_timing.json)
template <int CI>
static int launch_contract_lanes_ci(const ContractArgsT<float>& a, hipStream_t st) {
    const int MB = a.mx * a.my * a.mt, cpb = (MB + 63) / 64;
    const int cog = a.co % 5 == 0 ? 5 : 4, n_cg = (a.co + cog - 1) / cog;
    int n_bg = env_int("TCFD_CONTRACT_BG", 0);
    if (n_bg <= 0) n_bg = (int)std::max<long>(1, (1280 + 4L * cpb * n_cg - 1) / (4L * cpb * n_cg));   // ~1.3 waves per SIMD measured best
    n_bg = std::min(n_bg, a.b);
    const int nb = (a.b + n_bg - 1) / n_bg;
    n_bg = (a.b + nb - 1) / nb;
    const unsigned blocks = (unsigned)(((4 * cpb + 7) / 8) * 8 * n_cg * n_bg);
    if (cog == 5) hipLaunchKernelGGL((k_contract_lanes<CI, 5>), dim3(blocks), dim3(64), 0, st, a, n_cg, n_bg, nb, cpb);
    else hipLaunchKernelGGL((k_contract_lanes<CI, 4>), dim3(blocks), dim3(64), 0, st, a, n_cg, n_bg, nb, cpb);
    return 0;
}
static int launch_contract_lanes(const ContractArgsT<float>& a, hipStream_t st) {
    const long M = 4L * a.mx * a.my * a.mt;                  // 32-bit byte offsets inside one sample / one weight block
    if ((long)std::max(a.ci, a.co) * M * 8 >= (1L << 31) || (long)a.ci * a.co * (M / 4) * 8 >= (1L << 31)) return -1;
    switch (a.ci) {
        case 4: return launch_contract_lanes_ci<4>(a, st);
        case 5: return launch_contract_lanes_ci<5>(a, st);
        case 6: return launch_contract_lanes_ci<6>(a, st);
        case 8: return launch_contract_lanes_ci<8>(a, st);
        case 10: return launch_contract_lanes_ci<10>(a, st);
        case 12: return launch_contract_lanes_ci<12>(a, st);
        default: return -1;
    }
}
static int launch_contract_lanes(const ContractArgsT<double>&, hipStream_t) { return -1; }

template <typename T>
static int do_contract(ContractArgsT<T> a, int use_mfma, hipStream_t st) {
    FnoProfScope prof(FNO_K_CONTRACT, st);
    const int MB = a.mx * a.my * a.mt;
    // 8 modes per workgroup (64-byte runs of fp32 spectrum and weights); TCFD_CONTRACT_NM=16 (read per call) selects
    // whole 128-byte lines with half the workgroups: measured 40.5 against 39.2 us at the config-5 shape -- the launch is
    // ~1.4 rounds of workgroups moving in step (load burst, MFMAs, store burst), not a bandwidth or LDS problem
    const int force = env_int("TCFD_CONTRACT_NM", 0);
    const size_t lds16 = contract_lds<T, 16>(a), lds8 = contract_lds<T, 8>(a), lds4 = contract_lds<T, 4>(a);
    int rc = -1;
    if (use_mfma && env_int("TCFD_CONTRACT_LANES", 1)) rc = launch_contract_lanes(a, st);   // narrow fp32 layers; -1: not its shape
    if (rc < 0 && use_mfma && env_int("TCFD_CONTRACT_GEMM", 1)) rc = launch_contract_gemm(a, st);   // fp32 up to 32 output channels
    if (rc >= 0)
        ;
    else if (use_mfma && force == 16 && MB % 16 == 0 && lds16 <= 150 * 1024)
        rc = launch_contract_mfma<T, 16>(a, lds16, st);
    else if (use_mfma && MB % 4 == 0 && (force == 4 || (force == 0 && lds8 > 64 * 1024)) && lds4 <= 150 * 1024)
        rc = launch_contract_mfma<T, 4>(a, lds4, st);            // wide layers: 8 modes would leave one workgroup per CU
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

// Whether the plan's kernels take a call that keeps `t_keep` output steps: always for the FFT lengths; the pruned direct-DFT
// kernels of the other lengths hold a whole (Y x time) slab in LDS and know at most 16 time modes (launch_*_dft above).  A host asks
// before it picks the library over its own fallback (the Python layers: dense GEMM transforms).
extern "C" int tcfd_fno_plan_supports(const tcfd_fno_plan* p, int t_keep) {
    if (!p || t_keep <= 0 || t_keep > p->T_out) return 0;
    const bool dft_y = !fft_len(p->Y) || force_dft();
    if (!dft_y) return 1;
    if (p->mt > 16) return 0;
    const size_t ct = csize(p), rt = ct / 2;
    const size_t fwd = ((size_t)p->Y + (size_t)p->mt * p->Tp) * ct + (size_t)p->Y * p->mt * ct;
    const size_t Q = (size_t)2 * p->my * p->mt;
    const size_t inv = ((size_t)p->Y + (size_t)t_keep * p->mt) * ct + (Q + 2 * p->mt + 2 * (size_t)(p->my + 1) * p->mt) * ct +
                       (size_t)p->Y * t_keep * rt;
    return fwd <= 150 * 1024 && inv <= 150 * 1024;
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
// The same transforms with one real factor per kept time mode (kt_scale: mt values of the data's real precision in device memory)
// folded into their t-DFT tables: the adjoint of an r2c / c2r pair weighs the interior time modes by 2 resp. 1 / 2, which
// otherwise is an elementwise pass over the spectrum before and after the contraction of every layer's backward.
// acc / last: as tcfd_fno_inverse_trunc_acc / _last (at most one of them).
extern "C" int tcfd_fno_forward_trunc_kt(const tcfd_fno_plan* p, const void* v, void* vh, int batch, int c, double fwd_scale,
                                         const void* kt_scale, void* ws, size_t ws_bytes, void* stream) {
    KtScaleScope scope(kt_scale);
    return tcfd_fno_forward_trunc(p, v, vh, batch, c, fwd_scale, ws, ws_bytes, stream);
}
extern "C" int tcfd_fno_inverse_trunc_kt(const tcfd_fno_plan* p, const void* vh, void* out, const void* acc, const void* last,
                                         int batch, int c, int t_keep, double inv_scale, const void* kt_scale, void* ws,
                                         size_t ws_bytes, void* stream) {
    if (acc && last) return FAIL(TCFD_EINVAL, "fno_inverse_trunc_kt: acc and last are exclusive");
    KtScaleScope scope(kt_scale);
    if (last) return tcfd_fno_inverse_trunc_last(p, vh, out, last, batch, c, t_keep, inv_scale, ws, ws_bytes, stream);
    return tcfd_fno_inverse_trunc_acc(p, vh, out, acc, batch, c, t_keep, inv_scale, ws, ws_bytes, stream);
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
__global__ __launch_bounds__(256, 2) void k_contract_wgrad(WgradArgsT<T> a) {
    // 64 consecutive modes x 4 slices of the batch per workgroup (one wave each): the sum over the batch is the only serial
    // loop of this kernel, and with a whole batch per lane (round 4) a launch was 432 waves waiting on 32 dependent rounds of
    // loads (52 us at config 5).  The four partial sums meet in LDS; wave 0 adds them IN ORDER (deterministic) and stores.
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_raw[];
    T* part = reinterpret_cast<T*>(wg_raw);                       // [3 waves][IC * OC * 2][64 lanes]
    const int M = 4 * a.mx * a.my * a.mt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mode_raw = blockIdx.x * 64 + lane;
    const bool valid = mode_raw < M;
    const int mode = valid ? mode_raw : M - 1;
    const int i0 = blockIdx.y * IC, o0 = blockIdx.z * OC;
    const int kt = mode % a.mt;
    const int kyi = (mode / a.mt) % (2 * a.my);
    const int kxi = mode / (a.mt * 2 * a.my);
    const int ix = kxi >= a.mx, iy = kyi >= a.my;
    const int blk = ix + 2 * iy;
    const long MB = (long)a.mx * a.my * a.mt;
    const long wm = ((long)(kxi - ix * a.mx) * a.my + (kyi - iy * a.my)) * a.mt + kt;
    cx<T>* gw = blk == 0 ? a.gw[0] : blk == 1 ? a.gw[1] : blk == 2 ? a.gw[2] : a.gw[3];
    cx<T>* gb = blk == 0 ? a.gb[0] : blk == 1 ? a.gb[1] : blk == 2 ? a.gb[2] : a.gb[3];
    const int nb = (a.b + 3) / 4, b0 = wave * nb, b1 = min(a.b, b0 + nb);
    // IC input x OC output channels of ONE mode in registers: every spectrum value is read co / OC (vh) resp. ci / IC (gh) times
    T re[IC][OC], im[IC][OC];
#pragma unroll
    for (int v = 0; v < IC; ++v)
#pragma unroll
        for (int u = 0; u < OC; ++u) re[v][u] = im[v][u] = 0;
    T bre = 0, bim = 0;                                           // bias gradient: sum over (b, o) of gh, by the (0, *) workgroups
    const bool do_bias = blockIdx.y == 0;
    if constexpr (sizeof(T) == 4) {
        // fp32: (re, im) of an accumulator as one register pair, conj(v) g = v.re * (g.re, g.im) + v.im * (g.im, -g.re) as two
        // v_pk_fma_f32 (the half of v picked by op_sel)
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 acc[IC][OC];
#pragma unroll
        for (int v = 0; v < IC; ++v)
#pragma unroll
            for (int u = 0; u < OC; ++u) acc[v][u] = (f2){0.f, 0.f};
        for (int bb = b0; bb < b1; ++bb) {
            f2 vv[IC], gg[OC], gr[OC];
#pragma unroll
            for (int v = 0; v < IC; ++v) {
                const cx<T> t = i0 + v < a.ci ? a.vh[((long)bb * a.ci + i0 + v) * M + mode] : mk<T>((T)0, (T)0);
                vv[v] = (f2){t.x, t.y};
            }
#pragma unroll
            for (int u = 0; u < OC; ++u) {
                const cx<T> t = o0 + u < a.co ? a.gh[((long)bb * a.co + o0 + u) * M + mode] : mk<T>((T)0, (T)0);
                gg[u] = (f2){t.x, t.y};
                gr[u] = (f2){t.y, -t.x};
            }
#pragma unroll
            for (int v = 0; v < IC; ++v)
#pragma unroll
                for (int u = 0; u < OC; ++u) {
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[v][u]) : "v"(vv[v]), "v"(gg[u]));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[v][u]) : "v"(vv[v]), "v"(gr[u]));
                }
            if (do_bias) {
#pragma unroll
                for (int u = 0; u < OC; ++u) { bre += gg[u].x; bim += gg[u].y; }
            }
        }
#pragma unroll
        for (int v = 0; v < IC; ++v)
#pragma unroll
            for (int u = 0; u < OC; ++u) { re[v][u] = acc[v][u].x; im[v][u] = acc[v][u].y; }
    } else {
        for (int bb = b0; bb < b1; ++bb) {
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
            if (do_bias) {
#pragma unroll
                for (int u = 0; u < OC; ++u) { bre += gg[u].x; bim += gg[u].y; }
            }
        }
    }
    constexpr int NACC = IC * OC * 2 + 2;
    if (wave > 0) {
        T* mine = part + (size_t)(wave - 1) * NACC * 64 + lane;
#pragma unroll
        for (int v = 0; v < IC; ++v)
#pragma unroll
            for (int u = 0; u < OC; ++u) {
                mine[(size_t)((v * OC + u) * 2) * 64] = re[v][u];
                mine[(size_t)((v * OC + u) * 2 + 1) * 64] = im[v][u];
            }
        mine[(size_t)(NACC - 2) * 64] = bre;
        mine[(size_t)(NACC - 1) * 64] = bim;
    }
    __syncthreads();
    if (wave > 0 || !valid) return;
#pragma unroll 1
    for (int k = 0; k < 3; ++k) {
        const T* other = part + (size_t)k * NACC * 64 + lane;
#pragma unroll
        for (int v = 0; v < IC; ++v)
#pragma unroll
            for (int u = 0; u < OC; ++u) {
                re[v][u] += other[(size_t)((v * OC + u) * 2) * 64];
                im[v][u] += other[(size_t)((v * OC + u) * 2 + 1) * 64];
            }
        bre += other[(size_t)(NACC - 2) * 64];
        bim += other[(size_t)(NACC - 1) * 64];
    }
    if (gw) {
#pragma unroll
        for (int v = 0; v < IC; ++v)
#pragma unroll
            for (int u = 0; u < OC; ++u)
                if (i0 + v < a.ci && o0 + u < a.co) gw[((long)(i0 + v) * a.co + o0 + u) * MB + wm] = mk<T>(re[v][u], im[v][u]);
    }
    // the bias gradient needs every output channel: with one group of them (co <= OC) it is complete here, otherwise the
    // groups add theirs into the zeroed buffer one after another -- see the host side (it launches such shapes with gb = null
    // and a second, bias-only pass)
    if (gb && do_bias) gb[wm] = mk<T>(a.delta * bre, a.delta * bim);
}

// bias gradient alone, for layers with more output channels than one group holds: 64 modes x 4 slices of the (sample, channel)
// rows per workgroup, the four partial sums added in order through LDS (one lane per mode walking all b * co rows: 153 us at
// width 20)
template <typename T>
__global__ __launch_bounds__(256) void k_contract_bgrad(WgradArgsT<T> a) {
    __shared__ T part[3][2][64];
    const int M = 4 * a.mx * a.my * a.mt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mode_raw = blockIdx.x * 64 + lane;
    const bool valid = mode_raw < M;
    const int mode = valid ? mode_raw : M - 1;
    const int kt = mode % a.mt;
    const int kyi = (mode / a.mt) % (2 * a.my);
    const int kxi = mode / (a.mt * 2 * a.my);
    const int ix = kxi >= a.mx, iy = kyi >= a.my;
    const int blk = ix + 2 * iy;
    const long wm = ((long)(kxi - ix * a.mx) * a.my + (kyi - iy * a.my)) * a.mt + kt;
    cx<T>* gb = blk == 0 ? a.gb[0] : blk == 1 ? a.gb[1] : blk == 2 ? a.gb[2] : a.gb[3];
    const long rows = (long)a.b * a.co, per = (rows + 3) / 4, r0 = wave * per, r1 = r0 + per < rows ? r0 + per : rows;
    T re[4] = {0, 0, 0, 0}, im[4] = {0, 0, 0, 0};
    long r = r0;
    for (; r + 4 <= r1; r += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const cx<T> g = a.gh[(r + u) * M + mode];
            re[u] += g.x;
            im[u] += g.y;
        }
    }
    for (; r < r1; ++r) {
        const cx<T> g = a.gh[r * M + mode];
        re[0] += g.x;
        im[0] += g.y;
    }
    T sre = (re[0] + re[1]) + (re[2] + re[3]), sim = (im[0] + im[1]) + (im[2] + im[3]);
    if (wave > 0) { part[wave - 1][0][lane] = sre; part[wave - 1][1][lane] = sim; }
    __syncthreads();
    if (wave > 0 || !valid || !gb) return;
    for (int k = 0; k < 3; ++k) { sre += part[k][0][lane]; sim += part[k][1][lane]; }
    gb[wm] = mk<T>(a.delta * sre, a.delta * sim);
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
    FnoProfScope prof(FNO_K_CONTRACT_WGRAD, st);
    a.delta = (T)delta; a.b = batch; a.ci = cin; a.co = cout; a.mx = mx; a.my = my; a.mt = mt;
    const long M = 4L * mx * my * mt;
    if (M <= 0 || M > (1L << 30) || cin < 1 || cout < 1 || batch < 1) return FAIL(TCFD_EINVAL, "fno_contract_wgrad: bad shape");
    if constexpr (sizeof(T) == 4) {
        // wide fp32 layers: the per-mode product kernel (k = sample), every spectrum value read once.  TCFD_WGRAD_GEMM: 0 never,
        // 1 always (shapes permitting), default: above 12 channels
        const int mode = env_int("TCFD_WGRAD_GEMM", -1);
        const bool all_w = a.gw[0] && a.gw[1] && a.gw[2] && a.gw[3];
        if (all_w && cin <= 32 && cout <= 32 && (mode == 1 || (mode < 0 && std::max(cin, cout) > 12))) {
            ModesGemmArgs g;
            for (int k = 0; k < 4; ++k) { g.a[k] = a.vh; g.b[k] = a.gh; g.c[k] = a.gw[k]; g.bias[k] = nullptr; g.gb[k] = a.gb[k]; }
            g.a_blk = 0; g.b_blk = 0; g.c_blk = 1;
            g.sAk = cin; g.sAr = 1; g.sBk = cout; g.sBc = 1; g.sCr = cout;
            g.conj_a = -1.f; g.conj_b = 1.f; g.delta = (float)delta;
            g.R = cin; g.Cn = cout; g.K = batch; g.mx = mx; g.my = my; g.mt = mt;
            const int rc = launch_modes_gemm(g, (long)batch * cin * M, (long)batch * cout * M, (long)cin * cout * (M / 4), 1, st);
            if (rc >= 0) {
                if (rc == 0) HIP_TRY(hipGetLastError());
                return rc;
            }
        }
    }
    constexpr int OC = 10, IC = sizeof(T) == 8 ? 2 : 5;      // 100 resp. 80 accumulator registers
    const bool wide = cout > OC;                              // the bias gradient sums over ALL output channels: own pass then
    WgradArgsT<T> aw = a;
    if (wide) for (int k = 0; k < 4; ++k) aw.gb[k] = nullptr;
    bool any_w = false, any_b = false;
    for (int k = 0; k < 4; ++k) { any_w = any_w || a.gw[k]; any_b = any_b || a.gb[k]; }
    if (any_w || !wide) {
        auto kern = k_contract_wgrad<T, IC, OC>;
        const size_t lds = (size_t)3 * (IC * OC * 2 + 2) * 64 * sizeof(T);
        if (int rc = set_lds_attr(kern, lds)) return rc;
        // without weight gradients the bias alone needs one group of input channels only
        hipLaunchKernelGGL(kern, dim3((unsigned)((M + 63) / 64), any_w ? (cin + IC - 1) / IC : 1, (cout + OC - 1) / OC), dim3(256), lds,
                           st, aw);
        HIP_TRY(hipGetLastError());
    }
    if (wide && any_b) {
        hipLaunchKernelGGL(k_contract_bgrad<T>, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, st, a);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

extern "C" int tcfd_fno_contract_wgrad(const void* vh, const void* gh, void* const* gw, void* const* gb, double delta,
                                       int batch, int cin, int cout, int mx, int my, int mt, int dtype, void* stream) {
    if (!vh || !gh) return FAIL(TCFD_EINVAL, "fno_contract_wgrad: null argument");
    if (dtype == TCFD_C128) return do_contract_wgrad<double>(vh, gh, gw, gb, delta, batch, cin, cout, mx, my, mt, (hipStream_t)stream);
    if (dtype != TCFD_C64) return FAIL(TCFD_EINVAL, "fno_contract_wgrad: bad dtype %d", dtype);
    return do_contract_wgrad<float>(vh, gh, gw, gb, delta, batch, cin, cout, mx, my, mt, (hipStream_t)stream);
}

