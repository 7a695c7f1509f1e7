// tcfd_ns2d.hip -- MI355X (gfx950) kernels + C ABI for the batched 2-D
// pseudo-spectral RK4-CN vorticity step.  See include/tcfd.h for the boundary
// and DESIGN.md for the pass model.  Built from scratch; the reference path
// (torch_cfd/equations.py:328-358, 413-463; torch_cfd/spectral.py:41-115) is a
// stream of ~200 ATen launches per step, here it is three kernels per RK stage:
//
//   k_cols  MODE_A  : u -> {u^,v^,dx w^,dy w^} -> column inverse FFT -> 4 planes
//   k_rows_advect   : 4 row c2r -> -(dx w * vx + dy w * vy) -> row r2c
//   k_cols  MODE_CA : column FFT -> mask, forcing, h <- F + beta h,
//                     u <- (u + g h + mu L u)/(1 - mu L) -> (MODE_A of next stage)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <limits>
#include <mutex>
#include <vector>

// ---- compilation units ---------------------------------------------------------------------------------------------------
// This file is compiled TWICE, in parallel (torch-cfd_amd/_lib.py): unit 0 (-DTCFD_UNIT=0) holds the C ABI and every
// float64 kernel instantiation, unit 1 (-DTCFD_UNIT=1) the float32 instantiations behind seven `*_f32` entry points that unit
// 0's (dtype, n) dispatch forwards to.  Built without the macro it is one unit with both.  (A single unit took 7 minutes of
// hipcc once the 3 * 2^k and 5 * 2^k grids joined; in unit 1 the ABI functions are compiled as unused statics.)
#ifndef TCFD_UNIT
#define TCFD_UNIT (-1)
#endif
#if TCFD_UNIT == 1
#define TCFD_H_TYPES_ONLY
#define TCFD_API [[maybe_unused]] static
#else
#define TCFD_API extern "C"
#endif
#include "../../include/tcfd.h"
#include "tcfd_fft.hpp"

using namespace tcfd;

// ------------------------------------------------------------------ errors
#if TCFD_UNIT == 1
int tcfd_set_error(int code, const char* fmt, ...);   // unit 0's
#else
static thread_local char g_err[512] = "";
int tcfd_set_error(int code, const char* fmt, ...) {  // shared with tcfd_fno.hip
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#endif
#define fail(...) tcfd_set_error(__VA_ARGS__)
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(TCFD_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#if TCFD_UNIT != 1
extern "C" const char* tcfd_last_error(void) { return g_err; }
extern "C" int tcfd_version(void) { return TCFD_ABI_VERSION; }
#endif

// ------------------------------------------------------------------ per-size configuration
// COL_EPT / ROW_EPT = elements per lane of one transform in the column / row kernels,
// COLS = columns per tile of the column kernels, ROW_THREADS = workgroup size of the row kernels.
// TCFD_NT_COL_IN=1 (build time): the advection a column pass transforms is read exactly once and dead afterwards
#ifndef TCFD_NT_COL_IN
#define TCFD_NT_COL_IN 0
#endif
#if TCFD_NT_COL_IN
#define TCFD_COL_LD(p_) load_stream(p_)
#else
#define TCFD_COL_LD(p_) (*(p_))
#endif
template <typename T, int N> struct Cfg;
#define TCFD_CFG(T, N, CEPT_, COLS_, REPT_, RTHR_)                                     \
    template <> struct Cfg<T, N> {                                                     \
        static constexpr int COL_EPT = CEPT_, COLS = COLS_, ROW_EPT = REPT_, ROW_THREADS = RTHR_; \
    };
TCFD_CFG(double, 8, 8, 64, 8, 256)
TCFD_CFG(double, 16, 4, 16, 4, 256)
TCFD_CFG(double, 32, 8, 32, 8, 256)
TCFD_CFG(double, 64, 8, 32, 8, 256)
TCFD_CFG(double, 128, 8, 16, 8, 256)
TCFD_CFG(double, 256, 16, 16, 16, 256)
TCFD_CFG(double, 512, 8, 8, 8, 256)
TCFD_CFG(double, 1024, 16, 8, 8, 128)
TCFD_CFG(double, 2048, 16, 2, 16, 256)
TCFD_CFG(float, 8, 8, 64, 8, 256)
TCFD_CFG(float, 16, 4, 16, 4, 256)
TCFD_CFG(float, 32, 8, 32, 8, 256)
TCFD_CFG(float, 64, 8, 32, 8, 256)
TCFD_CFG(float, 128, 8, 16, 8, 256)
TCFD_CFG(float, 256, 16, 16, 16, 256)
TCFD_CFG(float, 512, 8, 16, 8, 256)
TCFD_CFG(float, 1024, 16, 8, 16, 256)
TCFD_CFG(float, 2048, 16, 8, 16, 256)
// n = 3 * 2^k: twelve elements per lane (radix 12 = 4 x 3 in registers, then radix-4 passes), plain Stockham tiles
TCFD_CFG(double, 96, 12, 32, 12, 256)
TCFD_CFG(double, 192, 12, 16, 12, 256)
TCFD_CFG(double, 384, 12, 8, 12, 256)
TCFD_CFG(double, 768, 12, 8, 12, 256)
TCFD_CFG(float, 96, 12, 32, 12, 256)
TCFD_CFG(float, 192, 12, 16, 12, 256)
TCFD_CFG(float, 384, 12, 16, 12, 256)
TCFD_CFG(float, 768, 12, 16, 12, 256)
TCFD_CFG(double, 1536, 12, 4, 12, 256)     // 4 (8) columns: a whole-column tile of 8 (16) would not fit the LDS
TCFD_CFG(float, 1536, 12, 8, 12, 256)
// n = 5 * 2^k: twenty elements per lane (radix 20 = 4 x 5 in registers, then radix 4 / 2)
TCFD_CFG(double, 80, 20, 32, 20, 256)
TCFD_CFG(double, 160, 20, 16, 20, 256)
TCFD_CFG(double, 320, 20, 8, 20, 256)
TCFD_CFG(double, 640, 20, 8, 20, 256)
TCFD_CFG(float, 80, 20, 32, 20, 256)
TCFD_CFG(float, 160, 20, 16, 20, 256)
TCFD_CFG(float, 320, 20, 16, 20, 256)
TCFD_CFG(float, 640, 20, 16, 20, 256)
TCFD_CFG(double, 1280, 20, 4, 20, 256)
TCFD_CFG(float, 1280, 20, 8, 20, 256)

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// ------------------------------------------------------------------ column kernels
enum ColMode {
    MODE_A = 0,     // u_in -> 4 planes
    MODE_CA = 1,    // adv -> FFT -> RK update (h, u_out) -> 4 planes
    MODE_C = 2,     // adv -> FFT -> RK update (h, u_out)
    MODE_F = 3,     // adv -> FFT -> out = mask*x + forcing
    MODE_RES = 4,   // adv -> FFT -> residual = wt - F - L w ; psi = -w/lap
    MODE_FWD = 5,   // generic column forward c2c:  in -> out * scale
    MODE_INV = 6    // generic column inverse c2c:  in -> out * scale
};

template <typename T>
struct ColArgs {
    const cx<T>* in;      // adv (MODE_CA/C/F/RES) or generic input
    const cx<T>* u_in;    // state read  (w for MODE_RES)
    const cx<T>* wt;      // MODE_RES only
    cx<T>* u_out;         // state write
    cx<T>* h;             // RK accumulator (read unless load_h == 0, written)
    cx<T>* planes;        // 4 output planes (MODE_A/CA)
    cx<T>* out;           // MODE_F / FWD / INV output ; MODE_RES residual
    cx<T>* psi;           // MODE_RES
    const cx<T>* w0;      // MODE_C with dwdt: the call's input (caller layout), dwdt = (u_new - w0) * dwdt_scale
    cx<T>* dwdt;          // or null
    const T* kx;          // [n]
    const T* ky;          // [m]
    const T* lin;         // [n*m]
    const T* mask;        // [n*m]
    const cx<T>* forcing; // [n*m] or null (dense form)
    // separable / sparse forms of the same tables (chosen at plan creation when they are exact):
    const T* mask_r;      // [n], mask = mask_r[i] * mask_c[j]
    const T* mask_c;      // [m]
    const T* lin_r;       // [n], linear_term = lin_r[i] + lin_c[j]
    const T* lin_c;       // [m]
    const int* f_ptr;     // [m+1] CSC column pointers of the non-zero forcing entries, or null
    const int* f_row;     // [nnz]
    const int* f_col;     // [nnz]
    const cx<T>* f_val;   // [nnz]
    int f_nnz;            // > 0: so few entries that every lane scans the whole list (uniform scalar loads)
    int sep;              // 1: use the separable mask / linear term
    int keep_cols;        // > 0: columns >= keep_cols and rows with mask_r == 0 carry F = h = 0 (pruned)
    const cx<T>* tw;      // [n]
    size_t plane_stride;  // elements between planes
    T beta, gdt, mu, scale;
    T dwdt_scale;
    T fa, mud;   // h <- fa F + beta h ;  u <- (base + gdt h + mu L base) / (1 - mud L)   (RK4-CN: fa = 1, mud = mu)
    int m;        // row pitch (elements) of caller-layout arrays: u_in, u_out, wt, out, psi
    int ldw;      // row pitch of workspace arrays (adv, h, planes): m rounded up to a 128-byte multiple
    int in_ld;    // MODE_FWD / MODE_INV: pitch of `in`
    int out_ld;   // MODE_FWD / MODE_INV: pitch of `out`
    int u_in_ld;  // MODE_A / CA / C: pitch of u_in (m for the caller's array, ldw for the internal state copy)
    int u_out_ld; // MODE_CA / C: pitch of u_out
    int ntiles;
    int batch;
    int load_h;
    int store_h;   // 0 at the last stage of a step: the accumulator is dead (the next step starts from h = 0)
    int ablate;    // timing ablations (TCFD_ABLATE bit mask; results are WRONG when non-zero): 1 skip the
                   // transforms, 2 skip the plane stores, 4 skip the table reads, 8 skip the h traffic,
                   // 64 skip the packing of the Nyquist column into the planes, 128 skip that column's update
    int nyq;       // 1: packed Nyquist column (see emit_planes): no lone tile for column m - 1, the lanes of column 0 own it too
    int nt_planes; // plane stores with the non-temporal hint (small cache-resident problems, see launch_cols)
    int nt_out;    // MODE_C: the new state and dw/dt go to the CALLER's arrays and are not read again by this call -- stored with
                   // the non-temporal hint they do not push the next chunk's working set out of the Infinity Cache
    int pair_xcd;  // block->tile map: 0 = batch fastest; LG = 2 / 4: the LG tiles that share a 128-byte line on one XCD
    cx<T>* h_out;  // RK accumulator write (NULL: == h)
};

// SP = 1 ("split"): the workgroup owns the rows of ONE parity q of the column (i = 2 s + q) and runs N/2-point
// transforms on them; the radix-2 butterfly that joins the two parities rides in the row kernel, which works
// on row pairs (r, r + N/2) anyway.  Workspace rows [0, N/2) then hold the even-part transforms E (resp. the
// folded sums S of the advection), rows [N/2, N) the odd parts O (resp. the twiddled differences D).
// XL = 1 (512-point tiles of 8 columns): the transforms are xl_col_fft512 (one LDS exchange + lane transpositions);
// register t of thread j then holds physical row xl_col_row(j) + t G instead of j + t G.
// The twiddles of a column workgroup's transforms, read ONCE per thread: a MODE_CA workgroup runs five transforms one
// after the other, and a table read inside every pass (behind that pass's barrier) is a memory round trip on its
// dependency chain each time: the s = 0 entry of every pass (tile_fft_rt: the powers come from squaring) for the short
// columns of small grids -- the latency-bound regime -- where the transform shape allows it; else nothing (table reads
// per pass; the long-column kernels sit at their 128-register cap, and the cross-lane tiles read two entries per transform).
template <typename T, int NT, int EPT, int XL>
struct ColTw {
    static constexpr bool RT = !XL && NT <= 256 && is_pow2c(NT) && is_pow2c(EPT) && pass_tw_count<NT, EPT>() > 0 && pass_tw_single<NT, EPT>();
    static constexpr int CNT = RT ? pass_tw_count<NT, EPT>() : 1;
    cx<T> w[CNT];
    __device__ __forceinline__ void load(const cx<T>* __restrict__ tw, int j) {
        if constexpr (RT) load_pass_tw<T, NT, EPT>(w, tw, j);
    }
};
template <typename T, int NT, int EPT, int DIR, int C, int XL, int XLT>
__device__ __forceinline__ void col_fft(cx<T> (&x)[EPT], cx<T>* lds, const cx<T>* __restrict__ tw,
                                        const ColTw<T, NT, EPT, XLT>& ctw, int j, int c) {
    if constexpr (XL) {   // two table entries, read per transform: 8 more live registers spill the fp64 kernels
        const XlColTw<T> t = xl_col_load_tw<T>(tw, j);
        xl_col_fft512<T, DIR>(x, lds, t, (int)threadIdx.x);
    } else if constexpr (!XLT && ColTw<T, NT, EPT, XLT>::RT) {
        NoHook nohook;
        tile_fft_rt<T, NT, EPT, DIR, C, false, true>(x, lds, j, c, nohook, ctw.w);
    } else {
        tile_fft<T, NT, EPT, DIR, C, false, true>(x, lds, tw, j, c);
    }
}

// f=0: u^ = 2 pi i ky psi   f=1: v^ = -2 pi i kx psi   f=2: dx w^ = 2 pi i kx w   f=3: dy w^ = 2 pi i ky w,
// psi = -w / lap with lap(0,0) patched to 1 (`origin`); `us` = w^ / n^2
template <typename T>
__device__ __forceinline__ cx<T> plane_value(int f, cx<T> us, T kx, T ky, bool origin) {
    constexpr T TWO_PI = (T)6.283185307179586476925286766559;
    constexpr T M4PI2 = (T)(-39.478417604357434475337963999505);
    if (f < 2) {
        T lap = M4PI2 * (kx * kx + ky * ky);
        if (origin) lap = (T)1;
        us = cscale(us, -fast_rcp(lap));
    }
    const T kk = (f == 0 || f == 3) ? ky : kx;
    cx<T> v = mul_i(cscale(us, TWO_PI * kk));
    if (f == 1) v = mk<T>(-v.x, -v.y);
    return v;
}

// PACKED NYQUIST COLUMN (a.nyq).  m = n/2 + 1 columns are n/2 / C full tiles plus ONE lone column, whose tile costs a
// whole workgroup: with 4 fields of 1024^2 per chunk that makes 520 workgroups for the 512 resident slots, and the
// 8 late ones stretch every launch by most of a workgroup's run time (measured: MODE_A 54 -> 39 us without them).
// The row pass consumes only the REAL parts of columns 0 and m - 1 of a plane (c2r semantics: Im of the DC and
// Nyquist coefficients of a row is dropped).  Re(IFFT(A)) = IFFT(herm A), herm A[k] = (A[k] + conj A[-k]) / 2, so
// both columns ride through ONE transform,
//     Z = herm(A_0) + i herm(A_nyq)   ->   IFFT(Z) = Re(IFFT A_0) + i Re(IFFT A_nyq),
// stored in column 0: the row pass takes its Nyquist element from the imaginary part of element 0.  (With the split
// plans the same holds per parity: the E / O half transforms and the row pass's butterfly are complex-linear.)
// Tile 0 builds Z through the exchange buffer (free between two transforms), one row per thread of the workgroup.
// The Nyquist column's own values wait in LDS (`un`: the rt_lin / rt_mask tables are dead by then and exactly that
// big): registers held across the four transforms would spill the fp64 kernels.
template <typename T, int N, int EPT, int C, int SP, int XL>
__device__ __forceinline__ void emit_planes(const ColArgs<T>& a, const cx<T> (&u)[EPT], cx<T>* lds, const T* rt_kx,
                                            size_t wbase, int j, int c, int jc, bool valid, int q,
                                            const ColTw<T, (N >> SP), EPT, XL>& ctw, bool nyq_tile, const cx<T>* un) {
    constexpr int NT = N >> SP;
    constexpr int G = NT / EPT;
    const T inv_n2 = (T)1 / ((T)N * (T)N);
    const T ky = valid ? a.ky[jc] : (T)0;
    const bool nyq_lane = nyq_tile && c == 0;
    const T ky_n = nyq_tile ? a.ky[a.m - 1] : (T)0;
#pragma unroll 1
    for (int f = 0; f < 4; ++f) {
        cx<T> x[EPT];
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            const int sl = j + t * G;
            const int i = SP ? 2 * sl + q : sl;
            const cx<T> v = plane_value<T>(f, cscale(u[t], inv_n2), rt_kx[sl], ky, i == 0 && jc == 0);
            x[t] = valid ? v : mk<T>((T)0, (T)0);
        }
        if (nyq_tile && !(a.ablate & 64)) {   // block-uniform.  Three short phases with the WHOLE workgroup (a row per thread), so that
            cx<T>* S0 = lds;           // tile 0 does not become the launch's long pole
            cx<T>* SZ = lds + NT;
            if (nyq_lane) {
#pragma unroll
                for (int t = 0; t < EPT; ++t) S0[j + t * G] = x[t];
            }
            __syncthreads();
            for (int sl = threadIdx.x; sl < NT; sl += C * G) {
                // row -i of the column: local index of the same parity
                const int ms = (SP && q) ? NT - 1 - sl : (NT - sl) % NT;
                const cx<T> v0 = S0[sl], m0 = S0[ms];
                const cx<T> vn = plane_value<T>(f, cscale(un[sl], inv_n2), rt_kx[sl], ky_n, false);
                const cx<T> mn = plane_value<T>(f, cscale(un[ms], inv_n2), rt_kx[ms], ky_n, false);
                const cx<T> h0 = mk<T>((T)0.5 * (v0.x + m0.x), (T)0.5 * (v0.y - m0.y));
                const cx<T> hn = mk<T>((T)0.5 * (vn.x + mn.x), (T)0.5 * (vn.y - mn.y));
                SZ[sl] = mk<T>(h0.x - hn.y, h0.y + hn.x);   // h0 + i hn
            }
            __syncthreads();
            if (nyq_lane) {
#pragma unroll
                for (int t = 0; t < EPT; ++t) x[t] = SZ[j + t * G];
            }
            __syncthreads();
        }
        const int jrow = XL ? xl_col_row(j) : j;   // physical row (mod G) of this thread's outputs
        if (!(a.ablate & 1)) col_fft<T, NT, EPT, +1, C, XL, XL>(x, lds, a.tw, ctw, j, c);
        if (valid && !(a.ablate & 2)) {
            cx<T>* dst = a.planes + (size_t)f * a.plane_stride + wbase + (size_t)(SP ? q * NT : 0) * a.ldw;
            if (a.nt_planes) {
#pragma unroll
                for (int t = 0; t < EPT; ++t) store_stream(dst + (size_t)(jrow + t * G) * a.ldw, x[t]);
            } else {
#pragma unroll
                for (int t = 0; t < EPT; ++t) dst[(size_t)(jrow + t * G) * a.ldw] = x[t];
            }
        }
    }
}

// x[t] (column FFT of the advection) -> F = mask * x + forcing, in place (equations.py:424-437)
template <typename T, int N, int EPT, int SP>
__device__ __forceinline__ void apply_mask_forcing(const ColArgs<T>& a, cx<T> (&x)[EPT], int j, int jc,
                                                   const T* rt_mask, T cm, int f_lo, int f_hi, int q) {
    constexpr int G = (N >> SP) / EPT;
#define TCFD_IROW(t_) (SP ? 2 * (j + (t_) * G) + q : j + (t_) * G)
    if (a.ablate & 4) return;
    if (a.sep) {
#pragma unroll
        for (int t = 0; t < EPT; ++t) x[t] = cscale(x[t], rt_mask[j + t * G] * cm);
    } else {
#pragma unroll
        for (int t = 0; t < EPT; ++t) x[t] = cscale(x[t], a.mask[(size_t)TCFD_IROW(t) * a.m + jc]);
    }
    if (a.f_nnz > 0) {  // a handful of entries in total (1 for Kolmogorov forcing): uniform loop
        for (int e = 0; e < a.f_nnz; ++e) {
            const int r = a.f_row[e];
            const int fc = a.f_col[e];      // all three reads of an entry before the branch: one round trip, not two
            const cx<T> v = a.f_val[e];
            if (fc == jc) {
#pragma unroll
                for (int t = 0; t < EPT; ++t)
                    if (r == TCFD_IROW(t)) x[t] = x[t] + v;
            }
        }
    } else if (a.f_ptr) {  // sparse, many entries: this column's slice of the CSC list
        for (int e = f_lo; e < f_hi; ++e) {
            const int r = a.f_row[e];
            const cx<T> v = a.f_val[e];
#pragma unroll
            for (int t = 0; t < EPT; ++t)
                if (r == TCFD_IROW(t)) x[t] = x[t] + v;
        }
    } else if (a.forcing) {
#pragma unroll
        for (int t = 0; t < EPT; ++t) x[t] = x[t] + a.forcing[(size_t)TCFD_IROW(t) * a.m + jc];
    }
#undef TCFD_IROW
}

// NYQ = 1: the variant that can run with a.nyq (packed Nyquist column); kept apart because the extra code costs the
// register-capped fp64 kernels a few spilled registers in EVERY workgroup, packed or not
template <typename T, int N, int EPT, int C, int MODE, int MINW, int SP = 0, int XL = 0, int NYQ = 0>
__global__ __launch_bounds__(C*((N >> SP) / EPT), MINW) void k_cols(ColArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem_raw);
    constexpr int NT = N >> SP;  // transform length (half the column when the workgroup owns one parity)
    constexpr int G = NT / EPT;
    static_assert(!(SP && (MODE == MODE_FWD || MODE == MODE_INV)), "plain transforms are not split");
    // parity of the spectral rows this workgroup owns: the LOW bit of the block id, so that the two parity workgroups
    // of a tile are dispatched next to each other -- to different XCDs.  (As blockIdx.y they were 256 ids apart, i.e.
    // in the two slots of ONE CU: with the packed Nyquist column both long workgroups of tile 0 shared that CU.)
    const int q = SP ? (int)(blockIdx.x & 1) : 0;
    const unsigned bx = SP ? blockIdx.x >> 1 : blockIdx.x;
#define TCFD_IROW(s_) (SP ? 2 * (s_) + q : (s_))
    const size_t wq = (size_t)(SP ? q * NT : 0);  // first workspace row of this parity
    const int c = threadIdx.x % C;
    const int j = threadIdx.x / C;
    // batch is the fastest block index: the blocks that share one tile of the (n, m) tables
    // (linear term, mask, forcing) run together and hit them in L2
    int tile;
    long b;
    if (a.pair_xcd) {
        // tiles narrower than a 128-byte line: the LG = 2 or 4 tiles that share a line are issued 8 block ids apart,
        // i.e. on the SAME XCD (block id % 8): their partial-line stores merge in that XCD's L2 into whole dirty
        // lines before the end-of-kernel write-back (each XCD writing its own 32 bytes of a line costs a fabric
        // transaction per piece: the plane stores were 7 of the 17 us of a 256^2 x 16 column pass)
        const unsigned LG = (unsigned)a.pair_xcd, span = 8 * LG;
        const unsigned q = bx / span, r = bx % span;
        const unsigned unit = q * 8 + (r % 8);
        if (unit >= (unsigned)(((a.ntiles + LG - 1) / LG) * a.batch)) return;
        tile = (int)(LG * (unit / a.batch) + r / 8);
        b = unit % a.batch;
    } else {
        tile = bx / a.batch;
        b = bx % a.batch;
    }
    const int jc = tile * C + c;
    const bool valid = jc < a.m;
    const bool nyq_tile = NYQ && a.nyq && tile == 0;   // block-uniform: the lanes c == 0 of this workgroup also own column m - 1
    const bool nyq_lane = nyq_tile && c == 0;
    const size_t colbase = (size_t)b * N * a.m + jc;   // element (b, 0, jc) of a caller-layout array
    const size_t wbase = (size_t)b * N * a.ldw + jc;   // same element of a workspace array

    // per-row constants (kx, separable parts of the linear term and of the mask) live in LDS behind the
    // exchange tile: every later use is an LDS read instead of a dependent global load; per-column
    // constants and the forcing column range are fetched up front, before the transform needs them
    T* rt_kx = reinterpret_cast<T*>(lds + lds_elems<NT, EPT, C, false>());
    T* rt_lin = rt_kx + NT;
    T* rt_mask = rt_lin + NT;
    cx<T>* un_lds = reinterpret_cast<cx<T>*>(rt_lin);   // [NT]: Nyquist-column state (a.nyq), overlays rt_lin + rt_mask
    constexpr bool NEEDS_TABLES = (MODE != MODE_FWD && MODE != MODE_INV);
    T col_ky = 0, col_lin = 0, col_mask = 1;
    int f_lo = 0, f_hi = 0;
    // Called AFTER the tile's own loads have been issued: the table reads (three loads whose results go to LDS, in a
    // loop the compiler does not move loads across) then ride in the shadow of the tile's memory round trip instead
    // of adding two of their own in front of it.  The null tables of a non-separable plan are read through `kx`.
    auto stage_tables = [&]() {
        if constexpr (NEEDS_TABLES) {
            const T* lin_r = a.sep ? a.lin_r : a.kx;
            const T* mask_r = a.sep ? a.mask_r : a.kx;
            for (int sl = threadIdx.x; sl < NT; sl += C * G) {  // indexed by the LOCAL row sl, spectral row IROW(sl)
                const int i = TCFD_IROW(sl);
                const T vk = a.kx[i], vl = lin_r[i], vm = mask_r[i];
                rt_kx[sl] = vk;
                rt_lin[sl] = a.sep ? vl : (T)0;
                rt_mask[sl] = a.sep ? vm : (T)1;
            }
            if (valid) {
                col_ky = a.ky[jc];
                if (a.sep) { col_lin = a.lin_c[jc]; col_mask = a.mask_c[jc]; }
                if (a.f_ptr && a.f_nnz == 0) { f_lo = a.f_ptr[jc]; f_hi = a.f_ptr[jc + 1]; }
            }
        }
    };

    ColTw<T, NT, EPT, XL> ctw;
    ctw.load(a.tw, j);
    cx<T> x[EPT];
    if constexpr (MODE == MODE_A) {
        {
            const size_t ub = (size_t)b * N * a.u_in_ld + jc;
#pragma unroll
            for (int t = 0; t < EPT; ++t)
                x[t] = valid ? a.u_in[ub + (size_t)TCFD_IROW(j + t * G) * a.u_in_ld] : mk<T>((T)0, (T)0);
        }
        stage_tables();
        __syncthreads();  // row tables visible
        if (nyq_tile) {   // rt_lin / rt_mask are not used in this mode; visible after the first barrier of emit_planes
            const cx<T>* uin = a.u_in + (size_t)b * N * a.u_in_ld + (a.m - 1);
            for (int sl = threadIdx.x; sl < NT; sl += C * G) un_lds[sl] = uin[(size_t)TCFD_IROW(sl) * a.u_in_ld];
        }
        emit_planes<T, N, EPT, C, SP, XL>(a, x, lds, rt_kx, wbase, j, c, jc, valid, q, ctw, nyq_tile, un_lds);
        return;
    } else {
        constexpr bool GENERIC = (MODE == MODE_FWD || MODE == MODE_INV);
        const int in_ld = GENERIC ? a.in_ld : a.ldw;
        const size_t inbase = (size_t)b * N * in_ld + jc;
        // pruned plans never write the advection for columns >= keep_cols: they are exact zeros
        const bool col_live = valid && (GENERIC || a.keep_cols == 0 || jc < a.keep_cols);
        const bool tile_live = GENERIC || a.keep_cols == 0 || tile * C < a.keep_cols;  // block-uniform
        constexpr int DIR = (MODE == MODE_INV) ? +1 : -1;
        constexpr bool XLF = XL && !GENERIC;   // forward transform of the advection: physical rows come in at xl_col_row
        const int jin = XLF ? xl_col_row(j) : j;
#pragma unroll
        for (int t = 0; t < EPT; ++t)
            x[t] = col_live ? TCFD_COL_LD(a.in + inbase + (wq + (size_t)(jin + t * G)) * in_ld) : mk<T>((T)0, (T)0);
        stage_tables();
        if (!(a.ablate & 1) && tile_live) col_fft<T, NT, EPT, DIR, C, XLF ? 1 : 0, XL>(x, lds, a.tw, ctw, j, c);
        if constexpr (NEEDS_TABLES) __syncthreads();  // row tables visible (a one-pass transform has no barrier)

        if constexpr (MODE == MODE_FWD || MODE == MODE_INV) {
            if (valid) {
                const size_t outbase = (size_t)b * N * a.out_ld + jc;
#pragma unroll
                for (int t = 0; t < EPT; ++t) a.out[outbase + (size_t)(j + t * G) * a.out_ld] = cscale(x[t], a.scale);
            }
            return;
        } else if constexpr (MODE == MODE_F) {
            if (valid) {
                apply_mask_forcing<T, N, EPT, SP>(a, x, j, jc, rt_mask, col_mask, f_lo, f_hi, q);
#pragma unroll
                for (int t = 0; t < EPT; ++t) a.out[colbase + (size_t)TCFD_IROW(j + t * G) * a.m] = x[t];
            }
            return;
        } else if constexpr (MODE == MODE_RES) {
            if (valid) {
                constexpr T M4PI2 = (T)(-39.478417604357434475337963999505);
                const T ky = col_ky;
                apply_mask_forcing<T, N, EPT, SP>(a, x, j, jc, rt_mask, col_mask, f_lo, f_hi, q);
                const T lc = col_lin;
#pragma unroll
                for (int t = 0; t < EPT; ++t) {
                    const int sl = j + t * G;
                    const int i = TCFD_IROW(sl);
                    const size_t g = colbase + (size_t)i * a.m;
                    const cx<T> F = x[t];
                    const cx<T> w = a.u_in[g];
                    if (a.out) {
                        const cx<T> wt = a.wt[g];
                        const T L = a.sep ? rt_lin[sl] + lc : a.lin[(size_t)i * a.m + jc];
                        a.out[g] = wt - F - cscale(w, L);
                    }
                    if (a.psi) {
                        const T kx = rt_kx[sl];
                        T lap = M4PI2 * (kx * kx + ky * ky);
                        if (i == 0 && jc == 0) lap = (T)1;
                        a.psi[g] = cscale(w, (T)-1 / lap);
                    }
                }
            }
            return;
        } else {  // MODE_CA / MODE_C : Runge-Kutta accumulate + Crank-Nicolson solve
            constexpr bool writer = true;
            // packed Nyquist column (tile 0, a row per thread): its reads go out first, in the shadow of the tile's own
            constexpr int NYQ_PER = (NT + C * G - 1) / (C * G);
            [[maybe_unused]] cx<T> nyq_u[NYQ_PER], nyq_w0[NYQ_PER];
            if (nyq_tile) {
#pragma unroll
                for (int r = 0; r < NYQ_PER; ++r) {
                    const int sl = threadIdx.x + r * C * G;
                    if (sl < NT) {
                        const int i = TCFD_IROW(sl);
                        nyq_u[r] = a.u_in[(size_t)b * N * a.u_in_ld + (a.m - 1) + (size_t)i * a.u_in_ld];
                        if constexpr (MODE == MODE_C) {
                            if (a.dwdt) nyq_w0[r] = a.w0[(size_t)b * N * a.m + (a.m - 1) + (size_t)i * a.m];
                        }
                    }
                }
            }
            if (valid) {
                // Every global read of the update is issued before the first value is used: the reads of one thread
                // are independent, but `h` and the state are updated in place through pointers that may alias, so
                // a load written after an earlier element's store is not moved above it -- 2 EPT memory round trips
                // one after the other.  In the cache-resident regimes (chunked 1024^2, small grids) a workgroup's
                // dependency chain IS the kernel time.
                const T lc = col_lin;
                const size_t ub = (size_t)b * N * a.u_in_ld + jc;
                apply_mask_forcing<T, N, EPT, SP>(a, x, j, jc, rt_mask, col_mask, f_lo, f_hi, q);
                // UB elements per batch of reads: everything at once where the registers are there (fp32, short
                // columns), 64 bytes' worth per array otherwise (the long-column kernels sit at the 128-register cap)
                constexpr int UB = EPT * (int)sizeof(cx<T>) <= 64 ? EPT : 64 / (int)sizeof(cx<T>);   // 8 fp32 / 4 fp64
#pragma unroll
                for (int t0 = 0; t0 < EPT; t0 += UB) {
                    cx<T> hv[UB], uv[UB];
                    [[maybe_unused]] cx<T> w0v[UB];
                    T Lv[UB];
                    bool h_live[UB];
#pragma unroll
                    for (int tt = 0; tt < UB; ++tt) {
                        const int sl = j + (t0 + tt) * G;
                        const int i = TCFD_IROW(sl);
                        // pruned entries: F = 0 at every stage, so h stays 0 and is not stored at all
                        h_live[tt] = !(a.ablate & 8) && (a.keep_cols == 0 || (col_live && rt_mask[sl] != (T)0));
                        hv[tt] = (h_live[tt] && a.load_h) ? a.h[wbase + (wq + (size_t)sl) * a.ldw] : mk<T>((T)0, (T)0);
                        uv[tt] = a.u_in[ub + (size_t)i * a.u_in_ld];
                        Lv[tt] = (a.ablate & 4) ? (T)-0.5 : (a.sep ? rt_lin[sl] + lc : a.lin[(size_t)i * a.m + jc]);
                        if constexpr (MODE == MODE_C) {
                            if (a.dwdt) w0v[tt] = a.w0[(size_t)b * N * a.m + jc + (size_t)i * a.m];
                        }
                    }
#pragma unroll
                    for (int tt = 0; tt < UB; ++tt) {
                        const int t = t0 + tt;
                        const int sl = j + t * G;
                        const int i = TCFD_IROW(sl);
                        const size_t gw = wbase + (wq + (size_t)sl) * a.ldw;  // h is stored parity-major when split
                        const cx<T> hn = cscale(x[t], a.fa) + cscale(hv[tt], a.beta);   // hv = 0: nothing to add
                        if (h_live[tt] && a.store_h && writer) a.h_out[gw] = hn;
                        const T L = Lv[tt];
                        const cx<T> u = uv[tt];
                        // u + gamma dt h + mu L u, then / (1 - mu L)     (equations.py:355-357)
                        cx<T> rhs = u + cscale(hn, a.gdt) + cscale(cscale(u, L), a.mu);
                        const T den = fast_rcp((T)1 - a.mud * L);
                        x[t] = cscale(rhs, den);
                        if constexpr (MODE == MODE_C) {
                            cx<T>* const po = a.u_out + (size_t)b * N * a.u_out_ld + jc + (size_t)i * a.u_out_ld;
                            if (writer) {
                                if (a.nt_out) store_stream(po, x[t]);
                                else *po = x[t];
                            }
                            if (a.dwdt) {   // (w_new - w_old) / (steps dt), equations.py:461-462, fused into the last stage
                                const size_t gi = (size_t)b * N * a.m + jc + (size_t)i * a.m;
                                const cx<T> dv = cscale(x[t] - w0v[tt], a.dwdt_scale);
                                if (a.nt_out) store_stream(a.dwdt + gi, dv);
                                else a.dwdt[gi] = dv;
                            }
                        } else {
                            if (writer) a.u_out[(size_t)b * N * a.u_out_ld + jc + (size_t)i * a.u_out_ld] = x[t];
                        }
                    }
                }
            }
            if (nyq_tile && !(a.ablate & 128)) {   // block-uniform
                // column m - 1, a row per thread: F = 0 there (the plan is pruned: keep_cols <= m - 1), so h stays 0 and
                // the stage is  u <- (u + mu L u) / (1 - mud L)
                const int jn = a.m - 1;
                const T lcn = a.sep ? a.lin_c[jn] : (T)0;
                constexpr int PER = NYQ_PER;
                cx<T> unew[PER];
#pragma unroll
                for (int r = 0; r < PER; ++r) {
                    const int sl = threadIdx.x + r * C * G;
                    if (sl < NT) {
                        const int i = TCFD_IROW(sl);
                        const cx<T> uo = nyq_u[r];
                        const T L = (a.ablate & 4) ? (T)-0.5 : (a.sep ? rt_lin[sl] + lcn : a.lin[(size_t)i * a.m + jn]);
                        const cx<T> rhs = uo + cscale(cscale(uo, L), a.mu);
                        unew[r] = cscale(rhs, fast_rcp((T)1 - a.mud * L));
                        if (writer) a.u_out[(size_t)b * N * a.u_out_ld + jn + (size_t)i * a.u_out_ld] = unew[r];
                        if constexpr (MODE == MODE_C) {
                            if (a.dwdt) {
                                const size_t gi = (size_t)b * N * a.m + jn + (size_t)i * a.m;
                                a.dwdt[gi] = cscale(unew[r] - nyq_w0[r], a.dwdt_scale);
                            }
                        }
                    }
                }
                if constexpr (MODE == MODE_CA) {
                    __syncthreads();   // every lane is done with rt_lin / rt_mask: the Nyquist column moves in
#pragma unroll
                    for (int r = 0; r < PER; ++r) {
                        const int sl = threadIdx.x + r * C * G;
                        if (sl < NT) un_lds[sl] = unew[r];
                    }
                }
            }
            if constexpr (MODE == MODE_CA)
                emit_planes<T, N, EPT, C, SP, XL>(a, x, lds, rt_kx, wbase, j, c, jc, valid, q, ctw, nyq_tile, un_lds);
        }
    }
#undef TCFD_IROW
}

// ------------------------------------------------------------------ row kernels
// One group of G lanes owns one PAIR of adjacent rows: two real rows ride through
// one complex transform (x0 + i x1), so c2r/r2c cost one complex FFT per two rows.

// Hermitian-extended load of the pair (A row, B row): sequence Z[e] = A~[e] + i B~[e]
// where A~ is the Hermitian completion of the half row with Im(DC), Im(Nyquist)
// dropped -- exactly what a c2r transform consumes (SURVEY note N2).
template <typename T, int N, int EPT>
__device__ __forceinline__ void load_herm_pair(cx<T> (&x)[EPT], const cx<T>* __restrict__ rowA,
                                               const cx<T>* __restrict__ rowB, int j) {
    constexpr int G = N / EPT;
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        const int e = j + t * G;
        if (t < EPT / 2) {
            cx<T> pa = rowA[e], pb = rowB[e];
            if (e == 0) { pa.y = 0; pb.y = 0; }
            x[t] = mk<T>(pa.x - pb.y, pa.y + pb.x);
        } else {
            const int k = N - e;  // 1 .. N/2
            cx<T> pa = rowA[k], pb = rowB[k];
            if (k == N / 2) { pa.y = 0; pb.y = 0; }
            // conj(a) + i conj(b)
            x[t] = mk<T>(pa.x + pb.y, pb.x - pa.y);
        }
    }
}

// x holds Z = FFT(r0 + i r1) distributed j + t*G; writes the half spectra of r0, r1.
template <typename T, int N, int EPT>
__device__ __forceinline__ void unpack_store_pair(const cx<T> (&x)[EPT], cx<T>* lds, cx<T>* __restrict__ out0,
                                                  cx<T>* __restrict__ out1, int j, bool valid, int kc = N) {
    constexpr int G = N / EPT;
    constexpr bool WG = (G > 64);
#pragma unroll
    for (int t = 0; t < EPT; ++t) lds[lds_addr<EPT, 1, true>(j + t * G, 0)] = x[t];
    group_sync<WG>();
    const T half = (T)0.5;
#pragma unroll
    for (int t = 0; t < EPT / 2; ++t) {
        const int k = j + t * G;
        const cx<T> A = x[t];
        const cx<T> Bm = lds[lds_addr<EPT, 1, true>((N - k) % N, 0)];
        const cx<T> S = mk<T>(A.x + Bm.x, A.y - Bm.y);  // A + conj(B) = 2 X0
        const cx<T> D = mk<T>(A.x - Bm.x, A.y + Bm.y);  // A - conj(B) = 2 i X1
        if (valid && k < kc) {  // kc: columns the consumer reads (2/3-rule pruning)
            out0[k] = cscale(S, half);
            out1[k] = mk<T>(D.y * half, -D.x * half);
        }
    }
    if (j == 0) {
        const cx<T> A = x[EPT / 2];  // element N/2
        if (valid && N / 2 < kc) {
            out0[N / 2] = mk<T>(A.x, (T)0);
            out1[N / 2] = mk<T>(A.y, (T)0);
        }
    }
    group_sync<WG>();
}

template <typename T, int N, int EPT, int THREADS_ = 256>
struct RowGeom {
    static constexpr int G = N / EPT;
    static constexpr int THREADS = G >= THREADS_ ? G : THREADS_;
    static constexpr int GROUPS = THREADS / G;
    static constexpr int LDS_PER_GROUP = lds_elems<N, EPT, 1, true>();
    static constexpr size_t LDS_BYTES = (size_t)GROUPS * LDS_PER_GROUP * sizeof(cx<T>);
};

// ---- row pass, software-pipelined ("v3") -------------------------------------------------
// The un-mirrored half rows of two planes as they come from HBM: k = j + G*t, t < EPT/2, plus the
// Nyquist element.  A group keeps TWO of these in registers: the one being consumed and the one in
// flight, so a load phase is never exposed (the plain kernel above alternates load and transform
// phases and reaches only ~3 TB/s at 1024^2).
template <typename T, int EPT>
struct RawPair {
    cx<T> a[EPT / 2], b[EPT / 2];
    cx<T> an, bn;  // element N/2 of both rows (only lane 0 of the group uses it)
};

// nyq (packed Nyquist column, see emit_planes): column N/2 of the planes is not there; the Nyquist element of a row
// is the imaginary part of its element 0 and comes out of pack_herm
// TCFD_NT_ROW_LOADS=1 (build time): the planes a row pass reads are read exactly once by that launch -- non-temporal loads
#ifndef TCFD_NT_ROW_LOADS
#define TCFD_NT_ROW_LOADS 0
#endif
#if TCFD_NT_ROW_LOADS
#define TCFD_ROW_LD(p_) load_stream(p_)
#else
#define TCFD_ROW_LD(p_) (*(p_))
#endif
template <typename T, int N, int EPT, int NYQ = 0>
__device__ __forceinline__ void load_raw(RawPair<T, EPT>& r, const cx<T>* __restrict__ rowA,
                                         const cx<T>* __restrict__ rowB, int j) {
    constexpr int G = N / EPT;
#pragma unroll
    for (int t = 0; t < EPT / 2; ++t) {
        r.a[t] = TCFD_ROW_LD(rowA + j + t * G);
        r.b[t] = TCFD_ROW_LD(rowB + j + t * G);
    }
    if constexpr (!NYQ) {
        r.an = rowA[N / 2];
        r.bn = rowB[N / 2];
    }
}

// Z[e] = A~[e] + i B~[e] (Hermitian completions, Im of DC / Nyquist dropped) in the j + t*G
// distribution.  The upper half is the mirror image of data owned by OTHER lanes: it goes through
// the group's LDS buffer (half an exchange) instead of being loaded from HBM a second time.
// SWZ: the XOR swizzle of the Stockham exchanges (lds_addr<.., true>).  The cross-lane row kernel (v7) has no stride-EPT
// stores -- its transform exchanges through xl_slot -- and passes SWZ = false: with the swizzle the MIRRORED accesses
// (lanes j -> slots N - j, descending) straddle two 8-element blocks with different XOR constants and collide pairwise
// (rocprofv3: SQ_LDS_BANK_CONFLICT 0.84 M of 4.03 M LDS-array cycles per chunk launch); unswizzled, descending consecutive
// 16-byte slots are conflict free for both the 8-lane store groups and the 16-lane load groups of the b128 instructions.
template <typename T, int N, int EPT, bool WG, int NYQ = 0, bool SWZ = true>
__device__ __forceinline__ void pack_herm(cx<T> (&x)[EPT], const RawPair<T, EPT>& r, cx<T>* lds, int j) {
    constexpr int G = N / EPT;
    // G % EPT^2 == 0: the swizzle term of lds_addr is the same for every t, so the mirrored slots N - j - t G and
    // the read slots j + t G are ONE address each plus compile-time offsets
    constexpr bool AFFINE = !SWZ || (G % (EPT * EPT) == 0);
    cx<T>* mir = lds + lds_addr<EPT, 1, SWZ>(N - j, 0);   // j = 0: slot N is never touched (k >= 1 below)
#pragma unroll
    for (int t = 0; t < EPT / 2; ++t) {
        const int k = j + t * G;
        cx<T> pa = r.a[t], pb = r.b[t];
        if (k == 0) { pa.y = 0; pb.y = 0; }
        x[t] = mk<T>(pa.x - pb.y, pa.y + pb.x);
        if (k >= 1) {
            const cx<T> m = mk<T>(pa.x + pb.y, pb.x - pa.y);
            if constexpr (AFFINE) mir[-t * G] = m;
            else lds[lds_addr<EPT, 1, SWZ>(N - k, 0)] = m;
        }
    }
    if (j == 0)   // lane 0 holds element 0 in a[0] / b[0]
        lds[lds_addr<EPT, 1, SWZ>(N / 2, 0)] = NYQ ? mk<T>(r.a[0].y, r.b[0].y) : mk<T>(r.an.x, r.bn.x);
    group_sync<WG>();
    const cx<T>* src = lds + lds_addr<EPT, 1, SWZ>(j, 0);
#pragma unroll
    for (int t = EPT / 2; t < EPT; ++t) {
        if constexpr (AFFINE) x[t] = src[t * G];
        else x[t] = lds[lds_addr<EPT, 1, SWZ>(j + t * G, 0)];
    }
    group_sync<WG>();
}

// ---- row pass of the VECTOR-JACOBIAN PRODUCT of the explicit terms -----------------------------------------------
// F(w) = M . R( -(dxw u + dyw v) ) [+ f],  u, v, dxw, dyw = I(a_f w).  For a cotangent g:  nbar = R^T(M g) is a real field
// (plane 4 holds the column-transformed M g / c), and the cotangents of the four inverse transforms are nbar times the
// PARTNER of each factor:   Y0 = nbar dxw (-> u^),  Y1 = nbar dyw (-> v^),  Y2 = nbar u (-> dxw^),  Y3 = nbar v (-> dyw^).
// One group per row pair, two rows per complex transform as in v5: five c2r, four products, four r2c; the Y rows replace
// the plane rows they were computed from (a pair's rows belong to its group alone).  Plain (non-split) plane layout.
template <typename T, int N, int EPT, int THR>
__global__ __launch_bounds__((RowGeom<T, N, EPT, THR>::THREADS)) void k_rows_vjp(cx<T>* __restrict__ planes, size_t plane_stride,
                                                                                 const cx<T>* __restrict__ gplane,
                                                                                 const cx<T>* __restrict__ tw, long npairs, int ld) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using Gm = RowGeom<T, N, EPT, THR>;
    constexpr int G = Gm::G;
    constexpr bool WG = (G > 64);
    const int grp = threadIdx.x / G;
    const int j = threadIdx.x % G;
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem_raw) + (size_t)grp * Gm::LDS_PER_GROUP;
    const long stride = (long)gridDim.x * Gm::GROUPS;
    long pair = (long)blockIdx.x * Gm::GROUPS + grp;
    const long iters = (npairs + stride - 1) / stride;   // every group runs the same number of iterations (workgroup barriers)
    for (long it = 0; it < iters; ++it, pair += stride) {
        const bool valid = pair < npairs;
        const long cur = valid ? pair : npairs - 1;
        const size_t off = (size_t)cur * 2 * (size_t)ld;
        RawPair<T, EPT> H;
        auto field = [&](cx<T>(&out)[EPT], const cx<T>* rowA) {
            load_raw<T, N, EPT>(H, rowA, rowA + ld, j);
            pack_herm<T, N, EPT, WG>(out, H, lds, j);
            tile_fft<T, N, EPT, +1, 1, true, WG>(out, lds, tw, j, 0);
        };
        auto emit = [&](const cx<T>(&nb)[EPT], const cx<T>(&fld)[EPT], cx<T>* rowA) {
            cx<T> y[EPT];
#pragma unroll
            for (int t = 0; t < EPT; ++t) y[t] = mk<T>(nb[t].x * fld[t].x, nb[t].y * fld[t].y);
            tile_fft<T, N, EPT, -1, 1, true, WG>(y, lds, tw, j, 0);
            unpack_store_pair<T, N, EPT>(y, lds, rowA, rowA + ld, j, valid);
        };
        cx<T> nb[EPT], fa[EPT], fb[EPT];
        cx<T>* P0 = planes + off;
        cx<T>* P1 = planes + plane_stride + off;
        cx<T>* P2 = planes + 2 * plane_stride + off;
        cx<T>* P3 = planes + 3 * plane_stride + off;
        field(nb, gplane + off);
        field(fa, P2);          // dx w
        field(fb, P0);          // u
        emit(nb, fa, P0);       // Y0 = nbar dxw
        emit(nb, fb, P2);       // Y2 = nbar u
        field(fa, P3);          // dy w
        field(fb, P1);          // v
        emit(nb, fa, P1);       // Y1 = nbar dyw
        emit(nb, fb, P3);       // Y3 = nbar v
    }
}

// ---- row pass, one plane per transform ("v5") --------------------------------------------------
// Round 1 rode two PLANES of one row through a complex transform, so the transformed velocity rows of BOTH rows of a pair
// stayed live while the gradient planes were transformed (256 VGPR + 84 AGPR, one wave per SIMD at 1024^2 fp64; kernels v3 / v4,
// removed in round 5 with the LDS-DMA staged v6 -- docs/history/DESIGN_rounds_1-4.md section 4).  Here the two ROWS of a pair ride
// through one transform of ONE plane:  Z = P(row a) + i P(row b)  ->  real part = field on row a, imaginary part = field on row b, and
//     p.x = -(vx_a dxw_a + vy_a dyw_a),  p.y = -(vx_b dxw_b + vy_b dyw_b)
// accumulates plane by plane (order u^, dx w^, v^, dy w^): one kept transform + the working one + the product +
// the next plane's half rows in flight -- half the live state, same loads / stores / transform count.
// SP = 0: rows (2p, 2p+1).  SP = 1 (split column plans): rows (r, r + N/2).  The planes then hold, for every field, the
// N/2-point column transforms of the even spectral rows (E, workspace rows [0, N/2)) and of the odd ones (O, rows [N/2, N)); the
// column transform is completed here,  a = E + w O,  b = E - w O,  w = exp(+2 pi i r / N), and the advection spectra A_r, A_{r+N/2}
// leave folded as S = A_r + A_{r+N/2},  D = (A_r - A_{r+N/2}) exp(-2 pi i r / N), which the two parity workgroups of the column
// pass transform with N/2 points each.
template <typename T, int N, int EPT, int THR, int SP, int MINW, int PF = 1, int NYQ = 0>
__global__ __launch_bounds__((RowGeom<T, N, EPT, THR>::THREADS), MINW) void k_rows_advect5(
    const cx<T>* __restrict__ planes, size_t plane_stride, cx<T>* __restrict__ adv, const cx<T>* __restrict__ tw,
    long npairs, int ld, int kc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using Gm = RowGeom<T, N, EPT, THR>;
    constexpr int G = Gm::G;
    constexpr bool WG = (G > 64);
    constexpr int N2 = N / 2;
    const int grp = threadIdx.x / G;
    const int j = threadIdx.x % G;
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem_raw) + (size_t)grp * Gm::LDS_PER_GROUP;
    const long stride = (long)gridDim.x * Gm::GROUPS;
    long pair = (long)blockIdx.x * Gm::GROUPS + grp;
    const long iters = (npairs + stride - 1) / stride;
    // first row of pair p; the second row sits `second` elements further
    auto row_a = [&](long p) -> size_t {
        return SP ? (size_t)((p / N2) * N + (p % N2)) * (size_t)ld : (size_t)p * 2 * (size_t)ld;
    };
    const size_t second = SP ? (size_t)N2 * ld : (size_t)ld;

    // PF = 1: half rows (a, b) -- or (E, O) -- of the plane whose turn is NEXT, in flight during the current transform;
    // PF = 0: loaded right before use (fewer live registers: occupancy hides the latency instead)
    RawPair<T, EPT> H;
    if constexpr (PF) {
        const size_t off = row_a(pair < npairs ? pair : npairs - 1);
        load_raw<T, N, EPT, NYQ>(H, planes + off, planes + off + second, j);
    }
    for (long it = 0; it < iters; ++it, pair += stride) {
        const bool valid = pair < npairs;
        const long cur = valid ? pair : npairs - 1;
        const long nxt = (pair + stride < npairs) ? pair + stride : npairs - 1;
        const size_t off = row_a(cur), offn = row_a(nxt);
        cx<T> wf = mk<T>((T)1, (T)0), wi = wf;
        if constexpr (SP) {
            wf = tw[(int)(cur % N2)];   // exp(-2 pi i r / N)
            wi = cconj(wf);
        }
        cx<T> za[EPT], x[EPT], p[EPT];
        // H -> Hermitian-packed sequence -> transform; the next plane's loads are issued in between
        auto field = [&](cx<T>(&out)[EPT], const cx<T>* thisA, const cx<T>* nextA) {
            if constexpr (!PF) load_raw<T, N, EPT, NYQ>(H, thisA, thisA + second, j);
            if constexpr (SP) {
#pragma unroll
                for (int t = 0; t < EPT / 2; ++t) {
                    const cx<T> o = cmul(H.b[t], wi);
                    H.b[t] = H.a[t] - o;
                    H.a[t] = H.a[t] + o;
                }
                if constexpr (!NYQ) {
                    const cx<T> o = cmul(H.bn, wi);
                    H.bn = H.an - o;
                    H.an = H.an + o;
                }
            }
            pack_herm<T, N, EPT, WG, NYQ>(out, H, lds, j);
            if constexpr (PF) load_raw<T, N, EPT, NYQ>(H, nextA, nextA + second, j);
            tile_fft<T, N, EPT, +1, 1, true, WG>(out, lds, tw, j, 0);
        };
        const cx<T>* P0 = planes + off;
        const cx<T>* P1 = planes + plane_stride + off;
        const cx<T>* P2 = planes + 2 * plane_stride + off;
        const cx<T>* P3 = planes + 3 * plane_stride + off;
        field(za, P0, P2);   // vx   (next: dx w)
        field(x, P2, P1);    // dx w (next: v^)
#pragma unroll
        for (int t = 0; t < EPT; ++t) p[t] = mk<T>(za[t].x * x[t].x, za[t].y * x[t].y);
        field(za, P1, P3);               // vy   (next: dy w)
        field(x, P3, planes + offn);     // dy w (next: u^ of the next pair)
#pragma unroll
        for (int t = 0; t < EPT; ++t) p[t] = mk<T>(-(p[t].x + za[t].x * x[t].x), -(p[t].y + za[t].y * x[t].y));

        const int jf = j;
        tile_fft<T, N, EPT, -1, 1, true, WG>(p, lds, tw, jf, 0);
        if constexpr (SP) {
            // unpack the two real-row spectra and fold them for the parity workgroups of the column pass
#pragma unroll
            for (int t = 0; t < EPT; ++t) lds[lds_addr<EPT, 1, true>(jf + t * G, 0)] = p[t];
            group_sync<WG>();
            const T half = (T)0.5;
            cx<T>* outS = adv + off;
            cx<T>* outD = adv + off + second;
#pragma unroll
            for (int t = 0; t < EPT / 2; ++t) {
                const int k = jf + t * G;
                const cx<T> A = p[t];
                const cx<T> Bm = lds[lds_addr<EPT, 1, true>((N - k) % N, 0)];
                const cx<T> X0 = mk<T>((A.x + Bm.x) * half, (A.y - Bm.y) * half);   // spectrum of row r
                const cx<T> X1 = mk<T>((A.y + Bm.y) * half, (Bm.x - A.x) * half);   // spectrum of row r + N/2
                if (valid && k < kc) {
                    outS[k] = X0 + X1;
                    outD[k] = cmul(X0 - X1, wf);
                }
            }
            if (jf == 0 && valid && N / 2 < kc) {
                const cx<T> A = p[EPT / 2];
                const cx<T> X0 = mk<T>(A.x, (T)0), X1 = mk<T>(A.y, (T)0);
                outS[N / 2] = X0 + X1;
                outD[N / 2] = cmul(X0 - X1, wf);
            }
            group_sync<WG>();
        } else {
            unpack_store_pair<T, N, EPT>(p, lds, adv + off, adv + off + second, jf, valid, kc);
        }
    }
}

// ---- row pass, cross-lane transforms ("v7", 1024 points on two-wave groups) ------------------------------------
// v5 with the three-exchange Stockham transform replaced by xl_fft1024 (tcfd_fft.hpp): one LDS exchange per
// transform, the other two are register <-> lane bit transpositions inside a wave (v_permlane32/16_swap, DPP).
// The four inverse transforms leave the physical rows in the fixed permutation `pi`; the point-wise product does
// not care, and the forward transform of the product starts from `pi` and ends in natural order.
// LDS stores per row pair: 18 -> 8 exchange-equivalents of 16 KB; workgroup barriers per pair: ~40 -> ~20.
template <typename T, int THR, int SP, int MINW, int PF, int NYQ = 0>
__global__ __launch_bounds__(128, MINW) void k_rows_advect7(
    const cx<T>* __restrict__ planes, size_t plane_stride, cx<T>* __restrict__ adv, const cx<T>* __restrict__ tw,
    long npairs, int ld, int kc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int N = 1024, EPT = 8, G = 128, N2 = N / 2;
    constexpr bool WG = true;
    constexpr bool SWZ7 = false;     // mirror accesses without the Stockham swizzle: no bank conflicts (see pack_herm)
    const int j = threadIdx.x;
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem_raw);
    const long stride = (long)gridDim.x;
    long pair = (long)blockIdx.x;
    const long iters = (npairs + stride - 1) / stride;
    auto row_a = [&](long p) -> size_t {
        return SP ? (size_t)((p / N2) * N + (p % N2)) * (size_t)ld : (size_t)p * 2 * (size_t)ld;
    };
    const size_t second = SP ? (size_t)N2 * ld : (size_t)ld;
    const XlTw<T> xtw = xl_load_tw<T>(tw, j);
    NoHook nohook;

    RawPair<T, EPT> H;
    if constexpr (PF) {
        const size_t off = row_a(pair < npairs ? pair : npairs - 1);
        load_raw<T, N, EPT, NYQ>(H, planes + off, planes + off + second, j);
    }
    for (long it = 0; it < iters; ++it, pair += stride) {
        const bool valid = pair < npairs;
        const long cur = valid ? pair : npairs - 1;
        const long nxt = (pair + stride < npairs) ? pair + stride : npairs - 1;
        const size_t off = row_a(cur), offn = row_a(nxt);
        cx<T> wf = mk<T>((T)1, (T)0), wi = wf;
        if constexpr (SP) {
            wf = tw[(int)(cur % N2)];   // exp(-2 pi i r / N)
            wi = cconj(wf);
        }
        cx<T> za[EPT], x[EPT], p[EPT];
        auto field = [&](cx<T>(&out)[EPT], const cx<T>* thisA, const cx<T>* nextA) {
            if constexpr (!PF) load_raw<T, N, EPT, NYQ>(H, thisA, thisA + second, j);
            if constexpr (SP) {
#pragma unroll
                for (int t = 0; t < EPT / 2; ++t) {
                    const cx<T> o = cmul(H.b[t], wi);
                    H.b[t] = H.a[t] - o;
                    H.a[t] = H.a[t] + o;
                }
                if constexpr (!NYQ) {
                    const cx<T> o = cmul(H.bn, wi);
                    H.bn = H.an - o;
                    H.an = H.an + o;
                }
            }
            pack_herm<T, N, EPT, WG, NYQ, SWZ7>(out, H, lds, j);
            if constexpr (PF) load_raw<T, N, EPT, NYQ>(H, nextA, nextA + second, j);
            xl_fft1024<T, +1, 1>(out, lds, xtw, j, nohook);
        };
        const cx<T>* P0 = planes + off;
        const cx<T>* P1 = planes + plane_stride + off;
        const cx<T>* P2 = planes + 2 * plane_stride + off;
        const cx<T>* P3 = planes + 3 * plane_stride + off;
        field(za, P0, P2);   // vx   (next: dx w)
        field(x, P2, P1);    // dx w (next: v^)
#pragma unroll
        for (int t = 0; t < EPT; ++t) p[t] = mk<T>(za[t].x * x[t].x, za[t].y * x[t].y);
        field(za, P1, P3);               // vy   (next: dy w)
        field(x, P3, planes + offn);     // dy w (next: u^ of the next pair)
#pragma unroll
        for (int t = 0; t < EPT; ++t) p[t] = mk<T>(-(p[t].x + za[t].x * x[t].x), -(p[t].y + za[t].y * x[t].y));

        xl_fft1024<T, -1, 1>(p, lds, xtw, j, nohook);   // pi -> natural order
        // unpack the two real-row spectra (mirror through the exchange buffer)
        const cx<T>* mir = lds + lds_addr<EPT, 1, SWZ7>(N - j, 0);
#pragma unroll
        for (int t = 0; t < EPT; ++t) lds[lds_addr<EPT, 1, SWZ7>(j + t * G, 0)] = p[t];
        group_sync<WG>();
        const T half = (T)0.5;
        cx<T>* out0 = adv + off;
        cx<T>* out1 = adv + off + second;
#pragma unroll
        for (int t = 0; t < EPT / 2; ++t) {
            const int k = j + t * G;
            const cx<T> A = p[t];
            const cx<T> Bm = (j == 0 && t == 0) ? p[0] : mir[-t * G];   // element N - k; k = 0 mirrors itself
            const cx<T> X0 = mk<T>((A.x + Bm.x) * half, (A.y - Bm.y) * half);   // spectrum of the first row
            const cx<T> X1 = mk<T>((A.y + Bm.y) * half, (Bm.x - A.x) * half);   // spectrum of the second row
            if (valid && k < kc) {
                if constexpr (SP) {   // folded for the parity workgroups of the column pass
                    out0[k] = X0 + X1;
                    out1[k] = cmul(X0 - X1, wf);
                } else {
                    out0[k] = X0;
                    out1[k] = X1;
                }
            }
        }
        if (j == 0 && valid && N2 < kc) {
            const cx<T> A = p[EPT / 2];
            const cx<T> X0 = mk<T>(A.x, (T)0), X1 = mk<T>(A.y, (T)0);
            if constexpr (SP) {
                out0[N2] = X0 + X1;
                out1[N2] = cmul(X0 - X1, wf);
            } else {
                out0[N2] = X0;
                out1[N2] = X1;
            }
        }
        group_sync<WG>();
    }
}

// real (rows, N) -> half spectrum (rows, m): first half of rfft2
template <typename T, int N, int EPT>
__global__ __launch_bounds__((RowGeom<T, N, EPT>::THREADS)) void k_rows_r2c(const T* __restrict__ in,
                                                                             cx<T>* __restrict__ out,
                                                                             const cx<T>* __restrict__ tw, long npairs,
                                                                             int m /* pitch of out */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using Gm = RowGeom<T, N, EPT>;
    constexpr int G = Gm::G;
    constexpr bool WG = (G > 64);
    const int grp = threadIdx.x / G;
    const int j = threadIdx.x % G;
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem_raw) + (size_t)grp * Gm::LDS_PER_GROUP;
    long pair = (long)blockIdx.x * Gm::GROUPS + grp;
    const bool valid = pair < npairs;
    if (!valid) pair = npairs - 1;
    const size_t row0 = (size_t)pair * 2;
    cx<T> p[EPT];
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        const int e = j + t * G;
        p[t] = mk<T>(in[row0 * N + e], in[(row0 + 1) * N + e]);
    }
    tile_fft<T, N, EPT, -1, 1, true, WG>(p, lds, tw, j, 0);
    unpack_store_pair<T, N, EPT>(p, lds, out + row0 * (size_t)m, out + (row0 + 1) * (size_t)m, j, valid);
}

// half spectrum (rows, m) -> real (rows, N): second half of irfft2 (scale applied by the column pass)
template <typename T, int N, int EPT>
__global__ __launch_bounds__((RowGeom<T, N, EPT>::THREADS)) void k_rows_c2r(const cx<T>* __restrict__ in,
                                                                             T* __restrict__ out,
                                                                             const cx<T>* __restrict__ tw, long npairs,
                                                                             int m /* pitch of in */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using Gm = RowGeom<T, N, EPT>;
    constexpr int G = Gm::G;
    constexpr bool WG = (G > 64);
    const int grp = threadIdx.x / G;
    const int j = threadIdx.x % G;
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem_raw) + (size_t)grp * Gm::LDS_PER_GROUP;
    long pair = (long)blockIdx.x * Gm::GROUPS + grp;
    const bool valid = pair < npairs;
    if (!valid) pair = npairs - 1;
    const size_t row0 = (size_t)pair * 2;
    cx<T> z[EPT];
    load_herm_pair<T, N, EPT>(z, in + row0 * (size_t)m, in + (row0 + 1) * (size_t)m, j);
    tile_fft<T, N, EPT, +1, 1, true, WG>(z, lds, tw, j, 0);
    if (valid) {
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            const int e = j + t * G;
            out[row0 * N + e] = z[t].x;
            out[(row0 + 1) * N + e] = z[t].y;
        }
    }
}

// Second half of irfft2 + the data-generation drivers' bilinear subsample in ONE pass (fno/data_gen/data_gen_McWilliams2d.py:
// 158-163: irfft2 -> F.interpolate(size = (N / S, N / S), mode = "bilinear")).  For an even integer factor S the sample point
// of output pixel (r, c) is (S r + (S - 1) / 2, S c + (S - 1) / 2): exactly between rows S r + S / 2 - 1 and S r + S / 2 and
// between the two columns of the same numbers, every weight 1 / 2.  A group transforms those two rows as ONE complex sequence
// (any two rows pack, they need not be neighbours in the pair index) -- so the rows the subsample never looks at are not
// transformed at all --, the horizontal neighbour is one lane away (element j + t G of lane j: S <= G keeps both in a wave),
// the vertical one is the other component of the same register.  Arithmetic in ATen's order, 0.5 (0.5 a + 0.5 b) + 0.5 (0.5 c +
// 0.5 d) with a, b on the upper row: at S = 2 (same row pairs as k_rows_c2r) the result is bit-identical to irfft2 followed by
// F.interpolate, beyond it agrees to rounding (which two rows share a transform shows in the last bit); the full-size field
// -- written and read again by the two-step path, 2.25 x the bytes of this one at S = 2 -- never exists.
template <typename T, int N, int EPT>
__global__ __launch_bounds__((RowGeom<T, N, EPT>::THREADS)) void k_rows_c2r_sub(const cx<T>* __restrict__ in,
                                                                                 T* __restrict__ out,
                                                                                 const cx<T>* __restrict__ tw, long nrows,
                                                                                 int m /* pitch of in */, int S) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using Gm = RowGeom<T, N, EPT>;
    constexpr int G = Gm::G;
    constexpr bool WG = (G > 64);
    const int grp = threadIdx.x / G;
    const int j = threadIdx.x % G;
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem_raw) + (size_t)grp * Gm::LDS_PER_GROUP;
    long orow = (long)blockIdx.x * Gm::GROUPS + grp;          // output row over all fields
    const bool valid = orow < nrows;
    if (!valid) orow = nrows - 1;
    const int NS = N / S;
    const long f = orow / NS;
    const int r = (int)(orow - f * NS);
    const size_t row0 = (size_t)f * N + (size_t)r * S + (S / 2 - 1);
    cx<T> z[EPT];
    load_herm_pair<T, N, EPT>(z, in + row0 * (size_t)m, in + (row0 + 1) * (size_t)m, j);
    tile_fft<T, N, EPT, +1, 1, true, WG>(z, lds, tw, j, 0);
    T* orow_p = out + (size_t)orow * NS;
#pragma unroll
    for (int t = 0; t < EPT; ++t) asm volatile("" : "+v"(z[t].x), "+v"(z[t].y));   // the transform ends here: nothing below
                                                                                  // may be contracted into its last stage
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        const int e = j + t * G;
        const T bx = __shfl_down(z[t].x, 1), by = __shfl_down(z[t].y, 1);
        if (valid && (e & (S - 1)) == S / 2 - 1) {
            const T up = (T)0.5 * z[t].x + (T)0.5 * bx, lo = (T)0.5 * z[t].y + (T)0.5 * by;
            orow_p[e / S] = (T)0.5 * up + (T)0.5 * lo;
        }
    }
}

// ------------------------------------------------------------------ small elementwise kernels
template <typename T>
__global__ void k_velocity(const cx<T>* __restrict__ w, cx<T>* __restrict__ uh, cx<T>* __restrict__ vh,
                           cx<T>* __restrict__ psi, const T* __restrict__ kxs, const T* __restrict__ kys, int n,
                           int m, size_t count) {
    constexpr T TWO_PI = (T)6.283185307179586476925286766559;
    constexpr T M4PI2 = (T)(-39.478417604357434475337963999505);
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < count;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int jc = (int)(idx % m);
        const int i = (int)((idx / m) % n);
        const T kx = kxs[i], ky = kys[jc];
        T lap = M4PI2 * (kx * kx + ky * ky);
        if (i == 0 && jc == 0) lap = (T)1;
        const cx<T> p = cscale(w[idx], (T)-1 / lap);
        if (psi) psi[idx] = p;
        if (uh) uh[idx] = mul_i(cscale(p, TWO_PI * ky));
        if (vh) vh[idx] = mul_mi(cscale(p, TWO_PI * kx));
    }
}

// ------------------------------------------------------------------ plan
struct ProfRec { int kind; hipEvent_t e0, e1; };
struct ProfState {
    bool on = false;
    int max_records = 0;
    std::vector<ProfRec> recs;
};

// hipGraph of ONE interior RK step (nstages x {row pass, fused column pass}, state upad -> upad): for small grids a
// step is 10 launches of 7-20 us each and launch gaps are a quarter of the time; a steps=k call replays the graph
// for its k-2 interior steps on a plan-owned stream (capture is not allowed on the legacy default stream PyTorch
// hands over), fenced by events against the caller's stream.
struct GraphState {
    hipStream_t stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipStream_t stream2 = nullptr;                 // batch-halves overlap (TCFD_OVERLAP): second lane
    hipEvent_t ev_out2 = nullptr, ev_rows = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph = nullptr;
    // the same step MULTI times in one graph: a graph launch costs ~10 us of idle device between replays, a sixth
    // of a 256^2 step; long calls replay this one and finish with the single-step graph
    static constexpr int MULTI = 16;
    hipGraphExec_t exec_multi = nullptr;
    hipGraph_t graph_multi = nullptr;
    // key of the captured sequence
    long batch = -1;
    int nstages = 0;
    void* ws = nullptr;
    std::vector<double> coef;
};

// Tuning switches (environment, read ONCE at plan creation and frozen in the plan: two plans of one process may
// differ, a plan never changes its kernels behind the caller's back).  -1 = automatic.
struct Tuning {
    int split;               // TCFD_SPLIT: radix-2 split of the column transform across the row pass
    int small_tiles;         // TCFD_SMALL_TILES: 4-column tiles / 64-lane row groups for small problems
    int two_wg;              // TCFD_TWO_WG: 128-VGPR cap (two workgroups per CU) for the 512-point fp64 column tiles
    int f32_cols8;           // TCFD_F32_COLS8: fp32 512-point column tiles of 8 columns (64-byte segments, pairs on one XCD) on the
                             // cross-lane transforms for the plane-emitting passes (1 = default); 0 = 16-column Stockham tiles
    int nt_out;              // TCFD_NT_OUT: non-temporal stores of the last stage's outputs (new state to the caller, dw/dt)
    int cols_lds_pad;        // TCFD_COLS_LDS_PAD: extra dynamic LDS bytes of the column kernels (experiments)
    int rows7_minw;          // TCFD_ROWS7_MINW: waves per SIMD of the fp32 cross-lane row kernel (2 / 4)
    int rows_blocks_per_cu;  // TCFD_ROWS_BLOCKS_PER_CU: persistent row-pass grid
    int pair_xcd;            // TCFD_PAIR_XCD: put the 2 / 4 column tiles that share a 128-byte line on one XCD (0 off, 2 pairs only)
    int ablate;              // TCFD_ABLATE: timing ablations (results are WRONG when non-zero)
    int nt_planes;           // TCFD_NT_PLANES: non-temporal plane stores (-1: for the small-tile launches only)
    int rows_v;              // TCFD_ROWS_V: 0 = per size; 5 = register-staged rows with Stockham exchanges, 7 = cross-lane
                             // transforms (1024 points fp64 only).  (4 / 6 -- round 1's two-plane kernels, LDS-DMA staging --
                             // were removed in round 5; the values now mean 5)
    int cols_xl;             // TCFD_COLS_XL: cross-lane column transforms where available (1 = default)
    int nyq_pack;            // TCFD_NYQ_PACK: packed Nyquist column in the step (1 = default where the plan allows it)
    int chunk;               // TCFD_CHUNK: fields per chunk of a batched call (0 = whole batch at once, -1 = cache sized)
    int graph;               // TCFD_GRAPH: hipGraph replay of interior steps (1 = on; off by default since round 4: plain stream
                             // launches measured 3-13 % faster than replay for every small problem, tests/micro/graph_crossover.py)
    int overlap;             // TCFD_OVERLAP: two half batches on two streams (opt-in experiment)
    int round_fields;        // n = 3 * 2^k: fields whose column tiles fill the resident workgroup slots exactly once (0: not used)
    size_t cache_bytes;      // last-level (Infinity Cache / MALL) size of the plan's device: what a chunk is sized for
    int cache_source;        // 0 = built-in 256 MB, 1 = KFD topology of this device, 2 = TCFD_CACHE_MB
};

struct tcfd_ns2d_plan {
    Tuning tune;
    std::mutex mu;    // guards the mutable side-cars below (the tables are immutable after creation)
    ProfState* prof;  // mutable side-car (tcfd_ns2d_profile_begin/end); null until first use
    GraphState* gs;   // mutable side-car: captured interior step (null until first use)
    int n, m, dtype;
    int ldw;        // workspace row pitch in elements: m rounded up so that a row is a multiple of 128 bytes
    void* tw;       // cx<T>[n]
    void* tw2;      // cx<T>[n/2]: table of the half-length column transforms (split plans)
    void* kx;       // T[n]
    void* ky;       // T[m]
    void* lin;      // T[n*m]
    void* mask;     // T[n*m]
    void* forcing;  // cx<T>[n*m] or null (dense form, used when the forcing is not sparse)
    // exact compact forms found at plan creation (the (n, m) tables cost ~2x the state's bytes per
    // column tile in L2 traffic; the reference's own tables are separable / sparse by construction)
    void* mask_r; void* mask_c; void* lin_r; void* lin_c;  // T[n], T[m], T[n], T[m]
    void* f_ptr; void* f_row; void* f_val; void* f_col;    // int[m+1], int[nnz], cx<T>[nnz], int[nnz]
    int f_nnz;
    int sep;        // mask and linear term are separable
    int f_sparse;   // forcing stored as CSC
    int nyq_a, nyq_ca;   // ... in the opening pass / in the fused passes of a step (set with nyq at plan creation)
    int nyq;        // packed Nyquist column in the step kernels (emit_planes): the plan is pruned (F = h = 0 in column
                    // m - 1), n >= 64 (every column tile width divides n / 2) and the row kernels are v5 / v7
    int keep_cols;  // > 0: F (hence the RK accumulator h) is identically zero for columns >= keep_cols and for
                    // rows with mask_r == 0 (2/3-rule mask, forcing inside the mask): those entries of the
                    // advection / h arrays are neither written nor read.  0: no pruning.
};

static bool supported_n(int n) {
    return (n >= 8 && n <= 2048 && (n & (n - 1)) == 0) || n == 96 || n == 192 || n == 384 || n == 768 ||   // 2^k, 3 * 2^k
           n == 1536 || n == 80 || n == 160 || n == 320 || n == 640 || n == 1280;                         // and 5 * 2^k
}

template <typename T>
static int upload(void** dst, const std::vector<T>& host) {
    HIP_TRY(hipMalloc(dst, host.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*dst, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

template <typename T>
static int plan_fill(tcfd_ns2d_plan* p, const double* kx, const double* ky, const double* lin, const double* mask,
                     const double* forcing) {
    const int n = p->n, m = p->m;
    std::vector<T> tw(2 * (size_t)n), a(n), b(m), l((size_t)n * m), k((size_t)n * m);
    for (int t = 0; t < n; ++t) {
        // long-double evaluation, rounded once to the working type
        const long double ang = -2.0L * 3.141592653589793238462643383279502884L * (long double)t / (long double)n;
        tw[2 * t] = (T)cosl(ang);
        tw[2 * t + 1] = (T)sinl(ang);
    }
    // exact values at the quadrant points
    tw[0] = 1; tw[1] = 0;
    tw[2 * (n / 4)] = 0; tw[2 * (n / 4) + 1] = -1;
    tw[2 * (n / 2)] = -1; tw[2 * (n / 2) + 1] = 0;
    tw[2 * (3 * n / 4)] = 0; tw[2 * (3 * n / 4) + 1] = 1;
    for (int i = 0; i < n; ++i) a[i] = (T)kx[i];
    for (int i = 0; i < m; ++i) b[i] = (T)ky[i];
    for (size_t i = 0; i < (size_t)n * m; ++i) { l[i] = (T)lin[i]; k[i] = (T)mask[i]; }
    int rc;
    if ((rc = upload(&p->tw, tw))) return rc;
    {
        std::vector<T> t2((size_t)n);  // W_{n/2}^t = W_n^{2t}
        for (int t = 0; t < n / 2; ++t) { t2[2 * t] = tw[2 * (2 * t)]; t2[2 * t + 1] = tw[2 * (2 * t) + 1]; }
        if ((rc = upload(&p->tw2, t2))) return rc;
    }
    if ((rc = upload(&p->kx, a))) return rc;
    if ((rc = upload(&p->ky, b))) return rc;
    if ((rc = upload(&p->lin, l))) return rc;
    if ((rc = upload(&p->mask, k))) return rc;
    const bool force_tables = env_int("TCFD_FORCE_TABLES", 0) != 0;
    // --- separable forms: mask = r[i]*c[j] (exact for 0/1 brick walls), lin = a[i] + b[j]
    {
        std::vector<T> mr(n, 0), mc(m, 0), lr(n), lc(m);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < m; ++j) {
                mr[i] = std::max(mr[i], k[(size_t)i * m + j]);
                mc[j] = std::max(mc[j], k[(size_t)i * m + j]);
            }
        bool ok = true;
        T lmax = 0;
        for (int j = 0; j < m; ++j) lc[j] = l[j];
        for (int i = 0; i < n; ++i) lr[i] = l[(size_t)i * m] - l[0];
        for (size_t i = 0; i < (size_t)n * m; ++i) lmax = std::max(lmax, (T)std::fabs(l[i]));
        const T tol = (T)8 * std::numeric_limits<T>::epsilon() * lmax;
        for (int i = 0; i < n && ok; ++i)
            for (int j = 0; j < m; ++j) {
                if (k[(size_t)i * m + j] != mr[i] * mc[j]) { ok = false; break; }
                if (std::fabs(l[(size_t)i * m + j] - (lr[i] + lc[j])) > tol) { ok = false; break; }
            }
        p->sep = (ok && !force_tables) ? 1 : 0;
        // prunable: mask_c is a 1...1 0...0 prefix pattern and every forcing entry sits where the mask is 1
        int kc = 0;
        while (kc < m && mc[kc] == (T)1) ++kc;
        bool prefix = p->sep && kc < m;
        for (int j = kc; j < m && prefix; ++j) prefix = (mc[j] == (T)0);
        for (int i = 0; i < n && prefix; ++i) prefix = (mr[i] == (T)0 || mr[i] == (T)1);
        if (prefix && forcing)
            for (size_t e = 0; e < (size_t)n * m && prefix; ++e)
                if ((forcing[2 * e] != 0 || forcing[2 * e + 1] != 0) && k[e] == (T)0) prefix = false;
        p->keep_cols = (prefix && !env_int("TCFD_NO_PRUNE", 0)) ? kc : 0;
        if (p->sep) {
            if ((rc = upload(&p->mask_r, mr))) return rc;
            if ((rc = upload(&p->mask_c, mc))) return rc;
            if ((rc = upload(&p->lin_r, lr))) return rc;
            if ((rc = upload(&p->lin_c, lc))) return rc;
        }
    }
    if (forcing) {
        std::vector<T> f(2 * (size_t)n * m);
        for (size_t i = 0; i < f.size(); ++i) f[i] = (T)forcing[i];
        size_t nnz = 0;
        for (size_t i = 0; i < (size_t)n * m; ++i) nnz += (f[2 * i] != 0 || f[2 * i + 1] != 0);
        if (nnz * 64 <= (size_t)n * m && !force_tables) {  // sparse: column-compressed list
            std::vector<int> ptr(m + 1, 0), row, col;
            std::vector<T> val;
            for (int j = 0; j < m; ++j) {
                for (int i = 0; i < n; ++i) {
                    const size_t e = (size_t)i * m + j;
                    if (f[2 * e] != 0 || f[2 * e + 1] != 0) {
                        row.push_back(i);
                        col.push_back(j);
                        val.push_back(f[2 * e]);
                        val.push_back(f[2 * e + 1]);
                    }
                }
                ptr[j + 1] = (int)row.size();
            }
            p->f_nnz = (int)row.size();
            if (row.empty()) { row.push_back(0); col.push_back(0); val.push_back(0); val.push_back(0); }
            p->f_sparse = 1;
            if ((rc = upload(&p->f_col, col))) return rc;
            if ((rc = upload(&p->f_ptr, ptr))) return rc;
            if ((rc = upload(&p->f_row, row))) return rc;
            if ((rc = upload(&p->f_val, val))) return rc;
        } else if ((rc = upload(&p->forcing, f))) {
            return rc;
        }
    }
    return 0;
}

TCFD_API void tcfd_ns2d_plan_destroy(tcfd_ns2d_plan* p) {
    if (!p) return;
    void* ptrs[] = {p->tw, p->tw2, p->kx, p->ky, p->lin, p->mask, p->forcing, p->mask_r, p->mask_c,
                    p->lin_r, p->lin_c, p->f_ptr, p->f_row, p->f_val, p->f_col};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    if (p->prof) {
        for (auto& r : p->prof->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        delete p->prof;
    }
    if (p->gs) {
        if (p->gs->exec) (void)hipGraphExecDestroy(p->gs->exec);
        if (p->gs->graph) (void)hipGraphDestroy(p->gs->graph);
        if (p->gs->exec_multi) (void)hipGraphExecDestroy(p->gs->exec_multi);
        if (p->gs->graph_multi) (void)hipGraphDestroy(p->gs->graph_multi);
        if (p->gs->ev_in) (void)hipEventDestroy(p->gs->ev_in);
        if (p->gs->ev_out) (void)hipEventDestroy(p->gs->ev_out);
        if (p->gs->ev_out2) (void)hipEventDestroy(p->gs->ev_out2);
        if (p->gs->ev_rows) (void)hipEventDestroy(p->gs->ev_rows);
        if (p->gs->stream2) (void)hipStreamDestroy(p->gs->stream2);
        if (p->gs->stream) (void)hipStreamDestroy(p->gs->stream);
        delete p->gs;
    }
    delete p;
}

// Size of the memory-side last-level cache (Infinity Cache / MALL) of the CURRENT device.  hipDeviceProp_t has no field
// for it (l2CacheSize is the per-XCD L2); the KFD topology does: the level-3 entry under
// /sys/class/kfd/kfd/topology/nodes/<node>/caches/, the node matched to the device by PCI location.  TCFD_CACHE_MB
// overrides; 256 MB (MI355X) when the topology cannot be read.
static size_t last_level_cache_bytes(int* source) {
    *source = 0;
    const int forced = env_int("TCFD_CACHE_MB", 0);
    if (forced > 0) { *source = 2; return (size_t)forced << 20; }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (size_t)256 << 20;
    const long want_loc = ((long)prop.pciBusID << 8) | ((long)prop.pciDeviceID << 3);
    auto read_props = [](const char* path, const char* const* keys, long* vals, int nk) {
        FILE* f = fopen(path, "r");
        if (!f) return false;
        char key[128];
        long long v;
        while (fscanf(f, "%127s %lld", key, &v) == 2)
            for (int i = 0; i < nk; ++i)
                if (!strcmp(key, keys[i])) vals[i] = (long)v;
        fclose(f);
        return true;
    };
    // Pass 1: the node whose PCI location is this device's.  Pass 2 (containers hide the GPU nodes' `properties` files but not
    // their cache entries): every node that lists a level-3 cache -- the GPUs of one box are of one kind, so the smallest
    // level-3 size found is this device's (and the safe choice if they were not).
    long any_l3 = 0;
    for (int node = 0; node < 128; ++node) {
        char path[256];
        snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/properties", node);
        const char* nkeys[] = {"simd_count", "location_id", "domain"};
        long nv[3] = {-1, -1, 0};
        const bool props = read_props(path, nkeys, nv, 3);
        snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/caches", node);
        if (!props && access(path, F_OK) != 0) { if (node > 16) break; else continue; }
        if (props && nv[0] == 0) continue;       // a CPU node
        long best = 0;
        for (int c = 0; c < 1024; ++c) {
            snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/caches/%d/properties", node, c);
            const char* ckeys[] = {"level", "size"};
            long cv[2] = {0, 0};
            if (!read_props(path, ckeys, cv, 2)) break;
            if (cv[0] >= 3 && cv[1] > best) best = cv[1];   // size is in KB
        }
        if (best <= 0) continue;
        if (props && nv[1] == want_loc && nv[2] == (long)prop.pciDomainID) { *source = 1; return (size_t)best << 10; }
        if (any_l3 == 0 || best < any_l3) any_l3 = best;
    }
    if (any_l3 > 0) { *source = 1; return (size_t)any_l3 << 10; }
    return (size_t)256 << 20;
}

static int fill_round_fields(tcfd_ns2d_plan* p);   // (needs the size dispatch, defined below)

TCFD_API int tcfd_ns2d_plan_create(tcfd_ns2d_plan** out, int n, int dtype, const double* kx, const double* ky,
                                     const double* linear_term, const double* mask, const double* forcing_hat) {
    if (!out || !kx || !ky || !linear_term || !mask) return fail(TCFD_EINVAL, "plan_create: null argument");
    if (!supported_n(n)) return fail(TCFD_EINVAL, "plan_create: n=%d is not a power of two in [8, 2048] (or 3 * 2^k in 96..1536, 5 * 2^k in 80..1280)", n);
    if (dtype != TCFD_C64 && dtype != TCFD_C128) return fail(TCFD_EINVAL, "plan_create: bad dtype %d", dtype);
    tcfd_ns2d_plan* p = new tcfd_ns2d_plan();   // value-initialised: every pointer / flag starts at zero
    p->n = n;
    p->m = n / 2 + 1;
    p->dtype = dtype;
    p->tune.split = env_int("TCFD_SPLIT", -1);
    p->tune.small_tiles = env_int("TCFD_SMALL_TILES", -1);
    p->tune.two_wg = env_int("TCFD_TWO_WG", 1);
    p->tune.f32_cols8 = env_int("TCFD_F32_COLS8", 1);
    p->tune.rows7_minw = env_int("TCFD_ROWS7_MINW", 4);
    p->tune.cols_lds_pad = env_int("TCFD_COLS_LDS_PAD", 0);
    p->tune.nt_out = env_int("TCFD_NT_OUT", 0);
    p->tune.rows_blocks_per_cu = env_int("TCFD_ROWS_BLOCKS_PER_CU", 0);
    p->tune.pair_xcd = env_int("TCFD_PAIR_XCD", 1);
    p->tune.ablate = env_int("TCFD_ABLATE", 0);
    p->tune.nt_planes = env_int("TCFD_NT_PLANES", -1);
    p->tune.rows_v = env_int("TCFD_ROWS_V", 0);
    p->tune.chunk = env_int("TCFD_CHUNK", -1);
    p->tune.cols_xl = env_int("TCFD_COLS_XL", 1);
    p->tune.nyq_pack = env_int("TCFD_NYQ_PACK", 1);
    p->tune.graph = env_int("TCFD_GRAPH", -1);
    p->tune.overlap = env_int("TCFD_OVERLAP", 0);
    p->tune.cache_bytes = last_level_cache_bytes(&p->tune.cache_source);
    {
        const int per_line = dtype == TCFD_C128 ? 8 : 16;  // complex elements per 128-byte line
        p->ldw = (p->m + per_line - 1) / per_line * per_line;
    }
    int rc = dtype == TCFD_C128 ? plan_fill<double>(p, kx, ky, linear_term, mask, forcing_hat)
                                : plan_fill<float>(p, kx, ky, linear_term, mask, forcing_hat);
    if (rc) {
        tcfd_ns2d_plan_destroy(p);
        return rc;
    }
    (void)fill_round_fields(p);
    p->nyq = (p->tune.nyq_pack && n >= 64 && (n & (n - 1)) == 0 && p->keep_cols > 0 && p->keep_cols <= p->m - 1 &&
              true) ? 1 : 0;
    // per pass (TCFD_NYQ_PACK = 3: the opening pass of a call only, for A/B runs; the row pass reads either form)
    p->nyq_a = p->nyq;
    p->nyq_ca = p->nyq && p->tune.nyq_pack != 3;
    *out = p;
    return 0;
}

TCFD_API int tcfd_ns2d_plan_info(const tcfd_ns2d_plan* p, int* separable, int* sparse_forcing, int* keep_cols) {
    if (!p) return fail(TCFD_EINVAL, "plan_info: null plan");
    if (separable) *separable = p->sep;
    if (sparse_forcing) *sparse_forcing = p->f_sparse;
    if (keep_cols) *keep_cols = p->keep_cols;
    return 0;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t field_bytes(const tcfd_ns2d_plan* p, long batch) {
    const size_t esz = p->dtype == TCFD_C128 ? 16 : 8;
    return align256((size_t)batch * p->n * p->ldw * esz);
}

static long chunk_fields(const tcfd_ns2d_plan* p, long batch);
// Scratch of the batched calls.  They run chunk by chunk through ONE chunk-sized set of 8 fields (h, adv, 4 planes,
// line-aligned state, second state), so the need does not grow with the batch beyond one chunk -- except for irfft2,
// which stages ONE field of the whole batch.  (1024^2 x 64 fp64: 0.55 GB instead of the 4.4 GB of an unchunked step.)
TCFD_API size_t tcfd_ns2d_workspace_bytes(const tcfd_ns2d_plan* p, long batch) {
    if (!p || batch <= 0) return 0;
    return std::max((size_t)8 * field_bytes(p, chunk_fields(p, batch)), field_bytes(p, batch));
}

// ------------------------------------------------------------------ optional per-launch event timing
// kinds: 0 cols MODE_A, 1 rows_advect, 2 cols MODE_CA, 3 cols MODE_C (+ dw/dt), 5 other   (4 was the separate dw/dt pass)
struct ProfScope {
    ProfState* ps;
    hipStream_t st;
    int idx = -1;
    ProfScope(const tcfd_ns2d_plan* p, int kind, hipStream_t s) : ps(p->prof), st(s) {
        if (!ps || !ps->on || (int)ps->recs.size() >= ps->max_records) { ps = nullptr; return; }
        ProfRec r;
        r.kind = kind;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) { ps = nullptr; return; }
        (void)hipEventRecord(r.e0, st);
        ps->recs.push_back(r);
        idx = (int)ps->recs.size() - 1;
    }
    ~ProfScope() {
        if (ps) (void)hipEventRecord(ps->recs[idx].e1, st);
    }
};

// ------------------------------------------------------------------ launch helpers
// Dynamic LDS above 64 KB needs a function attribute, and the attribute belongs to the function instance of ONE device:
// it is set once per (kernel, device), tracked by a bit mask at the launch site (one process may drive several GPUs
// from several host threads).
struct DevOnce { std::atomic<unsigned long long> mask{0}; };
template <typename K>
static int set_lds(DevOnce& once, K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (once.mask.load(std::memory_order_acquire) & bit) return 0;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)bytes));
    once.mask.fetch_or(bit, std::memory_order_release);
    return 0;
}

// tiles of C columns narrower than a 128-byte line: how many of them share a line (0: whole lines, nothing to group).
// TCFD_PAIR_XCD = 0 switches the XCD grouping off, 2 limits it to pairs (the round-1 form).
template <typename T, int C>
static int line_group(const tcfd_ns2d_plan* p) {
    constexpr int bytes = C * (int)sizeof(cx<T>);
    if (bytes >= 128 || p->tune.pair_xcd == 0) return 0;
    const int lg = 128 / bytes > 4 ? 4 : 128 / bytes;
    return p->tune.pair_xcd == 2 ? 2 : lg;
}

template <typename T, int N, int MODE, int EPT, int C, int MINW = 1, int SP = 0>
static int launch_cols_v(const tcfd_ns2d_plan* p, ColArgs<T> a, long batch, hipStream_t st) {
    constexpr int NT = N >> SP;
    constexpr int G = NT / EPT;
    // + TCFD_COLS_LDS_PAD bytes: an experiment knob that caps the workgroups a CU takes (occupancy A/B without a rebuild)
    const size_t lds = (size_t)lds_elems<NT, EPT, C, false>() * sizeof(cx<T>) + 3 * (size_t)NT * sizeof(T) + (size_t)p->tune.cols_lds_pad;
    a.m = p->m;
    a.ldw = p->ldw;
    a.ntiles = (p->m + C - 1) / C;
    if (a.nyq) a.ntiles = (p->m - 1) / C;   // packed Nyquist column: no lone tile (the caller checked the plan allows it)
    a.batch = (int)batch;
    a.kx = (const T*)p->kx;
    a.ky = (const T*)p->ky;
    a.lin = (const T*)p->lin;
    a.mask = (const T*)p->mask;
    a.forcing = (const cx<T>*)p->forcing;
    a.mask_r = (const T*)p->mask_r;
    a.mask_c = (const T*)p->mask_c;
    a.lin_r = (const T*)p->lin_r;
    a.lin_c = (const T*)p->lin_c;
    a.f_ptr = p->f_sparse ? (const int*)p->f_ptr : nullptr;
    a.f_col = (const int*)p->f_col;
    a.f_nnz = (p->f_sparse && p->f_nnz <= 32) ? p->f_nnz : 0;
    a.f_row = (const int*)p->f_row;
    a.f_val = (const cx<T>*)p->f_val;
    a.sep = p->sep;
    a.keep_cols = p->keep_cols;
    a.tw = (const cx<T>*)(SP ? p->tw2 : p->tw);
    a.pair_xcd = line_group<T, C>(p);
    a.ablate = p->tune.ablate;
    a.nt_out = (MODE == MODE_C && p->tune.nt_out && a.u_out_ld == p->m) ? 1 : 0;   // (the caller's array: pitch m, not ldw)
    long blocks = batch * a.ntiles;
    if (a.pair_xcd) blocks = ((((long)(a.ntiles + a.pair_xcd - 1) / a.pair_xcd) * batch + 7) / 8) * 8 * a.pair_xcd;
    if (!a.h_out) a.h_out = a.h;
    const dim3 grid((unsigned)(blocks * (SP ? 2 : 1))), block(C * G);
    const int kind = MODE == MODE_A ? 0 : MODE == MODE_CA ? 2 : MODE == MODE_C ? 3 : 5;
    auto launch = [&](auto kern, DevOnce& once) -> int {
        if (int rc_ = set_lds(once, kern, lds)) return rc_;
        ProfScope prof(p, kind, st);
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);
        HIP_TRY(hipGetLastError());
        return 0;
    };
    constexpr bool NYQ_MODE = (MODE == MODE_A || MODE == MODE_CA || MODE == MODE_C);
    // 512-point tiles of 8 columns (1024^2 split plans and 512^2, fp64): cross-lane transforms unless TCFD_COLS_XL=0
    constexpr bool XL_OK = (NT == 512 && EPT == 8 && C == 8 && MODE != MODE_FWD && MODE != MODE_INV);
    if constexpr (XL_OK) {
        if (p->tune.cols_xl) {
            a.nt_planes = p->tune.nt_planes > 0;
            static DevOnce once_x, once_xn;
            if constexpr (NYQ_MODE) {
                if (a.nyq) return launch(k_cols<T, N, EPT, C, MODE, MINW, SP, 1, 1>, once_xn);
            }
            return launch(k_cols<T, N, EPT, C, MODE, MINW, SP, 1, 0>, once_x);
        }
    }
    // 4-column tiles = the small, cache-resident problems: a whole pass's output would sit dirty in the L2s until the
    // end-of-kernel write-back; streamed out while the kernel still computes, 256^2 x 16 fp32 steps 4.6 % faster
    // (at 1024^2 the same hint costs 1-3 %)
    a.nt_planes = p->tune.nt_planes >= 0 ? p->tune.nt_planes : (C == 4 && EPT == 4);
    static DevOnce once, once_n;
    if constexpr (NYQ_MODE) {
        if (a.nyq) return launch(k_cols<T, N, EPT, C, MODE, MINW, SP, 0, 1>, once_n);
    }
    return launch(k_cols<T, N, EPT, C, MODE, MINW, SP, 0, 0>, once);
}

#ifndef TCFD_SPLIT_A_MINW4
#define TCFD_SPLIT_A_MINW4 1
#endif
// Split plans: the column transform is cut radix-2 across the row pass so that a column tile is half as big
// and two workgroups share a CU.  Default: n = 1024 (fp64: the full tile would fill the CU's LDS; fp32: the half
// tiles are 16 columns = full 128-byte lines instead of 8).
template <typename T, int N>
static bool use_split(const tcfd_ns2d_plan* p) {
    const int force = p->tune.split;
    if (N < 16 || !is_pow2c(N)) return false;   // n = 3 * 2^k / 5 * 2^k: whole-column tiles only (split measured slower at 768: 160 vs 181 steps/s)
    if (force >= 0) return force != 0;
    return N == 1024;   // (768: split 160 vs plain 181 steps/s at one round of workgroups per launch)   // fp64: 7.99 vs 8.6 ms/step; fp32 (16-column, 128-byte tiles of 512 rows): 5.61 vs 6.05 ms/step
}

template <typename T, int N, int MODE>
static int launch_cols_split(const tcfd_ns2d_plan* p, ColArgs<T> a, long batch, hipStream_t st) {
    constexpr int H = N >= 16 ? N / 2 : 8;
    constexpr int EPT = Cfg<T, H>::COL_EPT, C = Cfg<T, H>::COLS;
    // 128 VGPRs per lane = two 512-lane workgroups per CU for the half-size fp64 tiles
    constexpr int MINW = (C * (H / EPT) == 512 && (MODE != MODE_A || TCFD_SPLIT_A_MINW4)) ? 4 : 1;
    if constexpr (N == 1024 && sizeof(T) == 4 && (MODE == MODE_A || MODE == MODE_CA)) {   // (see launch_cols: the same at 1024^2 x 64: CA 1.86 -> 1.74 ms)
        if (p->tune.f32_cols8) return launch_cols_v<T, N, MODE, 8, 8, 4, 1>(p, a, batch, st);
    }
    if constexpr (N >= 16 && MODE != MODE_FWD && MODE != MODE_INV)
        return launch_cols_v<T, N, MODE, EPT, C, MINW, 1>(p, a, batch, st);
    else
        return fail(TCFD_EINVAL, "split launch of an unsupported mode");
}

template <typename T, int N, int MODE>
static int launch_cols(const tcfd_ns2d_plan* p, ColArgs<T> a, long batch, hipStream_t st) {
    if constexpr (!is_pow2c(N)) {   // n = 3 * 2^k / 5 * 2^k: whole-column Stockham tiles (no split / cross-lane / packed-Nyquist variants)
        return launch_cols_v<T, N, MODE, Cfg<T, N>::COL_EPT, Cfg<T, N>::COLS>(p, a, batch, st);
    } else
    if constexpr (MODE != MODE_FWD && MODE != MODE_INV) {
        if (use_split<T, N>(p)) return launch_cols_split<T, N, MODE>(p, a, batch, st);
    }
    // small problems (few column tiles for 256 CUs, working set cache resident): narrow 4-column tiles and
    // 4 elements per lane give ~4x more, shorter workgroups (measured at 256^2 x 16 fp32: CA 29 -> 21 us)
    if constexpr ((N == 128 || N == 256) && (MODE == MODE_A || MODE == MODE_CA || MODE == MODE_C)) {
        const int force = p->tune.small_tiles;
        const long tiles = batch * ((p->m + Cfg<T, N>::COLS - 1) / Cfg<T, N>::COLS);
        if (force == 1 || (force != 0 && tiles < 2 * 256)) return launch_cols_v<T, N, MODE, 4, 4>(p, a, batch, st);
    }
    // 512-point fp64 tiles are 64 KB: capping the update kernels at 128 VGPRs lets TWO workgroups share a CU,
    // so one tile's memory phases overlap the other's transforms (measured at 512^2 x 256: CA 1.05 -> 0.87 ms)
    if constexpr (N == 512 && sizeof(T) == 8 && (MODE == MODE_CA || MODE == MODE_C)) {
        if (p->tune.two_wg) return launch_cols_v<T, N, MODE, 8, 8, 4>(p, a, batch, st);
    }
    // fp32 512-point tiles: 8 columns on the cross-lane transforms (one LDS exchange instead of two, 512-thread workgroups: two
    // per CU) for the passes that emit planes; the update-only pass keeps the 16-column Stockham tiles (measured per step at
    // 512^2 x 64: CA 0.453 -> 0.411 ms, C 0.086 -> 0.094, profiles/r06_f32_variants.txt).  TCFD_F32_COLS8=0: 16 columns everywhere
    if constexpr (N == 512 && sizeof(T) == 4 && (MODE == MODE_A || MODE == MODE_CA)) {
        if (p->tune.f32_cols8) return launch_cols_v<T, N, MODE, 8, 8, 4>(p, a, batch, st);
    }
    return launch_cols_v<T, N, MODE, Cfg<T, N>::COL_EPT, Cfg<T, N>::COLS>(p, a, batch, st);
}

// persistent row-pass grid: `per_cu` workgroups per CU, grid-stride over the row pairs
static long rows_grid(const tcfd_ns2d_plan* p, long want, int dflt_per_cu) {
    const int per_cu = p->tune.rows_blocks_per_cu > 0 ? p->tune.rows_blocks_per_cu : dflt_per_cu;
    return std::min<long>(want, 256L * per_cu);
}

template <typename T, int N, int EPT, int THR, int SP, int MINW, int PF = 1, int NYQ = 0>
static int launch_rows_advect5(const tcfd_ns2d_plan* p, const cx<T>* planes, size_t plane_stride, cx<T>* adv,
                               long batch, hipStream_t st, int nyq = 0) {
    if constexpr (!NYQ) {
        if (nyq) return launch_rows_advect5<T, N, EPT, THR, SP, MINW, PF, 1>(p, planes, plane_stride, adv, batch, st, 0);
    }
    using Gm = RowGeom<T, N, EPT, THR>;
    auto kern = k_rows_advect5<T, N, EPT, THR, SP, MINW, PF, NYQ>;
    static DevOnce lds_once;
    if (int rc_ = set_lds(lds_once, kern, Gm::LDS_BYTES)) return rc_;
    const long npairs = batch * (N / 2);
    // resident workgroups per CU: MINW waves per SIMD x 4 SIMDs / waves per workgroup (LDS allows it for every size)
    constexpr int WAVES = (Gm::THREADS + 63) / 64;
    constexpr int PER_CU = (4 * MINW / WAVES) > 0 ? (4 * MINW / WAVES) : 1;
    const long blocks = rows_grid(p, (npairs + Gm::GROUPS - 1) / Gm::GROUPS, PER_CU);
    ProfScope prof(p, 1, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Gm::THREADS), Gm::LDS_BYTES, st, planes, plane_stride, adv,
                       (const cx<T>*)p->tw, npairs, p->ldw, p->keep_cols > 0 ? p->keep_cols : N);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T, int SP, int MINW, int PF, int NYQ = 0>
static int launch_rows_advect7(const tcfd_ns2d_plan* p, const cx<T>* planes, size_t plane_stride, cx<T>* adv,
                               long batch, hipStream_t st, int nyq = 0) {
    if constexpr (!NYQ) {
        if (nyq) return launch_rows_advect7<T, SP, MINW, PF, 1>(p, planes, plane_stride, adv, batch, st, 0);
    }
    auto kern = k_rows_advect7<T, 128, SP, MINW, PF, NYQ>;
    constexpr size_t lds = (size_t)(1024 + 1) * sizeof(cx<T>);   // + 1: the unpack reads slot N for lane 0 (value unused)
    const long npairs = batch * 512;
    const long blocks = rows_grid(p, npairs, 2 * MINW);   // two-wave workgroups: 2 per SIMD pair and wave slot
    ProfScope prof(p, 1, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(128), lds, st, planes, plane_stride, adv,
                       (const cx<T>*)p->tw, npairs, p->ldw, p->keep_cols > 0 ? p->keep_cols : 1024);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Row-pass kernel of a size.  5 = register-staged rows (default everywhere but 1024^2 fp64).
template <typename T, int N>
static constexpr int rows_default_version() {
    // 7 = cross-lane transforms (1024 points fp64): equal to 5 while the row pass streams from HBM (0.616 vs 0.607 ms
    // per launch on the whole batch) and ahead of it once the planes come from the Infinity Cache (chunked calls:
    // 7.54 vs 7.82 ms per step)
    // fp32 at 1024 points (round 6): 7 as well -- 1.36 against 2.08 ms per step of 64 fields at four waves per SIMD
    return N == 1024 ? 7 : 5;
}

template <typename T, int N>
static int launch_rows_advect(const tcfd_ns2d_plan* p, const cx<T>* planes, size_t plane_stride, cx<T>* adv, long batch,
                              hipStream_t st, int nyq = 0) {
    constexpr int EPT = Cfg<T, N>::ROW_EPT, THR = Cfg<T, N>::ROW_THREADS;
    if constexpr (!is_pow2c(N)) {
        // n = 3 * 2^k / 5 * 2^k: v5 (one plane per transform, next plane's rows in flight, one wave per SIMD); at 768^2 fp64 per
        // 7-field chunk launch: 43 us, against 53 for v3 (two planes per transform) and 124 for v5 capped at two waves per
        // SIMD (twelve complex fp64 per array spill there).  Only this one variant is instantiated for these sizes.
        return launch_rows_advect5<T, N, EPT, THR, 0, 1, 1>(p, planes, plane_stride, adv, batch, st, 0);
    } else {
    const bool split = use_split<T, N>(p);
    if constexpr (N == 128 || N == 256) {
        const int force = p->tune.small_tiles;
        if (!split && (force == 1 || (force != 0 && batch * (N / 2) < 16 * 256)))
            return launch_rows_advect5<T, N, 8, 64, 0, 1>(p, planes, plane_stride, adv, batch, st, nyq);
    }
    if constexpr (N == 1024 && sizeof(T) == 4) {
        if (p->tune.rows_v == 7 || p->tune.rows_v == 0) {   // cross-lane transforms in fp32 (half the registers of fp64: four waves
                                                            // per SIMD fit): per step at 1024^2 x 64 2.08 -> 1.36 ms against the Stockham rows
            if (p->tune.rows7_minw == 2) {
                if (split) return launch_rows_advect7<T, 1, 2, 1>(p, planes, plane_stride, adv, batch, st, nyq);
                return launch_rows_advect7<T, 0, 2, 1>(p, planes, plane_stride, adv, batch, st, nyq);
            }
            if (split) return launch_rows_advect7<T, 1, 4, 1>(p, planes, plane_stride, adv, batch, st, nyq);
            return launch_rows_advect7<T, 0, 4, 1>(p, planes, plane_stride, adv, batch, st, nyq);
        }
    }
    if constexpr (N == 1024 && sizeof(T) == 8) {
        if (p->tune.rows_v == 7 || p->tune.rows_v == 0) {   // cross-lane transforms: the default here
            if (split) return launch_rows_advect7<T, 1, 2, 1>(p, planes, plane_stride, adv, batch, st, nyq);
            return launch_rows_advect7<T, 0, 2, 1>(p, planes, plane_stride, adv, batch, st, nyq);
        }
        // v5 at this size: two waves per SIMD, rows loaded right before use (236 VGPRs, no spills)
        if (split) return launch_rows_advect5<T, N, EPT, THR, 1, 2, 0>(p, planes, plane_stride, adv, batch, st, nyq);
    }
    if constexpr (N >= 16) {
        if (split) return launch_rows_advect5<T, N, EPT, THR, 1, 1>(p, planes, plane_stride, adv, batch, st, nyq);
    }
    // 512-point fp32 rows: 164 registers with the packed arithmetic (tcfd_fft.hpp) -- three workgroups per CU (three waves per
    // SIMD) instead of one: the pass is bound by the latency of its dependent chains, not by issue (per step at 512^2 x 64:
    // 0.465 -> 0.352 ms, profiles/r06_rows_sweep.txt)
    if constexpr (N == 512 && sizeof(T) == 4)
        return launch_rows_advect5<T, N, EPT, THR, 0, 3>(p, planes, plane_stride, adv, batch, st, nyq);
    // ... and fp64 (240 registers): two per CU, 0.644 -> 0.587 ms per step (three: 0.80), profiles/r06_f64_rows_percu.txt
    if constexpr (N == 512 && sizeof(T) == 8)
        return launch_rows_advect5<T, N, EPT, THR, 0, 2>(p, planes, plane_stride, adv, batch, st, nyq);
    return launch_rows_advect5<T, N, EPT, THR, 0, 1>(p, planes, plane_stride, adv, batch, st, nyq);
    }
}

template <typename T>
struct Ws {
    cx<T>* h;
    cx<T>* adv;
    cx<T>* planes;
    cx<T>* upad;          // state in the line-aligned internal pitch (stages 1.. of a call)
    cx<T>* upad2;         // intermediate state of schedules whose later stages restart from the step's initial state
    size_t plane_stride;  // elements
};
template <typename T>
static Ws<T> carve(const tcfd_ns2d_plan* p, void* ws, long batch) {
    const size_t fb = field_bytes(p, batch);
    unsigned char* base = (unsigned char*)ws;
    Ws<T> w;
    w.h = (cx<T>*)base;
    w.adv = (cx<T>*)(base + fb);
    w.planes = (cx<T>*)(base + 2 * fb);
    w.upad = (cx<T>*)(base + 6 * fb);
    w.upad2 = (cx<T>*)(base + 7 * fb);
    w.plane_stride = fb / sizeof(cx<T>);
    return w;
}

// ------------------------------------------------------------------ two batch halves, software pipelined
// The fused column pass saturates the HBM copy rate while the row pass is LDS / latency bound and leaves ~40 % of the
// read bandwidth idle.  The batch elements are independent, so the step is run on two half batches on two streams,
// half B one row pass behind half A: B's row pass then shares the chip with A's column pass and vice versa.
// (Experiment, off by default -- see step_impl.)
template <typename T, int N>
static int step_overlap_impl(const tcfd_ns2d_plan* p, const void* w_in, void* w_out, void* dwdt, long batch, int nstages,
                             const double* beta, const double* gdt, const double* mu, const double* fa,
                             const double* mud, const int* base0, int steps, double inv_total_dt, void* ws,
                             hipStream_t st) {
    tcfd_ns2d_plan* mp = const_cast<tcfd_ns2d_plan*>(p);
    std::lock_guard<std::mutex> lock(mp->mu);   // the plan-owned streams / events serve one call at a time
    if (!mp->gs) mp->gs = new GraphState();
    GraphState* g = mp->gs;
    if (!g->stream) {
        HIP_TRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&g->ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g->ev_out, hipEventDisableTiming));
    }
    if (!g->stream2) {
        HIP_TRY(hipStreamCreateWithFlags(&g->stream2, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&g->ev_out2, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g->ev_rows, hipEventDisableTiming));
    }
    Ws<T> W = carve<T>(p, ws, batch);
    struct Half {
        ColArgs<T> a;
        const cx<T>* u_src; int u_src_ld;
        const cx<T>* w_in; cx<T>* w_out; cx<T>* dwdt;
        cx<T>* planes; cx<T>* adv; cx<T>* upad; cx<T>* upad2;
        long batch; hipStream_t q;
    } H[2];
    const long bA = batch / 2;
    for (int i = 0; i < 2; ++i) {
        const long b0 = i ? bA : 0;
        const size_t oc = (size_t)b0 * N * p->m, ow = (size_t)b0 * N * p->ldw;
        Half& h = H[i];
        h.batch = i ? batch - bA : bA;
        h.q = i ? g->stream2 : g->stream;
        h.w_in = (const cx<T>*)w_in + oc;
        h.w_out = (cx<T>*)w_out + oc;
        h.dwdt = dwdt ? (cx<T>*)dwdt + oc : nullptr;
        h.planes = W.planes + ow; h.adv = W.adv + ow; h.upad = W.upad + ow; h.upad2 = W.upad2 + ow;
        h.a = ColArgs<T>{};
        h.a.nyq = p->nyq_a;
        h.a.planes = h.planes; h.a.plane_stride = W.plane_stride; h.a.h = W.h + ow; h.a.in = h.adv;
        h.a.u_in = h.w_in; h.a.u_in_ld = p->m;
        h.u_src = h.w_in; h.u_src_ld = p->m;
    }
    HIP_TRY(hipEventRecord(g->ev_in, st));
    int rc;
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipStreamWaitEvent(H[i].q, g->ev_in, 0));
        if ((rc = launch_cols<T, N, MODE_A>(p, H[i].a, H[i].batch, H[i].q))) return rc;
        H[i].a.nyq = p->nyq_ca;
    }
    int planes_packed = p->nyq_a;
    auto cols = [&](Half& h, int k, bool last, const cx<T>* u0, int u0_ld) -> int {
        bool u0_needed_later = false;
        for (int k2 = k + 1; k2 < nstages; ++k2) u0_needed_later |= (base0 && base0[k2]);
        const bool from_u0 = base0 && base0[k];
        h.a.u_in = from_u0 ? u0 : h.u_src;
        h.a.u_in_ld = from_u0 ? u0_ld : h.u_src_ld;
        cx<T>* dst = h.upad;
        if (u0_needed_later && (const cx<T>*)dst == u0) dst = h.upad2;
        h.a.u_out = last ? h.w_out : dst;
        h.a.u_out_ld = last ? p->m : p->ldw;
        h.a.beta = (T)beta[k]; h.a.gdt = (T)gdt[k]; h.a.mu = (T)mu[k];
        h.a.fa = fa ? (T)fa[k] : (T)1;
        h.a.mud = mud ? (T)mud[k] : (T)mu[k];
        h.a.load_h = (k != 0);
        h.a.store_h = (k != nstages - 1);
        h.a.dwdt = last ? h.dwdt : nullptr;
        h.a.w0 = h.w_in;
        h.a.dwdt_scale = (T)inv_total_dt;
        int r = last ? launch_cols<T, N, MODE_C>(p, h.a, h.batch, h.q) : launch_cols<T, N, MODE_CA>(p, h.a, h.batch, h.q);
        h.u_src = h.a.u_out;
        h.u_src_ld = h.a.u_out_ld;
        return r;
    };
    for (int s = 0; s < steps; ++s) {
        const cx<T>* u0[2] = {H[0].u_src, H[1].u_src};
        const int u0_ld[2] = {H[0].u_src_ld, H[1].u_src_ld};
        for (int k = 0; k < nstages; ++k) {
            const bool last = (s == steps - 1) && (k == nstages - 1);
            if ((rc = launch_rows_advect<T, N>(p, H[0].planes, W.plane_stride, H[0].adv, H[0].batch, H[0].q, planes_packed))) return rc;
            HIP_TRY(hipEventRecord(g->ev_rows, H[0].q));
            HIP_TRY(hipStreamWaitEvent(H[1].q, g->ev_rows, 0));   // B stays one row pass behind A
            if ((rc = launch_rows_advect<T, N>(p, H[1].planes, W.plane_stride, H[1].adv, H[1].batch, H[1].q, planes_packed))) return rc;
            planes_packed = p->nyq_ca;
            if ((rc = cols(H[0], k, last, u0[0], u0_ld[0]))) return rc;
            if ((rc = cols(H[1], k, last, u0[1], u0_ld[1]))) return rc;
        }
    }
    HIP_TRY(hipEventRecord(g->ev_out, H[0].q));
    HIP_TRY(hipEventRecord(g->ev_out2, H[1].q));
    HIP_TRY(hipStreamWaitEvent(st, g->ev_out, 0));
    HIP_TRY(hipStreamWaitEvent(st, g->ev_out2, 0));
    return 0;
}

// ------------------------------------------------------------------ step driver
template <typename T, int N>
static int step_impl(const tcfd_ns2d_plan* p, const void* w_in, void* w_out, void* dwdt, long batch, int nstages,
                     const double* beta, const double* gdt, const double* mu, const double* fa, const double* mud,
                     const int* base0, int steps, double inv_total_dt, void* ws, hipStream_t st) {
    {
        // opt-in (TCFD_OVERLAP=1): measured +2.6 % at 1024^2 x 64 fp64 -- both kernels fill the VGPR file, so the CUs
        // time-slice the two halves instead of co-running them; not worth two streams by default
        if (batch >= 2 && p->tune.overlap == 1)
            return step_overlap_impl<T, N>(p, w_in, w_out, dwdt, batch, nstages, beta, gdt, mu, fa, mud, base0, steps,
                                           inv_total_dt, ws, st);
    }
    Ws<T> W = carve<T>(p, ws, batch);
    int rc;
    ColArgs<T> a{};
    a.planes = W.planes;
    a.plane_stride = W.plane_stride;
    a.h = W.h;
    a.in = W.adv;
    a.u_in = (const cx<T>*)w_in;
    a.u_in_ld = p->m;
    a.nyq = p->nyq_a;
    if ((rc = launch_cols<T, N, MODE_A>(p, a, batch, st))) return rc;
    int planes_packed = p->nyq_a;   // how the planes in the workspace were written: the row pass reads them that way
    cx<T>* h_cur = W.h;
    a.nyq = p->nyq_ca;
    // the caller's (n, m) rows are not 128-byte aligned (m is odd): only the first read and the last
    // write of a call touch that layout, every stage in between uses the aligned copy `upad`
    const cx<T>* u_src = (const cx<T>*)w_in;
    int u_src_ld = p->m;
    // one RK step on stream `q`; `first`/`final` select the caller-layout source / destination
    // General IMEX stage:  h <- fa_k F(u) + beta_k h ;  u <- (base + gdt_k h + mu_k L base) / (1 - mud_k L)  with
    // base = the current state, or (base0[k]) the state the step started from (IMEX RK2-CN, equations.py:190-228).
    // A stage whose successors still need the step's initial state writes to the second state buffer.
    auto run_step = [&](hipStream_t q, bool final) -> int {
        const cx<T>* u0 = u_src;
        const int u0_ld = u_src_ld;
        for (int k = 0; k < nstages; ++k) {
            int r;
            if ((r = launch_rows_advect<T, N>(p, W.planes, W.plane_stride, W.adv, batch, q, planes_packed))) return r;
            planes_packed = p->nyq_ca;
            const bool last = final && (k == nstages - 1);
            bool u0_needed_later = false;
            for (int k2 = k + 1; k2 < nstages; ++k2) u0_needed_later |= (base0 && base0[k2]);
            const bool from_u0 = base0 && base0[k];
            a.u_in = from_u0 ? u0 : u_src;
            a.u_in_ld = from_u0 ? u0_ld : u_src_ld;
            cx<T>* dst = W.upad;
            if (u0_needed_later && (const cx<T>*)dst == u0) dst = W.upad2;
            a.u_out = last ? (cx<T>*)w_out : dst;
            a.u_out_ld = last ? p->m : p->ldw;
            a.beta = (T)beta[k];
            a.gdt = (T)gdt[k];
            a.mu = (T)mu[k];
            a.fa = fa ? (T)fa[k] : (T)1;
            a.mud = mud ? (T)mud[k] : (T)mu[k];
            a.load_h = (k != 0);  // h starts from 0 every step (equations.py:353)
            a.store_h = (k != nstages - 1);
            a.h = h_cur;
            a.h_out = h_cur;
            a.dwdt = last ? (cx<T>*)dwdt : nullptr;
            a.w0 = (const cx<T>*)w_in;
            a.dwdt_scale = (T)inv_total_dt;
            r = last ? launch_cols<T, N, MODE_C>(p, a, batch, q) : launch_cols<T, N, MODE_CA>(p, a, batch, q);
            if (r) return r;
            u_src = a.u_out;
            u_src_ld = a.u_out_ld;
            h_cur = a.h_out;
        }
        return 0;
    };
    const long state_bytes = (long)batch * N * p->ldw * (long)sizeof(cx<T>);
    const int graph_env = p->tune.graph;
    const bool profiling = p->prof && p->prof->on;
    const bool use_graph = steps >= 3 && !profiling &&
                           graph_env == 1;   // opt-in since round 4 (plain launches are faster at every size measured)
    (void)state_bytes;
    int s0 = 0;
    if (use_graph) {
        tcfd_ns2d_plan* mp = const_cast<tcfd_ns2d_plan*>(p);
        // the captured graph, its stream and its fence events are plan-owned and mutable: host threads that share a
        // plan take turns here (the replays of one call are enqueued before the next call may re-capture)
        std::lock_guard<std::mutex> lock(mp->mu);
        if (!mp->gs) mp->gs = new GraphState();
        GraphState* g = mp->gs;
        if (!g->stream) {
            HIP_TRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&g->ev_in, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&g->ev_out, hipEventDisableTiming));
        }
        if ((rc = run_step(st, false))) return rc;   // step 1: caller layout -> upad (also sets every kernel attribute)
        std::vector<double> coef;
        for (int k = 0; k < nstages; ++k) {
            coef.push_back(beta[k]); coef.push_back(gdt[k]); coef.push_back(mu[k]);
            coef.push_back(fa ? fa[k] : 1.0); coef.push_back(mud ? mud[k] : mu[k]); coef.push_back(base0 ? base0[k] : 0);
        }
        auto capture = [&](int reps, hipGraph_t* graph, hipGraphExec_t* exec) -> int {
            HIP_TRY(hipStreamBeginCapture(g->stream, hipStreamCaptureModeThreadLocal));
            int r = 0;
            for (int i = 0; i < reps && !r; ++i) r = run_step(g->stream, false);         // upad -> upad
            hipError_t e = hipStreamEndCapture(g->stream, graph);
            if (r) return r;
            if (e != hipSuccess) return fail(TCFD_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
            HIP_TRY(hipGraphInstantiate(exec, *graph, nullptr, nullptr, 0));
            return 0;
        };
        if (!g->exec || g->batch != batch || g->nstages != nstages || g->ws != ws || g->coef != coef) {
            if (g->exec) { (void)hipGraphExecDestroy(g->exec); g->exec = nullptr; }
            if (g->graph) { (void)hipGraphDestroy(g->graph); g->graph = nullptr; }
            if (g->exec_multi) { (void)hipGraphExecDestroy(g->exec_multi); g->exec_multi = nullptr; }
            if (g->graph_multi) { (void)hipGraphDestroy(g->graph_multi); g->graph_multi = nullptr; }
            if ((rc = capture(1, &g->graph, &g->exec))) return rc;
            g->batch = batch; g->nstages = nstages; g->ws = ws; g->coef = coef;
        }
        int interior = steps - 2;
        if (interior >= 2 * GraphState::MULTI && !g->exec_multi &&
            (rc = capture(GraphState::MULTI, &g->graph_multi, &g->exec_multi)))
            return rc;
        HIP_TRY(hipEventRecord(g->ev_in, st));
        HIP_TRY(hipStreamWaitEvent(g->stream, g->ev_in, 0));
        if (g->exec_multi)
            for (; interior >= GraphState::MULTI; interior -= GraphState::MULTI)
                HIP_TRY(hipGraphLaunch(g->exec_multi, g->stream));
        for (; interior > 0; --interior) HIP_TRY(hipGraphLaunch(g->exec, g->stream));
        HIP_TRY(hipEventRecord(g->ev_out, g->stream));
        HIP_TRY(hipStreamWaitEvent(st, g->ev_out, 0));
        s0 = steps - 1;
    }
    for (int s = s0; s < steps; ++s)
        if ((rc = run_step(st, s == steps - 1))) return rc;
    return 0;
}

template <typename T, int N>
static int explicit_impl(const tcfd_ns2d_plan* p, const void* w, void* out, const void* wt, void* psi, bool residual,
                         long batch, void* ws, hipStream_t st) {
    Ws<T> W = carve<T>(p, ws, batch);
    int rc;
    ColArgs<T> a{};
    a.planes = W.planes;
    a.plane_stride = W.plane_stride;
    a.u_in = (const cx<T>*)w;
    a.u_in_ld = p->m;
    if ((rc = launch_cols<T, N, MODE_A>(p, a, batch, st))) return rc;
    if ((rc = launch_rows_advect<T, N>(p, W.planes, W.plane_stride, W.adv, batch, st))) return rc;
    a.in = W.adv;
    a.out = (cx<T>*)out;
    if (residual) {
        a.wt = (const cx<T>*)wt;
        a.psi = (cx<T>*)psi;
        return launch_cols<T, N, MODE_RES>(p, a, batch, st);
    }
    return launch_cols<T, N, MODE_F>(p, a, batch, st);
}

template <typename T, int N>
static int rfft2_impl(const tcfd_ns2d_plan* p, const void* x, void* out, long batch, hipStream_t st) {
    using Gm = RowGeom<T, N, Cfg<T, N>::ROW_EPT>;
    auto kern = k_rows_r2c<T, N, Cfg<T, N>::ROW_EPT>;
    static DevOnce lds_once;
    if (int rc_ = set_lds(lds_once, kern, Gm::LDS_BYTES)) return rc_;
    const long npairs = batch * (N / 2);
    const long blocks = (npairs + Gm::GROUPS - 1) / Gm::GROUPS;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Gm::THREADS), Gm::LDS_BYTES, st, (const T*)x, (cx<T>*)out,
                       (const cx<T>*)p->tw, npairs, p->m);
    HIP_TRY(hipGetLastError());
    ColArgs<T> a{};
    a.in = (const cx<T>*)out;
    a.out = (cx<T>*)out;
    a.in_ld = a.out_ld = p->m;
    a.scale = (T)1;
    return launch_cols<T, N, MODE_FWD>(p, a, batch, st);
}

template <typename T, int N>
static int irfft2_impl(const tcfd_ns2d_plan* p, const void* xh, void* out, long batch, void* ws, hipStream_t st) {
    ColArgs<T> a{};
    a.in = (const cx<T>*)xh;
    a.out = (cx<T>*)ws;
    a.in_ld = p->m;
    a.out_ld = p->ldw;
    a.scale = (T)1 / ((T)N * (T)N);
    int rc;
    if ((rc = launch_cols<T, N, MODE_INV>(p, a, batch, st))) return rc;
    using Gm = RowGeom<T, N, Cfg<T, N>::ROW_EPT>;
    auto kern = k_rows_c2r<T, N, Cfg<T, N>::ROW_EPT>;
    static DevOnce lds_once;
    {
        rc = set_lds(lds_once, kern, Gm::LDS_BYTES);
        if (rc) return rc;
    }
    const long npairs = batch * (N / 2);
    const long blocks = (npairs + Gm::GROUPS - 1) / Gm::GROUPS;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Gm::THREADS), Gm::LDS_BYTES, st, (const cx<T>*)ws, (T*)out,
                       (const cx<T>*)p->tw, npairs, p->ldw);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T, int N>
static int irfft2_sub_impl(const tcfd_ns2d_plan* p, const void* xh, void* out, long batch, int S, void* ws, hipStream_t st) {
    using Gm = RowGeom<T, N, Cfg<T, N>::ROW_EPT>;
    if (S < 2 || (S & (S - 1)) || S > Gm::G || S > 64)
        return fail(TCFD_EINVAL, "irfft2_subsample: factor %d is not a power of two in [2, %d] (n = %d)", S, Gm::G < 64 ? Gm::G : 64, N);
    ColArgs<T> a{};
    a.in = (const cx<T>*)xh;
    a.out = (cx<T>*)ws;
    a.in_ld = p->m;
    a.out_ld = p->ldw;
    a.scale = (T)1 / ((T)N * (T)N);
    int rc;
    if ((rc = launch_cols<T, N, MODE_INV>(p, a, batch, st))) return rc;
    auto kern = k_rows_c2r_sub<T, N, Cfg<T, N>::ROW_EPT>;
    static DevOnce lds_once;
    {
        rc = set_lds(lds_once, kern, Gm::LDS_BYTES);
        if (rc) return rc;
    }
    const long nrows = batch * (N / S);
    const long blocks = (nrows + Gm::GROUPS - 1) / Gm::GROUPS;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Gm::THREADS), Gm::LDS_BYTES, st, (const cx<T>*)ws, (T*)out,
                       (const cx<T>*)p->tw, nrows, p->ldw, S);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ (dtype, n) dispatch
#define TCFD_DISPATCH_N(T, n, CALL)                                       \
    switch (n) {                                                          \
        case 8: { constexpr int N_ = 8; return CALL; }                    \
        case 16: { constexpr int N_ = 16; return CALL; }                  \
        case 32: { constexpr int N_ = 32; return CALL; }                  \
        case 64: { constexpr int N_ = 64; return CALL; }                  \
        case 128: { constexpr int N_ = 128; return CALL; }                \
        case 256: { constexpr int N_ = 256; return CALL; }                \
        case 512: { constexpr int N_ = 512; return CALL; }                \
        case 1024: { constexpr int N_ = 1024; return CALL; }              \
        case 2048: { constexpr int N_ = 2048; return CALL; }              \
        case 96: { constexpr int N_ = 96; return CALL; }                  \
        case 192: { constexpr int N_ = 192; return CALL; }                \
        case 384: { constexpr int N_ = 384; return CALL; }                \
        case 768: { constexpr int N_ = 768; return CALL; }                \
        case 80: { constexpr int N_ = 80; return CALL; }                  \
        case 160: { constexpr int N_ = 160; return CALL; }                \
        case 320: { constexpr int N_ = 320; return CALL; }                \
        case 640: { constexpr int N_ = 640; return CALL; }                \
        case 1536: { constexpr int N_ = 1536; return CALL; }              \
        case 1280: { constexpr int N_ = 1280; return CALL; }              \
        default: return fail(TCFD_EINVAL, "unsupported n=%d", n);         \
    }
// TCFD_DISPATCHED(name, (parameters), (arguments), call): defines `static int name(parameters)` that runs `call` with T_ / N_
// bound to the plan's precision and grid size.  In a two-unit build the float32 half lives in unit 1 as `name_f32`.
#if TCFD_UNIT == 1
#define TCFD_DISPATCHED(NAME, PARAMS, ARGS, CALL)                         \
    int NAME##_f32 PARAMS {                                               \
        using T_ = float;                                                 \
        TCFD_DISPATCH_N(T_, (p)->n, CALL)                                 \
    }                                                                     \
    [[maybe_unused]] static int NAME PARAMS { return NAME##_f32 ARGS; }
#elif TCFD_UNIT == 0
#define TCFD_DISPATCHED(NAME, PARAMS, ARGS, CALL)                         \
    int NAME##_f32 PARAMS;                                                \
    static int NAME PARAMS {                                              \
        if ((p)->dtype == TCFD_C128) {                                    \
            using T_ = double;                                            \
            TCFD_DISPATCH_N(T_, (p)->n, CALL)                             \
        }                                                                 \
        return NAME##_f32 ARGS;                                           \
    }
#else
#define TCFD_DISPATCHED(NAME, PARAMS, ARGS, CALL)                         \
    static int NAME PARAMS {                                              \
        if ((p)->dtype == TCFD_C128) {                                    \
            using T_ = double;                                            \
            TCFD_DISPATCH_N(T_, (p)->n, CALL)                             \
        } else {                                                          \
            using T_ = float;                                             \
            TCFD_DISPATCH_N(T_, (p)->n, CALL)                             \
        }                                                                 \
    }
#endif

static int check_ws(const tcfd_ns2d_plan* p, long batch, void* ws, size_t bytes, size_t need) {
    if (!ws || bytes < need) return fail(TCFD_EWORKSPACE, "workspace %zu B < required %zu B", bytes, need);
    return 0;
}

TCFD_API int tcfd_ns2d_plan_chunking(const tcfd_ns2d_plan* p, long batch, long* fields_per_chunk, size_t* cache_bytes,
                                       int* cache_source) {
    if (!p || batch < 0) return fail(TCFD_EINVAL, "plan_chunking: bad argument");
    if (fields_per_chunk) *fields_per_chunk = batch > 0 ? chunk_fields(p, batch) : 0;
    if (cache_bytes) *cache_bytes = p->tune.cache_bytes;
    if (cache_source) *cache_source = p->tune.cache_source;
    return 0;
}

// n = 3 * 2^k: how many fields make ONE round of resident column workgroups.  These plans run whole-column Stockham tiles
// (768^2 fp64: 107 KB of LDS = one 512-lane workgroup per CU, 49 tiles per field): a cache-sized chunk of 7 fields is 343
// workgroups = 1.34 rounds, 5 fields are 245 = one round -- 157 -> 181 steps/s.
template <typename T, int N>
static int round_fields_impl(tcfd_ns2d_plan* p) {
    if constexpr (is_pow2c(N)) {
        p->tune.round_fields = 0;
    } else {
        constexpr int EPT = Cfg<T, N>::COL_EPT, C = Cfg<T, N>::COLS;
        constexpr size_t lds = (size_t)lds_elems<N, EPT, C, false>() * sizeof(cx<T>) + 3 * (size_t)N * sizeof(T);
        constexpr int threads = C * (N / EPT);
        const int by_lds = (int)std::max<size_t>(1, (160 * 1024) / lds), by_thr = std::max(1, 1024 / threads);
        const int per_cu = std::min(std::min(by_lds, by_thr), 2);      // the column kernels use > 128 registers at EPT = 12
        const int tiles = (p->m + C - 1) / C;
        p->tune.round_fields = std::max(1, 256 * per_cu / tiles);
    }
    return 0;
}
TCFD_DISPATCHED(fill_round_fields, (tcfd_ns2d_plan * p), (p), (round_fields_impl<T_, N_>(p)))

// Fields per chunk of a batched call.  TCFD_CHUNK > 0 forces it, 0 disables chunking, -1 (default) sizes the chunk so
// that its working set -- 4 planes + advection + RK accumulator + padded state, 7 workspace fields per batch element --
// fits the 256 MB Infinity Cache (measured on MI355X, steps/s per call: 1024^2 x 64 fp64 122.7 -> 129.3 at 4 fields
// per chunk, 114.5 at 5; 512^2 x 64 fp64 408 -> 501 at 16).  Chunks are balanced.
static long chunk_fields(const tcfd_ns2d_plan* p, long batch) {
    long c = p->tune.chunk;
    if (c == 0) return batch;
    if (c < 0) {
        const size_t per_field = 7 * (size_t)p->n * p->ldw * (p->dtype == TCFD_C128 ? 16 : 8);
        // 61/64 of the cache (244 of 256 MB, the measured optimum on MI355X: 4 fields of 1024^2 fp64 = 239 MB fit, 5 do not)
        c = (long)((p->tune.cache_bytes / 64 * 61) / per_field);
        // fewer than 3 fields per chunk: only at 2048^2, where ONE field is a whole round of workgroups (258 column tiles) and
        // most of its working set (237 of 244 MB) still fits: 16 fields 12.8 -> 11.5 ms per step with single-field chunks; at
        // 1536^2 / 1280^2 (1 - 2 fields would fit, 194 / 162 workgroups per field) the whole batch at once is faster
        // (6.3 vs 7.9 ms, 9.3 vs 13.0 ms: tests/micro/n2048_chunk.py)
        if (c < 3) return (c >= 1 && p->n >= 2048) ? 1 : batch;
        if (p->tune.round_fields > 0 && c > p->tune.round_fields) c = c / p->tune.round_fields * p->tune.round_fields;   // whole rounds
    }
    if (c >= batch) return batch;
    const long nchunks = (batch + c - 1) / c;
    return (batch + nchunks - 1) / nchunks;
}

TCFD_DISPATCHED(chunk_dispatch,
                (const tcfd_ns2d_plan* p, const void* w_in, void* w_out, void* dwdt, long batch, int nstages, const double* beta,
                 const double* gdt, const double* mu_num, const double* fa, const double* mu_den, const int* base0, int steps,
                 double inv_total_dt, void* ws, hipStream_t st),
                (p, w_in, w_out, dwdt, batch, nstages, beta, gdt, mu_num, fa, mu_den, base0, steps, inv_total_dt, ws, st),
                (step_impl<T_, N_>(p, w_in, w_out, dwdt, batch, nstages, beta, gdt, mu_num, fa, mu_den, base0, steps,
                                   inv_total_dt, ws, st)))

TCFD_API int tcfd_ns2d_step_imex(const tcfd_ns2d_plan* p, const void* w_in, void* w_out, void* dwdt, long batch,
                                   int nstages, const double* fa, const double* beta, const double* gdt,
                                   const double* mu_num, const double* mu_den, const int* base0, int steps,
                                   double inv_total_dt, void* ws, size_t ws_bytes, void* stream) {
    if (!p || !w_in || !w_out || !beta || !gdt || !mu_num) return fail(TCFD_EINVAL, "step: null argument");
    if (batch <= 0 || steps <= 0 || nstages <= 0) return fail(TCFD_EINVAL, "step: batch/steps/nstages must be > 0");
    if (dwdt && w_in == w_out) return fail(TCFD_EINVAL, "step: w_out may alias w_in only when dwdt is NULL");
    int rc = check_ws(p, batch, ws, ws_bytes, tcfd_ns2d_workspace_bytes(p, batch));
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    // Batch chunking (TCFD_CHUNK = fields per chunk): the batch elements are independent, so the call may run chunk by
    // chunk -- every stage of every step of one chunk back to back, all chunks through the SAME (chunk-sized) part of
    // the workspace.  With a chunk whose working set (4 planes + adv + h + state) fits the 256 MB Infinity Cache the
    // planes written by a column pass are still on die when the row pass reads them.
    const long chunk = chunk_fields(p, batch);
    if (chunk < batch) {
        const size_t esz = (p->dtype == TCFD_C128 ? 16 : 8) * (size_t)p->n * p->m;
        for (long b0 = 0; b0 < batch; b0 += chunk) {
            const long nb = std::min(chunk, batch - b0);
            const unsigned char* in = (const unsigned char*)w_in + b0 * esz;
            unsigned char* out = (unsigned char*)w_out + b0 * esz;
            unsigned char* dw = dwdt ? (unsigned char*)dwdt + b0 * esz : nullptr;
            rc = chunk_dispatch(p, in, out, dw, nb, nstages, beta, gdt, mu_num, fa, mu_den, base0, steps, inv_total_dt, ws, st);
            if (rc) return rc;
        }
        return 0;
    }
    return chunk_dispatch(p, w_in, w_out, dwdt, batch, nstages, beta, gdt, mu_num, fa, mu_den, base0, steps, inv_total_dt, ws,
                          st);
}

TCFD_API int tcfd_ns2d_step(const tcfd_ns2d_plan* p, const void* w_in, void* w_out, void* dwdt, long batch,
                              int nstages, const double* beta, const double* gdt, const double* mu, int steps,
                              double inv_total_dt, void* ws, size_t ws_bytes, void* stream) {
    return tcfd_ns2d_step_imex(p, w_in, w_out, dwdt, batch, nstages, nullptr, beta, gdt, mu, nullptr, nullptr, steps,
                               inv_total_dt, ws, ws_bytes, stream);
}

TCFD_DISPATCHED(explicit_dispatch,
                (const tcfd_ns2d_plan* p, const void* w, void* out, const void* wt, void* psi, bool residual, long batch, void* ws,
                 hipStream_t st),
                (p, w, out, wt, psi, residual, batch, ws, st), (explicit_impl<T_, N_>(p, w, out, wt, psi, residual, batch, ws, st)))
// F(w) / residual sweeps chunk by chunk like the steps (the planes of a chunk stay on die between the two passes)
static int explicit_chunked(const tcfd_ns2d_plan* p, const void* w, void* out, const void* wt, void* psi, bool residual,
                            long batch, void* ws, hipStream_t st) {
    const long chunk = chunk_fields(p, batch);
    if (chunk >= batch) return explicit_dispatch(p, w, out, wt, psi, residual, batch, ws, st);
    const size_t esz = (p->dtype == TCFD_C128 ? 16 : 8) * (size_t)p->n * p->m;
    auto at = [&](const void* base, long b0) -> unsigned char* {
        return base ? (unsigned char*)base + (size_t)b0 * esz : nullptr;
    };
    for (long b0 = 0; b0 < batch; b0 += chunk) {
        const long nb = std::min(chunk, batch - b0);
        int rc = explicit_dispatch(p, at(w, b0), at(out, b0), at(wt, b0), at(psi, b0), residual, nb, ws, st);
        if (rc) return rc;
    }
    return 0;
}

TCFD_API int tcfd_ns2d_explicit_terms(const tcfd_ns2d_plan* p, const void* w, void* out, long batch, void* ws,
                                        size_t ws_bytes, void* stream) {
    if (!p || !w || !out || batch <= 0) return fail(TCFD_EINVAL, "explicit_terms: bad argument");
    int rc = check_ws(p, batch, ws, ws_bytes, tcfd_ns2d_workspace_bytes(p, batch));
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    return explicit_chunked(p, w, out, nullptr, nullptr, false, batch, ws, st);
}

// Vector-Jacobian product of the explicit terms (the backward of tcfd_ns2d_explicit_terms): from the state w and the
// pre-weighted cotangent gm = mask * g / c it produces the four half spectra  X_f = rfft2(Y_f)  (k_rows_vjp), which the caller
// combines:  wbar = -(c / n^2) sum_f conj(a_f) X_f.  Three kinds of launches per chunk: the opening column pass of the forward
// (w -> 4 planes) + the column inverse transform of gm, the row pass above, four generic column forward transforms.
template <typename T, int N>
static int explicit_vjp_impl(const tcfd_ns2d_plan* p, const void* w, const void* gm, void* xout, size_t x_field_stride,
                             long batch, void* ws, hipStream_t st) {
    Ws<T> W = carve<T>(p, ws, batch);
    int rc;
    ColArgs<T> a{};
    a.planes = W.planes;
    a.plane_stride = W.plane_stride;
    a.u_in = (const cx<T>*)w;
    a.u_in_ld = p->m;
    a.nyq = 0;    // whole-column tiles, every column present (the row pass reads the plain layout)
    if ((rc = launch_cols_v<T, N, MODE_A, Cfg<T, N>::COL_EPT, Cfg<T, N>::COLS>(p, a, batch, st))) return rc;
    ColArgs<T> g{};
    g.in = (const cx<T>*)gm;
    g.out = W.adv;
    g.in_ld = p->m;
    g.out_ld = p->ldw;
    g.scale = (T)1;
    if ((rc = launch_cols<T, N, MODE_INV>(p, g, batch, st))) return rc;
    {
        constexpr int EPT = Cfg<T, N>::ROW_EPT, THR = Cfg<T, N>::ROW_THREADS;
        using Gm = RowGeom<T, N, EPT, THR>;
        auto kern = k_rows_vjp<T, N, EPT, THR>;
        static DevOnce lds_once;
        if (int rc_ = set_lds(lds_once, kern, Gm::LDS_BYTES)) return rc_;
        const long npairs = batch * (N / 2);
        const long blocks = rows_grid(p, (npairs + Gm::GROUPS - 1) / Gm::GROUPS, 4);
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Gm::THREADS), Gm::LDS_BYTES, st, W.planes, W.plane_stride,
                           (const cx<T>*)W.adv, (const cx<T>*)p->tw, npairs, p->ldw);
        HIP_TRY(hipGetLastError());
    }
    for (int f = 0; f < 4; ++f) {
        ColArgs<T> c{};
        c.in = W.planes + (size_t)f * W.plane_stride;
        c.out = (cx<T>*)xout + (size_t)f * x_field_stride;
        c.in_ld = p->ldw;
        c.out_ld = p->m;
        c.scale = (T)1;
        if ((rc = launch_cols<T, N, MODE_FWD>(p, c, batch, st))) return rc;
    }
    return 0;
}
TCFD_DISPATCHED(explicit_vjp_dispatch,
                (const tcfd_ns2d_plan* p, const void* w, const void* gm, void* xout, size_t xs, long batch, void* ws, hipStream_t st),
                (p, w, gm, xout, xs, batch, ws, st), (explicit_vjp_impl<T_, N_>(p, w, gm, xout, xs, batch, ws, st)))

TCFD_API int tcfd_ns2d_explicit_terms_vjp(const tcfd_ns2d_plan* p, const void* w, const void* gm, void* xout, long batch,
                                            void* ws, size_t ws_bytes, void* stream) {
    if (!p || !w || !gm || !xout || batch <= 0) return fail(TCFD_EINVAL, "explicit_terms_vjp: bad argument");
    int rc = check_ws(p, batch, ws, ws_bytes, tcfd_ns2d_workspace_bytes(p, batch));
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t esz = (p->dtype == TCFD_C128 ? 16 : 8) * (size_t)p->n * p->m;
    const size_t xs = (size_t)batch * p->n * p->m;            // elements between the four outputs
    const long chunk = chunk_fields(p, batch);
    for (long b0 = 0; b0 < batch; b0 += chunk) {
        const long nb = std::min(chunk, batch - b0);
        rc = explicit_vjp_dispatch(p, (const unsigned char*)w + (size_t)b0 * esz, (const unsigned char*)gm + (size_t)b0 * esz,
                                   (unsigned char*)xout + (size_t)b0 * esz, xs, nb, ws, st);
        if (rc) return rc;
    }
    return 0;
}

TCFD_API int tcfd_ns2d_stream_residual(const tcfd_ns2d_plan* p, const void* w, const void* wt, void* psi,
                                         void* residual, long batch, void* ws, size_t ws_bytes, void* stream) {
    if (!p || !w || batch <= 0 || (residual && !wt)) return fail(TCFD_EINVAL, "stream_residual: bad argument");
    int rc = check_ws(p, batch, ws, ws_bytes, tcfd_ns2d_workspace_bytes(p, batch));
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    return explicit_chunked(p, w, residual, wt, psi, true, batch, ws, st);
}

template <typename T>
static int velocity_impl(const tcfd_ns2d_plan* p, const void* w, void* uh, void* vh, void* psi, long batch,
                         hipStream_t st) {
    const size_t count = (size_t)batch * p->n * p->m;
    const unsigned blocks = (unsigned)std::min<size_t>((count + 255) / 256, 4096);
    hipLaunchKernelGGL(k_velocity<T>, dim3(blocks), dim3(256), 0, st, (const cx<T>*)w, (cx<T>*)uh, (cx<T>*)vh,
                       (cx<T>*)psi, (const T*)p->kx, (const T*)p->ky, p->n, p->m, count);
    HIP_TRY(hipGetLastError());
    return 0;
}

TCFD_API int tcfd_ns2d_velocity(const tcfd_ns2d_plan* p, const void* w, void* uh, void* vh, void* psi, long batch,
                                  void* stream) {
    if (!p || !w || batch <= 0) return fail(TCFD_EINVAL, "velocity: bad argument");
    hipStream_t st = (hipStream_t)stream;
    return p->dtype == TCFD_C128 ? velocity_impl<double>(p, w, uh, vh, psi, batch, st)
                                 : velocity_impl<float>(p, w, uh, vh, psi, batch, st);
}

// ------------------------------------------------------------------ weighted squared norm of half spectra
// partial[b][blk] = sum over the block's share of field b of  |z[b][e]|^2 * w2[e]   (double accumulation): the
// Fourier-domain norm of SobolevLoss (fno/losses.py:263-315) in ONE pass over the spectrum instead of five
// element-wise / reduction launches.
template <typename T>
__global__ __launch_bounds__(256) void k_weighted_sqnorm(const cx<T>* __restrict__ z, const T* __restrict__ w2,
                                                         double* __restrict__ partial, long elems, int blocks) {
    __shared__ double sh[4];
    // one-dimensional grid of batch * blocks workgroups (grid.y stops at 65535 fields; b * T of a loss has no such bound)
    const long b = blockIdx.x / blocks;
    const int blk = blockIdx.x - (unsigned)(b * blocks);
    const cx<T>* zb = z + (size_t)b * elems;
    double acc = 0.0;
    for (long e = (long)blk * 256 + threadIdx.x; e < elems; e += (long)blocks * 256) {
        const cx<T> v = zb[e];
        acc += (double)((v.x * v.x + v.y * v.y) * w2[e]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

TCFD_API int tcfd_weighted_sqnorm(const void* z, const void* w2, void* partial, long batch, long elems, int blocks,
                                    int dtype, void* stream) {
    if (!z || !w2 || !partial || batch <= 0 || elems <= 0 || blocks <= 0)
        return fail(TCFD_EINVAL, "weighted_sqnorm: bad argument");
    if (dtype != TCFD_C64 && dtype != TCFD_C128) return fail(TCFD_EINVAL, "weighted_sqnorm: bad dtype %d", dtype);
    if ((double)batch * blocks >= 2147483647.0) return fail(TCFD_EINVAL, "weighted_sqnorm: batch * blocks exceeds the grid limit");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(batch * blocks));
    if (dtype == TCFD_C128)
        hipLaunchKernelGGL(k_weighted_sqnorm<double>, grid, dim3(256), 0, st, (const cx<double>*)z, (const double*)w2,
                           (double*)partial, elems, blocks);
    else
        hipLaunchKernelGGL(k_weighted_sqnorm<float>, grid, dim3(256), 0, st, (const cx<float>*)z, (const float*)w2,
                           (double*)partial, elems, blocks);
    HIP_TRY(hipGetLastError());
    return 0;
}

TCFD_DISPATCHED(rfft2_dispatch, (const tcfd_ns2d_plan* p, const void* x, void* out, long batch, hipStream_t st), (p, x, out, batch, st),
                (rfft2_impl<T_, N_>(p, x, out, batch, st)))
TCFD_DISPATCHED(irfft2_dispatch, (const tcfd_ns2d_plan* p, const void* xh, void* out, long batch, void* ws, hipStream_t st),
                (p, xh, out, batch, ws, st), (irfft2_impl<T_, N_>(p, xh, out, batch, ws, st)))

template <typename T, int N>
static int irfft2_sub_max_impl(const tcfd_ns2d_plan*) {
    using Gm = RowGeom<T, N, Cfg<T, N>::ROW_EPT>;
    return Gm::G < 64 ? Gm::G : 64;
}
TCFD_DISPATCHED(irfft2_sub_max_dispatch, (const tcfd_ns2d_plan* p), (p), (irfft2_sub_max_impl<T_, N_>(p)))
TCFD_DISPATCHED(irfft2_sub_dispatch, (const tcfd_ns2d_plan* p, const void* xh, void* out, long batch, int S, void* ws, hipStream_t st),
                (p, xh, out, batch, S, ws, st), (irfft2_sub_impl<T_, N_>(p, xh, out, batch, S, ws, st)))

// largest factor tcfd_irfft2_subsample accepts for this plan (the lanes of one row transform, 64 at most); < 0: bad plan
TCFD_API int tcfd_irfft2_subsample_max_factor(const tcfd_ns2d_plan* p) {
    if (!p) return fail(TCFD_EINVAL, "irfft2_subsample_max_factor: null plan");
    return irfft2_sub_max_dispatch(p);
}
TCFD_API int tcfd_irfft2_subsample(const tcfd_ns2d_plan* p, const void* xh, void* out, long batch, int factor, void* ws,
                                   size_t ws_bytes, void* stream) {
    if (!p || !xh || !out || batch <= 0) return fail(TCFD_EINVAL, "irfft2_subsample: bad argument");
    if (factor < 2 || p->n % factor) return fail(TCFD_EINVAL, "irfft2_subsample: factor %d does not divide n = %d", factor, p->n);
    int rc = check_ws(p, batch, ws, ws_bytes, field_bytes(p, batch));
    if (rc) return rc;
    return irfft2_sub_dispatch(p, xh, out, batch, factor, ws, (hipStream_t)stream);
}

TCFD_API int tcfd_rfft2(const tcfd_ns2d_plan* p, const void* x, void* out, long batch, void* stream) {
    if (!p || !x || !out || batch <= 0) return fail(TCFD_EINVAL, "rfft2: bad argument");
    return rfft2_dispatch(p, x, out, batch, (hipStream_t)stream);
}

TCFD_API int tcfd_irfft2(const tcfd_ns2d_plan* p, const void* xh, void* out, long batch, void* ws, size_t ws_bytes,
                           void* stream) {
    if (!p || !xh || !out || batch <= 0) return fail(TCFD_EINVAL, "irfft2: bad argument");
    int rc = check_ws(p, batch, ws, ws_bytes, field_bytes(p, batch));
    if (rc) return rc;
    return irfft2_dispatch(p, xh, out, batch, ws, (hipStream_t)stream);
}

template <typename T, int N>
static int variant_impl(const tcfd_ns2d_plan* p, int* split, int* rows_kernel) {
    const bool sp = use_split<T, N>(p);
    if (split) *split = sp ? 1 : 0;
    if (rows_kernel) {
        int v = p->tune.rows_v;
        if (v == 4 || v == 6) v = 5;                 // (removed kernels)
        if (v != 5 && v != 7) v = rows_default_version<T, N>();
        if (v == 7 && N != 1024) v = 5;
        *rows_kernel = v;
    }
    return 0;
}

TCFD_DISPATCHED(variant_dispatch, (const tcfd_ns2d_plan* p, int* split, int* rows_kernel), (p, split, rows_kernel),
                (variant_impl<T_, N_>(p, split, rows_kernel)))
TCFD_API int tcfd_ns2d_plan_variant(const tcfd_ns2d_plan* p, int* split, int* rows_kernel) {
    if (!p) return fail(TCFD_EINVAL, "plan_variant: null plan");
    return variant_dispatch(p, split, rows_kernel);
}

// ------------------------------------------------------------------ test hook: the cross-lane transform alone
// One 128-lane group transforms `count` independent 1024-point sequences (complex128).  dir = +1: natural-order
// input, output in the permutation pi of xl_fft1024 (element t of lane j at out[seq][128 t + j]); dir = -1: input in
// that layout, natural-order output.  tests/test_ns2d_gpu.py checks both against numpy through the map of pi.
template <int DIR>
__global__ __launch_bounds__(128) void k_debug_xl(const cx<double>* __restrict__ in, cx<double>* __restrict__ out,
                                                  const cx<double>* __restrict__ tw) {
    __shared__ __attribute__((aligned(16))) cx<double> lds[1024];
    const int j = threadIdx.x;
    const XlTw<double> xtw = xl_load_tw<double>(tw, j);
    cx<double> x[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = in[(size_t)blockIdx.x * 1024 + 128 * t + j];
    NoHook nohook;
    xl_fft1024<double, DIR, 1>(x, lds, xtw, j, nohook);
#pragma unroll
    for (int t = 0; t < 8; ++t) out[(size_t)blockIdx.x * 1024 + 128 * t + j] = x[t];
}

TCFD_API int tcfd_debug_xl_fft1024(const tcfd_ns2d_plan* p, const void* in, void* out, int count, int dir, void* stream) {
    if (!p || !in || !out || count <= 0 || (dir != 1 && dir != -1)) return fail(TCFD_EINVAL, "debug_xl_fft1024: bad argument");
    if (p->n != 1024 || p->dtype != TCFD_C128) return fail(TCFD_EINVAL, "debug_xl_fft1024: needs a 1024^2 complex128 plan");
    hipStream_t st = (hipStream_t)stream;
    if (dir > 0)
        hipLaunchKernelGGL(k_debug_xl<+1>, dim3(count), dim3(128), 0, st, (const cx<double>*)in, (cx<double>*)out,
                           (const cx<double>*)p->tw);
    else
        hipLaunchKernelGGL(k_debug_xl<-1>, dim3(count), dim3(128), 0, st, (const cx<double>*)in, (cx<double>*)out,
                           (const cx<double>*)p->tw);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ profiling side-car
TCFD_API int tcfd_ns2d_profile_begin(tcfd_ns2d_plan* p, int max_records) {
    if (!p || max_records <= 0) return fail(TCFD_EINVAL, "profile_begin: bad argument");
    if (!p->prof) p->prof = new ProfState();
    for (auto& r : p->prof->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    p->prof->recs.clear();
    p->prof->recs.reserve(max_records);
    p->prof->max_records = max_records;
    p->prof->on = true;
    return 0;
}

TCFD_API int tcfd_ns2d_profile_end(tcfd_ns2d_plan* p, int capacity, int* count, int* kinds, float* ms) {
    if (!p || !p->prof || !count) return fail(TCFD_EINVAL, "profile_end: profiling was not started");
    ProfState* ps = p->prof;
    ps->on = false;
    int n = 0;
    for (auto& r : ps->recs) {
        HIP_TRY(hipEventSynchronize(r.e1));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, r.e0, r.e1));
        if (n < capacity && kinds && ms) { kinds[n] = r.kind; ms[n] = t; }
        ++n;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    ps->recs.clear();
    *count = n;
    return 0;
}

#if TCFD_UNIT != 1   // (a non-template kernel: one definition in the library)
// ---------------------------------------------------------------- HBM probe (bench.py reports it beside the 8 TB/s spec)
__global__ __launch_bounds__(256) void k_probe(const double2* __restrict__ src, double2* __restrict__ dst, size_t n16,
                                               int mode) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (mode == 0) {
        for (; i < n16; i += stride) dst[i] = src[i];
    } else if (mode == 1) {
        double acc = 0.0;
        for (; i < n16; i += stride) { const double2 v = src[i]; acc += v.x + v.y; }
        if (acc == 1.2345e300) reinterpret_cast<double*>(dst)[0] = acc;   // keeps the loads alive, never true
    } else {
        const double2 z = make_double2(0.0, 0.0);
        for (; i < n16; i += stride) dst[i] = z;
    }
}

TCFD_API int tcfd_hbm_probe(const void* src, void* dst, size_t bytes, int mode, int iters, float* ms, void* stream) {
    if (!dst || (mode != 2 && !src) || !ms || bytes < 16 || bytes % 16 || iters < 1 || mode < 0 || mode > 2)
        return fail(TCFD_EINVAL, "hbm_probe: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const size_t n16 = bytes / 16;
    const unsigned blocks = 256 * 16;  // 16 workgroups per CU, grid-stride
    hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, st, (const double2*)src, (double2*)dst, n16, mode);  // warm-up
    HIP_TRY(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, st, (const double2*)src, (double2*)dst, n16, mode);
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIP_TRY(hipGetLastError());
    *ms = t / iters;
    return 0;
}

#endif

// ---------------------------------------------------------------- strided device -> host copy of record slabs
// One record of `rows` samples lands in a host array whose sample pitch is a whole trajectory: a pitched copy on the caller's
// stream (asynchronous when the host side is page-locked), so the hand-over of a record overlaps the steps that follow it.
TCFD_API int tcfd_copy_rows_to_host(void* dst_host, size_t dst_pitch, const void* src_dev, size_t src_pitch,
                                      size_t row_bytes, size_t rows, void* stream) {
    if (!dst_host || !src_dev || row_bytes == 0 || dst_pitch < row_bytes || src_pitch < row_bytes)
        return fail(TCFD_EINVAL, "copy_rows_to_host: bad argument");
    if (rows == 0) return 0;
    HIP_TRY(hipMemcpy2DAsync(dst_host, dst_pitch, src_dev, src_pitch, row_bytes, rows, hipMemcpyDeviceToHost,
                             (hipStream_t)stream));
    return 0;
}

// Page-lock / release a range of ordinary host memory (hipHostRegister): the record hand-over locks the result of an
// ensemble job region by region on a helper thread -- ONE page-locked allocation of the whole result holds the runtime's
// memory lock for its full duration (0.36 s at 5.4 GB) and stalls every hipMalloc of the stepping thread behind it.
TCFD_API int tcfd_host_register(void* ptr, size_t bytes) {
    if (!ptr || bytes == 0) return fail(TCFD_EINVAL, "host_register: bad argument");
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return 0;
}
TCFD_API int tcfd_host_unregister(void* ptr) {
    if (!ptr) return fail(TCFD_EINVAL, "host_unregister: null pointer");
    HIP_TRY(hipHostUnregister(ptr));
    return 0;
}

// ---------------------------------------------------------------- stage bookkeeping of the DIFFERENTIABLE step
// With constant coefficients a stage of the low-storage RK / Crank-Nicolson schedule is linear in (f, h_prev, b):
//     h = fa f + beta h_prev ,    u = (b + gdt h + mu L b) / (1 - mud L)            (L: the real (n, m) linear term)
// -- written exactly as the fused forward kernels write it (k_cols, MODE_C), so a differentiable trajectory reproduces the
// plain one to round-off.  autograd.py ran this as ~10 element-wise tensor launches forward and ~20 backward per stage; these
// two kernels are the stage and its vector-Jacobian product in one launch each:
//     G = g_h + gdt r (.) g_u ,   g_f = fa G ,   g_hprev = beta G ,   g_b = (1 + mu L) r (.) g_u ,      r = 1 / (1 - mud L).
#if TCFD_UNIT != 1
template <typename T>
__global__ __launch_bounds__(256) void k_stage_update(const cx<T>* __restrict__ f, const cx<T>* __restrict__ hp,
                                                      const cx<T>* __restrict__ b, const T* __restrict__ Lt, T fa, T beta, T gdt,
                                                      T mu, T mud, cx<T>* __restrict__ h, cx<T>* __restrict__ u, long total,
                                                      long plane) {
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
        const T L = Lt[e % plane];
        cx<T> hn = cscale(f[e], fa);
        if (hp) hn = hn + cscale(hp[e], beta);
        const cx<T> bv = b[e];
        const cx<T> rhs = bv + cscale(hn, gdt) + cscale(cscale(bv, L), mu);
        h[e] = hn;
        u[e] = cscale(rhs, fast_rcp((T)1 - mud * L));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_stage_update_vjp(const cx<T>* __restrict__ gu, const cx<T>* __restrict__ gh,
                                                          const T* __restrict__ Lt, T fa, T beta, T gdt, T mu, T mud,
                                                          cx<T>* __restrict__ gf, cx<T>* __restrict__ ghp,
                                                          cx<T>* __restrict__ gb, long total, long plane) {
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
        const T L = Lt[e % plane];
        const cx<T> g = cscale(gu[e], fast_rcp((T)1 - mud * L));     // r (.) g_u
        cx<T> G = cscale(g, gdt);
        if (gh) G = G + gh[e];
        gf[e] = cscale(G, fa);
        if (ghp) ghp[e] = cscale(G, beta);
        gb[e] = g + cscale(cscale(g, L), mu);
    }
}
template <typename T>
static int stage_update_impl(const void* f, const void* hp, const void* b, const void* Lt, const double* c, void* h, void* u,
                             long batch, long plane, hipStream_t st) {
    const long total = batch * plane;
    const unsigned blocks = (unsigned)std::min<long>((total + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(k_stage_update<T>, dim3(blocks), dim3(256), 0, st, (const cx<T>*)f, (const cx<T>*)hp, (const cx<T>*)b,
                       (const T*)Lt, (T)c[0], (T)c[1], (T)c[2], (T)c[3], (T)c[4], (cx<T>*)h, (cx<T>*)u, total, plane);
    HIP_TRY(hipGetLastError());
    return 0;
}
template <typename T>
static int stage_update_vjp_impl(const void* gu, const void* gh, const void* Lt, const double* c, void* gf, void* ghp, void* gb,
                                 long batch, long plane, hipStream_t st) {
    const long total = batch * plane;
    const unsigned blocks = (unsigned)std::min<long>((total + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(k_stage_update_vjp<T>, dim3(blocks), dim3(256), 0, st, (const cx<T>*)gu, (const cx<T>*)gh, (const T*)Lt,
                       (T)c[0], (T)c[1], (T)c[2], (T)c[3], (T)c[4], (cx<T>*)gf, (cx<T>*)ghp, (cx<T>*)gb, total, plane);
    HIP_TRY(hipGetLastError());
    return 0;
}
#endif
// coef = {fa, beta, gdt, mu, mud}
TCFD_API int tcfd_ns2d_stage_update(const void* f, const void* h_prev, const void* b, const void* lin, const double* coef,
                                    void* h, void* u, long batch, long plane, int dtype, void* stream) {
#if TCFD_UNIT != 1
    if (!f || !b || !lin || !coef || !h || !u || batch <= 0 || plane <= 0) return fail(TCFD_EINVAL, "stage_update: bad argument");
    if (dtype == TCFD_C128) return stage_update_impl<double>(f, h_prev, b, lin, coef, h, u, batch, plane, (hipStream_t)stream);
    if (dtype == TCFD_C64) return stage_update_impl<float>(f, h_prev, b, lin, coef, h, u, batch, plane, (hipStream_t)stream);
    return fail(TCFD_EINVAL, "stage_update: bad dtype %d", dtype);
#else
    return 0;
#endif
}
TCFD_API int tcfd_ns2d_stage_update_vjp(const void* g_u, const void* g_h, const void* lin, const double* coef, void* g_f,
                                        void* g_hprev, void* g_b, long batch, long plane, int dtype, void* stream) {
#if TCFD_UNIT != 1
    if (!g_u || !lin || !coef || !g_f || !g_b || batch <= 0 || plane <= 0) return fail(TCFD_EINVAL, "stage_update_vjp: bad argument");
    if (dtype == TCFD_C128)
        return stage_update_vjp_impl<double>(g_u, g_h, lin, coef, g_f, g_hprev, g_b, batch, plane, (hipStream_t)stream);
    if (dtype == TCFD_C64)
        return stage_update_vjp_impl<float>(g_u, g_h, lin, coef, g_f, g_hprev, g_b, batch, plane, (hipStream_t)stream);
    return fail(TCFD_EINVAL, "stage_update_vjp: bad dtype %d", dtype);
#else
    return 0;
#endif
}

// wbar = sum_f post_f (.) X_f : the last step of the explicit terms' vector-Jacobian product (X: (4, batch, plane) half spectra
// from tcfd_ns2d_explicit_terms_vjp, post: (4, plane) complex tables -(c / n^2) conj(a_f)).  As tensor ops this was a broadcast
// multiply writing 4 planes per field and a reduction reading them back (0.34 ms per stage at 1024^2 x 8); one pass here.
#if TCFD_UNIT != 1
template <typename T>
__global__ __launch_bounds__(256) void k_vjp_combine(const cx<T>* __restrict__ X, const cx<T>* __restrict__ post,
                                                     cx<T>* __restrict__ out, long total, long plane) {
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
        const long t = e % plane;
        cx<T> acc = mk<T>((T)0, (T)0);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const cx<T> x = X[(long)f * total + e], p = post[(long)f * plane + t];
            acc = acc + mk<T>(x.x * p.x - x.y * p.y, x.x * p.y + x.y * p.x);
        }
        out[e] = acc;
    }
}
#endif
TCFD_API int tcfd_ns2d_vjp_combine(const void* X, const void* post, void* out, long batch, long plane, int dtype, void* stream) {
#if TCFD_UNIT != 1
    if (!X || !post || !out || batch <= 0 || plane <= 0) return fail(TCFD_EINVAL, "vjp_combine: bad argument");
    const long total = batch * plane;
    const unsigned blocks = (unsigned)std::min<long>((total + 255) / 256, 1 << 16);
    if (dtype == TCFD_C128)
        hipLaunchKernelGGL(k_vjp_combine<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const cx<double>*)X,
                           (const cx<double>*)post, (cx<double>*)out, total, plane);
    else if (dtype == TCFD_C64)
        hipLaunchKernelGGL(k_vjp_combine<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const cx<float>*)X,
                           (const cx<float>*)post, (cx<float>*)out, total, plane);
    else
        return fail(TCFD_EINVAL, "vjp_combine: bad dtype %d", dtype);
    HIP_TRY(hipGetLastError());
    return 0;
#else
    return 0;
#endif
}
