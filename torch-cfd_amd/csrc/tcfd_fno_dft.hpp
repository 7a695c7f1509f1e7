// tcfd_fno_dft.hpp -- spectral-convolution transforms for ANY spatial size (included by tcfd_fno.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "tcfd_fft.hpp"

using namespace tcfd;

// ------------------------------------------------------------------ transforms for ANY X / Y: pruned direct DFTs
// The FFT kernels above transform power-of-two X and Y.  Every other size -- 96^2 / 192^2 data, the X + 2p grid of
// SFNO(spatial_padding = p) (fno/sfno.py:313-328), non-square grids -- runs the SAME five-kernel pipeline with these
// three kernels in place of k_fwd_ty2 / k_x / k_inv_ty2.  Only 2mx x 2my x mt modes are ever formed (fno/sfno.py:379-381),
// so a direct DFT restricted to the kept rows costs Y * 2my * mt complex multiply-adds per slab -- at 2my = 48 of 96..272 a
// few times an FFT's arithmetic, on data that is read exactly once either way; no radix schedule, any length up to 1024.
// (Round 4 first served these sizes through thin library GEMMs with the kept rows of the DFT matrices, dense_fft.py: six
// passes over activation-sized intermediates, 6-16 x the fused kernels' time per grid point.)
//   k_fwd_ty_dft   slab [Y][T] real -> W1a[y][kt] (short real DFT in t, one lane per y) -> out[ky][kt], one lane per
//                  +-ky PAIR: a e^{-i th} and a e^{+i th} share their four products
//   k_x_dft        (bc, n_in, Q) -> (bc, n_out, Q): one lane per column q, 16 outputs per lane, the twiddle of (output, input)
//                  is lane uniform (LDS broadcast)
//   k_inv_ty_dft   W2[kyi][kt] -> S[y][kt] = sum_kyi W e^{+i th} (one lane per y) -> c2r in t in registers -> slab [Y][t_keep]
// Array-index conventions as above: truncated row kyi < my is ky = kyi, kyi >= my is array index Ys - 2my + kyi of the
// SOURCE grid (dropped when it falls outside an output grid of another size).
// MT: compile-time bound of the time modes (== mt, or the next bucket with run-time guards): with 16 predicated trips for
// mt = 5 the kernels issued three times the instructions they needed
template <typename T, int MT>
__global__ __launch_bounds__(256) void k_fwd_ty_dft(const T* __restrict__ v, cx<T>* __restrict__ w1, const cx<T>* __restrict__ tw_y,
                                                    const cx<T>* __restrict__ tw_tf, int Y, int T_in, int t_pad, int mt, int my,
                                                    T scale, int NS, long slabs, const T* __restrict__ kts) {
    typedef cx<T> cf;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Tp = T_in + t_pad, Q = 2 * my * mt;
    cf* twy = reinterpret_cast<cf*>(smem_raw);               // [Y]
    cf* twt = twy + Y;                                       // [mt][Tp]
    cf* A = twt + (size_t)mt * Tp;                           // [NS][Y][mt]
    const long base = (long)blockIdx.x * NS;
    const int count = (int)(slabs - base < NS ? slabs - base : NS);
    for (int i = threadIdx.x; i < Y; i += blockDim.x) twy[i] = tw_y[i];
    for (int i = threadIdx.x; i < mt * Tp; i += blockDim.x) twt[i] = kts ? cscale(tw_tf[i], kts[i / Tp]) : tw_tf[i];   // per-mode factor
    __syncthreads();
    // phase 1: W1a[y][kt] = sum_t v[y][t] w[kt][t_pad + t]
    for (int r = threadIdx.x; r < count * Y; r += blockDim.x) {
        const T* row = v + ((size_t)base * Y + r) * T_in;
        cf acc[MT];
#pragma unroll
        for (int k = 0; k < MT; ++k) acc[k] = mk<T>((T)0, (T)0);
        for (int t = 0; t < T_in; ++t) {
            const T x = row[t];
#pragma unroll
            for (int k = 0; k < MT; ++k)
                if (k < mt) {
                    const cf w = twt[(size_t)k * Tp + t_pad + t];
                    acc[k].x += x * w.x;
                    acc[k].y += x * w.y;
                }
        }
        cf* dst = A + (size_t)r * mt;
#pragma unroll
        for (int k = 0; k < MT; ++k)
            if (k < mt) dst[k] = acc[k];
    }
    __syncthreads();
    // phase 2: one lane per (slab, ky in [0, my]): out[+ky][kt] and out[-ky][kt]
    const int per = my + 1;
    for (int task = threadIdx.x; task < count * per; task += blockDim.x) {
        const int s = task / per, ky = task - s * per;
        const cf* a = A + (size_t)s * Y * mt;
        // a e^{-i th} and a e^{+i th} from P = sum a cos, R = sum a sin (four FMAs per term and time mode instead of four
        // products and eight sums):  out[+ky] = P - i R,  out[-ky] = P + i R
        cf Pc[MT], Rs[MT];
#pragma unroll
        for (int k = 0; k < MT; ++k) { Pc[k] = mk<T>((T)0, (T)0); Rs[k] = Pc[k]; }
        int idx = 0;
#pragma unroll 4
        for (int y = 0; y < Y; ++y) {
            const cf e = twy[idx];                 // (cos, -sin)
            idx += ky;
            if (idx >= Y) idx -= Y;
#pragma unroll
            for (int k = 0; k < MT; ++k)
                if (k < mt) {
                    const cf av = a[(size_t)y * mt + k];
                    Pc[k].x += av.x * e.x;  Pc[k].y += av.y * e.x;
                    Rs[k].x -= av.x * e.y;  Rs[k].y -= av.y * e.y;
                }
        }
        cf* dst = w1 + (size_t)(base + s) * Q;
#pragma unroll
        for (int k = 0; k < MT; ++k)
            if (k < mt) {
                if (ky < my) dst[(size_t)ky * mt + k] = mk<T>((Pc[k].x + Rs[k].y) * scale, (Pc[k].y - Rs[k].x) * scale);
                if (ky >= 1) dst[(size_t)(2 * my - ky) * mt + k] = mk<T>((Pc[k].x - Rs[k].y) * scale, (Pc[k].y + Rs[k].x) * scale);
            }
    }
}

// FWD: in (bc, n, Q) rows x -> out (bc, 2m, Q) kept rows;  INV: in (bc, 2m, Q) -> out (bc, n, Q), source grid ns
template <typename T, bool FWD>
__global__ __launch_bounds__(256) void k_x_dft(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, const cx<T>* __restrict__ tw_x,
                                               int n, int ns, int m, int Q) {
    typedef cx<T> cf;
    // (the twiddle of (output, input) is lane uniform; read through the scalar unit -- s_load per output and input row -- the
    //  kernel measured SLOWER than with this LDS copy of the table and broadcast reads: 154 vs 131 us at 96^2 x 320)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* tw = reinterpret_cast<cf*>(smem_raw);                // [n]
    for (int i = threadIdx.x; i < n; i += blockDim.x) tw[i] = tw_x[i];
    __syncthreads();
    constexpr int OC = 16;                                   // outputs per lane
    const int q = blockIdx.x * 256 + threadIdx.x;
    const size_t bc = blockIdx.z;
    const int o0 = blockIdx.y * OC;
    if (q >= Q) return;      // (after the workgroup's only barrier)
    const int n_in = FWD ? n : 2 * m, n_out = FWD ? 2 * m : n;
    auto arr = [&](int ki) { return ki < m ? ki : ns - 2 * m + ki; };   // array index of truncated row ki on its source grid
    cf acc[OC];
    int step[OC], idx[OC];
#pragma unroll
    for (int u = 0; u < OC; ++u) {
        acc[u] = mk<T>((T)0, (T)0);
        const int o = o0 + u;
        // the twiddle index advances by `step` per input row: FWD k_arr(o) per x, INV x = o per unit of k_arr
        step[u] = (o < n_out) ? (FWD ? arr(o) % n : o % n) : 0;
        idx[u] = 0;
    }
    const cf* src = in + (bc * n_in) * (size_t)Q + q;
    // eight input rows are requested before the first is used: one row per trip made the loop a chain of n_in memory round trips
    constexpr int PF = 8;
    for (int i0 = 0; i0 < n_in; i0 += PF) {
        cf xs[PF];
#pragma unroll
        for (int jj = 0; jj < PF; ++jj)
            if (i0 + jj < n_in) xs[jj] = src[(size_t)(i0 + jj) * Q];
#pragma unroll
        for (int jj = 0; jj < PF; ++jj) {
            const int i = i0 + jj;
            if (i >= n_in) break;
            bool live = true;
            if constexpr (!FWD) {
                const int a_i = arr(i);
                live = a_i < n;                              // a high row that falls outside a smaller output grid
                if (i == m) {                                // jump from the low block to the high block of array indices
#pragma unroll
                    for (int u = 0; u < OC; ++u) idx[u] = (int)(((long)step[u] * (a_i % n)) % n);
                }
            }
            const cf x = xs[jj];
            if (live) {
#pragma unroll
                for (int u = 0; u < OC; ++u) {
                    cf e = tw[idx[u]];
                    if constexpr (!FWD) e.y = -e.y;
                    acc[u].x += x.x * e.x - x.y * e.y;
                    acc[u].y += x.x * e.y + x.y * e.x;
                }
            }
#pragma unroll
            for (int u = 0; u < OC; ++u) {
                idx[u] += step[u];
                if (idx[u] >= n) idx[u] -= n;
            }
        }
    }
    cf* dst = out + (bc * n_out) * (size_t)Q + q;
#pragma unroll
    for (int u = 0; u < OC; ++u)
        if (o0 + u < n_out) dst[(size_t)(o0 + u) * Q] = acc[u];
}

template <typename T, int MT>
__global__ __launch_bounds__(1024) void k_inv_ty_dft(const cx<T>* __restrict__ w2, T* out, const cx<T>* __restrict__ tw_y,
                                                    const cx<T>* __restrict__ tw_ti, int Y, int Ys, int T_out, int t_keep, int mt,
                                                    int my, T scale, int NS, long slabs, const T* acc, const T* __restrict__ accb,
                                                    int accT, const T* __restrict__ kts) {
    typedef cx<T> cf;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Q = 2 * my * mt, t0 = T_out - t_keep;
    cf* twy = reinterpret_cast<cf*>(smem_raw);               // [Y]
    cf* twt = twy + Y;                                       // [t_keep][mt]
    cf* win = twt + (size_t)t_keep * mt;                     // [NS][Q + 2 mt]
    cf* ud = win + (size_t)NS * (Q + 2 * mt);                // [2][NS][my + 1][mt]: U and D of the paired form
    T* slab = reinterpret_cast<T*>(ud + (size_t)2 * NS * (my + 1) * mt);    // [NS][Y][t_keep]
    const long base = (long)blockIdx.x * NS;
    const int count = (int)(slabs - base < NS ? slabs - base : NS);
    for (int i = threadIdx.x; i < Y; i += blockDim.x) twy[i] = tw_y[i];
    for (int i = threadIdx.x; i < t_keep * mt; i += blockDim.x)
        twt[i] = kts ? cscale(tw_ti[(size_t)t0 * mt + i], kts[i % mt]) : tw_ti[(size_t)t0 * mt + i];
    {
        const cf* src = w2 + (size_t)base * Q;
        for (int i = threadIdx.x; i < count * Q; i += blockDim.x) win[(size_t)(i / Q) * (Q + 2 * mt) + (i % Q)] = src[i];
    }
    __syncthreads();
    // Ys == Y (no resampling in y): the slots of +ky and -ky meet in  W+ e^{+i th} + W- e^{-i th} = U cos + D sin  with
    // U = W+ + W-,  D = i (W+ - W-)  formed once per slab (they overwrite the slab's spectrum in LDS: rows [0, my] = U,
    // rows [my + 1, 2 my + 1] = D; `win` holds Q + 2 mt entries per slab for that), half the per-lane products
    const bool paired = (Ys == Y);
    const int wstride = Q + 2 * mt;
    if (paired) {
        for (int i = threadIdx.x; i < count * (my + 1) * mt; i += blockDim.x) {
            const int s = i / ((my + 1) * mt), rem = i - s * (my + 1) * mt;
            const int ky = rem / mt, k = rem - ky * mt;
            const cf* wq = win + (size_t)s * wstride;
            const cf wp = ky < my ? wq[(size_t)ky * mt + k] : mk<T>((T)0, (T)0);
            const cf wm = ky >= 1 ? wq[(size_t)(2 * my - ky) * mt + k] : mk<T>((T)0, (T)0);
            ud[i] = mk<T>(wp.x + wm.x, wp.y + wm.y);                                        // U
            ud[(size_t)count * (my + 1) * mt + i] = mk<T>(-(wp.y - wm.y), wp.x - wm.x);     // D = i (W+ - W-)
        }
        __syncthreads();
    }
    for (int r = threadIdx.x; r < count * Y; r += blockDim.x) {
        const int s = r / Y, y = r - s * Y;
        const cf* wq = win + (size_t)s * wstride;
        cf S[MT];
#pragma unroll
        for (int k = 0; k < MT; ++k) S[k] = mk<T>((T)0, (T)0);
        int idx = 0;
        if (paired) {
            const cf* U = ud + (size_t)s * (my + 1) * mt;
            const cf* D = U + (size_t)count * (my + 1) * mt;
#pragma unroll 4
            for (int ky = 0; ky <= my; ++ky) {
                const cf e = twy[idx];             // (cos, -sin)
                idx += y;
                if (idx >= Y) idx -= Y;
#pragma unroll
                for (int k = 0; k < MT; ++k)
                    if (k < mt) {
                        const cf u = U[(size_t)ky * mt + k], d = D[(size_t)ky * mt + k];
                        S[k].x += u.x * e.x - d.x * e.y;
                        S[k].y += u.y * e.x - d.y * e.y;
                    }
            }
        } else {
            for (int kyi = 0; kyi < 2 * my; ++kyi) {
                const int ka = kyi < my ? kyi : Ys - 2 * my + kyi;
                if (kyi == my) idx = (int)(((long)(ka % Y) * y) % Y);
                if (ka < Y) {
                    const cf e = twy[idx];             // e^{+i th} = conj
#pragma unroll
                    for (int k = 0; k < MT; ++k)
                        if (k < mt) {
                            const cf wv = wq[(size_t)kyi * mt + k];
                            S[k].x += wv.x * e.x + wv.y * e.y;
                            S[k].y += wv.y * e.x - wv.x * e.y;
                        }
                }
                idx += y;
                if (idx >= Y) idx -= Y;
            }
        }
        T* o = slab + (size_t)r * t_keep;
        for (int t = 0; t < t_keep; ++t) {
            T a = 0;
#pragma unroll
            for (int k = 0; k < MT; ++k)
                if (k < mt) {
                    const cf E = twt[(size_t)t * mt + k];
                    a += S[k].x * E.x - S[k].y * E.y;
                }
            o[t] = a * scale;
        }
    }
    __syncthreads();
    const size_t slab_elems = (size_t)Y * t_keep;
    const long total = (long)count * slab_elems;
    T* dst = out + (size_t)base * slab_elems;
    if (accb) {
        for (long i = threadIdx.x; i < total; i += blockDim.x) {
            const long row = i / t_keep;                     // (slab, y) row of the residual array
            if (accT > 0) dst[i] = slab[i] + accb[(size_t)((size_t)base * Y + row) * accT + (accT - 1)];
            else dst[i] = slab[i] + ((i - row * t_keep) == t_keep - 1 ? accb[(size_t)base * Y + row] : (T)0);   // accT < 0: last step only
        }
    } else if (acc) {
        const T* a = acc + (size_t)base * slab_elems;
        for (long i = threadIdx.x; i < total; i += blockDim.x) dst[i] = slab[i] + a[i];
    } else {
        for (long i = threadIdx.x; i < total; i += blockDim.x) dst[i] = slab[i];
    }
}

