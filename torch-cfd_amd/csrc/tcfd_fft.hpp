// tcfd_fft.hpp -- register/LDS Stockham FFT building blocks for gfx950 (CDNA4).
//
// Written from scratch for MI355X.  The reference (scaomath/torch-cfd) has no
// FFT of its own: every transform is a call into torch.fft (equations.py:415,
// 419,422).  Here a power-of-two transform of length N is held EPT elements
// per lane in registers; G = N/EPT lanes cooperate on one transform and
// exchange data through LDS between radix passes (autosort Stockham, so no
// bit-reversal pass).  Lane j owns elements  e = j + t*G,  t = 0..EPT-1  BEFORE
// and AFTER the transform, which lets the k-space arithmetic of the solver be
// fused around the transform without any extra LDS round trip.
//
// A "tile" is C independent transforms side by side, element-major in LDS
// ([e][c], c fastest) so that a wave touching C adjacent columns of a row-major
// (n, m) spectrum makes C*sizeof(complex)-byte contiguous HBM segments and
// conflict-free LDS rows.
#pragma once
#include <hip/hip_runtime.h>

namespace tcfd {

template <typename T>
struct alignas(2 * sizeof(T)) cx {
    T x, y;
};

template <typename T> __device__ __forceinline__ cx<T> mk(T a, T b) { cx<T> r; r.x = a; r.y = b; return r; }
template <typename T> __device__ __forceinline__ cx<T> operator+(cx<T> a, cx<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <typename T> __device__ __forceinline__ cx<T> operator-(cx<T> a, cx<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <typename T> __device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) {
    return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
template <typename T> __device__ __forceinline__ cx<T> cconj(cx<T> a) { return mk<T>(a.x, -a.y); }
template <typename T> __device__ __forceinline__ cx<T> cscale(cx<T> a, T s) { return mk<T>(a.x * s, a.y * s); }
// multiply by +i / -i
template <typename T> __device__ __forceinline__ cx<T> mul_i(cx<T> a) { return mk<T>(-a.y, a.x); }
template <typename T> __device__ __forceinline__ cx<T> mul_mi(cx<T> a) { return mk<T>(a.y, -a.x); }

// Reciprocal by v_rcp + Newton steps: ~1-2 ulp, a third of the instructions of an IEEE division
// (the column kernel needs three per spectral element per RK stage).
__device__ __forceinline__ float fast_rcp(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    return fmaf(r, fmaf(-d, r, 1.f), r);
}
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    return fma(r, fma(-d, r, 1.0), r);
}

// cos/sin(2*pi*k/16), k = 0..4, as literals (radix <= 16 only needs these).
__device__ __forceinline__ constexpr double cos16(int k) {
    return k == 0 ? 1.0 : k == 1 ? 0.92387953251128675613 : k == 2 ? 0.70710678118654752440
         : k == 3 ? 0.38268343236508977173 : 0.0;
}

// v <- v * exp(DIR * 2*pi*i * K / R) for compile-time K, R (R in {2,4,8,16}, 0 <= K < R/2)
template <int R, int K, int DIR, typename T>
__device__ __forceinline__ cx<T> rot(cx<T> v) {
    constexpr int k16 = K * (16 / R);  // angle in units of 2*pi/16, 0..7
    if constexpr (k16 == 0) {
        return v;
    } else if constexpr (k16 == 4) {
        return DIR > 0 ? mul_i(v) : mul_mi(v);
    } else if constexpr (k16 == 2) {
        constexpr T h = (T)0.70710678118654752440;
        // (c + i*s*DIR) with c = s = h
        return DIR > 0 ? mk<T>((v.x - v.y) * h, (v.x + v.y) * h) : mk<T>((v.x + v.y) * h, (v.y - v.x) * h);
    } else if constexpr (k16 == 6) {
        constexpr T h = (T)0.70710678118654752440;
        // c = -h, s = h
        return DIR > 0 ? mk<T>((-v.x - v.y) * h, (v.x - v.y) * h) : mk<T>((v.y - v.x) * h, (-v.x - v.y) * h);
    } else {
        constexpr T c = (T)(k16 < 4 ? cos16(k16) : -cos16(8 - k16));
        constexpr T s = (T)(k16 < 4 ? cos16(4 - k16) : cos16(k16 - 4)) * (T)DIR;
        return mk<T>(v.x * c - v.y * s, v.x * s + v.y * c);
    }
}

// In-register natural-order DFT of length R: v[q] <- sum_r v[r] exp(DIR*2*pi*i*r*q/R).
template <int R, int DIR, typename T>
struct Dft {
    static __device__ __forceinline__ void run(cx<T> (&v)[R]) {
        cx<T> e[R / 2], o[R / 2];
#pragma unroll
        for (int k = 0; k < R / 2; ++k) { e[k] = v[2 * k]; o[k] = v[2 * k + 1]; }
        Dft<R / 2, DIR, T>::run(e);
        Dft<R / 2, DIR, T>::run(o);
        combine<0>(v, e, o);
    }
    template <int K>
    static __device__ __forceinline__ void combine(cx<T> (&v)[R], cx<T> (&e)[R / 2], cx<T> (&o)[R / 2]) {
        if constexpr (K < R / 2) {
            cx<T> t = rot<R, K, DIR, T>(o[K]);
            v[K] = e[K] + t;
            v[K + R / 2] = e[K] - t;
            combine<K + 1>(v, e, o);
        }
    }
};
template <int DIR, typename T>
struct Dft<1, DIR, T> {
    static __device__ __forceinline__ void run(cx<T> (&)[1]) {}
};

// ---- synchronisation flavour of one transform group -------------------------
// WGSYNC: lanes of a group span several waves -> workgroup barrier.
// otherwise the group lives inside one wave: LDS is in-order per wave, so only
// the compiler has to be kept from reordering the exchange.
template <bool WGSYNC>
__device__ __forceinline__ void group_sync() {
    if constexpr (WGSYNC) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int EPT, int C, bool PAD>
__device__ __forceinline__ int lds_addr(int e, int c) {
    // PAD (a group owns the buffer alone, C == 1): the first pass stores element EPT*j + r from lane j -- a
    // stride-EPT pattern that lands every lane of a store group on the same banks.  XOR-ing the low log2(EPT)
    // index bits with the next ones spreads it over EPT different 16-byte slots, while every later access
    // (consecutive lanes <-> consecutive elements) only gets permuted inside aligned EPT-element blocks and
    // stays conflict free.  (A one-element pad per EPT block fixes the stores too, but makes every 16-byte
    // READ two-way conflicted: measured 28 % LDS conflict cycles in the row kernel.)
    if constexpr (PAD) e ^= (e / EPT) % EPT;
    return e * C + c;
}
template <int N, int EPT, int C, bool PAD>
__host__ __device__ constexpr int lds_elems() { return N * C; }

// twiddle: W[t] = exp(-2*pi*i*t/N); DIR > 0 uses the conjugate
template <int DIR, typename T>
__device__ __forceinline__ cx<T> ldtw(const cx<T>* __restrict__ tw, int idx) {
    cx<T> w = tw[idx];
    if constexpr (DIR > 0) w.y = -w.y;
    return w;
}

template <typename T, int N, int EPT, int DIR, int C, bool PAD, bool WGSYNC, int Ns>
struct Passes {
    static constexpr int REM = N / Ns;
    static constexpr int R = REM >= EPT ? EPT : REM;
    static constexpr int Q = EPT / R;
    static constexpr int G = N / EPT;
    static constexpr bool LAST = (Ns * R == N);

    static __device__ __forceinline__ void run(cx<T> (&x)[EPT], cx<T>* lds, const cx<T>* __restrict__ tw,
                                               int j, int c) {
#pragma unroll
        for (int s = 0; s < Q; ++s) {
            const int jb = j + s * G;
            const int k = jb & (Ns - 1);
            cx<T> v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = x[s + Q * r];
            if constexpr (Ns > 1) {
                // v[r] *= W^(r*base): only the log2(R) powers W^(base*2^b) are loaded; every v[r]
                // is multiplied by the ones whose bit is set in r (keeps 1 twiddle live instead of R)
                const int base = k * (N / (Ns * R));
#pragma unroll
                for (int bit = 1; bit < R; bit <<= 1) {
                    const cx<T> w = ldtw<DIR, T>(tw, base * bit);
#pragma unroll
                    for (int r = 1; r < R; ++r)
                        if (r & bit) v[r] = cmul(v[r], w);
                }
            }
            Dft<R, DIR, T>::run(v);
            if constexpr (LAST) {
#pragma unroll
                for (int r = 0; r < R; ++r) x[s + Q * r] = v[r];
            } else {
                const int o = (jb - k) * R + k;
#pragma unroll
                for (int r = 0; r < R; ++r) lds[lds_addr<EPT, C, PAD>(o + r * Ns, c)] = v[r];
            }
        }
        if constexpr (!LAST) {
            group_sync<WGSYNC>();
#pragma unroll
            for (int t = 0; t < EPT; ++t) x[t] = lds[lds_addr<EPT, C, PAD>(j + t * G, c)];
            group_sync<WGSYNC>();
            Passes<T, N, EPT, DIR, C, PAD, WGSYNC, Ns * R>::run(x, lds, tw, j, c);
        }
    }
};

// Unnormalised DFT (sign DIR) of the length-N sequence whose element j + t*G is
// x[t] in lane j of the group; result in the same distribution.  `lds` is the
// group's exchange buffer (lds_elems<N,EPT,C,PAD>() elements, shared by the C
// transforms of a tile).  The buffer is free again when the call returns.
template <typename T, int N, int EPT, int DIR, int C, bool PAD, bool WGSYNC>
__device__ __forceinline__ void tile_fft(cx<T> (&x)[EPT], cx<T>* lds, const cx<T>* __restrict__ tw, int j, int c) {
    Passes<T, N, EPT, DIR, C, PAD, WGSYNC, 1>::run(x, lds, tw, j, c);
}

}  // namespace tcfd
