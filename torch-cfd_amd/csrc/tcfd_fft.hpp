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

// e + i o  and  e - i o  (the quarter-turn butterflies): generic form; the packed fp32 form below is one instruction
template <typename T> __device__ __forceinline__ cx<T> add_i(cx<T> e, cx<T> o) { return mk<T>(e.x - o.y, e.y + o.x); }
template <typename T> __device__ __forceinline__ cx<T> sub_i(cx<T> e, cx<T> o) { return mk<T>(e.x + o.y, e.y - o.x); }
// v (c + i s) for real c, s
template <typename T> __device__ __forceinline__ cx<T> crot(cx<T> v, T c, T s) { return mk<T>(v.x * c - v.y * s, v.x * s + v.y * c); }

// ---- packed fp32 (gfx950: v_pk_add / mul / fma_f32 issue at the rate of ONE fp32 or fp64 vector instruction and do two) ----
// A cx<float> is an aligned (re, im) register pair, so complex add / subtract are one instruction instead of two, a complex
// product two (v_pk_mul + v_pk_fma with the halves picked by op_sel and the sign by neg_lo) instead of four, and e +- i o one
// (the swap and the sign ride on the operand modifiers).  hipcc packs the plain vector forms by itself (with broadcast
// op_sel for `.xx` and SGPR pairs for constants) but not a swapped operand: those three are inline asm.  Non-template
// overloads, so every butterfly / twiddle / k-space expression written on cx<T> picks them up for T = float.  The fp32
// solver kernels were bound by vector issue at the same instruction count as fp64 with half the bytes (round 5: 0.56 of the
// HBM peak on the C4 shard against 0.70 in fp64).  TCFD_PK32=0 compiles the scalar forms (cross-check / A-B).
#ifndef TCFD_PK32
#define TCFD_PK32 1
#endif
#ifndef TCFD_PK32_ADD
#define TCFD_PK32_ADD 1     // (the three groups can be switched off one by one: A/B builds)
#endif
#ifndef TCFD_PK32_CMUL
#define TCFD_PK32_CMUL 1
#endif
#ifndef TCFD_PK32_ROT
#define TCFD_PK32_ROT 1
#endif
#if TCFD_PK32
typedef float pk2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2f pk(cx<float> a) { return __builtin_bit_cast(pk2f, a); }
__device__ __forceinline__ cx<float> unpk(pk2f v) { return __builtin_bit_cast(cx<float>, v); }
#if TCFD_PK32_ADD
__device__ __forceinline__ cx<float> operator+(cx<float> a, cx<float> b) { return unpk(pk(a) + pk(b)); }
__device__ __forceinline__ cx<float> operator-(cx<float> a, cx<float> b) { return unpk(pk(a) - pk(b)); }
__device__ __forceinline__ cx<float> cscale(cx<float> a, float s) { return unpk(pk(a) * s); }
#endif
#if TCFD_PK32_CMUL
__device__ __forceinline__ cx<float> cmul(cx<float> a, cx<float> b) {
    pk2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(pk(a)), "v"(pk(b)));                 // (a.x b.x, a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"                 // + (-a.y b.y, a.y b.x)
        : "+v"(t) : "v"(pk(a)), "v"(pk(b)));
    return unpk(t);
}
#endif
#if TCFD_PK32_ROT
__device__ __forceinline__ cx<float> add_i(cx<float> e, cx<float> o) {      // (e.x - o.y, e.y + o.x)
    pk2f t;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(pk(e)), "v"(pk(o)));
    return unpk(t);
}
__device__ __forceinline__ cx<float> sub_i(cx<float> e, cx<float> o) {      // (e.x + o.y, e.y - o.x)
    pk2f t;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(t) : "v"(pk(e)), "v"(pk(o)));
    return unpk(t);
}
__device__ __forceinline__ cx<float> crot(cx<float> v, float c, float s) {  // v c + (i v) s
    pk2f t = pk(v) * c;
    const pk2f ss = {s, s};
    if (__builtin_constant_p(s)) {
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "+v"(t) : "v"(pk(v)), "s"(ss));
    } else {
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "+v"(t) : "v"(pk(v)), "v"(ss));
    }
    return unpk(t);
}
#endif
#endif

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
        // (c + i*s*DIR) with c = s = h:  h (v + i v)  resp.  h (v - i v)
        return cscale(DIR > 0 ? add_i(v, v) : sub_i(v, v), h);
    } else if constexpr (k16 == 6) {
        constexpr T h = (T)0.70710678118654752440;
        // c = -h, s = h:  h (i v - v) = -h (v - i v)  resp.  -h (v + i v)
        return cscale(DIR > 0 ? sub_i(v, v) : add_i(v, v), -h);
    } else {
        constexpr T c = (T)(k16 < 4 ? cos16(k16) : -cos16(8 - k16));
        constexpr T s = (T)(k16 < 4 ? cos16(4 - k16) : cos16(k16 - 4)) * (T)DIR;
        return crot(v, c, s);
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
            if constexpr (K * (16 / R) == 4) {      // t = +- i o: the quarter turn rides on the add / subtract
                v[K] = DIR > 0 ? add_i(e[K], o[K]) : sub_i(e[K], o[K]);
                v[K + R / 2] = DIR > 0 ? sub_i(e[K], o[K]) : add_i(e[K], o[K]);
            } else {
                cx<T> t = rot<R, K, DIR, T>(o[K]);
                v[K] = e[K] + t;
                v[K + R / 2] = e[K] - t;
            }
            combine<K + 1>(v, e, o);
        }
    }
};
template <int DIR, typename T>
struct Dft<1, DIR, T> {
    static __device__ __forceinline__ void run(cx<T> (&)[1]) {}
};

// ---- radix 3 (grids n = 3 * 2^k: 96, 192, 384, 768) ----------------------------------------------------------------
// three-point DFT of (a, b, c) in place: y_p = a + w^p b + w^2p c, w = exp(DIR 2 pi i / 3) = -1/2 + DIR i sqrt(3)/2
template <int DIR, typename T>
__device__ __forceinline__ void dft3(cx<T>& a, cx<T>& b, cx<T>& c) {
    constexpr T S3 = (T)0.86602540378443864676;   // sqrt(3) / 2
    const cx<T> t1 = b + c;
    const cx<T> t2 = mk<T>(a.x - (T)0.5 * t1.x, a.y - (T)0.5 * t1.y);
    const cx<T> d = cscale(b - c, S3);
    const cx<T> r = DIR > 0 ? mul_i(d) : mul_mi(d);          // DIR i sqrt(3)/2 (b - c)
    a = a + t1;
    b = t2 + r;
    c = t2 - r;
}
template <int DIR, typename T>
struct Dft<3, DIR, T> {
    static __device__ __forceinline__ void run(cx<T> (&v)[3]) { dft3<DIR, T>(v[0], v[1], v[2]); }
};
// v <- v * exp(DIR 2 pi i M / 12) for compile-time M in [0, 12)
template <int M, int DIR, typename T>
__device__ __forceinline__ cx<T> rot12(cx<T> v) {
    constexpr int m = ((M % 12) + 12) % 12;
    if constexpr (m == 0) return v;
    else if constexpr (m == 3) return DIR > 0 ? mul_i(v) : mul_mi(v);
    else if constexpr (m == 6) return mk<T>(-v.x, -v.y);
    else if constexpr (m == 9) return DIR > 0 ? mul_mi(v) : mul_i(v);
    else {
        constexpr double C30 = 0.86602540378443864676, C60 = 0.5;
        // cos / sin of m * 30 degrees
        constexpr double cc = m == 1 ? C30 : m == 2 ? C60 : m == 4 ? -C60 : m == 5 ? -C30 : m == 7 ? -C30 : m == 8 ? -C60 : m == 10 ? C60 : C30;
        constexpr double ss = m == 1 ? C60 : m == 2 ? C30 : m == 4 ? C30 : m == 5 ? C60 : m == 7 ? -C60 : m == 8 ? -C30 : m == 10 ? -C30 : -C60;
        constexpr T c = (T)cc, sn = (T)ss * (T)DIR;
        return mk<T>(v.x * c - v.y * sn, v.x * sn + v.y * c);
    }
}
// 12 = 4 x 3 (decimation in time over the factor 3): E_s = DFT4(v[3k + s]), twiddle W12^(s q), then 3-point DFTs
// over s give X[q + 4 p], p = 0..2
template <int DIR, typename T>
struct Dft<12, DIR, T> {
    static __device__ __forceinline__ void run(cx<T> (&v)[12]) {
        cx<T> e0[4], e1[4], e2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { e0[k] = v[3 * k]; e1[k] = v[3 * k + 1]; e2[k] = v[3 * k + 2]; }
        Dft<4, DIR, T>::run(e0);
        Dft<4, DIR, T>::run(e1);
        Dft<4, DIR, T>::run(e2);
        tw<0>(v, e0, e1, e2);
    }
    template <int Q>
    static __device__ __forceinline__ void tw(cx<T> (&v)[12], cx<T> (&e0)[4], cx<T> (&e1)[4], cx<T> (&e2)[4]) {
        if constexpr (Q < 4) {
            cx<T> a = e0[Q], b = rot12<Q, DIR, T>(e1[Q]), c = rot12<2 * Q, DIR, T>(e2[Q]);
            dft3<DIR, T>(a, b, c);
            v[Q] = a;
            v[Q + 4] = b;
            v[Q + 8] = c;
            tw<Q + 1>(v, e0, e1, e2);
        }
    }
};

// ---- radix 5 (grids n = 5 * 2^k: 80, 160, 320, 640) ----------------------------------------------------------------
// cos(2 pi m / 20), m = 0..19 (sin is the same table shifted by a quarter turn)
__host__ __device__ constexpr double cos20(int m) {
    constexpr double t[20] = {1.0, 0.95105651629515357212, 0.80901699437494742410, 0.58778525229247312917, 0.30901699437494742410,
                              0.0, -0.30901699437494742410, -0.58778525229247312917, -0.80901699437494742410, -0.95105651629515357212,
                              -1.0, -0.95105651629515357212, -0.80901699437494742410, -0.58778525229247312917, -0.30901699437494742410,
                              0.0, 0.30901699437494742410, 0.58778525229247312917, 0.80901699437494742410, 0.95105651629515357212};
    return t[((m % 20) + 20) % 20];
}
// v <- v * exp(DIR 2 pi i M / 20)
template <int M, int DIR, typename T>
__device__ __forceinline__ cx<T> rot20(cx<T> v) {
    constexpr int m = ((M % 20) + 20) % 20;
    if constexpr (m == 0) return v;
    else if constexpr (m == 5) return DIR > 0 ? mul_i(v) : mul_mi(v);
    else if constexpr (m == 10) return mk<T>(-v.x, -v.y);
    else if constexpr (m == 15) return DIR > 0 ? mul_mi(v) : mul_i(v);
    else {
        constexpr T c = (T)cos20(m), sn = (T)cos20(m - 5) * (T)DIR;     // sin(x) = cos(x - pi/2)
        return mk<T>(v.x * c - v.y * sn, v.x * sn + v.y * c);
    }
}
// five-point DFT in place: y_p = sum_s w^(p s) v_s, w = exp(DIR 2 pi i / 5)
template <int DIR, typename T>
__device__ __forceinline__ void dft5(cx<T>& a, cx<T>& b, cx<T>& c, cx<T>& d, cx<T>& e) {
    constexpr T C1 = (T)0.30901699437494742410, C2 = (T)-0.80901699437494742410;     // cos 72, cos 144
    constexpr T S1 = (T)0.95105651629515357212, S2 = (T)0.58778525229247312917;      // sin 72, sin 144
    const cx<T> t1 = b + e, t2 = c + d, t3 = b - e, t4 = c - d;
    const cx<T> m1 = mk<T>(a.x + C1 * t1.x + C2 * t2.x, a.y + C1 * t1.y + C2 * t2.y);
    const cx<T> m2 = mk<T>(a.x + C2 * t1.x + C1 * t2.x, a.y + C2 * t1.y + C1 * t2.y);
    const cx<T> n1 = mk<T>(S1 * t3.x + S2 * t4.x, S1 * t3.y + S2 * t4.y);
    const cx<T> n2 = mk<T>(S2 * t3.x - S1 * t4.x, S2 * t3.y - S1 * t4.y);
    const cx<T> r1 = DIR > 0 ? mul_i(n1) : mul_mi(n1), r2 = DIR > 0 ? mul_i(n2) : mul_mi(n2);
    a = a + t1 + t2;
    b = m1 + r1;
    e = m1 - r1;
    c = m2 + r2;
    d = m2 - r2;
}
template <int DIR, typename T>
struct Dft<5, DIR, T> {
    static __device__ __forceinline__ void run(cx<T> (&v)[5]) { dft5<DIR, T>(v[0], v[1], v[2], v[3], v[4]); }
};
// 20 = 4 x 5: E_s = DFT4(v[5k + s]), twiddle W20^(s q), five-point DFTs over s give X[q + 4 p], p = 0..4
template <int DIR, typename T>
struct Dft<20, DIR, T> {
    static __device__ __forceinline__ void run(cx<T> (&v)[20]) {
        cx<T> e[5][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int s = 0; s < 5; ++s) e[s][k] = v[5 * k + s];
#pragma unroll
        for (int s = 0; s < 5; ++s) Dft<4, DIR, T>::run(e[s]);
        tw<0>(v, e);
    }
    template <int Q>
    static __device__ __forceinline__ void tw(cx<T> (&v)[20], cx<T> (&e)[5][4]) {
        if constexpr (Q < 4) {
            cx<T> a = e[0][Q], b = rot20<Q, DIR, T>(e[1][Q]), c = rot20<2 * Q, DIR, T>(e[2][Q]),
                  d = rot20<3 * Q, DIR, T>(e[3][Q]), f = rot20<4 * Q, DIR, T>(e[4][Q]);
            dft5<DIR, T>(a, b, c, d, f);
            v[Q] = a;
            v[Q + 4] = b;
            v[Q + 8] = c;
            v[Q + 12] = d;
            v[Q + 16] = f;
            tw<Q + 1>(v, e);
        }
    }
};

__host__ __device__ constexpr bool is_pow2c(int n) { return n > 0 && (n & (n - 1)) == 0; }
// Radix of the Stockham pass that starts with Ns transformed elements per sub-sequence: the whole register set (EPT) while
// it divides what is left, then the largest power of two that divides both.  For power-of-two N this is min(EPT, rest);
// for N = 3 * 2^k with EPT = 12 it gives 12, 4, 4, ... (768 = 12 * 4 * 4 * 4: three exchanges, like 1024 = 8 * 8 * 8 * 2).
template <int N, int EPT, int Ns>
__host__ __device__ constexpr int pass_radix() {
    constexpr int REM = N / Ns;
    if (REM % EPT == 0) return EPT;
    for (int r = 16; r >= 2; r >>= 1)
        if (EPT % r == 0 && REM % r == 0) return r;
    return REM;
}

// ---- synchronisation flavour of one transform group -------------------------
// SYNC = 1: lanes of a group span several waves -> workgroup barrier (__syncthreads).
// SYNC = 0: the group lives inside one wave: LDS is in-order per wave, so only
//           the compiler has to be kept from reordering the exchange.
// SYNC = 2: workgroup barrier WITHOUT the release/acquire fence of __syncthreads: only the LDS
//           queue is drained (lgkmcnt).  hipcc puts `s_waitcnt vmcnt(0)` in front of a fenced
//           barrier, which would drain LDS-DMA loads (global_load_lds) that are meant to stay in
//           flight across the exchanges of a transform (row kernel v6).
template <int SYNC>
__device__ __forceinline__ void group_sync() {
    if constexpr (SYNC == 1) {
        __syncthreads();
    } else if constexpr (SYNC == 2) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// store that does not linger as a dirty line in this XCD's L2 (non-temporal hint)
template <typename T>
__device__ __forceinline__ void store_stream(cx<T>* p, cx<T> v) {
    typedef T vec2 __attribute__((ext_vector_type(2)));
    vec2 w;
    w.x = v.x;
    w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<vec2*>(p));
}

#if TCFD_PK32
// fp32: as an asm statement.  With the packed forms above hipcc merges the two arms of `if (nt) store_stream(p, v); else *p = v;`
// into ONE plain store (the arms became identical vector stores; merging drops the hint): every one of the 1,344 non-temporal
// stores of the fp32 solver unit had disappeared, and the small cache-resident problems lost the 5 % the hint is there for.
__device__ __forceinline__ void store_stream(cx<float>* p, cx<float> v) {
    asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(p), "v"(pk(v)) : "memory");
}
#endif

// load with the same hint: data that is read exactly once by this launch
template <typename T>
__device__ __forceinline__ cx<T> load_stream(const cx<T>* p) {
    typedef T vec2 __attribute__((ext_vector_type(2)));
    const vec2 w = __builtin_nontemporal_load(reinterpret_cast<const vec2*>(p));
    return mk<T>(w.x, w.y);
}

// called once by tile_fft after the FIRST exchange barrier of a transform (every lane of the group has then
// consumed whatever the input registers were built from); default: nothing
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};

template <int EPT, int C, bool PAD>
__device__ __forceinline__ int lds_addr(int e, int c) {
    // PAD (a group owns the buffer alone, C == 1): the first pass stores element EPT*j + r from lane j -- a
    // stride-EPT pattern that lands every lane of a store group on the same banks.  XOR-ing the low log2(EPT)
    // index bits with the next ones spreads it over EPT different 16-byte slots, while every later access
    // (consecutive lanes <-> consecutive elements) only gets permuted inside aligned EPT-element blocks and
    // stays conflict free.  (A one-element pad per EPT block fixes the stores too, but makes every 16-byte
    // READ two-way conflicted: measured 28 % LDS conflict cycles in the row kernel.)
    if constexpr (PAD) {
        if constexpr ((EPT & (EPT - 1)) == 0) e ^= (e / EPT) % EPT;
        else e = e - e % EPT + (e % EPT + e / EPT) % EPT;   // EPT not a power of two (12, 20): a rotation instead of the XOR
    }
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

// TWSQ = 1: ONE twiddle load per pass and lane; the powers W^(2 base), W^(4 base) ... come from squaring it and the
// sub-butterflies s > 0 of a short last pass from a constant rotation of the s = 0 twiddle (a few ulp of extra
// round-off in the twiddles for 7 -> 3 table reads per 1024-point transform; the loads are what a compiler hoists
// out of a loop over transforms and then has to keep in registers: 36 -> 12 VGPRs in fp64).
// `jt` is the lane index used for the twiddle INDEX only (callers may pass an asm-opaque copy of it as `j` so that
// the address arithmetic is redone per transform while the twiddle loads stay loop-invariant).
template <typename T>
__device__ __forceinline__ cx<T> csquare(cx<T> w) { return mk<T>((w.x - w.y) * (w.x + w.y), (T)2 * w.x * w.y); }
#if TCFD_PK32 && TCFD_PK32_CMUL
__device__ __forceinline__ cx<float> csquare(cx<float> w) { return cmul(w, w); }   // two packed instructions instead of five
#endif

// TWSQ = 2: as 1, but the per-pass twiddles are not loaded at all: the caller read them once (load_pass_tw, forward
// sign) into `trg[P]`, P = index of the pass among those with a twiddle.  This is what keeps them in registers
// across a loop whose body contains `asm volatile(... : "memory")` statements (loads cannot be hoisted over those).
template <typename T, int N, int EPT, int Ns = 1, int P = 0>
__device__ __forceinline__ void load_pass_tw(cx<T>* trg, const cx<T>* __restrict__ tw, int jt) {
    constexpr int REM = N / Ns;
    constexpr int R = REM >= EPT ? EPT : REM;
    if constexpr (Ns > 1) trg[P] = tw[(jt & (Ns - 1)) * (N / (Ns * R))];
    if constexpr (Ns * R < N) load_pass_tw<T, N, EPT, Ns * R, (Ns > 1 ? P + 1 : P)>(trg, tw, jt);
}
template <int N, int EPT, int Ns = 1>
constexpr int pass_tw_count() {
    constexpr int REM = N / Ns;
    constexpr int R = REM >= EPT ? EPT : REM;
    if constexpr (Ns * R < N) return (Ns > 1 ? 1 : 0) + pass_tw_count<N, EPT, Ns * R>();
    else return (Ns > 1 ? 1 : 0);
}
// every pass with a twiddle needs only its s = 0 entry (Q == 1 or the constant-rotation chain applies)
template <int N, int EPT, int Ns = 1>
constexpr bool pass_tw_single() {
    constexpr int REM = N / Ns;
    constexpr int R = REM >= EPT ? EPT : REM;
    constexpr int Q = EPT / R;
    constexpr int G = N / EPT;
    constexpr bool ok = (Ns == 1) || (Q == 1) ||
                        ((Q * G <= Ns) && ((16 * G) % (Ns * R) == 0) && ((Q - 1) * ((16 * G) / (Ns * R)) < 8));
    if constexpr (Ns * R < N) return ok && pass_tw_single<N, EPT, Ns * R>();
    else return ok;
}

template <typename T, int N, int EPT, int DIR, int C, bool PAD, int WGSYNC, int Ns, typename Hook = NoHook, int TWSQ = 0,
          int P = 0>
struct Passes {
    static constexpr int REM = N / Ns;
    static constexpr int R = pass_radix<N, EPT, Ns>();
    static constexpr int Q = EPT / R;
    static constexpr bool P2 = is_pow2c(N) && is_pow2c(EPT);   // the closed-form swizzle addresses and twiddle chains below
    static_assert(P2 || TWSQ == 0, "squared / register twiddles are for power-of-two transforms");
    static constexpr int G = N / EPT;
    static constexpr bool LAST = (Ns * R == N);
    // sub-butterfly s of a pass with Q > 1: k_s = k_0 + s G when nothing wraps, i.e. the twiddle is the s = 0 one
    // turned by s * G * N / (Ns R) table steps; usable when that is a whole number of sixteenths of a turn
    static constexpr bool CHAIN = P2 && TWSQ && Q > 1 && (Q * G <= Ns) && ((16 * G) % (Ns * R) == 0) &&
                                  ((Q - 1) * ((16 * G) / (Ns * R)) < 8);
    static constexpr int STEP16 = CHAIN ? (16 * G) / (Ns * R) : 0;

    template <int S>
    static __device__ __forceinline__ cx<T> chain_tw(cx<T> w0) {
        return rot<16, (S * STEP16) % 8, DIR, T>(w0);
    }

    static __device__ __forceinline__ void run(cx<T> (&x)[EPT], cx<T>* lds, const cx<T>* __restrict__ tw,
                                               int j, int c, Hook& hook, int jt, const cx<T>* trg = nullptr) {
        if constexpr (Ns == 1 && LAST) hook();   // single-pass transform: no exchange at all
        cx<T> w_first = mk<T>((T)1, (T)0);
        if constexpr (TWSQ == 2 && Ns > 1) {
            w_first = trg[P];
            if constexpr (DIR > 0) w_first.y = -w_first.y;
        }
        run_s<0>(x, lds, tw, j, c, jt, w_first);
        if constexpr (!LAST) {
            group_sync<WGSYNC>();
            if constexpr (Ns == 1) hook();
            if constexpr (P2 && PAD && C == 1 && G % (EPT * EPT) == 0) {
                // the swizzle term ((e / EPT) % EPT) is the same for e = j + t G: one address, constant offsets
                const cx<T>* src = lds + lds_addr<EPT, 1, true>(j, 0);
#pragma unroll
                for (int t = 0; t < EPT; ++t) x[t] = src[t * G];
            } else {
#pragma unroll
                for (int t = 0; t < EPT; ++t) x[t] = lds[lds_addr<EPT, C, PAD>(j + t * G, c)];
            }
            group_sync<WGSYNC>();
            Passes<T, N, EPT, DIR, C, PAD, WGSYNC, Ns * R, Hook, TWSQ, (Ns > 1 ? P + 1 : P)>::run(x, lds, tw, j, c, hook, jt,
                                                                                                  trg);
        }
    }

    template <int S>
    static __device__ __forceinline__ void run_s(cx<T> (&x)[EPT], cx<T>* lds, const cx<T>* __restrict__ tw, int j,
                                                 int c, int jt, cx<T>& w_first) {
        if constexpr (S < Q) {
            const int jb = j + S * G;
            const int k = jb % Ns;                 // (a mask for power-of-two Ns: Ns is a compile-time constant)
            cx<T> v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = x[S + Q * r];
            if constexpr (Ns > 1) {
                // v[r] *= W^(r*base): only the log2(R) powers W^(base*2^b) are needed; every v[r]
                // is multiplied by the ones whose bit is set in r (keeps 1 twiddle live instead of R)
                const int base = ((jt + S * G) % Ns) * (N / (Ns * R));
                if constexpr (TWSQ) {
                    cx<T> w;
                    if constexpr (CHAIN && S > 0) {
                        w = chain_tw<S>(w_first);
                    } else if constexpr (TWSQ == 2) {
                        static_assert(S == 0, "register twiddles: every pass must need its s = 0 entry only");
                        w = w_first;
                    } else {
                        w = ldtw<DIR, T>(tw, base);
                        if constexpr (S == 0) w_first = w;
                    }
#pragma unroll
                    for (int bit = 1; bit < R; bit <<= 1) {
#pragma unroll
                        for (int r = 1; r < R; ++r)
                            if (r & bit) v[r] = cmul(v[r], w);
                        if (2 * bit < R) w = csquare(w);
                    }
                } else {
#pragma unroll
                    for (int bit = 1; bit < R; bit <<= 1) {
                        const cx<T> w = ldtw<DIR, T>(tw, base * bit);
#pragma unroll
                        for (int r = 1; r < R; ++r)
                            if (r & bit) v[r] = cmul(v[r], w);
                    }
                }
            }
            Dft<R, DIR, T>::run(v);
            if constexpr (LAST) {
#pragma unroll
                for (int r = 0; r < R; ++r) x[S + Q * r] = v[r];
            } else if constexpr (P2 && PAD && C == 1 && Q == 1 && (Ns == 1 || Ns == EPT || Ns % (EPT * EPT) == 0)) {
                // The swizzled slot of element e_r = o + r Ns, o = (jb - k) R + k, in closed form (R = EPT here):
                //   Ns = 1      : (e/EPT) % EPT = jb % EPT            -> jb R + (r ^ (jb % EPT))
                //   Ns = EPT    : (e/EPT) % EPT = r                   -> (jb - k) R + r EPT + (k ^ r)
                //   EPT^2 | Ns  : (e/EPT) % EPT = (k/EPT) % EPT       -> (o ^ that) + r Ns      (one address + offsets)
                // i.e. one XOR + one add per store instead of the generic divide / modulo / XOR chain per element
                // (this arithmetic is redone per transform in the kernels that cannot afford ~40 address registers).
                if constexpr (Ns == 1) {
                    const int cj = jb & (EPT - 1);
                    cx<T>* dst = lds + jb * R;
#pragma unroll
                    for (int r = 0; r < R; ++r) dst[r ^ cj] = v[r];
                } else if constexpr (Ns == EPT) {
                    cx<T>* dst = lds + (jb - k) * R;
#pragma unroll
                    for (int r = 0; r < R; ++r) dst[r * EPT + (k ^ r)] = v[r];
                } else {
                    const int o = (jb - k) * R + k;
                    cx<T>* dst = lds + (o ^ ((k / EPT) & (EPT - 1)));
#pragma unroll
                    for (int r = 0; r < R; ++r) dst[r * Ns] = v[r];
                }
            } else {
                const int o = (jb - k) * R + k;
#pragma unroll
                for (int r = 0; r < R; ++r) lds[lds_addr<EPT, C, PAD>(o + r * Ns, c)] = v[r];
            }
            run_s<S + 1>(x, lds, tw, j, c, jt, w_first);
        }
    }
};

// Unnormalised DFT (sign DIR) of the length-N sequence whose element j + t*G is
// x[t] in lane j of the group; result in the same distribution.  `lds` is the
// group's exchange buffer (lds_elems<N,EPT,C,PAD>() elements, shared by the C
// transforms of a tile).  The buffer is free again when the call returns.
template <typename T, int N, int EPT, int DIR, int C, bool PAD, int WGSYNC>
__device__ __forceinline__ void tile_fft(cx<T> (&x)[EPT], cx<T>* lds, const cx<T>* __restrict__ tw, int j, int c) {
    NoHook hook;
    Passes<T, N, EPT, DIR, C, PAD, WGSYNC, 1, NoHook, 0>::run(x, lds, tw, j, c, hook, j);
}
template <typename T, int N, int EPT, int DIR, int C, bool PAD, int WGSYNC, typename Hook>
__device__ __forceinline__ void tile_fft(cx<T> (&x)[EPT], cx<T>* lds, const cx<T>* __restrict__ tw, int j, int c,
                                         Hook& hook) {
    Passes<T, N, EPT, DIR, C, PAD, WGSYNC, 1, Hook, 0>::run(x, lds, tw, j, c, hook, j);
}
// squared-twiddle form with the per-pass twiddles handed over in registers (see Passes, TWSQ = 2)
template <typename T, int N, int EPT, int DIR, int C, bool PAD, int WGSYNC, typename Hook>
__device__ __forceinline__ void tile_fft_rt(cx<T> (&x)[EPT], cx<T>* lds, int j, int c, Hook& hook, const cx<T>* trg) {
    static_assert(pass_tw_single<N, EPT>(), "register twiddles need one table entry per pass");
    Passes<T, N, EPT, DIR, C, PAD, WGSYNC, 1, Hook, 2>::run(x, lds, nullptr, j, c, hook, j, trg);
}

}  // namespace tcfd

namespace tcfd {

// =====================================================================================================
// 1024-point transform on a 128-lane group (two waves), 8 elements per lane, with ONE LDS exchange.
//
// The Stockham passes above move all 1024 elements through LDS between every two radix passes (three exchanges,
// six workgroup barriers).  Here only the exchange that crosses the wave boundary goes through LDS; the other two
// are transpositions of REGISTER-index bits with LANE-index bits inside a wave:
//     lane bits 5, 4  <->  v_permlane32_swap / v_permlane16_swap   (one instruction per dword and register pair)
//     lane bits 3, 2  <->  v_mov_dpp row_ror:8 / row_shl:4 + row_shr:4 with a bank mask (two per dword and pair;
//                          a DPP "bank" is four ADJACENT lanes, so the mask selects on lane bits 3 and 2 -- lane
//                          bits 1, 0 would need an extra select and are left alone: they carry output digits)
// which works because nothing has to be SORTED: the inverse transforms of the row pass feed a point-wise product,
// so their output may sit in any fixed permutation `pi` of the physical row as long as the forward transform of the
// product starts from the same permutation.  Decimation in frequency, in place (index bits n9..n0 of the input,
// registers t = (t2 t1 t0), lanes l5..l0, wave w; digits of the output index k appear where the input digits were):
//
//   natural layout           t = n[9:7]            (w, l) = n[6:0]              (element j + 128 t in lane j)
//   radix 8 over t           t = k[2:0]            twiddle W_1024^(t j)
//   LDS exchange             t = n[6:4]            w l1 l0 = k[2:0],  l5 l4 l3 l2 = n[3:0] =: m
//   radix 8 over t           t = k[5:3]            twiddle W_128^(t m)
//   t2 <-> l5, t1 <-> l4     t2 t1 = n[3:2]        l5 l4 = k[5:4]
//   radix 4 over (t2 t1)     t2 t1 = k[7:6]        twiddle W_16^((t2 t1) (l3 l2))
//   t2 <-> l3, t1 <-> l2     t2 t1 = n[1:0]        l3 l2 = k[7:6]
//   radix 4 over (t2 t1)     t2 t1 = k[9:8]
//
//   pi:  k = (w l1 l0) + 8 (l5 l4 t0) + 64 (l3 l2) + 256 (t2 t1)
//
// DIR = +1 runs this (natural -> pi); DIR = -1 runs the transposed flow graph (pi -> natural): the DFT matrix is
// symmetric, so the same stages in reverse order with conjugated roots are the forward transform.
// (tests/micro/xlane_fft_model.py is the numpy model of this data flow.)
// =====================================================================================================
template <int LANEBIT>
__device__ __forceinline__ void xswap_word(unsigned& a, unsigned& b) {
    if constexpr (LANEBIT == 5) {
        const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        a = r[0];
        b = r[1];
    } else if constexpr (LANEBIT == 4) {
        const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        a = r[0];
        b = r[1];
    } else if constexpr (LANEBIT == 3) {
        // partner lane ^ 8 = row_ror:8 (0x128); receivers: banks {2,3} (lanes 8..15 of a row) resp. {0,1}
        const unsigned na = (unsigned)__builtin_amdgcn_update_dpp((int)a, (int)b, 0x128, 0xF, 0xC, false);
        const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x128, 0xF, 0x3, false);
        a = na;
        b = nb;
    } else {
        static_assert(LANEBIT == 2, "lane bits 5, 4, 3, 2 only");
        // a (register bit 0) receives in the lanes with l2 = 1 (banks 1, 3) from lane - 4: row_shr:4 (0x114);
        // b (register bit 1) receives in the lanes with l2 = 0 (banks 0, 2) from lane + 4: row_shl:4 (0x104)
        const unsigned na = (unsigned)__builtin_amdgcn_update_dpp((int)a, (int)b, 0x114, 0xF, 0xA, false);
        const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x104, 0xF, 0x5, false);
        a = na;
        b = nb;
    }
}
// a: register whose index bit is 0, b: register whose index bit is 1.  Afterwards the register bit and the lane bit
// have traded places: a holds, in the lanes whose LANEBIT is set, what b held in the partner lanes, and vice versa.
template <int LANEBIT>
__device__ __forceinline__ void xswap(cx<double>& a, cx<double>& b) {
    unsigned long long ax = __builtin_bit_cast(unsigned long long, a.x), ay = __builtin_bit_cast(unsigned long long, a.y);
    unsigned long long bx = __builtin_bit_cast(unsigned long long, b.x), by = __builtin_bit_cast(unsigned long long, b.y);
    unsigned w[8] = {(unsigned)ax, (unsigned)(ax >> 32), (unsigned)ay, (unsigned)(ay >> 32),
                     (unsigned)bx, (unsigned)(bx >> 32), (unsigned)by, (unsigned)(by >> 32)};
#pragma unroll
    for (int i = 0; i < 4; ++i) xswap_word<LANEBIT>(w[i], w[4 + i]);
    a.x = __builtin_bit_cast(double, (unsigned long long)w[0] | ((unsigned long long)w[1] << 32));
    a.y = __builtin_bit_cast(double, (unsigned long long)w[2] | ((unsigned long long)w[3] << 32));
    b.x = __builtin_bit_cast(double, (unsigned long long)w[4] | ((unsigned long long)w[5] << 32));
    b.y = __builtin_bit_cast(double, (unsigned long long)w[6] | ((unsigned long long)w[7] << 32));
}
template <int LANEBIT>
__device__ __forceinline__ void xswap(cx<float>& a, cx<float>& b) {
    unsigned w[4] = {__builtin_bit_cast(unsigned, a.x), __builtin_bit_cast(unsigned, a.y),
                     __builtin_bit_cast(unsigned, b.x), __builtin_bit_cast(unsigned, b.y)};
    xswap_word<LANEBIT>(w[0], w[2]);
    xswap_word<LANEBIT>(w[1], w[3]);
    a.x = __builtin_bit_cast(float, w[0]);
    a.y = __builtin_bit_cast(float, w[1]);
    b.x = __builtin_bit_cast(float, w[2]);
    b.y = __builtin_bit_cast(float, w[3]);
}
// trade register-index bit REGBIT (of 3) with lane bit LANEBIT for all 8 registers
template <int REGBIT, int LANEBIT, typename T>
__device__ __forceinline__ void xtranspose(cx<T> (&x)[8]) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (!(t & (1 << REGBIT))) xswap<LANEBIT>(x[t], x[t | (1 << REGBIT)]);
}

// per-lane twiddles of the three twiddled stages, forward sign (exp(-2 pi i e / 1024)); loaded once per kernel
template <typename T>
struct XlTw {
    cx<T> w1;   // W^j        j = lane index in the group (0..127)
    cx<T> w2;   // W^(8 m)    m = (l5 l4 l3 l2)
    cx<T> w3;   // W^(64 c)   c = (l3 l2)
};
template <typename T>
__device__ __forceinline__ XlTw<T> xl_load_tw(const cx<T>* __restrict__ tw, int j) {
    const int m = (j & 63) >> 2;
    XlTw<T> r;
    r.w1 = tw[j];
    r.w2 = tw[8 * m];
    r.w3 = tw[64 * (m & 3)];
    return r;
}
// x[t] *= w^(digit of t over the register bits listed in MASK), powers of w by squaring
template <int DIR, int NBITS, int SHIFT, typename T>
__device__ __forceinline__ void xl_twiddle(cx<T> (&x)[8], cx<T> w) {
    if constexpr (DIR > 0) w.y = -w.y;
#pragma unroll
    for (int b = 0; b < NBITS; ++b) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if ((t >> SHIFT) & (1 << b)) x[t] = cmul(x[t], w);
        if (b + 1 < NBITS) w = csquare(w);
    }
}
template <int DIR, typename T>
__device__ __forceinline__ void xl_radix4_pairs(cx<T> (&x)[8]) {   // radix 4 over (t2 t1), for t0 = 0 and 1
#pragma unroll
    for (int t0 = 0; t0 < 2; ++t0) {
        cx<T> v[4] = {x[t0], x[2 + t0], x[4 + t0], x[6 + t0]};
        Dft<4, DIR, T>::run(v);
        x[t0] = v[0]; x[2 + t0] = v[1]; x[4 + t0] = v[2]; x[6 + t0] = v[3];
    }
}
// LDS slot of element (q, jj) of the exchange, q = k[2:0] digit, jj = n[6:0]: rows of 128; bits 3:2 of the column are
// XORed with (q1 q0) so that the 16 lanes a 16-byte read serves together (4 values of m x 4 values of (q1 q0), the
// same 16 a) fall on 16 different 16-byte bank groups
__device__ __forceinline__ int xl_slot(int q, int jj) { return q * 128 + (jj ^ ((q & 3) << 2)); }

template <typename T, int DIR, int SYNC, typename Hook>
__device__ __forceinline__ void xl_fft1024(cx<T> (&x)[8], cx<T>* lds, const XlTw<T>& tw, int j, Hook& hook) {
    const int q_lane = ((j >> 6) << 2) | (j & 3);   // (w l1 l0)
    const int m_lane = (j & 63) >> 2;               // (l5 l4 l3 l2)
    // lane-side view of the exchange buffer: element (q_lane, 16 a + m_lane) for a = 0..7: one address + 16 a
    cx<T>* lane_side = lds + q_lane * 128 + (m_lane ^ ((q_lane & 3) << 2));
    if constexpr (DIR > 0) {
        Dft<8, DIR, T>::run(x);
        xl_twiddle<DIR, 3, 0>(x, tw.w1);
#pragma unroll
        for (int q = 0; q < 8; ++q) lds[xl_slot(q, j)] = x[q];
        group_sync<SYNC>();
        hook();
#pragma unroll
        for (int a = 0; a < 8; ++a) x[a] = lane_side[16 * a];
        group_sync<SYNC>();
        Dft<8, DIR, T>::run(x);
        xl_twiddle<DIR, 3, 0>(x, tw.w2);
        xtranspose<2, 5>(x);
        xtranspose<1, 4>(x);
        xl_radix4_pairs<DIR>(x);
        xl_twiddle<DIR, 2, 1>(x, tw.w3);
        xtranspose<2, 3>(x);
        xtranspose<1, 2>(x);
        xl_radix4_pairs<DIR>(x);
    } else {
        xl_radix4_pairs<DIR>(x);
        xtranspose<1, 2>(x);
        xtranspose<2, 3>(x);
        xl_twiddle<DIR, 2, 1>(x, tw.w3);
        xl_radix4_pairs<DIR>(x);
        xtranspose<1, 4>(x);
        xtranspose<2, 5>(x);
        xl_twiddle<DIR, 3, 0>(x, tw.w2);
        Dft<8, DIR, T>::run(x);
#pragma unroll
        for (int a = 0; a < 8; ++a) lane_side[16 * a] = x[a];
        group_sync<SYNC>();
        hook();
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = lds[xl_slot(q, j)];
        group_sync<SYNC>();
        xl_twiddle<DIR, 3, 0>(x, tw.w1);
        Dft<8, DIR, T>::run(x);
    }
}

}  // namespace tcfd

namespace tcfd {

// =====================================================================================================
// 512-point transforms of a column tile (8 columns side by side, 8 elements per lane, 512 threads = 8 waves) with
// ONE LDS exchange.  Thread tid = 8 j + c (c = column, j = 0..63): lane bits (l5 l4 l3) = j & 7, wave = j >> 3.
// Index bits n8..n0 of the input, registers t, decimation in frequency:
//
//   natural layout           t = n[8:6]      wave = n[5:3]     (l5 l4 l3) = n[2:0]      (element j + 64 t)
//   radix 8 over t           t = k[2:0]      twiddle W_512^(t j)
//   LDS exchange             t <-> wave      (lanes stay: every wave moves whole 64-lane rows, conflict free)
//   radix 8 over t           t = k[5:3]      twiddle W_64^(t m), m = (l5 l4 l3)
//   t <-> (l5 l4 l3)         v_permlane32_swap, v_permlane16_swap, DPP row_ror:8 (lane bits 5, 4, 3)
//   radix 8 over t           t = k[8:6]
//
// so register t of thread j ends up holding output element  swap3(j) + 64 t,  swap3(j) = ((j & 7) << 3) | (j >> 3):
// the caller simply stores (DIR = +1) or loads (DIR = -1, the transposed flow graph) its rows at that index -- both
// sides of the transform stay in natural order in memory.  tests/micro/xcol_fft_model.py is the numpy model.
// =====================================================================================================
template <typename T>
struct XlColTw {
    cx<T> w1;   // W_512^j
    cx<T> w2;   // W_512^(8 (j & 7))
};
template <typename T>
__device__ __forceinline__ XlColTw<T> xl_col_load_tw(const cx<T>* __restrict__ tw512, int j) {
    XlColTw<T> r;
    r.w1 = tw512[j];
    r.w2 = tw512[8 * (j & 7)];
    return r;
}
__device__ __forceinline__ int xl_col_row(int j) { return ((j & 7) << 3) | (j >> 3); }

template <typename T, int DIR>
__device__ __forceinline__ void xl_col_fft512(cx<T> (&x)[8], cx<T>* lds, const XlColTw<T>& tw, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    cx<T>* by_reg_wave = lds + wave * 64 + lane;          // slot (q = register, a = wave): + 512 q
    cx<T>* by_wave_reg = lds + wave * 512 + lane;         // slot (q = wave, a = register): + 64 a
    if constexpr (DIR > 0) {
        Dft<8, DIR, T>::run(x);
        xl_twiddle<DIR, 3, 0>(x, tw.w1);
#pragma unroll
        for (int q = 0; q < 8; ++q) by_reg_wave[512 * q] = x[q];
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 8; ++a) x[a] = by_wave_reg[64 * a];
        __syncthreads();
        Dft<8, DIR, T>::run(x);
        xl_twiddle<DIR, 3, 0>(x, tw.w2);
        xtranspose<2, 5>(x);
        xtranspose<1, 4>(x);
        xtranspose<0, 3>(x);
        Dft<8, DIR, T>::run(x);
    } else {
        Dft<8, DIR, T>::run(x);
        xtranspose<0, 3>(x);
        xtranspose<1, 4>(x);
        xtranspose<2, 5>(x);
        xl_twiddle<DIR, 3, 0>(x, tw.w2);
        Dft<8, DIR, T>::run(x);
#pragma unroll
        for (int a = 0; a < 8; ++a) by_wave_reg[64 * a] = x[a];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = by_reg_wave[512 * q];
        __syncthreads();
        xl_twiddle<DIR, 3, 0>(x, tw.w1);
        Dft<8, DIR, T>::run(x);
    }
}

}  // namespace tcfd
