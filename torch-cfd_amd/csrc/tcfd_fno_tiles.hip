// tcfd_fno_tiles.hip -- backward of the fused pointwise block of an SFNO layer for EVERY width up to 32, gfx950
//   out = act2( W2 . act1(W1 . x + b1) + b2  [+ Ws . s + bs | + s[..., -1:]] )        (fno/base.py:86-111, fno/sfno.py:607-614)
// The reference trains at width 10 (fno/train.py:293), 16 (fno/sfno_pytest.py:258-270) and 20 (its notebooks), with ReLU or GELU
// (fno/train.py:303).  The weights live in LDS as ready-made operand fragments and every channel dimension is tiled, so ONE
// kernel serves every width 4 ... 32.  (Rounds 3 / 4 had a register-resident kernel for widths <= 14 -- 71 products per 16 points
// at width 10, the transpositions as products with the identity -- and LDS-staged multi-wave kernels before that; this kernel
// measured faster at width 10 and replaced them)
//
// A wave owns 16 points at a time.  Lane l = (q, c) = (l >> 4, l & 15); v_mfma_f32_16x16x4_f32 takes A[row c][k q] and
// B[k q][col c] from lane (q, c) and leaves D[row 4q + r][col c] in register r.  A D tile of a matrix M therefore is, register by
// register, the B operand of X . M and the A operand of M^T . Y (k-step r contracts over the rows {4q + r}).
//
// Every tensor is read ONCE, as 16-byte lanes along the points: lane (q, c) holds channel slot c at points 4q .. 4q + 3, which is
// the D-tile layout of the POINT-major matrix ("OT": rows = points, columns = channels).  The channel-major orientation a
// contraction over channels needs is the TRANSPOSE of such a tile, and a 16 x 16 transposition is one ds_write_b128 + four
// ds_read_b32 through a wave-private 1.25 KB of LDS (pitch 20 floats: conflict free both ways) -- ~20 LDS cycles against the
// 4 x 32 matrix-pipe cycles of a product with the identity, and the LDS pipe is idle here otherwise.  Per group of 16 points:
//   g2^T = dout (.) act2'(.)                       registers (ReLU: sign of the saved output y; others: the saved pre-activation z2)
//   x, g2     <- transposes of x^T, g2^T            (TI + TO tiles)
//   dh^T  = g2 ^T-chain  W2                          A = g2 tile,  B = W2 fragment     TM x ksteps(CO)
//   z1^T  = x  ^T-chain  W1^T + b1                   A = x tile,   B = W1 fragment     TM x ksteps(CI)
//   h^T = act1(z1^T),  g1^T = dh^T (.) act1'(z1^T)   registers
//   g1    <- transposes of g1^T                      (TM tiles)
//   dx^T  = g1 . W1   (A = g1 tile),  ds^T = g2 . Ws (A = g2 tile)                      TI x (ksteps(CM) + ksteps(CO))
//   dW2 += g2^T ^T h^T,  dW1 += g1^T ^T x^T,  dWs += g2^T ^T s^T    (contractions over the 16 points, OT tiles on both sides)
//   db1, db2: a constant-1 channel in a free slot of the last x / s tile, so they are a column of dW1 / dWs (plain adds only
//             where no slot is free -- CI a multiple of 16 -- or the block has no skip convolution)
// The forward output (ReLU) or pre-activation (GELU, SiLU, tanh) of the block comes from the caller: nothing of z2 is recomputed,
// and the hidden layer exists in ONE orientation.  Matrix instructions per 16 points: width 10: 59 (the round-4 kernel: 71),
// 16: 88, 20: 119 + 77 small ones (TilesGeom::REM4: the products whose N or M side is the 4-channel tile as 4 x 4 x 1 blocks), 24: 244, 32: 352.  A partly filled channel tile deals its channels to the rows 4q + r with r < ceil(n / 4)
// (slot -> channel map `Ch::chan`), so the k-steps that would multiply padding are never issued.
// Weight fragments: groups of four consecutive k-steps interleaved per lane, one ds_read_b128 at an immediate offset per group.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "tcfd_fno_common.hpp"
#include "tcfd_fno_pw.hpp"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// a channel dimension of N channels cut into 16-slot tiles
template <int N>
struct Ch {
    static constexpr int T = (N + 15) / 16;
    __host__ __device__ static constexpr int n(int t) { return N - 16 * t >= 16 ? 16 : N - 16 * t; }
    __host__ __device__ static constexpr int rv(int t) { return (n(t) + 3) / 4; }          // k-steps that hold channels
    __host__ __device__ static constexpr int chan(int t, int slot) {                          // channel of a tile slot, or -1
        const int q = slot >> 2, r = slot & 3, i = q * rv(t) + r;
        return (r < rv(t) && i < n(t)) ? 16 * t + i : -1;
    }
    // a slot of the LAST tile that holds no channel (-1 when N is a multiple of 16): a constant-1 "channel" put there makes the
    // bias gradient fall out of the weight-gradient products (column `free_slot` of the last tile) instead of costing vector adds
    __host__ __device__ static constexpr int free_slot() {
        for (int s = 0; s < 16; ++s)
            if (chan(T - 1, s) < 0) return s;
        return -1;
    }
};

__device__ __forceinline__ float relu_bits(float z) {      // max(0, z) on the bit pattern: ONE v_max_i32 (see tcfd_fno.hip pw_relu)
    const int zi = __float_as_int(z);
    return __int_as_float(zi > 0 ? zi : 0);
}

// cotangent through an activation whose saved value is `v`: the OUTPUT for ReLU (y > 0 <=> z > 0), the PRE-activation otherwise
template <int ACT>
__device__ __forceinline__ float gate_saved(float g, float v) {
    if constexpr (ACT == 0) return g;
    else if constexpr (ACT == 1) return v > 0.f ? g : 0.f;
    else { float h, d; pw_act_pair<ACT>(v, h, d); return g * d; }
}
__device__ __forceinline__ float gate_saved_rt(float g, float v, int act) {
    switch (act) {
        case 0: return g;
        case 1: return gate_saved<1>(g, v);
        case 2: return gate_saved<2>(g, v);
        case 3: return gate_saved<3>(g, v);
        default: return gate_saved<4>(g, v);
    }
}

template <int CI, int CM, int CO>
struct TilesGeom {
    using IT = Ch<CI>;
    using HT = Ch<CM>;
    using OT = Ch<CO>;
    static constexpr int TI = IT::T, TM = HT::T, TO = OT::T;
    static constexpr int G_W1A = 0;                          // [t][ti]  B of z1^T:  W1[hid(t, c)][ci(ti, 4q + j)]
    static constexpr int G_W2B = G_W1A + TM * TI;            // [t][to]  B of dh^T:  W2[co(to, 4q + r)][hid(t, c)]
    static constexpr int G_W1B = G_W2B + TM * TO;            // [t][ti]  B of dx^T:  W1[hid(t, 4q + r)][ci(ti, c)]
    static constexpr int G_WSB = G_W1B + TM * TI;            // [to][ti] B of ds^T:  Ws[co(to, 4q + r)][ci(ti, c)]
    // REM4 (width 20): the last ci / co tile holds 4 channels.  Wherever such a tile is the N or M side of a product it would fill
    // a quarter of a 16 x 16 x 4 instruction (77 of the 196 per 16 points); those products run as v_mfma_f32_4x4x1_16b instead --
    // 16 independent 4 x 4 blocks, block (q, mb) = lanes 16 q + 4 mb .. + 3, 11 cycles against 32 (tests/micro/mfma_4x4.hip) --
    // with the OTHER operand taken as it is (a 4-row / 4-column slice of a 16 x 16 x 4 operand IS a block operand) and the
    // 4-channel operand replicated over mb by one lane shuffle.  Their B fragments:
    static constexpr bool REM4 = TI > 1 && TO > 1 && IT::n(TI - 1) == 4 && OT::n(TO - 1) == 4;
    static constexpr int G_W1R = G_WSB + TO * TI;            // [t]      W1[hid(t, 4q + r)][ci 16 (TI-1) + (c & 3)]   (dx^T, last ci tile)
    static constexpr int G_WSR = G_W1R + (REM4 ? TM : 0);    // [to]     Ws[co(to, 4q + r)][ci 16 (TI-1) + (c & 3)]    (ds^T, last ci tile)
    static constexpr int NG = G_WSR + (REM4 ? TO : 0);       // groups of 4 fragments = 1 KB each
    static constexpr int SCR = (TI + TO > TM ? TI + TO : TM);   // transposition tiles per wave
    static constexpr int TILE = 16 * 20;                     // floats per tile (pitch 20)
    static constexpr size_t LDS = ((size_t)NG * 256 + (size_t)4 * SCR * TILE) * sizeof(float);
    // skip_mode 2 with the t-sum in the kernel (PwBwdArgs::ds_tsum): per wave, CO channels x 80 points of dL/dz2 (pitch 84: the
    // 16-byte lane writes of neighbouring channels fall on distinct banks)
    static constexpr int TS_PITCH = 84;
    static constexpr size_t TSUM = (size_t)4 * CO * TS_PITCH * sizeof(float);    // (one row per CHANNEL: width 10 keeps 4 workgroups per CU)
    static constexpr size_t lds_of(int mode) { return LDS + (mode == 2 ? TSUM : 0); }
};

#define PWB_MFMA(A_, B_, C_) __builtin_amdgcn_mfma_f32_16x16x4f32((A_), (B_), (C_), 0, 0, 0)
#define PWB_M44(A_, B_, C_) __builtin_amdgcn_mfma_f32_4x4x1f32((A_), (B_), (C_), 0, 0, 0)

// MODE = skip_mode (0 none, 1 skip convolution, 2 broadcast last slice).  ACT >= 0: both activations are that code at compile
// time; ACT = -1: a.act1 / a.act2 at run time (one uniform switch per tile set).  WPS = waves per SIMD the registers must allow.
template <int CI, int CM, int CO, int MODE, int ACT, int WPS>
__global__ __launch_bounds__(256, WPS) void k_pwb_tiles(PwBwdArgs a) {
    using Gm = TilesGeom<CI, CM, CO>;
    using IT = typename Gm::IT;
    using HT = typename Gm::HT;
    using OT = typename Gm::OT;
    using Lay = PwBwdGeom<CI, CM, CO, true>;
    constexpr int TI = Gm::TI, TM = Gm::TM, TO = Gm::TO;
    constexpr bool REM4 = Gm::REM4;
    constexpr int TIF = REM4 ? TI - 1 : TI, TOF = REM4 ? TO - 1 : TO;   // tiles that take the N / M side of a 16 x 16 x 4 product
    constexpr int CIR = 16 * (TI - 1), COR = 16 * (TO - 1);             // first channel of the 4-channel tiles (REM4)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const ldsw = smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63, q = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- weight fragments -> LDS (once per workgroup): float (group g, lane l, k-step i) at g * 256 + 4 l + i = g * 256 + tid.
    // Every load is unconditional (index clamped, value selected afterwards): all of a thread's loads are in flight together
    // (a load behind a lane condition sits in its own exec-masked block and is waited for on the spot)
    {
        const int fl = tid >> 2, fi = tid & 3, fq = fl >> 4, fc = fl & 15;
        auto pick = [](const float* w, int idx, bool ok) { const float v = w[ok ? idx : 0]; return ok ? v : 0.f; };
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int hc = HT::chan(t, fc), hk = HT::chan(t, 4 * fq + fi);
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
                const int ik = IT::chan(ti, 4 * fq + fi), ic = IT::chan(ti, fc);
                ldsw[(Gm::G_W1A + t * TI + ti) * 256 + tid] = pick(a.w1, hc * CI + ik, hc >= 0 && ik >= 0);
                ldsw[(Gm::G_W1B + t * TI + ti) * 256 + tid] = pick(a.w1, hk * CI + ic, hk >= 0 && ic >= 0);
            }
#pragma unroll
            for (int to = 0; to < TO; ++to) {
                const int ok = OT::chan(to, 4 * fq + fi);
                ldsw[(Gm::G_W2B + t * TO + to) * 256 + tid] = pick(a.w2t, hc * CO + ok, hc >= 0 && ok >= 0);
            }
        }
        if constexpr (MODE == 1) {
#pragma unroll
            for (int to = 0; to < TO; ++to) {
                const int ok = OT::chan(to, 4 * fq + fi);
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) {
                    const int ic = IT::chan(ti, fc);
                    ldsw[(Gm::G_WSB + to * TI + ti) * 256 + tid] = pick(a.wst, ic * CO + ok, ok >= 0 && ic >= 0);
                }
                if constexpr (REM4) ldsw[(Gm::G_WSR + to) * 256 + tid] = pick(a.wst, (CIR + (fc & 3)) * CO + ok, ok >= 0);
            }
        }
        if constexpr (REM4) {
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int hk = HT::chan(t, 4 * fq + fi);
                ldsw[(Gm::G_W1R + t) * 256 + tid] = pick(a.w1, hk * CI + CIR + (fc & 3), hk >= 0);
            }
        }
    }
    __syncthreads();
    const f4* const wf = reinterpret_cast<const f4*>(ldsw) + lane;            // group g: wf[g * 64]
    float* const sc = smem + Gm::NG * 256 + wave * (Gm::SCR * Gm::TILE);       // this wave's transposition tiles
    float* const sc_w = sc + c * 20 + 4 * q;                                   // ds_write_b128: LDS[slot c][pt 4q .. 4q + 3]
    const float* const sc_r = sc + (4 * q) * 20 + c;                           // 4 x ds_read_b32: LDS[slot 4q + r][pt c]
    auto put = [&](int k, f4 v) { *reinterpret_cast<f4*>(sc_w + k * Gm::TILE) = v; };
    auto get = [&](int k) { const float* p = sc_r + k * Gm::TILE; return f4{p[0], p[20], p[40], p[60]}; };

    // ---- per-lane constants
    float b1v[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) { const int hc = HT::chan(t, c); b1v[t] = (hc >= 0 && a.b1) ? a.b1[hc] : 0.f; }
    f4 accW2[TO][TM], accW1[TM][TI], accWs[TO][TI];
    // REM4: the 4 x 4 block accumulators of the products with the 4-channel tiles (the [.][TI - 1] / [TO - 1][.] entries above stay unused)
    f4 accW1R[TM], accW2R[TM], accWsRa[TO], accWsRb[TI], accWsRc = f4{0.f, 0.f, 0.f, 0.f};
    float accB1[TM], accB2[TO];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        accB1[t] = 0.f;
#pragma unroll
        for (int to = 0; to < TO; ++to) accW2[to][t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) accW1[t][ti] = f4{0.f, 0.f, 0.f, 0.f};
        accW1R[t] = accW2R[t] = f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int to = 0; to < TO; ++to) {
        accB2[to] = 0.f;
        accWsRa[to] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) { accWs[to][ti] = f4{0.f, 0.f, 0.f, 0.f}; accWsRb[ti] = f4{0.f, 0.f, 0.f, 0.f}; }
    }

    const int gpb = (int)((a.P + 15) / 16);                 // groups of 16 points per batch element
    const int total = gpb * a.batch;                        // (the launcher checks the 31-bit range)
    const int wid = blockIdx.x * 4 + wave, wstride = gridDim.x * 4;
    const unsigned P4 = (unsigned)a.P * 4u;
    constexpr unsigned OOB = 0xffffffffu;                    // beyond every buffer: the bounds check returns 0
    constexpr int BSLOT = REM4 ? -1 : IT::free_slot();        // slot of the constant-1 channel beside x / s, or -1 (CI % 16 == 0; REM4:
                                                              // the last tile's products are 4 x 4 blocks with no spare column)
    unsigned i_off[TI], o_off[TO];                           // byte offset of this lane's channel row inside ONE sample, or OOB
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) { const int ch = IT::chan(ti, c); i_off[ti] = ch >= 0 ? (unsigned)ch * P4 : OOB; }
#pragma unroll
    for (int to = 0; to < TO; ++to) { const int ch = OT::chan(to, c); o_off[to] = ch >= 0 ? (unsigned)ch * P4 : OOB; }

    struct In {
        f4 xb[TI], sb[TI], dzb[TO], yob[TO];
    };
    // Buffer descriptors are built per SAMPLE (wave-uniform scalar arithmetic): offsets stay 32-bit whatever the batch size, and
    // a lane with nothing to read -- padding slot, point beyond P -- passes OOB and gets 0: no lane condition guards a load, the
    // next group's loads are straight-line code that stays in flight across the current group's products.
    auto load = [&](int G, In& in) {
        const int b = G / gpb;
        const unsigned pb = (unsigned)(G - b * gpb) * 16u + 4u * q;
        const bool live = pb < (unsigned)a.P;                 // P % 4 == 0: a 16-byte lane is all live or all dead
        const unsigned po = pb * 4u;
        const int ibytes = (int)((unsigned)CI * P4), obytes = (int)((unsigned)CO * P4);
        const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)b * CI * a.P, 0, ibytes, 0x00020000);
        const auto rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dout) + (size_t)b * CO * a.P, 0, obytes, 0x00020000);
        const float one = (BSLOT >= 0 && c == BSLOT && live) ? 1.f : 0.f;      // the constant-1 channel (0 in every other lane)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
            const unsigned off = (live && i_off[ti] != OOB) ? i_off[ti] + po : OOB;
            const float k1 = ti == TI - 1 ? one : 0.f;
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
            in.xb[ti] = f4{__uint_as_float(v.x) + k1, __uint_as_float(v.y) + k1, __uint_as_float(v.z) + k1, __uint_as_float(v.w) + k1};
            if constexpr (MODE == 1) {
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.s) + (size_t)b * CI * a.P, 0, ibytes, 0x00020000);
                const u4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                in.sb[ti] = f4{__uint_as_float(w.x) + k1, __uint_as_float(w.y) + k1, __uint_as_float(w.z) + k1, __uint_as_float(w.w) + k1};
            } else {
                in.sb[ti] = f4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            const unsigned off = (live && o_off[to] != OOB) ? o_off[to] + po : OOB;
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rd, off, 0, 0);
            in.dzb[to] = f4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
            if (a.out) {                                      // (kernel-argument uniform)
                const auto ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.out) + (size_t)b * CO * a.P, 0, obytes, 0x00020000);
                const u4 w = __builtin_amdgcn_raw_buffer_load_b128(ry, off, 0, 0);
                in.yob[to] = f4{__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)};
            } else {
                in.yob[to] = f4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    // One group of loads in flight per wave, issued a whole group ahead of its use (two in flight measured no faster: the kernel
    // is bound by instruction issue, not by memory latency); the two input buffers alternate, so no register is ever copied.
    float* const tsb = smem + Gm::NG * 256 + 4 * (Gm::SCR * Gm::TILE) + wave * (CO * Gm::TS_PITCH);   // (MODE 2, ds_tsum) this wave's rows
    auto body = [&](const int G, const In& cur, const int sub = 0) {
        const int b = G / gpb;
        const long pb = (long)(G - b * gpb) * 16 + 4 * q;
        const bool live = pb < a.P;

        // ---- g2^T from the saved output / pre-activation; x^T and g2^T go through the transposition tiles
        f4 g2T[TO];
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            if constexpr (ACT == 2) {                       // GELU: the gate of two values at a time on packed math
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    v2f hv, dv;
                    pw_gelu_pair_pk(v2f{cur.yob[to][r], cur.yob[to][r + 1]}, hv, dv);
                    g2T[to][r] = cur.dzb[to][r] * dv.x;
                    g2T[to][r + 1] = cur.dzb[to][r + 1] * dv.y;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    g2T[to][r] = ACT >= 0 ? gate_saved<(ACT >= 0 ? ACT : 0)>(cur.dzb[to][r], cur.yob[to][r])
                                          : gate_saved_rt(cur.dzb[to][r], cur.yob[to][r], a.act2);
            }
        }
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) put(ti, cur.xb[ti]);
#pragma unroll
        for (int to = 0; to < TO; ++to) put(TI + to, g2T[to]);
        // REM4: the 4-channel tiles as block operands -- lane (q, 4 mb + j) takes channel j of its row of 16 lanes, which the
        // point-major layout keeps in lane (q, 4 j)
        f4 xrep = f4{0.f, 0.f, 0.f, 0.f}, srep = xrep, g2rep = xrep;
        if constexpr (REM4) {
            const int src = (lane & 48) | ((lane & 3) << 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xrep[r] = __shfl(cur.xb[TI - 1][r], src, 64);
                g2rep[r] = __shfl(g2T[TO - 1][r], src, 64);
                if constexpr (MODE == 1) srep[r] = __shfl(cur.sb[TI - 1][r], src, 64);
            }
        }
        // register-only work while the tiles are on their way: the skip convolution's weight gradient, the output bias gradient
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            if constexpr (!(MODE == 1 && BSLOT >= 0)) accB2[to] += (g2T[to][0] + g2T[to][1]) + (g2T[to][2] + g2T[to][3]);
            if constexpr (MODE == 1) {
                if (to < TOF) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int ti = 0; ti < TIF; ++ti) accWs[to][ti] = PWB_MFMA(g2T[to][r], cur.sb[ti][r], accWs[to][ti]);
                        if constexpr (REM4) accWsRa[to] = PWB_M44(g2T[to][r], srep[r], accWsRa[to]);       // (co tile) x (4 ci)
                    }
                }
            }
        }
        if constexpr (REM4 && MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int ti = 0; ti < TIF; ++ti) accWsRb[ti] = PWB_M44(g2rep[r], cur.sb[ti][r], accWsRb[ti]);   // (4 co) x (ci tile)
                accWsRc = PWB_M44(g2rep[r], srep[r], accWsRc);                                                 // (4 co) x (4 ci)
            }
        }
        if constexpr (MODE == 2) {     // dL/dz2 itself is the skip gradient before its t-sum (tcfd_sum_t_into_last) ...
            if (a.ds && !a.ds_tsum) {
#pragma unroll
                for (int to = 0; to < TO; ++to) {
                    const int ch = OT::chan(to, c);
                    if (ch >= 0 && live) *reinterpret_cast<f4*>(a.ds + ((size_t)b * CO + ch) * a.P + pb) = g2T[to];
                }
            } else if (a.ds) {
                // ... or summed over t HERE: a wave takes `tsum_groups` consecutive groups (1 when T | 16, 5 when T | 80: whole
                // rows of T steps), parks their dL/dz2 in its LDS rows and, after the last of them, adds each row's T values in
                // order and stores (b, co, P / T) -- the 839 MB tensor and the pass that summed it are gone (config 5: 0.3 ms)
                const int gin = G - b * gpb;
#pragma unroll
                for (int to = 0; to < TO; ++to) {
                    const int ch = OT::chan(to, c);
                    if (ch >= 0) *reinterpret_cast<f4*>(tsb + ch * Gm::TS_PITCH + 16 * sub + 4 * q) = g2T[to];
                }
                if (sub == a.tsum_groups - 1) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int rows = 16 * a.tsum_groups / a.T;                    // rows of T steps in the super-group
                    const long row0 = (long)(gin - sub) * 16 / a.T;               // first of them inside the sample
                    const long rows_per_sample = a.P / a.T;
#pragma unroll
                    for (int to = 0; to < TO; ++to) {
                        const int ch = OT::chan(to, c);
                        for (int row = q; row < rows; row += 4) {                  // lane (q, c): slot c, rows q, q + 4, ...
                            const float* src = tsb + (ch >= 0 ? ch : 0) * Gm::TS_PITCH + row * a.T;
                            float sum = 0.f;
                            for (int t = 0; t < a.T; ++t) sum += src[t];
                            if (ch >= 0) a.ds[((size_t)b * CO + ch) * rows_per_sample + row0 + row] = sum;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        f4 xa[TI], g2[TO];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) xa[ti] = get(ti);
#pragma unroll
        for (int to = 0; to < TO; ++to) g2[to] = get(TI + to);
        __builtin_amdgcn_wave_barrier();

        // ---- dh^T[t] = sum over co of g2 (A) x W2 (B);  z1^T[t] = b1 + sum over ci of x (A) x W1^T (B).  Two hidden tiles at a
        // time so that consecutive matrix instructions never wait for each other's accumulator (40-cycle dependent latency)
        f4 dhT[TM], zT[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) { dhT[t] = f4{0.f, 0.f, 0.f, 0.f}; zT[t] = f4{b1v[t], b1v[t], b1v[t], b1v[t]}; }
#pragma unroll
        for (int t0 = 0; t0 < TM; t0 += 2) {
#pragma unroll
            for (int to = 0; to < TO; ++to) {
                const f4 w0 = wf[(Gm::G_W2B + t0 * TO + to) * 64];
                const f4 w1 = wf[(Gm::G_W2B + (t0 + 1 < TM ? t0 + 1 : t0) * TO + to) * 64];
#pragma unroll
                for (int r = 0; r < OT::rv(to); ++r) {
                    dhT[t0] = PWB_MFMA(g2[to][r], w0[r], dhT[t0]);
                    if (t0 + 1 < TM) dhT[t0 + 1] = PWB_MFMA(g2[to][r], w1[r], dhT[t0 + 1]);
                }
            }
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
                const f4 w0 = wf[(Gm::G_W1A + t0 * TI + ti) * 64];
                const f4 w1 = wf[(Gm::G_W1A + (t0 + 1 < TM ? t0 + 1 : t0) * TI + ti) * 64];
#pragma unroll
                for (int j = 0; j < IT::rv(ti); ++j) {
                    zT[t0] = PWB_MFMA(xa[ti][j], w0[j], zT[t0]);
                    if (t0 + 1 < TM) zT[t0 + 1] = PWB_MFMA(xa[ti][j], w1[j], zT[t0 + 1]);
                }
            }
        }
        // ---- h^T = act1(z1^T) (in zT),  g1^T = dh^T (.) act1'(z1^T) (in dhT)
        if constexpr (ACT == 1) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dhT[t][r] = zT[t][r] > 0.f ? dhT[t][r] : 0.f;
                    zT[t][r] = relu_bits(zT[t][r]);
                }
        } else {
#define PWB_ACT_ALL(ACT_)                                                  \
    _Pragma("unroll") for (int t = 0; t < TM; ++t)                         \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                    \
            float hv, dv;                                                  \
            pw_act_pair<(ACT_)>(zT[t][r], hv, dv);                           \
            zT[t][r] = hv;                                                 \
            dhT[t][r] *= dv;                                               \
        }
            if constexpr (ACT == 2) {
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        v2f hv, dv;
                        pw_gelu_pair_pk(v2f{zT[t][r], zT[t][r + 1]}, hv, dv);
                        zT[t][r] = hv.x; zT[t][r + 1] = hv.y;
                        dhT[t][r] *= dv.x; dhT[t][r + 1] *= dv.y;
                    }
            } else if constexpr (ACT >= 0) { PWB_ACT_ALL(ACT >= 0 ? ACT : 0) }
            else {
                switch (a.act1) {
                    case 1: PWB_ACT_ALL(1) break;
                    case 2: PWB_ACT_ALL(2) break;
                    case 3: PWB_ACT_ALL(3) break;
                    case 4: PWB_ACT_ALL(4) break;
                    default: break;
                }
            }
#undef PWB_ACT_ALL
        }
        if (a.dx) {
#pragma unroll
            for (int t = 0; t < TM; ++t) put(t, dhT[t]);
        }
        // ---- weight gradients: contractions over the 16 points (while the g1 tiles are on their way)
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            if constexpr (BSLOT < 0) accB1[t] += (dhT[t][0] + dhT[t][1]) + (dhT[t][2] + dhT[t][3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int to = 0; to < TOF; ++to) accW2[to][t] = PWB_MFMA(g2T[to][r], zT[t][r], accW2[to][t]);
#pragma unroll
                for (int ti = 0; ti < TIF; ++ti) accW1[t][ti] = PWB_MFMA(dhT[t][r], cur.xb[ti][r], accW1[t][ti]);
                if constexpr (REM4) {
                    accW2R[t] = PWB_M44(g2rep[r], zT[t][r], accW2R[t]);      // (4 co) x (hidden tile): rows = co, columns = hidden slots
                    accW1R[t] = PWB_M44(dhT[t][r], xrep[r], accW1R[t]);      // (hidden tile) x (4 ci)
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- dx^T = g1 (A, the transposed tiles) x W1 (B);  ds^T = g2 (A) x Ws (B).  Two accumulators per output tile (dependent latency)
        if (a.dx) {
            f4 dxT[TI][2];
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) dxT[ti][0] = dxT[ti][1] = f4{0.f, 0.f, 0.f, 0.f};
            f4 dxR[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const f4 g1 = get(t);
#pragma unroll
                for (int ti = 0; ti < TIF; ++ti) {
                    const f4 w = wf[(Gm::G_W1B + t * TI + ti) * 64];
#pragma unroll
                    for (int r = 0; r < HT::rv(t); ++r) dxT[ti][r & 1] = PWB_MFMA(g1[r], w[r], dxT[ti][r & 1]);
                }
                if constexpr (REM4) {                        // block (q, mb): rows = points 4 mb .., ONE hidden slot 4 q + r per instruction
                    const f4 w = wf[(Gm::G_W1R + t) * 64];
#pragma unroll
                    for (int r = 0; r < HT::rv(t); ++r) dxR[r & 1] = PWB_M44(g1[r], w[r], dxR[r & 1]);
                }
            }
#pragma unroll
            for (int ti = 0; ti < TIF; ++ti) {
                const int ch = IT::chan(ti, c);
                if (ch >= 0 && live) *reinterpret_cast<f4*>(a.dx + ((size_t)b * CI + ch) * a.P + pb) = dxT[ti][0] + dxT[ti][1];
            }
            if constexpr (REM4) {                            // the four q hold the partial sums over their hidden slots
                f4 v = dxR[0] + dxR[1];
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] += __shfl_xor(v[i], 16, 64); v[i] += __shfl_xor(v[i], 32, 64); }
                const long pr = (long)(G - b * gpb) * 16 + 4 * (c >> 2);          // lane (0, 4 mb + j): channel j at points 4 mb ..
                if (q == 0 && pr < a.P) *reinterpret_cast<f4*>(a.dx + ((size_t)b * CI + CIR + (c & 3)) * a.P + pr) = v;
            }
        }
        if constexpr (MODE == 1) {
            if (a.ds) {
                f4 dsT[TI][2];
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) dsT[ti][0] = dsT[ti][1] = f4{0.f, 0.f, 0.f, 0.f};
                f4 dsR[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int to = 0; to < TO; ++to) {
#pragma unroll
                    for (int ti = 0; ti < TIF; ++ti) {
                        const f4 w = wf[(Gm::G_WSB + to * TI + ti) * 64];
#pragma unroll
                        for (int r = 0; r < OT::rv(to); ++r) dsT[ti][r & 1] = PWB_MFMA(g2[to][r], w[r], dsT[ti][r & 1]);
                    }
                    if constexpr (REM4) {
                        const f4 w = wf[(Gm::G_WSR + to) * 64];
#pragma unroll
                        for (int r = 0; r < OT::rv(to); ++r) dsR[r & 1] = PWB_M44(g2[to][r], w[r], dsR[r & 1]);
                    }
                }
#pragma unroll
                for (int ti = 0; ti < TIF; ++ti) {
                    const int ch = IT::chan(ti, c);
                    if (ch >= 0 && live) *reinterpret_cast<f4*>(a.ds + ((size_t)b * CI + ch) * a.P + pb) = dsT[ti][0] + dsT[ti][1];
                }
                if constexpr (REM4) {
                    f4 v = dsR[0] + dsR[1];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { v[i] += __shfl_xor(v[i], 16, 64); v[i] += __shfl_xor(v[i], 32, 64); }
                    const long pr = (long)(G - b * gpb) * 16 + 4 * (c >> 2);
                    if (q == 0 && pr < a.P) *reinterpret_cast<f4*>(a.ds + ((size_t)b * CI + CIR + (c & 3)) * a.P + pr) = v;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    if (MODE == 2 && a.ds_tsum) {
        // a wave owns whole super-groups: groups ng * s + 0 .. ng - 1 for s = wid, wid + wstride, ...  (the host checked
        // P % (16 ng) == 0); the position inside the super-group is carried along, not divided out
        const int ng = a.tsum_groups, supers = total / ng;
        int g_cur = ng * wid, sub_cur = 0;                               // the group a body call works on
        auto next = [&](int& g, int& sub) { if (++sub == ng) { sub = 0; g += ng * (wstride - 1) + 1; } else ++g; };
        const int g_end = ng * supers;
        if constexpr (WPS == 2 && TM <= 4) {
            In A, B;
            load(g_cur < g_end ? g_cur : total - 1, A);
            while (g_cur < g_end) {
                int g_n = g_cur, sub_n = sub_cur;
                next(g_n, sub_n);
                load(g_n < g_end ? g_n : total - 1, B);
                __builtin_amdgcn_sched_barrier(0);
                body(g_cur, A, sub_cur);
                if (g_n >= g_end) break;
                g_cur = g_n; sub_cur = sub_n;
                next(g_n, sub_n);
                load(g_n < g_end ? g_n : total - 1, A);
                __builtin_amdgcn_sched_barrier(0);
                body(g_cur, B, sub_cur);
                g_cur = g_n; sub_cur = sub_n;
            }
        } else {
            In A;
            load(g_cur < g_end ? g_cur : total - 1, A);
            while (g_cur < g_end) {
                int g_n = g_cur, sub_n = sub_cur;
                next(g_n, sub_n);
                In B;
                load(g_n < g_end ? g_n : total - 1, B);
                __builtin_amdgcn_sched_barrier(0);
                body(g_cur, A, sub_cur);
                A = B;
                g_cur = g_n; sub_cur = sub_n;
            }
        }
    } else if constexpr (WPS == 2 && TM <= 4) {                         // narrow widths: the two input buffers alternate (no register copies)
        In A, B;
        load(wid < total ? wid : total - 1, A);
        for (int G = wid; G < total; G += 2 * wstride) {
            load(G + wstride < total ? G + wstride : total - 1, B);     // unconditional (clamped): no branch around the prefetch
            __builtin_amdgcn_sched_barrier(0);                           // ... and issued HERE, a whole group ahead of its use
            body(G, A);
            if (G + wstride >= total) break;
            load(G + 2 * wstride < total ? G + 2 * wstride : total - 1, A);
            __builtin_amdgcn_sched_barrier(0);
            body(G + wstride, B);
        }
    } else {                                                            // wide widths: one body (two would spill), the buffer is copied
        In A;
        load(wid < total ? wid : total - 1, A);
        for (int G = wid; G < total; G += wstride) {
            In B;
            load(G + wstride < total ? G + wstride : total - 1, B);
            __builtin_amdgcn_sched_barrier(0);
            body(G, A);
            A = B;
        }
    }

    // ---- this wave's row of partial sums, in the layout of tcfd_fno_pointwise_bwd:
    //      A (COP x CB) = [dW2 | db2 | dWs],  B (CM1 x CIP) = [dW1 | db1]
    float* out = a.partials + (size_t)wid * Lay::TOTAL;
    float* o1 = out + Lay::N_A;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int hc = HT::chan(t, c);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hq = HT::chan(t, 4 * q + r);
#pragma unroll
            for (int to = 0; to < TOF; ++to) {
                const int oq = OT::chan(to, 4 * q + r);
                if (oq >= 0 && hc >= 0) out[oq * Lay::CB + hc] = accW2[to][t][r];
            }
#pragma unroll
            for (int ti = 0; ti < TIF; ++ti) {
                const int ic = IT::chan(ti, c);
                if (hq >= 0 && ic >= 0) o1[hq * Lay::CIP + ic] = accW1[t][ti][r];
            }
        }
        if constexpr (REM4) {      // block accumulators: register i of lane (q, 4 mb + j), summed over the four q (their point groups)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float w2 = accW2R[t][i], w1 = accW1R[t][i];
                w2 += __shfl_xor(w2, 16, 64); w2 += __shfl_xor(w2, 32, 64);
                w1 += __shfl_xor(w1, 16, 64); w1 += __shfl_xor(w1, 32, 64);
                const int hr = HT::chan(t, 4 * (c >> 2) + i);                 // dW1: row = hidden slot 4 mb + i, column = ci CIR + j
                if (q == 0 && hc >= 0) out[(COR + i) * Lay::CB + hc] = w2;    // dW2: row = co COR + i, column = hidden slot c
                if (q == 0 && hr >= 0) o1[hr * Lay::CIP + CIR + (c & 3)] = w1;
            }
        }
        if constexpr (BSLOT >= 0) {                            // db1 = column BSLOT of the last ci tile of dW1
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hq = HT::chan(t, 4 * q + r);
                if (c == BSLOT && hq >= 0) o1[hq * Lay::CIP + CI] = accW1[t][TI - 1][r];
            }
        } else {
            float s = accB1[t];                               // lanes (q, c), q = 0 .. 3, hold partial sums of the same channel
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (q == 0 && hc >= 0) o1[hc * Lay::CIP + CI] = s;
        }
    }
#pragma unroll
    for (int to = 0; to < TO; ++to) {
        const int oc = OT::chan(to, c);
        if constexpr (MODE == 1 && BSLOT >= 0) {               // db2 (= dbs) = column BSLOT of the last ci tile of dWs
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oq = OT::chan(to, 4 * q + r);
                if (c == BSLOT && oq >= 0) out[oq * Lay::CB + CM] = accWs[to][TI - 1][r];
            }
        } else {
            float s = accB2[to];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (q == 0 && oc >= 0) out[oc * Lay::CB + CM] = s;
        }
        if constexpr (MODE == 1) {
            if (to < TOF) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int oq = OT::chan(to, 4 * q + r);
#pragma unroll
                    for (int ti = 0; ti < TIF; ++ti) {
                        const int ic = IT::chan(ti, c);
                        if (oq >= 0 && ic >= 0) out[oq * Lay::CB + CM + 1 + ic] = accWs[to][ti][r];
                    }
                }
                if constexpr (REM4) {                          // (co tile) x (4 ci): row = co slot 4 mb + i, column = ci CIR + j
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = accWsRa[to][i];
                        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                        const int orow = OT::chan(to, 4 * (c >> 2) + i);
                        if (q == 0 && orow >= 0) out[orow * Lay::CB + CM + 1 + CIR + (c & 3)] = v;
                    }
                }
            }
        }
    }
    if constexpr (REM4 && MODE == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int ti = 0; ti < TIF; ++ti) {                 // (4 co) x (ci tile): row = co COR + i, column = ci slot c
                float v = accWsRb[ti][i];
                v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                const int ic = IT::chan(ti, c);
                if (q == 0 && ic >= 0) out[(COR + i) * Lay::CB + CM + 1 + ic] = v;
            }
            float v = accWsRc[i];                              // (4 co) x (4 ci): every mb holds the same block
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (q == 0 && c < 4) out[(COR + i) * Lay::CB + CM + 1 + CIR + c] = v;
        }
    }
}
#undef PWB_MFMA
#undef PWB_M44

template <int CI, int CM, int CO, int MODE, int ACT, int WPS>
int launch_tiles(PwBwdArgs a, int batch, int max_rows, int* dims, hipStream_t st) {
    FnoProfScope prof(FNO_K_POINTWISE_BWD, st);
    using Gm = TilesGeom<CI, CM, CO>;
    a.batch = batch;
    auto kern = k_pwb_tiles<CI, CM, CO, MODE, ACT, WPS>;
    static int lds_set_dev[64] = {0};
    int dev = 0, cus = 256, per_cu = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !lds_set_dev[dev]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Gm::lds_of(MODE)));
        if (dev >= 0 && dev < 64) lds_set_dev[dev] = 1;
    }
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, Gm::lds_of(MODE)));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const long groups = ((a.P + 15) / 16) * batch;
    if (groups >= (1L << 30)) return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: %ld groups of 16 points exceed the kernel's index range", groups);
    if ((size_t)(CI > CO ? CI : CO) * (size_t)a.P * 4 >= ((size_t)1 << 32))
        return FAIL(TCFD_EINVAL, "fno_pointwise_bwd: a SAMPLE of 4 GiB and more is beyond the kernel's 32-bit buffer offsets");
    long blocks = std::min<long>({(groups + 3) / 4, (long)max_rows / 4, (long)std::max(per_cu, 1) * cus});
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), Gm::lds_of(MODE), st, a);
    HIP_TRY(hipGetLastError());
    dims[5] = (int)(blocks * 4);     // every wave writes its row, also one that found no work
    return 0;
}

// both activations ReLU, both GELU (the reference's choices, fno/train.py:303), or anything at run time
template <int CI, int CM, int CO, int MODE, int WPS>
int launch_tiles_act(const PwBwdArgs& a, int batch, int max_rows, int* dims, hipStream_t st) {
    // width 20 with ReLU fits two waves per SIMD (8 spilled registers in mode 1, none in mode 2): 5.12 -> 4.47 ms per launch; its
    // GELU / run-time variants would spill 34 ... 52 and stay at one
    constexpr int WPS_RELU = (CI == 20 || CI == 24) ? 2 : WPS;
    if (a.act1 == 1 && a.act2 == 1) return launch_tiles<CI, CM, CO, MODE, 1, WPS_RELU>(a, batch, max_rows, dims, st);
    if (a.act1 == 2 && a.act2 == 2) return launch_tiles<CI, CM, CO, MODE, 2, WPS>(a, batch, max_rows, dims, st);
    return launch_tiles<CI, CM, CO, MODE, -1, WPS>(a, batch, max_rows, dims, st);
}

template <int CI, int CM, int CO, int WPS>
int launch_tiles_mode(const PwBwdArgs& a, int batch, int max_rows, int* dims, hipStream_t st) {
    using Lay = PwBwdGeom<CI, CM, CO, true>;
    dims[0] = Lay::COP; dims[1] = Lay::CB; dims[2] = Lay::CM1; dims[3] = Lay::CIP; dims[4] = Lay::TOTAL; dims[5] = 0;
    if (!a.x) {                                               // layout query: dims[5] = rows a launch that fills the device writes
        int dev = 0, cus = 256, per_cu = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) {
            auto kern = k_pwb_tiles<CI, CM, CO, 1, 1, WPS>;   // (every MODE / ACT variant of a width has the same WPS and LDS)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TilesGeom<CI, CM, CO>::LDS);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, TilesGeom<CI, CM, CO>::LDS) == hipSuccess)
                dims[5] = std::max(per_cu, 1) * cus * 4;
        }
        (void)hipGetLastError();
        return 0;
    }
    if (a.skip_mode == 1) return launch_tiles_act<CI, CM, CO, 1, WPS>(a, batch, max_rows, dims, st);
    if (a.skip_mode == 2) return launch_tiles_act<CI, CM, CO, 2, WPS>(a, batch, max_rows, dims, st);
    return launch_tiles<CI, CM, CO, 0, -1, WPS>(a, batch, max_rows, dims, st);
}


// ------------------------------------------------------------------ the FORWARD block on the same tiles (wide layers)
//   out = act2( W2 . act1(W1 . x + b1) + b2  [+ Ws . s + bs | + s[..., -1:]] ),  optionally also the pre-activation (training)
// k_pointwise keeps the channels of a point in registers and runs 9 W^2 multiply-adds per point on the vector unit: above
// width 20 it is one point per lane (two need > 128 registers) and reaches ~55 TFLOP/s of the 157 (width 32: 7.1 ms per launch
// at config-5 size).  Here the same products run as v_mfma_f32_16x16x4_f32 on 16-point tiles with the machinery of the
// backward kernel above: x^T, s^T read once as 16-byte lanes and transposed through wave-private LDS, z1^T = x W1^T + b1,
// h^T = act1(z1^T), its transpose h, out^T = h-chain W2^T + s-chain Ws^T + biases, weights as LDS fragments.  Matrix
// instructions per 16 points: width 16: 36, 20: 75, 24: 96, 32: 144 (no padding at 16 / 32).  Exact fp32 multiply-adds in another
// order than k_pointwise's: equal to rounding, not bit for bit.
template <int CI, int CM, int CO>
struct TilesFwdGeom {
    using IT = Ch<CI>;
    using HT = Ch<CM>;
    using OT = Ch<CO>;
    static constexpr int TI = IT::T, TM = HT::T, TO = OT::T;
    static constexpr int G_W1A = 0;                          // [t][ti]  B of z1^T:   W1[hid(t, c)][ci(ti, 4q + j)]
    static constexpr int G_W2C = G_W1A + TM * TI;            // [t][to]  B of out^T:  W2[co(to, c)][hid(t, 4q + r)]
    static constexpr int G_WSC = G_W2C + TM * TO;            // [ti][to] B of out^T:  Ws[co(to, c)][ci(ti, 4q + j)]
    static constexpr int NG = G_WSC + TI * TO;
    static constexpr int SCR = (2 * TI > TM ? 2 * TI : TM);
    static constexpr int TILE = 16 * 20;
    static constexpr size_t LDS = ((size_t)NG * 256 + (size_t)4 * SCR * TILE) * sizeof(float);
};

#define PWF_MFMA(A_, B_, C_) __builtin_amdgcn_mfma_f32_16x16x4f32((A_), (B_), (C_), 0, 0, 0)
template <int CI, int CM, int CO, int MODE, int ACT, int WPS>
__global__ __launch_bounds__(256, WPS) void k_pwf_tiles(PwArgs a, int batch) {
    using Gm = TilesFwdGeom<CI, CM, CO>;
    using IT = typename Gm::IT;
    using HT = typename Gm::HT;
    using OT = typename Gm::OT;
    constexpr int TI = Gm::TI, TM = Gm::TM, TO = Gm::TO;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const ldsw = smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63, q = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const int fl = tid >> 2, fi = tid & 3, fq = fl >> 4, fc = fl & 15;
        auto pick = [](const float* w, int idx, bool ok) { const float v = w[ok ? idx : 0]; return ok ? v : 0.f; };
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int hc = HT::chan(t, fc), hk = HT::chan(t, 4 * fq + fi);
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
                const int ik = IT::chan(ti, 4 * fq + fi);
                ldsw[(Gm::G_W1A + t * TI + ti) * 256 + tid] = pick(a.w1, hc * CI + ik, hc >= 0 && ik >= 0);
            }
#pragma unroll
            for (int to = 0; to < TO; ++to) {
                const int oc = OT::chan(to, fc);
                ldsw[(Gm::G_W2C + t * TO + to) * 256 + tid] = pick(a.w2t, hk * CO + oc, hk >= 0 && oc >= 0);
            }
        }
        if constexpr (MODE == 1) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
                const int ik = IT::chan(ti, 4 * fq + fi);
#pragma unroll
                for (int to = 0; to < TO; ++to) {
                    const int oc = OT::chan(to, fc);
                    ldsw[(Gm::G_WSC + ti * TO + to) * 256 + tid] = pick(a.wst, ik * CO + oc, ik >= 0 && oc >= 0);
                }
            }
        }
    }
    __syncthreads();
    const f4* const wf = reinterpret_cast<const f4*>(ldsw) + lane;
    float* const sc = smem + Gm::NG * 256 + wave * (Gm::SCR * Gm::TILE);
    float* const sc_w = sc + c * 20 + 4 * q;
    const float* const sc_r = sc + (4 * q) * 20 + c;
    auto put = [&](int k, f4 v) { *reinterpret_cast<f4*>(sc_w + k * Gm::TILE) = v; };
    auto get = [&](int k) { const float* p = sc_r + k * Gm::TILE; return f4{p[0], p[20], p[40], p[60]}; };

    float b1v[TM], b2v[TO];
#pragma unroll
    for (int t = 0; t < TM; ++t) { const int hc = HT::chan(t, c); b1v[t] = (hc >= 0 && a.b1) ? a.b1[hc] : 0.f; }
#pragma unroll
    for (int to = 0; to < TO; ++to) {
        const int oc = OT::chan(to, c);
        b2v[to] = oc >= 0 ? ((a.b2 ? a.b2[oc] : 0.f) + ((MODE == 1 && a.bs) ? a.bs[oc] : 0.f)) : 0.f;
    }
    const int gpb = (int)((a.P + 15) / 16);
    const int total = gpb * batch;
    const int wid = blockIdx.x * 4 + wave, wstride = gridDim.x * 4;
    const unsigned P4 = (unsigned)a.P * 4u;
    constexpr unsigned OOB = 0xffffffffu;
    unsigned i_off[TI];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) { const int ch = IT::chan(ti, c); i_off[ti] = ch >= 0 ? (unsigned)ch * P4 : OOB; }
    const long sP = MODE == 2 ? (a.P / a.T) * a.sT : 0;

    struct In {
        f4 xb[TI], sb[TI];
        float sl[TO][4];
    };
    auto load = [&](int G, In& in) {
        const int b = G / gpb;
        const unsigned pb = (unsigned)(G - b * gpb) * 16u + 4u * q;
        const bool live = pb < (unsigned)a.P;
        const unsigned po = pb * 4u;
        const int ibytes = (int)((unsigned)CI * P4);
        const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)b * CI * a.P, 0, ibytes, 0x00020000);
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
            const unsigned off = (live && i_off[ti] != OOB) ? i_off[ti] + po : OOB;
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 2);        // (aux 2: non-temporal, read once)
            in.xb[ti] = f4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
            if constexpr (MODE == 1) {
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.s) + (size_t)b * CI * a.P, 0, ibytes, 0x00020000);
                const u4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 2);
                in.sb[ti] = f4{__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)};
            } else {
                in.sb[ti] = f4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int to = 0; to < TO; ++to)
#pragma unroll
            for (int r = 0; r < 4; ++r) in.sl[to][r] = 0.f;
        if constexpr (MODE == 2) {     // the skip's last time slice, broadcast over t: lane (q, c) needs it at its four points
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.s) + (size_t)b * CO * sP, 0, (int)((unsigned)(CO * sP) * 4u), 0x00020000);
#pragma unroll
            for (int to = 0; to < TO; ++to) {
                const int oc = OT::chan(to, c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned pt = pb + r;
                    const unsigned off = (live && oc >= 0) ? ((unsigned)oc * (unsigned)sP + (pt / (unsigned)a.T) * (unsigned)a.sT + (unsigned)(a.sT - 1)) * 4u : OOB;
                    in.sl[to][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
                }
            }
        }
    };

    auto body = [&](const int G, const In& cur) {
        const int b = G / gpb;
        const long pb = (long)(G - b * gpb) * 16 + 4 * q;
        const bool live = pb < a.P;
        // ---- x^T (and s^T) through the transposition tiles
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
            put(ti, cur.xb[ti]);
            if constexpr (MODE == 1) put(TI + ti, cur.sb[ti]);
        }
        __builtin_amdgcn_wave_barrier();
        f4 xa[TI], sa[MODE == 1 ? TI : 1];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
            xa[ti] = get(ti);
            if constexpr (MODE == 1) sa[ti] = get(TI + ti);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- out^T starts from the biases and the skip convolution (independent of the hidden layer)
        f4 oT[TO][2];
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            oT[to][0] = f4{b2v[to] + cur.sl[to][0], b2v[to] + cur.sl[to][1], b2v[to] + cur.sl[to][2], b2v[to] + cur.sl[to][3]};
            oT[to][1] = f4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (MODE == 1) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                for (int to = 0; to < TO; ++to) {
                    const f4 w = wf[(Gm::G_WSC + ti * TO + to) * 64];
#pragma unroll
                    for (int j = 0; j < IT::rv(ti); ++j) oT[to][j & 1] = PWF_MFMA(sa[ti][j], w[j], oT[to][j & 1]);
                }
        }
        // ---- z1^T = b1 + x-chain W1^T, two hidden tiles at a time
        f4 zT[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) zT[t] = f4{b1v[t], b1v[t], b1v[t], b1v[t]};
#pragma unroll
        for (int t0 = 0; t0 < TM; t0 += 2) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
                const f4 w0 = wf[(Gm::G_W1A + t0 * TI + ti) * 64];
                const f4 w1 = wf[(Gm::G_W1A + (t0 + 1 < TM ? t0 + 1 : t0) * TI + ti) * 64];
#pragma unroll
                for (int j = 0; j < IT::rv(ti); ++j) {
                    zT[t0] = PWF_MFMA(xa[ti][j], w0[j], zT[t0]);
                    if (t0 + 1 < TM) zT[t0 + 1] = PWF_MFMA(xa[ti][j], w1[j], zT[t0 + 1]);
                }
            }
        }
        // ---- h^T = act1(z1^T), transposed tile by tile
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            if constexpr (ACT == 2) {                      // GELU two values at a time on packed math
                const v2f g0 = gelu_pk(v2f{zT[t][0], zT[t][1]}), g1 = gelu_pk(v2f{zT[t][2], zT[t][3]});
                zT[t] = f4{g0.x, g0.y, g1.x, g1.y};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (ACT == 1) zT[t][r] = relu_bits(zT[t][r]);
                    else zT[t][r] = pw_act(zT[t][r], a.act1);
                }
            }
            put(t, zT[t]);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- out^T += h-chain W2^T
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const f4 h = get(t);
#pragma unroll
            for (int to = 0; to < TO; ++to) {
                const f4 w = wf[(Gm::G_W2C + t * TO + to) * 64];
#pragma unroll
                for (int r = 0; r < HT::rv(t); ++r) oT[to][r & 1] = PWF_MFMA(h[r], w[r], oT[to][r & 1]);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            const int oc = OT::chan(to, c);
            f4 z = oT[to][0] + oT[to][1], y;
            if constexpr (ACT == 2) {
                const v2f g0 = gelu_pk(v2f{z[0], z[1]}), g1 = gelu_pk(v2f{z[2], z[3]});
                y = f4{g0.x, g0.y, g1.x, g1.y};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (ACT == 1) y[r] = relu_bits(z[r]);
                    else y[r] = pw_act(z[r], a.act2);
                }
            }
            if (oc >= 0 && live) {
                const size_t o = ((size_t)b * CO + oc) * a.P + pb;
                if (a.pre) __builtin_nontemporal_store(z, reinterpret_cast<f4*>(a.pre + o));
                __builtin_nontemporal_store(y, reinterpret_cast<f4*>(a.out + o));
            }
        }
    };
    {
        In A, B;                                             // the two input buffers alternate: no register copies
        load(wid < total ? wid : total - 1, A);
        for (int G = wid; G < total; G += 2 * wstride) {
            load(G + wstride < total ? G + wstride : total - 1, B);
            __builtin_amdgcn_sched_barrier(0);
            body(G, A);
            if (G + wstride >= total) break;
            load(G + 2 * wstride < total ? G + 2 * wstride : total - 1, A);
            __builtin_amdgcn_sched_barrier(0);
            body(G + wstride, B);
        }
    }
}
#undef PWF_MFMA

template <int CI, int CM, int CO, int MODE, int ACT, int WPS>
int launch_fwd_tiles(const PwArgs& a, int batch, hipStream_t st) {
    FnoProfScope prof(FNO_K_POINTWISE, st);
    using Gm = TilesFwdGeom<CI, CM, CO>;
    auto kern = k_pwf_tiles<CI, CM, CO, MODE, ACT, WPS>;
    int dev = 0, cus = 256, per_cu = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Gm::LDS));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, Gm::LDS));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const long groups = ((a.P + 15) / 16) * batch;
    long blocks = std::min<long>((groups + 3) / 4, (long)std::max(per_cu, 1) * cus);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), Gm::LDS, st, a, batch);
    HIP_TRY(hipGetLastError());
    return 0;
}
template <int CI, int CM, int CO, int MODE, int WPS>
int launch_fwd_tiles_act(const PwArgs& a, int batch, hipStream_t st) {
    if (a.act1 == 1 && a.act2 == 1) return launch_fwd_tiles<CI, CM, CO, MODE, 1, WPS>(a, batch, st);
    if (a.act1 == 2 && a.act2 == 2) return launch_fwd_tiles<CI, CM, CO, MODE, 2, WPS>(a, batch, st);
    return launch_fwd_tiles<CI, CM, CO, MODE, -1, WPS>(a, batch, st);
}
template <int CI, int CM, int CO, int WPS>
int launch_fwd_tiles_mode(const PwArgs& a, int batch, hipStream_t st) {
    if (a.skip_mode == 1) return launch_fwd_tiles_act<CI, CM, CO, 1, WPS>(a, batch, st);
    if (a.skip_mode == 2) return launch_fwd_tiles_act<CI, CM, CO, 2, WPS>(a, batch, st);
    return launch_fwd_tiles<CI, CM, CO, 0, -1, WPS>(a, batch, st);
}

}  // namespace

// The widths the reference's own code uses (10: fno/train.py:293; 16: fno/sfno_pytest.py:261; 20: its notebooks) and the
// even widths around them (4 ... 16, 24, 32), with its channel expansion of 4 (fno/sfno.py:479).  The kernel needs the block's saved output /
// pre-activation unless the output activation is the identity.
int tcfd_pwb_tiles_dispatch(const PwBwdArgs& a, int batch, int ci, int cm, int co, int max_rows, int* dims, hipStream_t st,
                            int* handled) {
    *handled = 0;
    if (a.pe || a.per_sample || a.P % 4 != 0) return 0;                                      // (known at the layout query too)
    if (a.x && (max_rows < 4 || !a.w1 || (a.act2 != 0 && !a.out))) return 0;
#define PWT_CASE(CI_, CM_, CO_, WPS_)                                                          \
    if (ci == CI_ && cm == CM_ && co == CO_) {                                                  \
        *handled = 1;                                                                           \
        return launch_tiles_mode<CI_, CM_, CO_, WPS_>(a, batch, max_rows, dims, st);            \
    }
    PWT_CASE(4, 16, 4, 2) PWT_CASE(6, 24, 6, 2) PWT_CASE(8, 32, 8, 2) PWT_CASE(10, 40, 10, 2) PWT_CASE(12, 48, 12, 2)
    PWT_CASE(14, 56, 14, 2) PWT_CASE(16, 64, 16, 2) PWT_CASE(20, 80, 20, 1) PWT_CASE(24, 96, 24, 1) PWT_CASE(32, 128, 32, 1)
#undef PWT_CASE
    return 0;
}

// Forward block of the wide layers on the matrix pipe (k_pwf_tiles): widths 24 / 32 with cm = 4 ci, shared weights, P % 4 == 0,
// the (batch, C, P) layout.  Measured at the config-5 grid, per launch (profiles/r05_pw_fwd_tiles_timing.json): width 24 2.61 ms
// against 3.19 for k_pointwise, width 32 3.75 against 6.60 -- and width 16 1.76 against 1.45, width 20 2.13 against 1.66 (36 / 75
// products per 16 points leave the per-tile overhead -- two LDS transpositions, address arithmetic -- uncovered), so those two
// stay on the packed vector kernel unless TCFD_PW_FWD_TILES=2 asks for them (cross-check); =0 keeps k_pointwise everywhere.
int tcfd_pwf_tiles_dispatch(const PwArgs& a, int batch, int ci, int cm, int co, hipStream_t st, int* handled) {
    *handled = 0;
    const int mode = env_int("TCFD_PW_FWD_TILES", 1);
    if (!mode || a.pe || a.frame || !a.w1 || a.P % 4 != 0 || a.w2_bstride || a.b2_bstride) return 0;
    if (a.skip_mode == 2 && (a.T <= 0 || a.sT <= 0)) return 0;
    if ((size_t)(ci > co ? ci : co) * (size_t)a.P * 4 >= ((size_t)1 << 32)) return 0;
    if (((uintptr_t)a.x | (uintptr_t)a.out | (uintptr_t)a.pre | (uintptr_t)(a.skip_mode == 1 ? a.s : nullptr)) % 16 != 0) return 0;
#define PWF_CASE(CI_, CM_, CO_, WPS_)                                                          \
    if (ci == CI_ && cm == CM_ && co == CO_) {                                                  \
        *handled = 1;                                                                           \
        return launch_fwd_tiles_mode<CI_, CM_, CO_, WPS_>(a, batch, st);                        \
    }
    PWF_CASE(24, 96, 24, 2) PWF_CASE(32, 128, 32, 2)
    if (mode >= 2) { PWF_CASE(16, 64, 16, 2) PWF_CASE(20, 80, 20, 2) }
#undef PWF_CASE
    return 0;
}
