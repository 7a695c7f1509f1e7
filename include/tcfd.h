/* tcfd.h -- C ABI of libtcfd_hip.so: MI355X (gfx950) kernels for the torch-cfd
 * spectral hot path.  Plain C: raw device pointers, sizes, a HIP stream handle
 * passed as void*.  No torch types.
 *
 * The reference (scaomath/torch-cfd) has no FFI: the path sits behind Python
 * nn.Module operators.  Each entry point below names the reference interface
 * it replaces (paths relative to the reference checkout).  The Python classes
 * in torch-cfd_amd/ keep those operator signatures and call these symbols via
 * ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every function returns 0 on success, a negative tcfd_status otherwise and
 *     never throws; tcfd_last_error() returns a thread-local message.
 *   - the CALLER owns every data buffer and the workspace (PyTorch's caching
 *     allocator in practice); the library owns only what a plan holds
 *     (twiddle tables, wavenumbers, linear/mask/forcing tables).
 *   - every call is asynchronous on the given stream; no internal sync.
 *   - a plan's TABLES are immutable after creation.  A plan may be used from several
 *     streams and host threads as long as every concurrent call has its own
 *     workspace (calls that share a workspace must be stream-ordered by the
 *     caller): the only mutable plan state -- the captured hipGraph of the
 *     multi-step calls on small grids, its plan-owned stream / fence events, and
 *     the two-stream TCFD_OVERLAP experiment -- is guarded by a mutex inside the
 *     plan, so such calls are ENQUEUED one after the other (they still run
 *     asynchronously).  tcfd_ns2d_profile_begin/end is a single-threaded debugging
 *     aid and is not covered.  Distinct plans are fully independent.
 *   - tuning switches (TCFD_* environment variables) are read once,
 *     at plan creation, and frozen in the plan.
 *   - complex data are interleaved (re, im) pairs of the plan's real type,
 *     half spectra are (batch, n, m) row-major with m = n/2 + 1.  Solver plans:
 *     n = 2^k (8 ... 2048), 3 * 2^k (96 ... 1536) or 5 * 2^k (80 ... 1280); every
 *     other even n is served above this boundary (dense device transforms,
 *     torch-cfd_amd/mixed_radix.py).  FNO plans: X, Y in [4, 1024] (FFT kernels for
 *     2^k, 3 * 2^k, 5 * 2^k; pruned direct DFTs otherwise, tcfd_fno_plan_supports).
 */
#ifndef TCFD_H
#define TCFD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    TCFD_OK = 0,
    TCFD_EINVAL = -1,    /* bad argument (size not a supported power of two, null pointer ...) */
    TCFD_ENOMEM = -2,    /* device allocation for plan tables failed */
    TCFD_EHIP = -3,      /* a HIP runtime call / kernel launch failed */
    TCFD_EWORKSPACE = -4 /* workspace smaller than tcfd_ns2d_workspace_bytes() */
} tcfd_status;

typedef enum { TCFD_C64 = 0, TCFD_C128 = 1 } tcfd_dtype;

typedef struct tcfd_ns2d_plan tcfd_ns2d_plan;
typedef struct tcfd_fno_plan tcfd_fno_plan;

#define TCFD_ABI_VERSION 7   /* what tcfd_version() of a library built from THIS header returns */

#ifndef TCFD_H_TYPES_ONLY   /* (the library's second compilation unit wants the types without the prototypes) */

const char* tcfd_last_error(void);
/* ABI revision of the library that was loaded.  It changes whenever an entry point changes its argument list or the
 * meaning of an argument (round 3 turned the float scalars of the tcfd_fno_* calls into doubles and gave tcfd_fno_contract
 * a dtype: revision 1 -> 4; round 5: 6, tcfd_fno_pointwise_pre / _bwd_saved / _profile_*, tcfd_fno_spectral_conv_pointwise
 * removed; 7: tcfd_sobolev_loss_backward, tcfd_fno_forward_trunc_kt / _inverse_trunc_kt added -- a host written against 7 needs them).  A host compares it with the TCFD_ABI_VERSION it was written against BEFORE the
 * first call: a stale prebuilt library would otherwise be called with the wrong argument layout and return garbage
 * (torch-cfd_amd/_lib.py::load does). */
int tcfd_version(void);

/* ---- plan: constant tables of one NavierStokes2DSpectral instance -----------
 * Replaces NavierStokes2DSpectral._initialize (torch_cfd/equations.py:394-403)
 * plus the per-call re-evaluation of the forcing (equations.py:429-437) and of
 * the patched Laplacian (torch_cfd/spectral.py:41-46).
 * Host inputs (double, converted to the plan's real type on upload):
 *   kx[n], ky[m]        ordinal wavenumbers of the rows / columns of the half
 *                        spectrum (torch_cfd/grids.py:197-201; kx[i] = kx2d[i,0])
 *   linear_term[n*m]    nu*laplace - drag           (equations.py:401)
 *   mask[n*m]           2/3-rule filter, or all ones when smooth=False (:424-425)
 *   forcing_hat[2*n*m]  complex forcing term added to F, or NULL (:429-437)
 */
int tcfd_ns2d_plan_create(tcfd_ns2d_plan** plan, int n, int dtype, const double* kx, const double* ky,
                          const double* linear_term, const double* mask, const double* forcing_hat);
void tcfd_ns2d_plan_destroy(tcfd_ns2d_plan* plan);

/* Which exact compact forms the plan found in its tables: separable mask / linear term (1-D vectors
 * instead of (n, m) tables), sparse forcing (CSC list), and keep_cols > 0 when F and the RK
 * accumulator are identically zero outside the 2/3-rule mask so those entries are never stored. */
int tcfd_ns2d_plan_info(const tcfd_ns2d_plan* plan, int* separable, int* sparse_forcing, int* keep_cols);

/* Which kernel variants the plan launches: split = 1 when the column transform is cut radix-2 across the
 * row pass; rows_kernel = 6 (LDS-DMA staged row pass), 5 (register staged) or 4 (round-1 kernel). */
int tcfd_ns2d_plan_variant(const tcfd_ns2d_plan* plan, int* split, int* rows_kernel);

/* How a batched call on this plan is cut into chunks: `*fields_per_chunk` = batch elements per chunk for a call with
 * `batch` fields (= batch when the call is not chunked), `*cache_bytes` = size of the device's memory-side last-level
 * cache (Infinity Cache) the chunk is sized for, `*cache_source` = 0 built-in default (256 MB), 1 read from the KFD
 * topology of the device at plan creation, 2 TCFD_CACHE_MB.  Any output pointer may be NULL. */
int tcfd_ns2d_plan_chunking(const tcfd_ns2d_plan* plan, long batch, long* fields_per_chunk, size_t* cache_bytes,
                            int* cache_source);

/* Test hook: the cross-lane 1024-point transform of the row pass on its own (complex128, `count` sequences of
 * 1024 elements, one 128-lane group each).  dir = +1: natural-order input -> output in the kernel's internal
 * permutation (register t of lane j at out[1024 s + 128 t + j]); dir = -1: the reverse.  Needs a 1024^2
 * complex128 plan (for its twiddle table). */
int tcfd_debug_xl_fft1024(const tcfd_ns2d_plan* plan, const void* in, void* out, int count, int dir, void* stream);

/* Bytes of caller-owned scratch needed by the calls below for `batch` fields.  Batched calls run in cache-sized chunks
 * through one chunk-sized set of fields, so this stops growing with the batch at one chunk (plus one field of the whole
 * batch for tcfd_irfft2): 0.55 GB at 1024^2 x 64 complex128. */
size_t tcfd_ns2d_workspace_bytes(const tcfd_ns2d_plan* plan, long batch);

/* ---- RK4-CN time stepping ----------------------------------------------------
 * Replaces NavierStokes2DSpectral.forward (equations.py:452-463) driving
 * RK4CrankNicolsonStepper.forward (equations.py:328-358): `steps` steps of
 *     h <- F(u) + beta[k] h ;  u <- (u + gdt[k] h + mu[k] L u) / (1 - mu[k] L)
 * over nstages stages, with gdt[k] = gamma_k*dt and mu[k] = dt/2*(alpha_{k+1}-alpha_k)
 * precomputed by the caller in the precision the reference would use.
 *   w_in   (batch, n, m) complex, read only        w_out  (batch, n, m) complex
 *   dwdt   (batch, n, m) complex or NULL: (w_out - w_in) * inv_total_dt
 * w_out may alias w_in only when dwdt is NULL.
 */
int tcfd_ns2d_step(const tcfd_ns2d_plan* plan, const void* w_in, void* w_out, void* dwdt, long batch,
                   int nstages, const double* beta, const double* gdt, const double* mu, int steps,
                   double inv_total_dt, void* workspace, size_t workspace_bytes, void* stream);

/* The same fused step for the reference's other IMEX schedules (IMEXStepper._imex / _rk2_crank_nicolson,
 * torch_cfd/equations.py:174-228).  Stage k:
 *     h <- fa[k] F(u) + beta[k] h ;   u <- (base + gdt[k] h + mu_num[k] L base) / (1 - mu_den[k] L)
 * with base = the current state, or, where base0[k] != 0, the state the STEP started from.
 *   forward-backward Euler / IMEX-CN (order 1, 1.5):  1 stage,  fa 1, beta 0, gdt dt, mu_num (1-alpha) dt, mu_den alpha dt
 *   RK2-CN (order 2): 2 stages, fa {1, alpha}, beta {0, 1-alpha}, gdt dt, mu_num = mu_den = beta_cn dt, base0 {0, 1}
 * fa, mu_den, base0 may be NULL (1, mu_num, all 0: tcfd_ns2d_step).  Everything else as tcfd_ns2d_step. */
int tcfd_ns2d_step_imex(const tcfd_ns2d_plan* plan, const void* w_in, void* w_out, void* dwdt, long batch, int nstages,
                        const double* fa, const double* beta, const double* gdt, const double* mu_num,
                        const double* mu_den, const int* base0, int steps, double inv_total_dt, void* workspace,
                        size_t workspace_bytes, void* stream);

/* F(w): NavierStokes2DSpectral.explicit_terms (equations.py:413-441). */
int tcfd_ns2d_explicit_terms(const tcfd_ns2d_plan* plan, const void* w, void* out, long batch,
                             void* workspace, size_t workspace_bytes, void* stream);

/* Vector-Jacobian product of the explicit terms (what torch autograd builds from the ~40 ops of
 * NavierStokes2DSpectral._explicit_terms, torch_cfd/equations.py:413-438, when a state requires grad).  With
 * F(w) = mask . rfft2(-(dxw u + dyw v)) and a cotangent g of F:  the caller passes gm = mask * g / c  (c = 1 on the DC
 * and Nyquist columns of the half spectrum, 2 elsewhere: the adjoint of the r2c transform) and receives the four half
 * spectra X_f (xout: (4, batch, n, m), f = u^, v^, dxw^, dyw^ as in the forward's planes); the cotangent of w is
 *     wbar = -(c / n^2) * sum_f conj(a_f) X_f ,   a_0 = -2 pi i ky / lap, a_1 = 2 pi i kx / lap, a_2 = 2 pi i kx, a_3 = 2 pi i ky
 * (lap = -4 pi^2 |k|^2 with lap(0,0) := 1).  Workspace as tcfd_ns2d_explicit_terms. */
int tcfd_ns2d_explicit_terms_vjp(const tcfd_ns2d_plan* plan, const void* w, const void* gm, void* xout, long batch,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* wbar (batch, plane) = sum_f post_f (.) X_f: the closing sum of the vector-Jacobian product above (X = xout, post (4, plane)
 * complex tables -(c / n^2) conj(a_f) built by the caller) in one pass. */
int tcfd_ns2d_vjp_combine(const void* X, const void* post, void* out, long batch, long plane, int dtype, void* stream);

/* Stage bookkeeping of the differentiable step with constant coefficients (what autograd derives for the low-storage RK /
 * Crank-Nicolson stage of torch_cfd/equations.py:139-160, 355-357 when only the STATE requires grad), one launch each way:
 *     h = fa f + beta h_prev ,   u = (b + gdt h + mu L b) / (1 - mud L)          coef = {fa, beta, gdt, mu, mud}
 *     G = g_h + gdt r g_u ,   g_f = fa G ,  g_hprev = beta G ,  g_b = (1 + mu L) r g_u ,   r = 1 / (1 - mud L)
 * f, h_prev, b, h, u and the cotangents: (batch, plane) complex of dtype (TCFD_C64 / TCFD_C128); lin: the (plane) real linear
 * term in the matching precision.  h_prev / g_h / g_hprev may be NULL (first stage: no previous h; last stage: nothing reads h). */
int tcfd_ns2d_stage_update(const void* f, const void* h_prev, const void* b, const void* lin, const double* coef, void* h,
                           void* u, long batch, long plane, int dtype, void* stream);
int tcfd_ns2d_stage_update_vjp(const void* g_u, const void* g_h, const void* lin, const double* coef, void* g_f,
                               void* g_hprev, void* g_b, long batch, long plane, int dtype, void* stream);

/* psi = -w/lap and residual = w_t - F(w) - L w in one call: the record step of
 * get_trajectory_imex (fno/data_gen/solvers.py:245-247, torch_cfd/spectral.py:113,
 * equations.py:405-411).  Either output may be NULL. */
int tcfd_ns2d_stream_residual(const tcfd_ns2d_plan* plan, const void* w, const void* wt, void* psi,
                              void* residual, long batch, void* workspace, size_t workspace_bytes,
                              void* stream);

/* (u_hat, v_hat, psi_hat) of vorticity_to_velocity (torch_cfd/spectral.py:87-115);
 * any output may be NULL. */
int tcfd_ns2d_velocity(const tcfd_ns2d_plan* plan, const void* w, void* u_hat, void* v_hat, void* psi,
                       long batch, void* stream);

/* ---- plain transforms with torch.fft semantics (tests, IC / output side) -------
 * rfft2:  real (batch, n, n) -> complex (batch, n, m), unnormalised.
 * irfft2: complex (batch, n, m) -> real (batch, n, n), 1/n^2, imaginary parts of
 *         the DC / Nyquist columns ignored AFTER the column transform (the c2r
 *         semantics the reference relies on, SURVEY note N2). */
int tcfd_rfft2(const tcfd_ns2d_plan* plan, const void* x_real, void* out_hat, long batch, void* stream);
int tcfd_irfft2(const tcfd_ns2d_plan* plan, const void* x_hat, void* out_real, long batch, void* workspace,
                size_t workspace_bytes, void* stream);

/* irfft2 followed by F.interpolate(size = (n / factor, n / factor), mode = "bilinear") of the data-generation
 * drivers (fno/data_gen/data_gen_McWilliams2d.py:158-163) in one pass over the spectrum: complex (batch, n, m) ->
 * real (batch, n / factor, n / factor); bit-identical to the two calls at factor 2, equal to rounding beyond (the two
 * rows a pixel needs share one complex transform here and ride through two different ones in tcfd_irfft2; the rows the
 * subsample never looks at are not transformed).  factor: a power of two >= 2 that divides n
 * (at most the lanes of one row transform, 64 at most; TCFD_EINVAL otherwise -- the caller then runs the two calls).
 * Same workspace as tcfd_irfft2. */
int tcfd_irfft2_subsample(const tcfd_ns2d_plan* plan, const void* x_hat, void* out_real, long batch, int factor,
                          void* workspace, size_t workspace_bytes, void* stream);
/* The largest factor tcfd_irfft2_subsample takes for this plan (a power of two: 4 at n = 80, 8 at n = 64 / 96 / 160, 16 at
 * n = 128 / 256 / 320, ... 64 at most): a host asks before choosing between the one-pass and the two-call path. */
int tcfd_irfft2_subsample_max_factor(const tcfd_ns2d_plan* plan);

/* ---- FNO / SFNO spectral convolution (fp32 and fp64) -------------------------------
 * Replaces SpectralConv.forward (fno/base.py:229-237) with SpectralConvS.spectral_conv
 * (fno/sfno.py:364-391), SpectralConvT.forward (fno/sfno.py:433-457) and
 * SpectralConv3d.forward (fno/fno3d.py:86-116):
 *     out = irfftn( contract( rfftn(pad_t(v)) ), s = (X, Y, T_out) )[..., -t_keep:]
 * with pruned transforms (only the 2mx x 2my x mt kept modes are produced / consumed).
 *   plan: grid (X, Y), each in [4, 1024]: 2^k (8 ... 1024), 3 * 2^k (96 ... 768) and 5 * 2^k (80 ... 640) run the FFT
 *         kernels, every other size pruned direct DFTs on the kept rows (same results, 3-5 x the time per grid point;
 *         csrc/tcfd_fno_dft.hpp); T_in input steps, t_pad zeros prepended
 *         (SpectralConvT temporal_padding), T_out = irfftn length in t, modes (mx, my, mt).
 *   precision: a plan is fp32 (TCFD_C64: real data fp32, spectra / weights complex64 -- tcfd_fno_plan_create and
 *         _resample) or fp64 (TCFD_C128, tcfd_fno_plan_create_dtype: FNOBase.double(), fno/base.py:342-349); every
 *         array of a call has the plan's precision.  Scalars (delta, scales) are double in both cases.
 *   v        (batch, cin, X, Y, T_in) real          out (batch, cout, X, Y, t_keep) real
 *   weights  4 pointers, each (cin, cout, mx, my, mt) interleaved complex, block order
 *            ix + 2*iy = [lo-x lo-y, hi-x lo-y, lo-x hi-y, hi-x hi-y]  (sfno.py:374-386; the
 *            complex weights1..4 of fno3d.py:36-79 have the same memory layout)
 *   bias     NULL or 4 pointers (mx, my, mt) complex, added as delta * bias (sfno.py:388-390)
 *   fwd_scale / inv_scale   norm="backward": 1 and 1/(X*Y*T_out)
 *   use_mfma 1: per-mode products on v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64; 0: plain VALU kernel */
int tcfd_fno_plan_create(tcfd_fno_plan** plan, int X, int Y, int T_in, int t_pad, int T_out, int mx, int my, int mt);
/* Plan of an INVERSE transform onto a grid (X, Y) of a truncated spectrum taken from a grid (Xs, Ys): what
 * SpectralConv.forward(v, out_mesh_size) does through irfftn(s = out_mesh_size) (fno/base.py:229-237) -- torch pads or
 * trims the spectrum array at its end, so the high-frequency block keeps its array indices [Xs - mx, Xs) x [Ys - my, Ys).
 * Only tcfd_fno_inverse_trunc may be called with such a plan. */
int tcfd_fno_plan_create_resample(tcfd_fno_plan** plan, int X, int Y, int T_in, int t_pad, int T_out, int mx, int my, int mt,
                                  int Xs, int Ys);
/* 1 when the plan's kernels take a call that keeps `t_keep` output steps, 0 when not: FFT lengths always do; the pruned direct-DFT
 * kernels of the other lengths hold one (Y x time) slab in 150 KB of LDS and know at most 16 time modes.  A host asks before it
 * prefers the library to its own fallback. */
int tcfd_fno_plan_supports(const tcfd_fno_plan* plan, int t_keep);
/* The general constructor: as _resample (Xs = X, Ys = Y for an ordinary plan) with the precision, TCFD_C64 or TCFD_C128. */
int tcfd_fno_plan_create_dtype(tcfd_fno_plan** plan, int X, int Y, int T_in, int t_pad, int T_out, int mx, int my, int mt,
                               int Xs, int Ys, int dtype);
void tcfd_fno_plan_destroy(tcfd_fno_plan* plan);
size_t tcfd_fno_workspace_bytes(const tcfd_fno_plan* plan, int batch, int cin, int cout);
int tcfd_fno_spectral_conv(const tcfd_fno_plan* plan, const void* v, const void* const* weights,
                           const void* const* bias, double delta, void* out, int batch, int cin, int cout,
                           int t_keep, double fwd_scale, double inv_scale, int use_mfma, void* workspace,
                           size_t workspace_bytes, void* stream);
/* The two halves of the convolution on their own, for layers that post-process the spectrum between contraction
 * and inverse transform (SpectralConvT(postprocess=HelmholtzProjection), fno/sfno.py:449):
 *   forward_trunc: v (batch, c, X, Y, T_in) real -> vh (batch, c, 2mx, 2my, mt) complex, kept modes only
 *   inverse_trunc: vh -> out (batch, c, X, Y, t_keep) real.  Workspace: tcfd_fno_workspace_bytes(plan, batch, c, c). */
int tcfd_fno_forward_trunc(const tcfd_fno_plan* plan, const void* v, void* vh, int batch, int c, double fwd_scale,
                           void* workspace, size_t workspace_bytes, void* stream);
int tcfd_fno_inverse_trunc(const tcfd_fno_plan* plan, const void* vh, void* out, int batch, int c, int t_keep,
                           double inv_scale, void* workspace, size_t workspace_bytes, void* stream);
/* out = acc + inverse_trunc(vh): acc has the shape of out and may be out itself (NULL: plain inverse_trunc).  Training: the
 * gradient of a layer input that also feeds the skip path is finished by the transform that produces its spectral part. */
int tcfd_fno_inverse_trunc_acc(const tcfd_fno_plan* plan, const void* vh, void* out, const void* acc, int batch, int c,
                               int t_keep, double inv_scale, void* workspace, size_t workspace_bytes, void* stream);
/* tcfd_fno_forward_trunc / tcfd_fno_inverse_trunc_acc / _last with ONE real factor per kept time mode (kt_scale: plan mt values
 * of the data's real precision, device memory; NULL = none) folded into the transform's t-DFT table.  The adjoints of the r2c /
 * c2r time transforms weigh the interior time modes by 2 resp. 1 / 2 (torch's convention for the gradient of a half spectrum):
 * with this argument a layer's backward needs no elementwise pass over its spectra.  acc / last: as in
 * tcfd_fno_inverse_trunc_acc / tcfd_fno_inverse_trunc_last, at most one of them non-NULL. */
int tcfd_fno_forward_trunc_kt(const tcfd_fno_plan* plan, const void* v, void* vh, int batch, int c, double fwd_scale,
                              const void* kt_scale, void* ws, size_t ws_bytes, void* stream);
int tcfd_fno_inverse_trunc_kt(const tcfd_fno_plan* plan, const void* vh, void* out, const void* acc, const void* last, int batch,
                              int c, int t_keep, double inv_scale, const void* kt_scale, void* ws, size_t ws_bytes, void* stream);
/* The contraction alone on truncated spectra (batch, c, 2mx, 2my, mt), dtype TCFD_C64 / TCFD_C128 (tests). */
int tcfd_fno_contract(const void* vin, const void* const* weights, const void* const* bias, double delta,
                      void* vout, int batch, int cin, int cout, int mx, int my, int mt, int use_mfma, int dtype,
                      void* stream);
/* Gradient w.r.t. the spectrum: gv (batch, cin_fwd, ..) = sum_o conj(W[i][o]) gh (batch, cout_fwd, ..), reading the FORWARD
 * blocks (cin_fwd, cout_fwd, mx, my, mt) in place (no conjugate-transposed copy). */
int tcfd_fno_contract_adjoint(const void* gh, const void* const* weights, void* gv, int batch, int cout_fwd, int cin_fwd,
                              int mx, int my, int mt, int use_mfma, int dtype, void* stream);
/* Its weight / bias gradient (training): gw[k] (cin, cout, mx, my, mt) = sum_b conj(vh) gh over corner k = ix + 2 iy,
 * gb[k] (mx, my, mt) = delta sum_{b, o} gh; entries (or the whole array) may be NULL.  What autograd derives for the einsum of fno/sfno.py:376-389. */
int tcfd_fno_contract_wgrad(const void* vh, const void* gh, void* const* gw, void* const* gb, double delta, int batch,
                            int cin, int cout, int mx, int my, int mt, int dtype, void* stream);

/* Fused pointwise block of an SFNO layer (fp32, channel-major (batch, C, P) tensors, P = X*Y*T):
 *     out = act2( W2.act1(W1.x + b1) + b2  [+ Ws.skip + bs  |  + skip[..., -1:]] )
 * = PointwiseFFN (fno/base.py:86-111) + the 1x1x1 skip convolution + sum + activation of one layer
 * (fno/sfno.py:607-614), the lifting tail act(v[..., -1:] + mlp(.)) (fno/sfno.py:258-259), or a single 1x1x1
 * convolution when w1 is NULL.  w2t / wst are the TRANSPOSED weight matrices ((cm, co) and (ci, co)).
 * act: 0 none, 1 ReLU, 2 GELU(erf), 3 SiLU, 4 tanh.  skip_mode: 0 none, 1 convolution of `skip` (batch, ci, P),
 * 2 broadcast of the last time slice of `skip` (batch, co, P/T*skip_T).
 * Returns TCFD_EINVAL for channel combinations that are not instantiated. */
int tcfd_fno_pointwise(const void* x, const void* skip, void* out, const void* w1, const void* b1, const void* w2t,
                       const void* b2, const void* wst, const void* bs, int batch, int ci, int cm, int co, long P,
                       int T, int skip_T, int act1, int act2, int skip_mode, long w2_bstride, long b2_bstride,
                       const void* pe, void* stream);
/* The same, also storing the block's PRE-activation z2 (batch, co, P) into `pre` (NULL: exactly the call above): the training
 * forward of a block whose output activation is neither ReLU nor the identity -- its backward reads act2'(z2) from it
 * (tcfd_fno_pointwise_bwd_out) instead of recomputing W2.h + Ws.skip. */
int tcfd_fno_pointwise_pre(const void* x, const void* skip, void* out, void* pre, const void* w1, const void* b1, const void* w2t,
                           const void* b2, const void* wst, const void* bs, int batch, int ci, int cm, int co, long P,
                           int T, int skip_T, int act1, int act2, int skip_mode, long w2_bstride, long b2_bstride,
                           const void* pe, void* stream);
/* The same block in float64 (every array double; no positional-encoding input; w2_bstride / b2_bstride as in
 * tcfd_fno_pointwise: per-batch-element offsets of w2t / b2, 0 = shared): what an SFNO converted
 * with .double() (fno/base.py:342-349) runs.  Widths 4, 6, 8, 10, 12, 16, 20, 24, 32, any hidden width. */
int tcfd_fno_pointwise_f64(const void* x, const void* skip, void* out, const void* w1, const void* b1, const void* w2t,
                           const void* b2, const void* wst, const void* bs, int batch, int ci, int cm, int co, long P, int T,
                           int skip_T, int act1, int act2, int skip_mode, long w2_bstride, long b2_bstride, void* stream);
/* pe (ci, P) or NULL: when given, x is ONE channel (batch, 1, P) and the block input is x + pe[c] -- the lifting
 * operator's "input + positional encoding" (fno/sfno.py:109-113) without materialising the (batch, ci, P) tensor.
 * w2_bstride / b2_bstride: element offsets of w2t / b2 per batch element (0 = shared weights); a per-sample
 * affine map such as a folded LayerNormnd (fno/base.py:61-83) then rides in the single-layer form.
 * tcfd_row_moments: sum and sum of squares (double, stats[rows][2]) of every row of a (rows, L) fp32 matrix --
 * the statistics of LayerNormnd / GroupNorm(1 group) with the rows cut across many workgroups. */
int tcfd_row_moments(const void* x, void* stats, int rows, long L, void* stream);
/* The same for float64 rows. */
int tcfd_row_moments_f64(const void* x, void* stats, int rows, long L, void* stream);
/* Small reductions of the SFNO training step (fp32 data):
 *   sum_rows: out (cols) double = column sums of in (rows, cols); scratch = tcfd_sum_rows_slices(rows) * cols doubles.
 *     The per-wave partial weight-gradient rows of tcfd_fno_pointwise_bwd are added with it.
 *   sum_t_into_last: g (rows, sT) = 0 except g[r][sT-1] = sum_t d[r][t], d (rows, T): the gradient of the skip input of
 *     skip_mode 2 (only its last time slice is used, fno/sfno.py:258-259) from the full dL/dz2. */
int tcfd_sum_rows_slices(long rows);
int tcfd_sum_rows(const void* in, void* out, void* scratch, long rows, long cols, void* stream);
/* tcfd_sum_rows whose final pass writes sub-matrices of the summed row as fp32 into dense tensors of their own: nseg <= 8 segments,
 * segs = nseg x {src_off, nrows, ncols, pitch} (host longs: element (r, c) of segment k is sum[src_off + r * pitch + c]), dst = nseg
 * device pointers (host array), segment k written as (nrows, ncols) row-major.  The parameter gradients of a pointwise block leave
 * the [dW2 | db2 | dWs] / [dW1 | db1] layout of tcfd_fno_pointwise_bwd this way.  scratch as for tcfd_sum_rows. */
int tcfd_sum_rows_scatter(const void* in, void* scratch, long rows, long cols, int nseg, const long* segs, void* const* dst,
                          void* stream);
int tcfd_sum_t_into_last(const void* d, void* g, long rows, int T, int sT, void* stream);

/* ---- per-launch event timing (measurement aid; no reference counterpart) -------
 * Between profile_begin and profile_end every kernel the plan launches is
 * bracketed by a pair of HIP events recorded on the launch stream (up to
 * max_records launches).  profile_end synchronises on them and returns, per
 * launch, the kernel kind (0 column pass A, 1 row pass, 2 column pass C+A,
 * 3 column pass C incl. dw/dt, 5 other; 4 is unused) and its duration in milliseconds.
 * Mutates the plan: not to be used concurrently with other calls on it. */
int tcfd_ns2d_profile_begin(tcfd_ns2d_plan* plan, int max_records);
int tcfd_ns2d_profile_end(tcfd_ns2d_plan* plan, int capacity, int* count, int* kinds, float* ms);

/* Per-launch event timing of the FNO kernels (measurement aid, process-wide; the counterpart of tcfd_ns2d_profile_begin / _end).
 * Between begin and end every transform / contraction / pointwise launch is bracketed by a pair of HIP events recorded on its
 * launch stream (up to max_records launches).  profile_end synchronises on them and returns, per launch, the kind
 * (0 forward t/y transform, 1 forward x transform, 2 contraction, 3 inverse x transform, 4 inverse t/y transform, 5 two-layer
 * pointwise block, 6 its backward, 7 single-layer pointwise forms, 8 contraction weight gradient, 9 other, 10 backward of the
 * single-layer forms) and its duration in milliseconds.  Not for concurrent use with measured work on other threads. */
int tcfd_fno_profile_begin(int max_records);
int tcfd_fno_profile_end(int capacity, int* count, int* kinds, float* ms);

/* Backward of tcfd_fno_pointwise (shared weights): one pass over x / skip / dout recomputes the
 * block per point and writes dx (batch, ci, P), dskip (skip_mode 1: (batch, ci, P), may be NULL; skip_mode 2:
 * dL/d(pre-activation) (batch, co, P), which the caller sums over t into the skip's last time slice; skip_mode 3 = skip_mode 2 with
 * that sum done by the kernel: dskip is (batch, co, P / T) -- two-layer blocks on the tiled kernel with T | 16 or T | 80 and P a
 * multiple of 16 resp. 80, TCFD_EINVAL otherwise, also at the layout query) and per-wave partial weight
 * gradients into `partials` (max_waves rows).  On return dims[6] = {COP, CB, CM1, CIP, floats per row, rows
 * written}; a row holds two row-major zero-padded tiles, A (COP x CB) then B (CM1 x CIP), with ch = cm (single
 * layer: ci):  A[o][0:ch] = dW2[o][.],  A[o][ch] = db2[o] = dbs[o],  A[o][ch+1 : ch+1+ci] = dWs[o][.];
 * B[m][0:ci] = dW1[m][.],  B[m][ci] = db1[m].  The caller sums the rows (deterministic reduction).
 * x == NULL: layout query, only dims is filled.  Replaces torch autograd through fno/base.py:86-111 +
 * fno/sfno.py:607-614.  TCFD_EINVAL for channel combinations that are not instantiated. */
int tcfd_fno_pointwise_bwd(const void* x, const void* skip, const void* dout, void* dx, void* dskip, const void* w1,
                           const void* b1, const void* w2t, const void* b2, const void* wst, const void* bs,
                           void* partials, int max_waves, int* dims, int batch, int ci, int cm, int co, long P, int T,
                           int skip_T, int act1, int act2, int skip_mode, int per_sample, void* stream);
/* The same with what the forward kept handed over in `out` (batch, co, P) (NULL: exactly the call above):
 *   act2 = ReLU:            the block's forward OUTPUT -- its mask is read from it (y > 0 <=> z2 > 0: the mask the forward kernel
 *                           applied; torch's own ReLU backward reads the saved result the same way);
 *   act2 = GELU/SiLU/tanh:  the block's PRE-activation z2 (tcfd_fno_pointwise_pre) -- act2'(z2) is evaluated from it.
 * Nothing of z2 = W2.h + Ws.skip is then recomputed, and for the two-layer form with P % 4 == 0 the call runs the tiled
 * all-matrix-instruction kernel of csrc/tcfd_fno_tiles.hip at the widths 10, 16, 20, 24, 32 with cm = 4 ci (59 / 88 / 196 / 244 /
 * 352 v_mfma_f32_16x16x4_f32 per 16 points).  Without `out` those widths above 14 return "not instantiated". */
int tcfd_fno_pointwise_bwd_out(const void* x, const void* skip, const void* dout, const void* out, void* dx, void* dskip,
                               const void* w1, const void* b1, const void* w2t, const void* b2, const void* wst, const void* bs,
                               void* partials, int max_waves, int* dims, int batch, int ci, int cm, int co, long P, int T,
                               int skip_T, int act1, int act2, int skip_mode, int per_sample, void* stream);
/* What tcfd_fno_pointwise_bwd_out wants in `out` for the two-layer block ci -> cm -> co at P points per sample: 0 nothing is
 * read, 1 the forward output, 2 the pre-activation (tcfd_fno_pointwise_pre).  A host asks BEFORE the forward, so that it only
 * keeps / produces a tensor the backward kernel will read. */
int tcfd_fno_pointwise_bwd_saved(int ci, int cm, int co, long P, int act1, int act2);
/* The single-layer form (w1 NULL, no skip, no activations) whose input is x1 (batch, 1, P) + pe (ci, P) -- the `pe` mode of
 * tcfd_fno_pointwise -- so the weight gradients of the lifting operator's projection need no materialised (batch, ci, P) input. */
int tcfd_fno_pointwise_bwd_pe(const void* x1, const void* pe, const void* dout, void* dx, const void* w2t, const void* b2,
                              void* partials, int max_waves, int* dims, int batch, int ci, int co, long P, int per_sample,
                              void* stream);
/* Per-sample sums of outer products over the points, on MFMA: partials (waves_per_sample, batch, R16, C16) row-major with
 * R16 = 16 ceil(co / 16), C16 = 16 ceil((C + 1) / 16) and, after adding the waves (tcfd_sum_rows),
 * M[o][c] = sum_p dy[b][o][p] xin[b][c][p] for c < C and M[o][C] = sum_p dy[b][o][p]; xin = x
 * (batch, C, P), or x (batch, P) + pe (C, P) when pe is given.  What the backward of proj(LayerNormnd(xin)) (fno/sfno.py:252-254,
 * fno/base.py:61-83) needs from the data.  C <= 47, co <= 32, P % 16 == 0, waves_per_sample a multiple of 4. */
int tcfd_fno_sample_outer_sums(const void* dy, const void* x, const void* pe, void* partials, int batch, int c, int co, long P,
                               int waves_per_sample, void* stream);
/* per_sample = 1: the rows written (dims[5], a multiple of batch) are per-SAMPLE partial sums, row r belongs to batch
 * element r % batch -- what the backward of a LayerNorm folded into the convolution needs (its statistics differ per
 * sample); dx may then be NULL (only the sums are wanted). */

/* Weighted squared norm of half spectra, the reduction behind SobolevLoss (fno/losses.py:263-315):
 *   partial[b][blk] = sum_e |z[b][e]|^2 * w2[e]  over block blk's share of the `elems` complex entries of field b,
 * accumulated in double; the caller adds the `blocks` partials of a field.  z (batch, elems) complex64 / complex128
 * (dtype TCFD_C64 / TCFD_C128), w2 (elems) real of the matching precision, partial (batch, blocks) double. */
int tcfd_weighted_sqnorm(const void* z, const void* w2, void* partial, long batch, long elems, int blocks, int dtype,
                         void* stream);

/* Output operator glue (fno/sfno.py:313-328) without its two extra passes over the data:
 *   tcfd_fno_reduce_frames: the 1x1x1 channel reduction (ci -> 1) writes its T latent steps BEHIND the last frame of the
 *     network input, out (b, 1, P / T * (T + 1)) = cat([frame[..., -1:], conv1x1(x)], dim=t) -- the reference's torch.cat.
 *     x (b, ci, P), w2t (ci, 1), b2 (1) or NULL, frame (b, P / T, frame_T), fp32; T even.
 *   tcfd_fno_inverse_trunc_residual: tcfd_fno_inverse_trunc whose store loop adds res[..., -1:] (res (batch * c, X, Y, res_T))
 *     to every kept step -- the reference's `v_res[..., -1:] + conv(...)[..., -out_steps:]`. */
int tcfd_fno_reduce_frames(const void* x, void* out, const void* w2t, const void* b2, const void* frame, int frame_T, int batch,
                           int ci, long P, int T, void* stream);
int tcfd_fno_inverse_trunc_residual(const tcfd_fno_plan* p, const void* vh, void* out, const void* res, int res_T, int batch, int c,
                                    int t_keep, double inv_scale, void* ws, size_t ws_bytes, void* stream);
/* out = transform, plus last (batch * c, X, Y) at the LAST kept step only (training: the gradient of an input of which only the
 * last time slice was used joins the adjoint of the forward transform in its store loop, compact -- no zero-filled
 * activation-sized tensor in between; fno/sfno.py:258-259). */
int tcfd_fno_inverse_trunc_last(const tcfd_fno_plan* p, const void* vh, void* out, const void* last, int batch, int c, int t_keep,
                                double inv_scale, void* ws, size_t ws_bytes, void* stream);

/* Lifting operator, proj(LayerNormnd(v + q)) with v ONE channel (fno/sfno.py:252-254, fno/base.py:61-83): the three
 * per-sample sums over v (sum, sum of squares, dot product with the table's channel sum qs) and, from them and the table's
 * constants sq = sum q, sq2 = sum q^2 (one double each, on the device), the per-sample folded weights for the `pe` form of
 * tcfd_fno_pointwise:  w2t (b, C, co) = W[o][c] gamma[c] rstd_b,  fb (b, co) = sum_c (beta[c] - gamma[c] mu_b rstd_b) W[o][c] +
 * bias[o].  Two launches.  v (b, P), qs (P), W (co, C), bias (co) / gamma (C) / beta (C) fp32 or NULL; moments (b, 2) double
 * or NULL receives (sum, sum of squares) of every sample's (C, P) block; scratch: (b, 3) doubles. */
int tcfd_fno_lift_fold(const void* v, const void* qs, const void* sq, const void* sq2, const void* W, const void* bias,
                       const void* gamma, const void* beta, double eps, void* w2t, void* fb, void* moments, void* scratch,
                       int batch, int C, int co, long P, void* stream);

/* The kept modes of the lifting operator's projection WITHOUT the projection (fno/sfno.py:252-256: proj(norm(v + table)) is a
 * per-sample affine map of the one input channel, and the truncated transform that follows it is linear):
 *     out[b, o, k] = sum_c w2t[b, c, o] (vh[b, k] + table[c, k]) + fb[b, o] table[C, k]
 * vh (batch, K) kept modes of the one-channel input, table (C + 1, K) kept modes of the C table channels and of the constant-1
 * field (same plan, same padding, same normalisation: formed once per mesh), w2t (batch, C, co) / fb (batch, co) from
 * tcfd_fno_lift_fold, out (batch, co, K); complex64 / fp32.  One launch; C <= 32. */
int tcfd_fno_lift_spectrum(const void* vh, const void* table, const void* w2t, const void* fb, void* out, int batch, int C,
                           int co, long K, void* stream);

/* ---- SobolevLoss in three launches (fno/losses.py:263-315; BASELINE config 5 "forward + loss") -------------------------
 * x, y: (batch, n, n, nt) real, TIME-LAST and contiguous, read in place (no permuted copies, no x - y tensor).
 *   pass 1  per (b, row) slab: d = x - y while staging, one complex n-point FFT per time step of d_t + i y_t, Hermitian
 *           separation -> half-spectrum planes (field, b, t, row, ky) in the workspace
 *   pass 2  n-point FFT down 128-byte column tiles, |.|^2 * w2 accumulated in double (no spectrum written)
 *   pass 3  the scalar (relative / mesh-weighted / time-averaged / batch mean or sum, losses.py:297-314)
 * The plan holds the twiddle table of one (n, precision): n = 2^k in [16, 1024], 3 * 2^k in [96, 768] or 5 * 2^k in
 * [80, 640]; dtype TCFD_C64 = float data,
 * TCFD_C128 = double data.  w2: (n, n/2+1) real table of the data's precision = weight^2 of the half spectrum with the
 * Hermitian multiplicity (1 on the DC / Nyquist columns, else 2) and the fft-norm scale folded in -- the caller builds it
 * once per (n, order, alpha, cutoff, norm).  nfields = 2: both ||w (x - y)^|| and ||w y^|| (relative loss); 1: the first only
 * (y may be NULL: the norm of x).  out: ONE scalar of the data's precision (device memory); sums (optional, device):
 * (nfields, batch, nt) doubles = the per-time squared norms.  mesh_weighted: 0 off, 1 on, 2 on with the unit norm of a
 * non-relative loss divided by n in float32 (the reference's torch.ones(bsz) / n under a float32 default dtype,
 * losses.py:297-308: differs from 1 by 1.5e-8 at n = 80, not at all when n is a power of two).  tcfd_sobolev_loss_supported: 0 when nt time steps of an
 * n-point row do not fit one workgroup (the caller then composes the loss from tcfd_rfft2 + tcfd_weighted_sqnorm). */
typedef struct tcfd_loss_plan tcfd_loss_plan;
int tcfd_loss_plan_create(tcfd_loss_plan** out, int n, int dtype);
void tcfd_loss_plan_destroy(tcfd_loss_plan* p);
size_t tcfd_loss_workspace_bytes(const tcfd_loss_plan* p, long batch, int nt, int nfields);
int tcfd_sobolev_loss_supported(const tcfd_loss_plan* p, int nt, int nfields);
int tcfd_sobolev_loss(const tcfd_loss_plan* p, const void* x, const void* y, const void* w2, long batch, int nt, int nfields,
                      int relative, int mesh_weighted, int time_average, int reduction, void* out, void* sums, void* ws,
                      size_t ws_bytes, void* stream);
/* The gradient of that scalar with respect to x (the autograd side of SobolevLoss, fno/losses.py:263-315 under
 * loss.backward()): grad (batch, n, n, nt), the layout of x.  x, y, nfields and the four flags as in the forward call; wf: w2
 * WITHOUT the Hermitian multiplicity (the (n, n/2+1) weights of the full spectrum, fft-norm scale folded in); sums: what the
 * forward call left in its `sums` argument; gout: ONE scalar of the data's precision in device memory (the incoming
 * gradient).  y is a constant here (a caller whose target needs a gradient composes the loss from tcfd_rfft2 instead).
 * Three launches; the workspace of a one-field forward call suffices (tcfd_loss_workspace_bytes(p, batch, nt, 1)). */
int tcfd_sobolev_loss_backward(const tcfd_loss_plan* p, const void* x, const void* y, const void* wf, const void* sums,
                               const void* gout, long batch, int nt, int nfields, int relative, int mesh_weighted,
                               int time_average, int reduction, void* grad, void* ws, size_t ws_bytes, void* stream);

/* STREAM-style device probe (measurement aid, SURVEY 8d "verify with a device STREAM-style probe"): `iters`
 * launches of a 16-byte-per-lane grid-stride kernel over `bytes` (a multiple of 16) of caller-owned device memory,
 * timed with HIP events on `stream`.  mode 0: copy src -> dst (2*bytes of traffic per launch); 1: read-only
 * sum of src (dst = 8 bytes of scratch); 2: write-only fill of dst (src unused).  *ms = average launch
 * duration.  Synchronises the stream (it is a benchmark, not part of the path). */
int tcfd_hbm_probe(const void* src, void* dst, size_t bytes, int mode, int iters, float* ms, void* stream);

/* Pitched device -> host copy on `stream`: `rows` rows of `row_bytes` bytes, source rows `src_pitch` bytes apart
 * (device), destination rows `dst_pitch` bytes apart (host).  Asynchronous with respect to the host when the
 * destination is page-locked.  The output side of the trajectory recorder uses it to drop one record of a batch
 * shard into its (sample, record, y, x) slot of the host dataset while the next steps run (the reference does four
 * blocking `.cpu()` calls per record, fno/data_gen/solvers.py:246-250). */
int tcfd_copy_rows_to_host(void* dst_host, size_t dst_pitch, const void* src_dev, size_t src_pitch, size_t row_bytes,
                           size_t rows, void* stream);

/* Page-lock / release a range of ordinary (pageable) host memory, so that tcfd_copy_rows_to_host into it is a true
 * asynchronous DMA.  The ensemble driver locks its result region by region from a helper thread while the steps run: one
 * page-locked allocation of the whole result would hold the runtime's memory lock for its whole duration and stall every
 * device allocation of the stepping thread behind it (measured: 0.14 - 0.28 s stalls at 5.4 GB).  The caller keeps the
 * memory alive until it has been unregistered. */
int tcfd_host_register(void* ptr, size_t bytes);
int tcfd_host_unregister(void* ptr);

#endif /* TCFD_H_TYPES_ONLY */

#ifdef __cplusplus
}
#endif
#endif
