/* The C ABI on its own: no Python, no PyTorch -- plain C99 against include/tcfd.h and the HIP runtime.
 *
 * Known-answer run: the Taylor-Green vortex  w0 = 2k cos(kx) cos(ky)  on [0, 2 pi)^2 is an exact solution of the
 * unforced vorticity equation that just decays, w(t) = w0 exp(-2 nu k^2 t) (the advection term vanishes).  The program
 * builds the operator tables the way NavierStokes2DSpectral._initialize does (torch_cfd/equations.py:394-403,
 * brick_wall_filter_2d torch_cfd/spectral.py:78-84), writes w0's half spectrum by hand, advances 100 RK4-CN steps with
 * tcfd_ns2d_step (Carpenter-Kennedy coefficients, equations.py:294-317) and compares with the analytic decay.
 *
 *   gcc -std=c99 -O2 examples/c_abi_taylor_green.c -Iinclude -I/opt/rocm/include -Ltorch-cfd_amd/csrc -L/opt/rocm/lib \
 *       -ltcfd_hip -lamdhip64 -lm -Wl,-rpath,$PWD/torch-cfd_amd/csrc -Wl,-rpath,/opt/rocm/lib -o c_abi_taylor_green
 *   ./c_abi_taylor_green          (tests/test_ns2d_gpu.py::test_c_abi_without_python builds and runs it)
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "tcfd.h"

#define CHECK_HIP(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "HIP: %s\n", hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_TCFD(e) do { if ((e) != 0) { fprintf(stderr, "tcfd: %s\n", tcfd_last_error()); return 3; } } while (0)

int main(void) {
    const int n = 64, m = n / 2 + 1, k = 2, steps = 100, batch = 3;
    const double PI = 3.14159265358979323846, L = 2 * PI, nu = 1e-2, dt = 1e-2;
    double *kx = malloc(sizeof(double) * n), *ky = malloc(sizeof(double) * m);
    double *lin = malloc(sizeof(double) * n * m), *mask = calloc((size_t)n * m, sizeof(double));
    for (int i = 0; i < n; ++i) kx[i] = (i < n / 2 ? i : i - n) / L;        /* fftfreq(n, d = L / n) */
    for (int j = 0; j < m; ++j) ky[j] = (j < n / 2 ? j : j - n) / L;        /* the Nyquist column is -n/2 / L */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) lin[i * m + j] = nu * (-4 * PI * PI * (kx[i] * kx[i] + ky[j] * ky[j]));
    {   /* 2/3-rule brick wall, with the reference's  -int(2/3*n) // 2  row count for the upper block */
        const int kr = (int)(2.0 / 3.0 * n), lo = kr / 2, hi = (kr + 1) / 2, cols = (int)(2.0 / 3.0 * m);
        for (int i = 0; i < n; ++i)
            if (i < lo || i >= n - hi)
                for (int j = 0; j < cols; ++j) mask[i * m + j] = 1.0;
    }
    tcfd_ns2d_plan* plan = NULL;
    CHECK_TCFD(tcfd_ns2d_plan_create(&plan, n, TCFD_C128, kx, ky, lin, mask, NULL));

    /* half spectrum of w0 (unnormalised forward transform): k n^2 / 2 at (row k, col k) and (row n-k, col k) */
    const size_t field = (size_t)n * m * 2, bytes = sizeof(double) * field * batch;
    double* h0 = calloc(field * batch, sizeof(double));
    for (int b = 0; b < batch; ++b) {
        const double amp = (b + 1) * k * (double)n * n / 2;                   /* three amplitudes, same decay */
        h0[b * field + ((size_t)k * m + k) * 2] = amp;
        h0[b * field + ((size_t)(n - k) * m + k) * 2] = amp;
    }
    void *w = NULL, *w_out = NULL, *dwdt = NULL, *ws = NULL;
    const size_t ws_bytes = tcfd_ns2d_workspace_bytes(plan, batch);
    CHECK_HIP(hipMalloc(&w, bytes));
    CHECK_HIP(hipMalloc(&w_out, bytes));
    CHECK_HIP(hipMalloc(&dwdt, bytes));
    CHECK_HIP(hipMalloc(&ws, ws_bytes));
    CHECK_HIP(hipMemcpy(w, h0, bytes, hipMemcpyHostToDevice));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));

    const double alphas[6] = {0, 0.1496590219993, 0.3704009573644, 0.6222557631345, 0.9582821306748, 1};
    const double betas[5] = {0, -0.4178904745, -1.192151694643, -1.697784692471, -1.514183444257};
    const double gammas[5] = {0.1496590219993, 0.3792103129999, 0.8229550293869, 0.6994504559488, 0.1530572479681};
    double gdt[5], mu[5];
    for (int s = 0; s < 5; ++s) { gdt[s] = gammas[s] * dt; mu[s] = 0.5 * dt * (alphas[s + 1] - alphas[s]); }
    CHECK_TCFD(tcfd_ns2d_step(plan, w, w_out, dwdt, batch, 5, betas, gdt, mu, steps, 1.0 / (steps * dt), ws, ws_bytes,
                              (void*)stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    double* h1 = malloc(bytes);
    CHECK_HIP(hipMemcpy(h1, w_out, bytes, hipMemcpyDeviceToHost));

    const double decay = exp(-2 * nu * k * k * dt * steps);
    double err2 = 0, ref2 = 0;
    for (size_t i = 0; i < field * batch; ++i) {
        const double e = h1[i] - h0[i] * decay;
        err2 += e * e;
        ref2 += h0[i] * h0[i] * decay * decay;
    }
    const double rel = sqrt(err2 / ref2);
    printf("tcfd version %d, %d x %d^2 fields, %d RK4-CN steps: rel-L2 error vs exp(-2 nu k^2 t) = %.3e\n", tcfd_version(),
           batch, n, steps, rel);
    tcfd_ns2d_plan_destroy(plan);
    (void)hipFree(w); (void)hipFree(w_out); (void)hipFree(dwdt); (void)hipFree(ws);
    if (!(rel < 1e-8)) { printf("FAIL\n"); return 1; }
    printf("PASS\n");
    return 0;
}
