// Micro-benchmark (not part of the library): what a kernel boundary costs on this GPU next to a grid barrier inside
// a persistent kernel -- the numbers behind the small-grid (C2) design notes in DESIGN.md.
//   (1) graph replay of a chain of near-empty kernels: time per kernel
//   (2) persistent kernel, G workgroups, barrier over groups of `team` workgroups: time per barrier
// build: hipcc --offload-arch=gfx950 -O3 tests/micro/grid_barrier.hip -o tests/micro/build/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_tiny(float* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] += 1.0f;
}

// sense-free counting barrier: every team has a monotonically increasing counter; phase k is complete when the
// counter reaches k * team.  Bounded spin (returns an error flag instead of hanging the GPU).
__device__ __forceinline__ bool team_barrier(unsigned* ctr, unsigned target, int* err) {
    __shared__ int failed;
    if (threadIdx.x == 0) failed = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(ctr, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 20)) { *err = 1; failed = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
    return !failed;
}

__global__ __launch_bounds__(256) void k_persist(unsigned* ctrs, float* data, int team, int iters, int work, int* err) {
    const int t = blockIdx.x / team;   // teams of consecutive block ids
    unsigned* ctr = ctrs + 32 * t;
    float acc = 0;
    for (int it = 1; it <= iters; ++it) {
        // a little memory work: read what a neighbour of the team wrote in the previous phase, write own slot
        const int me = blockIdx.x, nb = t * team + (blockIdx.x + 1) % team;
        for (int w = 0; w < work; ++w) acc += data[(size_t)nb * 4096 + w * 256 + threadIdx.x];
        for (int w = 0; w < work; ++w) data[(size_t)me * 4096 + w * 256 + threadIdx.x] = acc + it;
        if (!team_barrier(ctr, (unsigned)it * team, err)) return;
    }
}

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* buf; CK(hipMalloc(&buf, 64 << 20)); CK(hipMemset(buf, 0, 64 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // (1) kernel chain in a graph
    for (int nblk : {256, 2048}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int k = 0; k < 64; ++k) hipLaunchKernelGGL(k_tiny, dim3(nblk), dim3(256), 0, st, buf, nblk * 256);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph chain, %4d blocks/kernel: %.2f us per kernel\n", nblk, ms * 1e3 / (20 * 64));
        // plain stream launches
        for (int k = 0; k < 64; ++k) hipLaunchKernelGGL(k_tiny, dim3(nblk), dim3(256), 0, st, buf, nblk * 256);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < 1280; ++k) hipLaunchKernelGGL(k_tiny, dim3(nblk), dim3(256), 0, st, buf, nblk * 256);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("stream chain, %4d blocks/kernel: %.2f us per kernel\n", nblk, ms * 1e3 / 1280);
    }
    // (2) persistent kernel barriers
    unsigned* ctrs; int* err;
    CK(hipMalloc(&ctrs, 1 << 20)); CK(hipMalloc(&err, 4));
    for (int G : {256, 512, 1024}) for (int team : {8, 16, 32, 64, G}) for (int work : {0, 4}) {
        CK(hipMemsetAsync(ctrs, 0, 1 << 20, st)); CK(hipMemsetAsync(err, 0, 4, st));
        const int iters = 2000;
        hipLaunchKernelGGL(k_persist, dim3(G), dim3(256), 0, st, ctrs, buf, team, 10, work, err);
        CK(hipMemsetAsync(ctrs, 0, 1 << 20, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_persist, dim3(G), dim3(256), 0, st, ctrs, buf, team, iters, work, err);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("persistent G=%4d team=%4d work=%d: %.2f us per phase%s\n", G, team, work, ms * 1e3 / iters, herr ? "  (SPIN LIMIT HIT)" : "");
    }
    return 0;
}
