// Does element-wise (VALU) work hide behind v_mfma_f32_16x16x4_f32 on gfx950?  Every wave runs ITERS x { M matrix
// instructions on independent accumulators ; V fused multiply-adds on other registers }, with W waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tests/micro/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int M, int V>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f4 acc[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the same with v_mfma_f32_16x16x16_bf16 (the matrix engine proper: 4 passes for 4 x the K of the fp32 instruction)
typedef short s4 __attribute__((ext_vector_type(4)));
template <int M, int V>
__global__ __launch_bounds__(256) void kb(float* out, int iters) {
    f4 acc[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const s4 av = s4{(short)0x3f80, (short)0x3f00, (short)0x3e80, (short)0x3f80}, bv = s4{(short)0x3f80, (short)0x3f80, (short)0x3f00, (short)0x3e80};
    float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// fp64: v_mfma_f64_16x16x4_f64 beside v_fma_f64 (would a radix-16 stage of the row pass's 1024-point transform
// on the matrix pipe run BESIDE the fp64 butterflies of the vector pipe?)
typedef double d4 __attribute__((ext_vector_type(4)));
template <int M, int V>
__global__ __launch_bounds__(256) void kd(float* out, int iters) {
    d4 acc[4] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j & 7] = __builtin_fma(v[j & 7], 1.0001, 0.5);
        __builtin_amdgcn_sched_barrier(0);
    }
    double s = 0;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}
template <int M, int V>
static void rund(int wgs_per_cu, float* d) {
    const int iters = 20000, blocks = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kd<M, V>), dim3(blocks), dim3(256), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kd<M, V>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("f64 16x16x4: M=%d V=%2d waves/SIMD=%d : %8.3f ms  -> %6.1f ns per iteration\n", M, V, wgs_per_cu, ms, ms * 1e6 / iters);
}
template <int M, int V>
static void runb(int wgs_per_cu, float* d) {
    const int iters = 20000, blocks = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kb<M, V>), dim3(blocks), dim3(256), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kb<M, V>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("bf16 16x16x16: M=%d V=%2d waves/SIMD=%d : %8.3f ms  -> %6.1f ns per iteration\n", M, V, wgs_per_cu, ms, ms * 1e6 / iters);
}
template <int M, int V>
static void run(int wgs_per_cu, float* d) {
    const int iters = 20000, blocks = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<M, V>), dim3(blocks), dim3(256), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<M, V>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // cycles per iteration per SIMD at 2.4 GHz (nominal): one workgroup = 4 waves = one wave per SIMD
    printf("M=%d V=%2d waves/SIMD=%d : %8.3f ms  -> %6.1f ns per iteration of ONE wave slot (matrix alone would be %d x 32 cycles)\n", M, V,
           wgs_per_cu, ms, ms * 1e6 / iters, M * wgs_per_cu);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w = 1; w <= 2; ++w) {
        run<4, 0>(w, d); run<4, 8>(w, d); run<4, 16>(w, d); run<4, 32>(w, d); run<0, 32>(w, d);
    }
    for (int w = 1; w <= 2; ++w) {
        runb<4, 0>(w, d); runb<4, 8>(w, d); runb<4, 16>(w, d); runb<4, 32>(w, d);
    }
    for (int w = 1; w <= 2; ++w) {
        rund<4, 0>(w, d); rund<4, 8>(w, d); rund<4, 16>(w, d); rund<4, 32>(w, d); rund<0, 32>(w, d);
    }
    return 0;
}
