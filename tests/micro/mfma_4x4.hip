// v_mfma_f32_4x4x1_16b_f32 on gfx950: operand layout and issue / dependent rates (is a quarter-filled 16 x 16 x 4 tile -- the
// 4-channel remainder of a 20-channel layer -- cheaper as 16 independent 4 x 4 x 1 blocks?).
//   hipcc --offload-arch=gfx950 -O3 tests/micro/mfma_4x4.hip -o /tmp/m44 && /tmp/m44
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

// layout probe: every lane supplies a = lane + 1 (A) and b = 100 + lane (B); D of each lane / register is printed by the host
__global__ void k_layout(float* out) {
    const int l = threadIdx.x;
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), (float)(100 + l), d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}

template <int DEP>
__global__ void k_rate(float* out, int iters, long long* cyc) {
    const int l = threadIdx.x;
    f4 d[8];
    for (int i = 0; i < 8; ++i) d[i] = f4{0.f, 0.f, 0.f, 0.f};
    float a = l * 0.001f, b = 1.f + l * 0.002f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = DEP ? 0 : i;                               // DEP: all eight into ONE accumulator (a dependent chain)
            d[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d[k], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    f4 s = d[0];
    for (int i = 1; i < 8; ++i) s += d[i];
    out[blockIdx.x * 64 + l] = s[0] + s[1] + s[2] + s[3];
    if (l == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int DEP>
__global__ void k_rate16(float* out, int iters, long long* cyc) {     // the same with 16 x 16 x 4
    const int l = threadIdx.x;
    f4 d[8];
    for (int i = 0; i < 8; ++i) d[i] = f4{0.f, 0.f, 0.f, 0.f};
    float a = l * 0.001f, b = 1.f + l * 0.002f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = DEP ? 0 : i;
            d[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d[k], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    f4 s = d[0];
    for (int i = 1; i < 8; ++i) s += d[i];
    out[blockIdx.x * 64 + l] = s[0] + s[1] + s[2] + s[3];
    if (l == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    k_layout<<<1, 64>>>(out);
    std::vector<float> h(256);
    hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost);
    // expected if block = lane / 4, A row i = lane % 4, B column j = lane % 4, D[i][j] in register i of lane (block, j):
    //   D = (4 block + i + 1) * (100 + 4 block + j)
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l / 4, j = l % 4;
            const float want = (float)(4 * blk + r + 1) * (float)(100 + 4 * blk + j);
            if (h[l * 4 + r] != want) ok = 0;
        }
    printf("layout (block = lane / 4; A row = lane %% 4; B col = lane %% 4; D[row = register][col = lane %% 4]): %s\n", ok ? "CONFIRMED" : "NOT this");
    if (!ok) for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    const int iters = 4096;
    long long c;
    auto run = [&](auto kern, const char* name) {
        kern<<<1, 64>>>(out, iters, cyc); hipDeviceSynchronize();
        kern<<<1, 64>>>(out, iters, cyc); hipDeviceSynchronize();
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-44s %.2f clock-counter ticks per instruction (one wave)\n", name, (double)c / (8.0 * iters));
    };
    run(k_rate<0>, "4x4x1_16b, 8 independent accumulators");
    run(k_rate<1>, "4x4x1_16b, one dependent chain");
    run(k_rate16<0>, "16x16x4, 8 independent accumulators");
    run(k_rate16<1>, "16x16x4, one dependent chain");
    return 0;
}
