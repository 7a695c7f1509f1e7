// Micro-benchmark: achievable HBM bandwidth of the access patterns used by the column kernels
// (tiles of C complex columns x all rows of a row-major (B, n, pitch) array) vs contiguous streaming.
// Build: hipcc --offload-arch=gfx950 -O3 tests/micro/access_patterns.hip -o tests/micro/access_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct alignas(16) c128 { double x, y; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// contiguous grid-stride copy, 16 B per lane, UNR loads in flight
template <int UNR>
__global__ void k_copy(const c128* __restrict__ in, c128* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride * UNR) {
        c128 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) if (i + u * stride < n) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNR; ++u) if (i + u * stride < n) out[i + u * stride] = v[u];
    }
}

// column-tile copy: block = (tile of C columns, batch b, chunk of ROWS rows); lane -> (row = tid / C, col = tid % C)
template <int C, int EPT>
__global__ void k_tile_copy(const c128* __restrict__ in, c128* __restrict__ out, int n, int m, int pitch, int batch,
                            int ntiles, int row_chunks) {
    const int rows_per_pass = blockDim.x / C;
    int id = blockIdx.x;
    const int b = id % batch; id /= batch;
    const int chunk = id % row_chunks; id /= row_chunks;
    const int tile = id;
    const int c = threadIdx.x % C, j = threadIdx.x / C;
    const int jc = tile * C + c;
    if (jc >= m) return;
    const int rows_per_chunk = n / row_chunks;
    const size_t base = ((size_t)b * n + (size_t)chunk * rows_per_chunk) * pitch + jc;
    for (int r0 = 0; r0 < rows_per_chunk; r0 += rows_per_pass * EPT) {
        c128 v[EPT];
#pragma unroll
        for (int t = 0; t < EPT; ++t) v[t] = in[base + (size_t)(r0 + j + t * rows_per_pass) * pitch];
#pragma unroll
        for (int t = 0; t < EPT; ++t) out[base + (size_t)(r0 + j + t * rows_per_pass) * pitch] = v[t];
    }
}

template <typename F>
float time_ms(F f, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main() {
    const int n = 1024, m = 513, B = 64;
    for (int pitch : {513, 520}) {
        size_t elems = (size_t)B * n * pitch;
        c128 *in, *out;
        CK(hipMalloc(&in, elems * 16)); CK(hipMalloc(&out, elems * 16));
        CK(hipMemset(in, 1, elems * 16));
        double gb = 2.0 * B * n * m * 16 / 1e9;
        if (pitch == 513) {
            float ms = time_ms([&] { k_copy<8><<<2048, 256>>>(in, out, elems); });
            printf("contiguous copy            : %.3f ms  %.0f GB/s (r+w)\n", ms, 2.0 * elems * 16 / 1e9 / (ms * 1e-3));
        }
#define TILE(C, EPT, THR, CHUNKS) { int nt = (m + C - 1) / C; \
        float ms = time_ms([&] { k_tile_copy<C, EPT><<<nt * B * CHUNKS, THR>>>(in, out, n, m, pitch, B, nt, CHUNKS); }); \
        printf("pitch %d tile C=%2d (%4d B) EPT=%2d thr=%4d chunks=%d: %.3f ms  %.0f GB/s\n", pitch, C, C * 16, EPT, THR, CHUNKS, ms, gb / (ms * 1e-3)); }
        TILE(8, 16, 512, 1)
        TILE(8, 16, 256, 1)
        TILE(8, 8, 256, 4)
        TILE(8, 4, 256, 8)
        TILE(4, 16, 256, 1)
        TILE(16, 16, 512, 1)
        TILE(16, 8, 256, 4)
        TILE(32, 8, 256, 4)
        TILE(64, 4, 256, 4)
        CK(hipFree(in)); CK(hipFree(out));
    }
    return 0;
}
