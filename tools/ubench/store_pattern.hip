// Epilogue store shapes: every lane stores 16 B; a wave instruction covers (a) 16 rows x 64 B or (b) 8 rows x 128 B of a
// row-major [rows][128 halfs] matrix (row stride 256 B).  Does the half-line shape cost store-path time?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/store_pattern.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int FULL>
__global__ void __launch_bounds__(512) k(_Float16* out, int rows_per_wg, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8 v; for (int i = 0; i < 8; ++i) v[i] = (_Float16)(lane + i);
    for (int it = 0; it < iters; ++it) {
        // a wave owns 64 rows x 128 channels per iteration = 16 KB = 16 store instructions
        const size_t row0 = ((size_t)(blockIdx.x * iters + it) * 8 + wave) * 64;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            size_t row; int ch;
            if (FULL) { row = row0 + s * 4 + (lane >> 4); ch = (lane & 15) * 8; }          // 4 rows x 256 B per instruction (whole rows)
            else if (FULL == 0) { row = row0 + (s >> 2) * 16 + (lane & 15); ch = (s & 3) * 32 + (lane >> 4) * 8; }   // 16 rows x 64 B
            *reinterpret_cast<half8*>(out + row * 128 + ch) = v;
        }
    }
}
template <int FULL>
__global__ void __launch_bounds__(512) k2(_Float16* out, int rows_per_wg, int iters) {      // 8 rows x 128 B
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8 v; for (int i = 0; i < 8; ++i) v[i] = (_Float16)(lane + i);
    for (int it = 0; it < iters; ++it) {
        const size_t row0 = ((size_t)(blockIdx.x * iters + it) * 8 + wave) * 64;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const size_t row = row0 + (s >> 1) * 8 + (lane & 7);
            const int ch = (s & 1) * 64 + (lane >> 3) * 8;
            *reinterpret_cast<half8*>(out + row * 128 + ch) = v;
        }
    }
}
int main() {
    const int grid = 256, iters = 2;                   // 256 x 2 x 512 rows x 256 B = 67 MB
    _Float16* out; (void)hipMalloc(&out, (size_t)grid * iters * 512 * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int v = 0; v < 3; ++v) {
        float best = 1e9;
        for (int rep = 0; rep < 8; ++rep) {
            (void)hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, out, 512, iters);
            else if (v == 1) hipLaunchKernelGGL(k2<0>, dim3(grid), dim3(512), 0, 0, out, 512, iters);
            else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, out, 512, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%s: %.1f us for %.1f MB = %.2f TB/s\n", v == 0 ? "16 rows x 64 B per instruction " : v == 1 ? "8 rows x 128 B per instruction " : "4 rows x 256 B per instruction ",
               best * 1e3, grid * iters * 512 * 256 / 1e6, grid * iters * 512 * 256 / (best * 1e-3) / 1e12);
    }
    return 0;
}
