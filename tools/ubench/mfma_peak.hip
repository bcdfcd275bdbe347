// Sustained v_mfma_f32_16x16x32_f16 rate and shader clock of the chip: NW waves per workgroup, one workgroup per CU-slot,
// 32 independent accumulator tiles per wave (the register blocking of conv_wide_kernel<8, 8, ...>), operands from registers.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int LDS_READS>
__global__ void __launch_bounds__(512, 1) k(float* out, long long* cyc, int iters, const _Float16* src) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[96 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 96 * 1024 / 16; i += blockDim.x) reinterpret_cast<half8*>(smem)[i] = reinterpret_cast<const half8*>(src)[i & 1023];
    __syncthreads();
    floatx4 acc[8][4];
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = floatx4{0, 0, 0, 0};
    half8 A[8], B[4];
    for (int a = 0; a < 8; ++a) A[a] = reinterpret_cast<const half8*>(smem)[a * 64 + lane];
    for (int b = 0; b < 4; ++b) B[b] = reinterpret_cast<const half8*>(smem)[(8 + b) * 64 + lane];
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (LDS_READS) {
            const unsigned char* p = smem + ((it & 31) * 12) * 1024 + (lane << 4);
#pragma unroll
            for (int a = 0; a < 8; ++a) A[a] = *reinterpret_cast<const half8*>(p + a * 1024);
#pragma unroll
            for (int b = 0; b < 4; ++b) B[b] = *reinterpret_cast<const half8*>(p + (8 + b) * 1024);
        }
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[a], B[b], acc[a][b], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int R> static void run(const char* name, int grid, int iters, const _Float16* src, bool randomData) {
    float* out; long long* cyc;
    hipMalloc(&out, (size_t)grid * 512 * 4); hipMalloc(&cyc, grid * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<R>, dim3(grid), dim3(512), 0, 0, out, cyc, iters, src);
    hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<R>, dim3(grid), dim3(512), 0, 0, out, cyc, iters, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<long long> h(grid); hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += v; c /= grid;
    const double flop = (double)grid * 8 * iters * 32 * 16 * 16 * 32 * 2;
    printf("%-34s %s data: %8.1f us  %7.1f TFLOP/s  %9.0f cycles/workgroup (s_memtime 100 MHz ticks x?)  cyc/MFMA/SIMD %.2f  eff clock %.3f GHz\n", name, randomData ? "random" : "zero  ", ms * 1e3,
           flop / ms / 1e9, c, c / (iters * 32.0 * 2), c / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main() {
    _Float16* src; hipMalloc(&src, 1024 * 16);
    for (int rnd = 0; rnd < 2; ++rnd) {
        std::vector<_Float16> h(8192);
        for (auto& v : h) v = rnd ? (_Float16)((rand() % 2001 - 1000) * 1e-3f) : (_Float16)0.f;
        hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice);
        run<0>("registers only, 256 workgroups", 256, 4000, src, rnd);
        run<1>("12 ds_read_b128 / 32 MFMA, 256 wg", 256, 4000, src, rnd);
    }
    return 0;
}
