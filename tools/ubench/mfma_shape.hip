// The same LDS-fed loop on the two dense fp16 MFMA shapes: does v_mfma_f32_32x32x16_f16 (half the operand-register reads and half the issued instructions per flop)
// deliver more under the power budget than v_mfma_f32_16x16x32_f16?  Wave tile 128 x 64 in both cases: 12 ds_read_b128 per K = 32, 128 accumulator registers;
// 8 waves per CU, random fp16 data.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int BARRIER, int LDSREAD>
__global__ void __launch_bounds__(512, 1) k(float* out, long long* cyc, int iters, const _Float16* src, int seed) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[96 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 96 * 1024 / 16; i += blockDim.x) reinterpret_cast<half8*>(smem)[i] = reinterpret_cast<const half8*>(src)[i & 1023];
    __syncthreads();
    floatx4 acc[8][4];
    floatx16 acc32[4][2];
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = floatx4{0, 0, 0, 0};
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) acc32[a][b][i] = 0.f;
    half8 A[8], B[4];
    for (int a = 0; a < 8; ++a) A[a] = reinterpret_cast<const half8*>(smem)[lane + 64 * a];
    for (int b = 0; b < 4; ++b) B[b] = reinterpret_cast<const half8*>(smem)[lane + 64 * (8 + b)];
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {                // one K = 32 step of the 128 x 64 tile
            const unsigned char* p = smem + (((it * 2 + st) & 7) * 12) * 1024 + (lane << 4);
            if (LDSREAD) {
#pragma unroll
                for (int b = 0; b < 4; ++b) B[b] = *reinterpret_cast<const half8*>(p + (8 + b) * 1024);
#pragma unroll
                for (int a = 0; a < 8; ++a) A[a] = *reinterpret_cast<const half8*>(p + a * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (SHAPE == 16) {
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[a], B[b], acc[a][b], 0, 0, 0);
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc32[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[ks * 4 + a], B[ks * 2 + b], acc32[a][b], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) s += acc32[a][b][0] + acc32[a][b][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int BAR, int RD> static void run(const _Float16* src) {
    const int grid = 256, iters = 2000;
    float* out; long long* cyc;
    (void)hipMalloc(&out, (size_t)grid * 512 * 4); (void)hipMalloc(&cyc, grid * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, BAR, RD>), dim3(grid), dim3(512), 0, 0, out, cyc, iters, src, 1);
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<SHAPE, BAR, RD>), dim3(grid), dim3(512), 0, 0, out, cyc, iters, src, 1);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<long long> h(grid); (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += v; c /= grid;
    const double flop = (double)grid * 8 * iters * 2 * 128 * 64 * 32 * 2;
    printf("shape %2d  barrier %d  lds reads %d: %8.1f us %7.1f TFLOP/s  cycles per K=32 tile step per SIMD %.1f  clock %.3f GHz\n", SHAPE, BAR, RD, ms * 1e3, flop / ms / 1e9,
           c / (iters * 2.0 * 2), c / (ms * 1e6));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    _Float16* src; (void)hipMalloc(&src, 1024 * 16);
    std::vector<_Float16> h(8192);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
    (void)hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<16, 0, 0>(src); run<32, 0, 0>(src);
        run<16, 0, 1>(src); run<32, 0, 1>(src);
        run<16, 1, 1>(src); run<32, 1, 1>(src);
    }
    return 0;
}
