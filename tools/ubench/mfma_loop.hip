// What an MFMA inner loop of the conv_wide shape loses to (a) a workgroup barrier every 64 MFMAs per wave, (b) scalar / vector
// bookkeeping instructions between the MFMA batches.  8 waves per CU, 32 accumulator tiles per wave, 12 ds_read_b128 per 32 MFMAs.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_loop.hip -o /tmp/mfma_loop && /tmp/mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int BARRIER, int FILL>       // FILL: extra dependent integer instructions per batch of 16 MFMAs (half SALU, half VALU)
__global__ void __launch_bounds__(512, 1) k(float* out, long long* cyc, int iters, const _Float16* src, int seed) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[96 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 96 * 1024 / 16; i += blockDim.x) reinterpret_cast<half8*>(smem)[i] = reinterpret_cast<const half8*>(src)[i & 1023];
    __syncthreads();
    floatx4 acc[8][4];
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = floatx4{0, 0, 0, 0};
    half8 A[8], B[4];
    int sx = seed, vx = lane + seed;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const unsigned char* p = smem + (((it * 2 + st) & 7) * 12) * 1024 + (lane << 4) + ((sx + vx) & 0);
#pragma unroll
            for (int b = 0; b < 4; ++b) B[b] = *reinterpret_cast<const half8*>(p + (8 + b) * 1024);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int a = 0; a < 4; ++a) A[a] = *reinterpret_cast<const half8*>(p + (h * 4 + a) * 1024);
#pragma unroll
                for (int f = 0; f < FILL / 2; ++f) {
                    asm volatile("s_mul_i32 %0, %0, 0x55555557" : "+s"(sx));
                    asm volatile("v_mad_u32_u24 %0, %0, 3, %0" : "+v"(vx));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[h * 4 + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[a], B[b], acc[h * 4 + a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = (float)(sx + vx);
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int BAR, int FILL> static void run(const _Float16* src) {
    const int grid = 256, iters = 2000;
    float* out; long long* cyc;
    (void)hipMalloc(&out, (size_t)grid * 512 * 4); (void)hipMalloc(&cyc, grid * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<BAR, FILL>), dim3(grid), dim3(512), 0, 0, out, cyc, iters, src, 1);
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<BAR, FILL>), dim3(grid), dim3(512), 0, 0, out, cyc, iters, src, 1);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<long long> h(grid); (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += v; c /= grid;
    const double flop = (double)grid * 8 * iters * 64 * 16 * 16 * 32 * 2;
    printf("barrier/64 MFMA %d  fill %2d instr / 16 MFMA: %8.1f us %7.1f TFLOP/s  cycles per MFMA per SIMD %.2f  clock %.3f GHz\n", BAR, FILL, ms * 1e3, flop / ms / 1e9,
           c / (iters * 64.0 * 2), c / (ms * 1e6));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    _Float16* src; (void)hipMalloc(&src, 1024 * 16);
    std::vector<_Float16> h(8192);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
    (void)hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice);
    run<0, 0>(src); run<1, 0>(src); run<0, 16>(src); run<0, 40>(src); run<0, 80>(src); run<1, 40>(src); run<1, 80>(src);
    return 0;
}
