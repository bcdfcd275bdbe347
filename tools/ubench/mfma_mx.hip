// v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950: (1) the operand layout and the scale semantics, checked against a host
// product; (2) what a conv_wide-shaped inner loop gains when the two correction terms of a split-precision product
// (a_lo w_hi + a_hi w_lo) run as fp8 (e4m3) K = 128 blocks instead of two more fp16 MFMAs per k-step.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_mx.hip -o /tmp/mfma_mx && /tmp/mfma_mx
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef int intx4 __attribute__((ext_vector_type(4)));

static float e4m3(unsigned char b) {           // OCP e4m3fn
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + (float)m / 8.f, e - 7);
    return s ? -v : v;
}

__global__ void one(const intx8* a, const intx8* b, floatx4* d, const int* sa, const int* sb) {
    const int l = threadIdx.x;
    floatx4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, 0, sa[l], 0, sb[l]);
    d[l] = c;
}

static void semantics() {
    std::vector<unsigned char> A(64 * 32), B(64 * 32);
    std::vector<int> SA(64), SB(64);
    for (auto& v : A) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v = 0x38; }
    for (auto& v : B) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v = 0x38; }
    for (int l = 0; l < 64; ++l) { SA[l] = 120 + (l % 7) + (l >> 4); SB[l] = 125 + (l % 5); }     // byte 0 = E8M0 exponent
    intx8 *da, *db; floatx4* dd; int *dsa, *dsb;
    (void)hipMalloc(&da, 2048); (void)hipMalloc(&db, 2048); (void)hipMalloc(&dd, 1024); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256);
    (void)hipMemcpy(da, A.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(db, B.data(), 2048, hipMemcpyHostToDevice);
    (void)hipMemcpy(dsa, SA.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, SB.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, da, db, dd, dsa, dsb);
    std::vector<float> D(256);
    (void)hipMemcpy(D.data(), dd, 1024, hipMemcpyDeviceToHost);
    // lane (r, g) of A pairs byte j with byte j of lane (c, g) of B (whatever k index the hardware gives it), and a lane's scale byte
    // applies to its own 32 values
    double worst = 0, scale = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const int col = l & 15, row = 4 * (l >> 4) + i;      // D[row of A][row of B]: lane = B's row, registers = four consecutive rows of A
            double ref = 0;
            for (int g = 0; g < 4; ++g)
                for (int j = 0; j < 32; ++j) {
                    const int la = row + 16 * g, lb = col + 16 * g;
                    ref += (double)e4m3(A[la * 32 + j]) * ldexp(1.0, SA[la] - 127) * (double)e4m3(B[lb * 32 + j]) * ldexp(1.0, SB[lb] - 127);
                }
            worst = fmax(worst, fabs(ref - D[l * 4 + i])); scale = fmax(scale, fabs(ref));
        }
    printf("semantics (lane (r, g) holds 32 values of k-block g with ITS scale byte; D[4 (l >> 4) + i][l & 15]): max |diff| %.3e of %.3e\n", worst, scale);
}

// conv_wide shape: 8 waves, 8 channel tiles x 4 pixel tiles per wave.  MODE 0: 27 fp16 k-steps (split precision as three fp16 products);
// MODE 1: 9 fp16 k-steps + 5 fp8 K = 128 steps (two taps x [lo | hi] of 32 channels each); MODE 2: 9 fp16 steps only (the fp16 frame)
template <int MODE>
__global__ void __launch_bounds__(512, 1) loopk(float* out, long long* cyc, int iters, const _Float16* src) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[128 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 128 * 1024 / 16; i += blockDim.x) reinterpret_cast<half8*>(smem)[i] = reinterpret_cast<const half8*>(src)[i & 4095];
    __syncthreads();
    floatx4 acc[8][4];
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = floatx4{0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    const int scale = 127 - 11;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        constexpr int NF = MODE == 0 ? 27 : 9, NX = MODE == 1 ? 5 : 0;
#pragma unroll 1
        for (int st = 0; st < NF; ++st) {
            const unsigned char* p = smem + (((it + st) & 7) * 12) * 1024 + (lane << 4);
            half8 A[8], B[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) B[b] = *reinterpret_cast<const half8*>(p + (8 + b) * 1024);
#pragma unroll
            for (int a = 0; a < 8; ++a) A[a] = *reinterpret_cast<const half8*>(p + a * 1024);
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[a], B[b], acc[a][b], 0, 0, 0);
            if ((st & 1) == 1) __builtin_amdgcn_s_barrier();
        }
#pragma unroll 1
        for (int st = 0; st < NX; ++st) {
            const unsigned char* p = smem + (((it + st) & 3) * 24) * 1024 + (lane << 5);
            intx8 B[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) B[b] = *reinterpret_cast<const intx8*>(p + (16 + 2 * b) * 1024);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                intx8 A[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) A[a] = *reinterpret_cast<const intx8*>(p + (h * 4 + a) * 2048);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        acc[h * 4 + a][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[a], B[b], acc[h * 4 + a][b], 0, 0, 0, scale, 0, 127);
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> static double run(const _Float16* src) {
    const int grid = 256, iters = 200;
    float* out; long long* cyc;
    (void)hipMalloc(&out, (size_t)grid * 512 * 4); (void)hipMalloc(&cyc, grid * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((loopk<MODE>), dim3(grid), dim3(512), 0, 0, out, cyc, iters, src);
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((loopk<MODE>), dim3(grid), dim3(512), 0, 0, out, cyc, iters, src);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<long long> h(grid); (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += v; c /= grid;
    // "product" flops: one split-precision (or fp16) product of 32 channels x 9 taps x 32 tiles per iteration
    const double flop = (double)grid * 8 * iters * 9 * 32 * 16 * 16 * 32 * 2;
    printf("mode %d (%s): %8.1f us, %7.1f TFLOP/s of products, cycles per iteration %.0f, clock %.3f GHz\n", MODE,
           MODE == 0 ? "3 x fp16" : MODE == 1 ? "fp16 + fp8 MX cross terms" : "fp16 only", ms * 1e3, flop / ms / 1e9, c / iters, c / (ms * 1e6));
    (void)hipFree(out); (void)hipFree(cyc);
    return ms;
}

int main() {
    semantics();
    _Float16* src; (void)hipMalloc(&src, 65536);
    std::vector<_Float16> h(32768);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
    (void)hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
    const double t0 = run<0>(src), t1 = run<1>(src), t2 = run<2>(src);
    printf("3 x fp16 / (fp16 + MX) = %.2f;  (fp16 + MX) / fp16 = %.2f\n", t0 / t1, t1 / t2);
    return 0;
}
