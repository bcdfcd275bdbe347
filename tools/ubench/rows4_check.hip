// rows4Sum / rows4Max (csrc/device_utils.h: v_permlane16_swap + v_permlane32_swap) against the __shfl_xor(16) / __shfl_xor(32) form they replaced: bit for bit, in every lane.
//   hipcc --offload-arch=gfx950 -O3 -I../../dsvt-ai-trt_amd/csrc -I../../include rows4_check.hip -o rows4_check.bin && ./rows4_check.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
#include "plugin_base.h"
#include "device_utils.h"
using namespace dsvt;
__global__ void k(const float* in, float* o) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    float v = in[i];
    float a = v; a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
    float m = v; m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
    o[4 * i] = a; o[4 * i + 1] = rows4Sum(v); o[4 * i + 2] = m; o[4 * i + 3] = rows4Max(v);
}
int main() {
    const int N = 64 * 4096;
    std::vector<float> h(N); std::mt19937 g(1); std::normal_distribution<float> d(0.f, 1.f);
    for (auto& x : h) x = d(g) * std::exp(d(g) * 3.f);
    float *di, *dout; hipMalloc(&di, N * 4); hipMalloc(&dout, N * 16);
    hipMemcpy(di, h.data(), N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(N / 64), dim3(64), 0, 0, di, dout);
    std::vector<float> o(4 * N); hipMemcpy(o.data(), dout, N * 16, hipMemcpyDeviceToHost);
    long bs = 0, bm = 0;
    for (int i = 0; i < N; ++i) { bs += memcmp(&o[4 * i], &o[4 * i + 1], 4) != 0; bm += memcmp(&o[4 * i + 2], &o[4 * i + 3], 4) != 0; }
    printf("lanes whose sum differs: %ld of %d, whose max differs: %ld\n", bs, N, bm);
    return bs || bm;
}
