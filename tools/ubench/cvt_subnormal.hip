// does the device flush fp16-subnormal results of a float -> half conversion?  hipcc --offload-arch=gfx950 cvt_subnormal.hip -o cvt_subnormal.bin && ./cvt_subnormal.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* in, unsigned short* out, float* mf, int n) {
    int i = threadIdx.x;
    if (i < n) { _Float16 h = (_Float16)in[i]; out[i] = *reinterpret_cast<unsigned short*>(&h);
                 float v = in[i] + 1.0f; _Float16 hi = (_Float16)v; _Float16 lo = (_Float16)(v - (float)hi); out[n + i] = *reinterpret_cast<unsigned short*>(&lo); }
    // MFMA with a subnormal A operand: D = A (16x32, all = x) * B (32x16, all = 1) -> every element 32 x
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)3.0e-5f; b[j] = (_Float16)1.0f; }
    unsigned short sub = 0x0200;            // 2^-15 as a raw fp16 subnormal
    _Float16 s = *reinterpret_cast<_Float16*>(&sub);
    half8 a2; for (int j = 0; j < 8; ++j) a2[j] = s;
    floatx4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b, floatx4{0, 0, 0, 0}, 0, 0, 0);
    if (i == 0) { mf[0] = d[0]; mf[1] = 32.0f * 3.0517578125e-05f; }
}
int main() {
    const int n = 6; float h[n] = {3.0e-5f, 6.0e-5f, 6.2e-5f, 1.0e-5f, 5.0e-8f, 1.0e-7f};
    float* din; unsigned short* dout; float* mf;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, 2 * n * 2); hipMalloc(&mf, 8);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, mf, n);
    unsigned short o[2 * n]; float m[2];
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost); hipMemcpy(m, mf, 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("cvt(%g) = 0x%04x   lo(1 + %g) = 0x%04x\n", h[i], o[i], h[i], o[n + i]);
    printf("mfma with subnormal A (2^-15 x 32 ones): %g (expected %g)\n", m[0], m[1]);
    return 0;
}
