// do v_cvt_f16_f32 and v_cvt_pk_f16_f32 round exact ties the same way?  (round 3: rare 1-ulp inconsistencies between a packed hi fragment and
// the residual computed from a separately converted hi)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned short* out, int n) {
    int i = threadIdx.x;
    if (i >= n) return;
    float a = in[i], b = in[(i + 1) % n];
    _Float16 s = (_Float16)a;                                        // scalar conversion
    half2v p = {(_Float16)a, (_Float16)b};                            // pair: the compiler may pick v_cvt_pk_f16_f32
    unsigned int raw = *reinterpret_cast<unsigned int*>(&p);
    out[3 * i] = *reinterpret_cast<unsigned short*>(&s); out[3 * i + 1] = (unsigned short)(raw & 0xffff);
    unsigned int pk; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
    out[3 * i + 2] = (unsigned short)(pk & 0xffff);
}
int main() {
    const int n = 8;
    float h[n] = {1.0f + 0x1p-11f, 1.0f + 3 * 0x1p-11f, 0.2f, 0.15f + 0.0f, 0.125f + 0x1p-14f, 0.125f + 3 * 0x1p-14f, 0.3f, -(1.0f + 0x1p-11f)};
    float* din; unsigned short* dout;
    (void)hipMalloc(&din, sizeof(h)); (void)hipMalloc(&dout, 3 * n * 2);
    (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, n);
    unsigned short o[3 * n];
    (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%.10g: scalar 0x%04x  pair 0x%04x  v_cvt_pk_f16_f32 0x%04x\n", h[i], o[3 * i], o[3 * i + 1], o[3 * i + 2]);
    return 0;
}
