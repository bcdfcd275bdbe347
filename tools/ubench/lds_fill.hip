// How fast can a CU fill its LDS from L2-resident global memory?  (Round 6: conv_rows_kernel's slabs track the number of LDS-DMA requests they carry,
// ~85 cycles of request path per 1 KB piece with eight waves issuing.)  Eight waves per workgroup, one workgroup per CU, each wave moves N pieces of 1 KB:
//   v0  buffer_load_dwordx4 ... lds, contiguous 1 KB rows            (the weight rows)
//   v1  buffer_load_dwordx4 ... lds, 16 segments of 64 B, stride 768 B (a halo piece of a [hi | lo | hi] x 128-channel tensor)
//   v2  global_load_dwordx4 -> VGPRs -> ds_write_b128, contiguous
//   v3  global_load_dwordx4 -> VGPRs -> ds_write_b128, 64 B segments
//   v4  buffer_load_dword ... lds (256 B per request), contiguous
//   v5  v1 with 128 B segments (stride 768 B): a 64-channel phase
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_fill.hip -o /tmp/lf && /tmp/lf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void* lds_t;
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ void __launch_bounds__(512, 1) k(const unsigned char* __restrict__ src, size_t bytes, int N, unsigned* sink, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[128 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, (int)bytes, 0x00020000);
    // every workgroup of an XCD walks the same 2 MB window (L2-resident after the first pass)
    const uint32_t base = (uint32_t)(((blockIdx.x & 7) * 2u) << 20);
    uint32_t voff;
    if (V == 0 || V == 2 || V == 4) voff = (uint32_t)lane * (V == 4 ? 4u : 16u);
    else if (V == 5) voff = (uint32_t)(lane >> 3) * 768u + (uint32_t)(lane & 7) * 16u;
    else voff = (uint32_t)(lane >> 2) * 768u + (uint32_t)(lane & 3) * 16u;
    const unsigned long long t0 = clock64();
    unsigned acc = 0;
    if (V == 0 || V == 1 || V == 4 || V == 5) {
        for (int i = 0; i < N; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t so = base + (uint32_t)(((i + j) * 8 + wave) * (V == 0 ? 1024 : V == 4 ? 256 : 16 * 768) & 0x1FFFFF);
                if constexpr (V == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t)(smem + ((j * 8 + wave) * 1024)), 4, (int)voff, (int)so, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t)(smem + ((j * 8 + wave) * 1024)), 16, (int)voff, (int)so, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        for (int i = 0; i < N; i += 8) {
            u4 r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t so = base + (uint32_t)(((i + j) * 8 + wave) * (V == 2 ? 1024 : 16 * 768) & 0x1FFFFF);
                r[j] = *reinterpret_cast<const u4*>(src + so + voff);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<u4*>(smem + (j * 8 + wave) * 1024 + lane * 16) = r[j];
        }
    }
    __syncthreads();
    acc += *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4);
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t bytes = 64u << 20;
    unsigned char* src; (void)hipMalloc(&src, bytes); (void)hipMemset(src, 1, bytes);
    unsigned* sink; (void)hipMalloc(&sink, 4);
    unsigned long long* cyc; (void)hipMallocManaged(&cyc, 256 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int N = 512;                      // pieces per wave: 8 waves x 512 KB = 4 MB per CU
    const char* names[] = {"buffer_load_dwordx4 lds, contiguous 1 KB      ", "buffer_load_dwordx4 lds, 16 x 64 B stride 768 ", "global_load_dwordx4 + ds_write_b128, contiguous",
                           "global_load_dwordx4 + ds_write_b128, 16 x 64 B ", "buffer_load_dword lds, contiguous 256 B        ", "buffer_load_dwordx4 lds, 8 x 128 B stride 768  "};
    for (int v = 0; v < 6; ++v) {
        float best = 1e9; unsigned long long c = 0;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0);
            switch (v) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, src, bytes, N, sink, cyc); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, src, bytes, N, sink, cyc); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, src, bytes, N, sink, cyc); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, src, bytes, N, sink, cyc); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, src, bytes, N, sink, cyc); break;
                default: hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, src, bytes, N, sink, cyc); break;
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) { best = ms; c = 0; for (int i = 0; i < 256; ++i) c += cyc[i]; c /= 256; }
        }
        const double per_req = (double)c / (8.0 * N);          // CU cycles per request (eight waves share the path)
        const double kb = v == 4 ? 0.25 : 1.0;
        printf("%s: %8.1f us, %7.0f cycles per workgroup, %6.1f cycles of CU time per request, %5.1f B/cycle/CU, %5.2f TB/s chip\n", names[v], best * 1e3, (double)c, per_req, kb * 1024 / per_req,
               256.0 * 8 * N * kb * 1024 / (best * 1e-3) / 1e12);
    }
    return 0;
}
