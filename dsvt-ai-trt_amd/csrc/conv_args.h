// conv_args.h -- what the convolution translation units (conv.hip, conv_rows.hip) share: the launch arguments, the split-precision store helpers
// and the slab-end wait.  (Moved out of conv.hip in round 6, unchanged.)
#pragma once
#include "plugin_base.h"
#include "device_utils.h"

namespace dsvt {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int CNB = 128;          // output channels per workgroup
constexpr int CPX = 128;          // pixels per workgroup
constexpr int CNT = CNB / 16;     // n-tiles

struct ConvArgs {
    const _Float16* in; int H, W, Cin;           // NHWC input
    const _Float16* wt;                          // [CoutRows][KH*KW][Cin]
    const float* bias;                           // [Cout] (per real output channel) or nullptr
    const _Float16* res; int res_ld;             // residual NHWC on the OUTPUT grid, or nullptr
    void* out; int out_ld, out_coff, out_f32;    // NHWC output, channel stride / offset
    int Ho, Wo;                                  // GEMM pixel grid (= conv output grid before pixel shuffle)
    int CoutRows;                                // rows of wt = up*up*Cout
    int Cout;                                    // real output channels
    int KH, KW, stride, pad, up, relu;
    int wide;                                    // halo kernel: 16-byte epilogue accesses are legal (channel strides / offsets % 8, Cout % 16, fp16 output)
    int nb;                                      // images per launch (>= 1): image b = pixels b*H*W .. of `in`, b*Ho*up*Wo*up .. of `out` / `res`
    // split precision (round 3): activations travel as the fp16 triple [hi | lo | hi] along the channel axis (hi = fp16(v), lo = fp16(v - hi);
    // the third plane repeats hi so that ONE plain convolution over 3 Cin channels with weight rows [w_hi | w_hi | w_lo] is the fp32-grade
    // product hi w_hi + lo w_hi + hi w_lo).  split_out > 0: the epilogue writes the three planes itself, split_out = plane stride in channels
    // (out_ld = 3 * split_out); res_split > 0: the residual tensor is such a triple, its value is hi + lo (exact to 2^-22), plane stride res_split.
    // The kernels are instantiated twice (template parameter SPL): the fp16 frame's instantiations do not carry the split epilogue's registers.
    int split_out, res_split;
    // round 4, the cheaper fp32-grade product: the two correction terms lo w_hi + hi w_lo need ~5 bits, so they run as OCP fp8 (e4m3) blocks of
    // v_mfma_scale_f32_16x16x128_f8f6f4 (2.4 x the fp16 rate, tools/ubench/mfma_mx.hip) instead of two more fp16 MFMAs per k-step.  The third plane of
    // a split tensor then holds the fp8 operands ("x8 plane"): per 32 channels 64 bytes [lo8 0..15 | hi8 0..15 | lo8 16..31 | hi8 16..31] with
    // lo8 = e4m3(2^11 (v - hi)), hi8 = e4m3(v), both saturated at +-448 (x8Store).
    //   x8_out : 1 = the epilogue writes [hi | lo | x8] instead of [hi | lo | hi]; 2 = [hi | - | x8]: the lo plane is left untouched, for tensors that only
    //            the fp16 + fp8 K loop reads (split_output = 3)
    //   alias3 : (kernels that still walk three fp16 planes) first channel of the third plane, whose phases read plane 0 instead; 0 = off
    //   xscale : (conv_wide_kernel<.., MX>) E8M0 scale byte per weight row: 127 - 11 - e with w_hi8 = e4m3(2^e w_hi), w_lo8 = e4m3(2^(e + 11) w_lo)
    int x8_out, alias3;
    int res_x8;                                  // the residual triple has no lo plane ([hi | - | x8]): its value is hi + 2^-11 lo8 (to 2^-15 relative)
    const unsigned char* xscale;
    unsigned long long* trace;                   // debugging (DSVT_CONV_TRACE=1, tools/trace_conv.py): s_memtime stamps of waves 0 and NW/2, or nullptr
    int variant;                                 // plugin field "kernel_variant": 0 = the launcher's choice; 1 = never conv_rows_kernel (tests: the round-5 kernel of the same layer, bit for bit)
};
constexpr int CONV_TRACE_N = 512;               // stamps per traced wave


// hi / lo planes of N consecutive channels (saturated: |v| beyond the fp16 range gives +-65504, not inf)
template <int N, class HV>
__device__ __forceinline__ void splitPlanes(const float (&v)[N], HV& hi, HV& lo) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const _Float16 h = (_Float16)__builtin_fminf(__builtin_fmaxf(v[i], -65504.f), 65504.f);
        hi[i] = h; lo[i] = (_Float16)__builtin_fminf(__builtin_fmaxf(v[i] - (float)h, -65504.f), 65504.f);
    }
}
// x8 plane of N (4 or 8) consecutive channels starting at plane channel c (c % N == 0); hi = the fp16 plane values already computed
template <int N, class HV>
__device__ __forceinline__ void x8Store(unsigned char* xplane, int c, const float (&v)[N], const HV& hi) {
    unsigned lo8[N / 4], hi8[N / 4];
#pragma unroll
    for (int i = 0; i < N; i += 4) {
        lo8[i / 4] = packE4m3((v[i] - (float)hi[i]) * 2048.f, (v[i + 1] - (float)hi[i + 1]) * 2048.f, (v[i + 2] - (float)hi[i + 2]) * 2048.f, (v[i + 3] - (float)hi[i + 3]) * 2048.f);
        hi8[i / 4] = packE4m3(v[i], v[i + 1], v[i + 2], v[i + 3]);
    }
    unsigned char* p = xplane + x8Offset(c);
    if constexpr (N == 8) { *reinterpret_cast<uint2*>(p) = make_uint2(lo8[0], lo8[1]); *reinterpret_cast<uint2*>(p + 16) = make_uint2(hi8[0], hi8[1]); }
    else { *reinterpret_cast<unsigned*>(p) = lo8[0]; *reinterpret_cast<unsigned*>(p + 16) = hi8[0]; }
}
// store of eight consecutive channels of one output pixel: plain fp16, or (SPL) the [hi | lo | hi] / [hi | lo | x8] planes when split_out is set
template <bool SPL>
__device__ __forceinline__ void storeHalf8(const ConvArgs& a, const float (&v)[8], size_t opix, int co) {
    _Float16* o = static_cast<_Float16*>(a.out) + opix * a.out_ld + a.out_coff + co;
    if (SPL && a.split_out) {
        half8 hi, lo;
        splitPlanes<8>(v, hi, lo);
        *reinterpret_cast<half8*>(o) = hi;
        if (a.x8_out != 2) *reinterpret_cast<half8*>(o + a.split_out) = lo;          // (x8_out 2: no consumer reads the lo plane -- a third of the store burst)
        if (a.x8_out == 3) {}                                                         // (x8_out 3: the tensor is only ever a residual -- hi + lo --: no third plane)
        else if (a.x8_out) x8Store<8>(reinterpret_cast<unsigned char*>(static_cast<_Float16*>(a.out) + opix * a.out_ld + 2 * a.split_out), a.out_coff + co, v, hi);
        else *reinterpret_cast<half8*>(o + 2 * a.split_out) = hi;
    } else {
        half8 h;
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = (_Float16)v[i];
        *reinterpret_cast<half8*>(o) = h;
    }
}


typedef __attribute__((address_space(1))) const void* glds_src_t;
typedef __attribute__((address_space(3))) void* glds_dst_t;

// end of a K slab: this wave's LDS-DMA requests older than the youngest `keep` have landed, then the workgroup barrier
// publishes them (LDS-DMA data is ordered for a ds_read only by the issuing wave's vmcnt followed by a barrier)
__device__ __forceinline__ void slabWait(int keep) {
    switch (keep) {          // wave-uniform
        case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void slabBarrier(int keep) {
    slabWait(keep);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// conv_rows.hip: the ky-row-slab kernel of the three-product 3 x 3 stride-1 layers with 128-channel chunks (see its header); returns < 0 when the
// layer is not one of its shapes (the caller falls back to conv_wide_kernel)
bool convRowsEligible(const ConvArgs& a, int ncu);
int launchConvRows(const ConvArgs& a, const _Float16* Wp, int ncu, hipStream_t stream);
bool convRows64Eligible(const ConvArgs& a, int ncu);          // (64-channel chunks on 24-row items)
int launchConvRows64(const ConvArgs& a, const _Float16* Wp, int ncu, hipStream_t stream);
bool convRowsSmallEligible(const ConvArgs& a);                    // (64-channel chunks on 16- or 8-row items: launches that do not fill the chip with the items above)
int launchConvRowsSmall(const ConvArgs& a, const _Float16* Wp, int rowsPerWave, int ncu, hipStream_t stream);

}  // namespace dsvt
