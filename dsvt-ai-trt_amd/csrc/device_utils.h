// device_utils.h -- small wave64 / workgroup primitives shared by the HIP kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dsvt {

constexpr int kWave = 64;   // CDNA wavefront width

// compute units of the CURRENT device (host side; cached per device ordinal: a process may drive several devices, and a partitioned or
// smaller part must not inherit the first device's count)
inline int deviceCUs() {
    static int n[64] = {0};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
    if (!n[d]) { hipDeviceProp_t p; n[d] = hipGetDeviceProperties(&p, d) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256; }
    return n[d];
}

__device__ __forceinline__ int laneId() { return threadIdx.x & (kWave - 1); }

template <class T> __device__ __forceinline__ T waveSum(T v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ uint32_t waveMinU(uint32_t v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o, kWave); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ float waveMaxF(float v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
    return v;
}

// sum / maximum over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (the four rows of an MFMA accumulator column), in every lane: gfx950's
// v_permlane16_swap / v_permlane32_swap exchange the rows inside the VALU (with both operands the same register, result[0] + result[1] is
// "mine + my partner's" in every lane); __shfl_xor is a ds_bpermute round trip through the LDS crossbar each.  Same additions, same bits.
__device__ __forceinline__ float rows4Sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float rows4Max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// inclusive scan across the 64 lanes of a wave: six DPP adds -- row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then row_bcast15 into rows 1 and
// 3 and row_bcast31 into rows 2 and 3 (lanes without a source add `old` = 0).  (Six __shfl_up steps are six ds_bpermute round trips through the
// LDS crossbar, ~0.4 us of pure latency per scan: sp_scan's three workgroup scans were most of its 8 us.)
__device__ __forceinline__ uint32_t waveInclusiveScan(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (uint32_t)x;
}

// Exclusive scan of one uint32 per thread over a workgroup of NT threads (NT multiple of 64,
// <= 1024).  `smem` needs NT/64 + 1 words.  Returns the exclusive prefix; *total = sum.
template <int NT>
__device__ __forceinline__ uint32_t blockExclusiveScan(uint32_t v, uint32_t* smem, uint32_t* total) {
    constexpr int NW = NT / kWave;
    const int lane = laneId(), wave = threadIdx.x / kWave;
    uint32_t inc = waveInclusiveScan(v);
    if (lane == kWave - 1) smem[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < NW ? smem[lane] : 0;
        uint32_t wi = waveInclusiveScan(w);
        if (lane < NW) smem[lane] = wi - w;
        if (lane == NW - 1) smem[NW] = wi;
    }
    __syncthreads();
    uint32_t res = smem[wave] + inc - v;
    *total = smem[NW];
    __syncthreads();
    return res;
}

// The same for K values per thread at once (one set of barriers).  `smem` needs K * (NT/64 + 1) words; v[] becomes the exclusive
// prefixes, total[] the sums.
template <int NT, int K>
__device__ __forceinline__ void blockExclusiveScanK(uint32_t (&v)[K], uint32_t* smem, uint32_t (&total)[K]) {
    constexpr int NW = NT / kWave;
    const int lane = laneId(), wave = threadIdx.x / kWave;
    uint32_t inc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { inc[k] = waveInclusiveScan(v[k]); if (lane == kWave - 1) smem[k * (NW + 1) + wave] = inc[k]; }
    __syncthreads();
    for (int k = wave; k < K; k += NW) {            // the wave totals of value k are scanned by wave k mod NW
        const uint32_t w = lane < NW ? smem[k * (NW + 1) + lane] : 0;
        const uint32_t wi = waveInclusiveScan(w);
        if (lane < NW) smem[k * (NW + 1) + lane] = wi - w;
        if (lane == NW - 1) smem[k * (NW + 1) + NW] = wi;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) { const uint32_t res = smem[k * (NW + 1) + wave] + inc[k] - v[k]; total[k] = smem[k * (NW + 1) + NW]; v[k] = res; }
    __syncthreads();
}

// ---- the "x8 plane" of a split-precision tensor (conv.hip ConvArgs::x8_out; written by the convolutions and by Map2Bev) ----
// four values -> four OCP e4m3 bytes (round to nearest even; v_cvt_pk_fp8_f32 turns anything beyond the format's 448 into NaN, hence the clamp)
__device__ __forceinline__ unsigned packE4m3(float a, float b, float c, float d) {
    a = __builtin_fminf(__builtin_fmaxf(a, -448.f), 448.f); b = __builtin_fminf(__builtin_fmaxf(b, -448.f), 448.f);
    c = __builtin_fminf(__builtin_fmaxf(c, -448.f), 448.f); d = __builtin_fminf(__builtin_fmaxf(d, -448.f), 448.f);
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    return (unsigned)p;
}
// the lo parts (v - hi, to four bits: e4m3 of 2^11 (v - hi)) of N (4 or 8) consecutive channels from their packed lo8 bytes
template <int N>
__device__ __forceinline__ void x8DecodeLo(const unsigned (&w)[N / 4], float (&lo)[N]) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        const auto a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], true);
        lo[4 * i] = a[0] * (1.f / 2048.f); lo[4 * i + 1] = a[1] * (1.f / 2048.f); lo[4 * i + 2] = b[0] * (1.f / 2048.f); lo[4 * i + 3] = b[1] * (1.f / 2048.f);
    }
}
// byte offset, inside a pixel's x8 plane, of the lo8 byte of channel c (its hi8 byte is 16 further): [lo8 0..15 | hi8 0..15 | lo8 16..31 | hi8 16..31] per 32 channels
__device__ __forceinline__ int x8Offset(int c) { return (c >> 5) * 64 + ((c >> 4) & 1) * 32 + (c & 15); }

}  // namespace dsvt
