// pfn.hip -- DsvtPillarFeatureNetPlugin: the two PFN layers + both scatter-max reductions of the voxel feature
// encoder in ONE launch that never writes a per-point activation.
//
// Reference wiring (src/dsvt-ai-trt.cpp:565-589): x0 = ReLU(BN(FC0(f))) [Nk,96] -> TorchScatterMax -> concat
// [x0 | max_pillar(x0)] [Nk,192] -> x1 = ReLU(BN(FC1(cat))) [Nk,192] -> TorchScatterMax -> pillar features [P,192].
// As separate launches that is ~1 GB of traffic per 180k-point frame (six kernels, 0.33 ms).  Three facts remove it:
//   * FC1 is linear in the two halves of its input:  FC1(cat) = W1a x0 + W1b m_p + b1 with m_p = max_pillar(x0);
//     the second term is per PILLAR;
//   * ReLU and "+ constant" are monotone, so  max_points ReLU(W1a x0 + t_p) = ReLU(max_points(W1a x0) + t_p):
//     both per-point maxima, m_p and U_p = max_points(W1a x0), come out of ONE pass over the points;
//   * x0 is consumed from registers (the layer-0 tile is the B operand of layer 1, see mlp.hip).
// Tiles (round 2): the median pillar holds ONE point and the mean 4.8, so "one pillar per 16-point MFMA tile" spent 3.6x the matrix work on
// padding.  Pillars with <= 4 points are packed FOUR to a tile -- pillar q takes points 4q..4q+3, which in the x0 / u tile layouts is
// exactly lane group g = q, so their maxima need no cross-lane step at all -- and only the larger pillars keep whole tiles (1..3 each):
// 38k tiles -> 17k on the 180k-point frame.  A row of a GEMM tile depends on nothing but its own point, so the features are the same bits.
// A workgroup owns 16 consecutive pillars; the compact point ids of a pillar are consecutive (Points2Features' canonical
// order: pillar-major, then slot) and every 16-point MFMA tile belongs to one pillar, so the maxima are plain register
// reductions: nothing is atomic, in LDS or in global memory.  After the point pass the workgroup computes t = W1b m (16 x 96 x 192 on the matrix cores, W1b fragments straight from L2) and
// writes  vfeat_p = ReLU(U_p + t_p + b1)  as fp32 and fp16.
// Arithmetic: layer 0 on v_mfma_f32_16x16x4_f32 (fp32: the inputs are raw metric coordinates), both halves of layer 1 on
// v_mfma_f32_16x16x32_f16 (operands rounded to fp16, fp32 accumulate); maxima, bias, ReLU in fp32.
#include "plugin_base.h"
#include "device_utils.h"
#include <cstdio>
#include <cstdlib>

namespace dsvt {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int PF_IN = 10, PF_C0 = 96, PF_C1 = 192;
constexpr int PF_PB = 16;              // pillars per workgroup (= one MFMA row tile of the per-pillar GEMM)
constexpr int PF_NW = 4;               // waves per workgroup (four pillars of a group each; 8 waves measured 12 % slower)

struct PfnArgs {
    const float* feat;                 // [Nk, 10]
    const uint32_t* pidx; int T;       // [P, T] compact point ids (first entry = the pillar's first row)
    const uint32_t* pcnt;              // [P]
    const uint32_t* pillar_num; int max_pillars;
    const float* w0; const float* b0;  // [96][12] (k padded with zeros), [96]
    const _Float16* w1a;               // fragment-ordered [3 k-steps][12 tiles][64 lanes][8], k-permuted (chained operand); SPLIT: the same image of w_lo follows
    const _Float16* w1b;               // fragment-ordered [3 k-steps][12 tiles][64 lanes][8], natural k; SPLIT: the same image of w_lo follows
    const float* b1;                   // [192]
    float* out; _Float16* out16;       // vfeat [P,192] + fp16 copy
    unsigned long long* trace;         // debugging: phase timestamps of workgroup 0
    int dbg;                           // timing ablations (wrong results): 1 no m atomics, 2 no U atomics, 4 no layer-1 MFMA, 8 no epilogue
    int pack;                          // 1: pillars with <= 4 points share tiles (default); 0: one pillar per tile (round-1 layout, A/B switch)
};

__device__ __forceinline__ uint32_t pfnKey(float f) {            // larger float <=> larger key; every key > 0
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float pfnUnkey(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// max over the four 16-lane groups (lanes l, l ^ 16, l ^ 32, l ^ 48) on the VALU: v_permlane16_swap / v_permlane32_swap with both
// operands equal broadcast the even / odd rows (lower / upper half) into two registers
__device__ __forceinline__ float maxOverLaneGroups(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float w = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const uint32_t x = __float_as_uint(w);
    const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// SPLIT (round 3, the fp32-grade mode): both halves of layer 1 with hi / lo fp16 operand pairs, three MFMAs per product (see
// linear.hip linear_split_rows_kernel); layer 0 is fp32 MFMA either way.  W1a hi + lo = 72 KB of LDS: one workgroup per CU.
template <bool SPLIT>
__global__ void __launch_bounds__(64 * PF_NW)
pfn_kernel(PfnArgs a)
{
    // (row strides 100 / 196 dwords: with 96 / 192 the sixteen pillar rows of a 16-byte read fall on two / one bank group,
    // SQ_LDS_BANK_CONFLICT was 40 % of the kernel's LDS cycles)
    // sM holds fp16: it is only ever read as the fp16 B operand of the per-pillar GEMM (rounding at the store = rounding at the read).
    // (52.7 KB of LDS in all; a third resident workgroup per CU was measured: 82 vs 79 us, the group loop is not latency-starved.)
    constexpr int SM_LD = PF_C0 + 8, SU_LD = PF_C1 + 4;
    __shared__ __attribute__((aligned(16))) _Float16 sM[(SPLIT ? 2 : 1) * PF_PB * SM_LD];   // max_pillar(x0) as fp16 (SPLIT: hi rows, then lo rows), 208-byte rows (conflict-free b128 reads)  3.3 KB
    __shared__ __attribute__((aligned(16))) uint32_t sU[PF_PB * SU_LD];   // max_pillar(W1a x0), float bits                 12 KB
    __shared__ uint32_t sStart[PF_PB + 1];
    __shared__ uint32_t sSmall[PF_PB], sLarge[PF_PB], sNum[2];           // pillars of the group with <= 4 points / more, and how many of each
    __shared__ __attribute__((aligned(16))) _Float16 sW1[(SPLIT ? 2 : 1) * 3 * 12 * 512];                              // 36 KB (SPLIT: w_hi image, then w_lo image)
    uint32_t P = *a.pillar_num; if (P > (uint32_t)a.max_pillars) P = a.max_pillars;
    const uint32_t ngroups = (P + PF_PB - 1) / PF_PB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    if (blockIdx.x >= ngroups) return;
    // persistent workgroup: W1a (36 KB) and the layer-0 fragments are loaded once, groups of 16 pillars round-robin
    for (int i = tid; i < (SPLIT ? 2 : 1) * 3 * 12 * 64; i += 64 * PF_NW)
        *reinterpret_cast<uint4*>(&sW1[i * 8]) = *reinterpret_cast<const uint4*>(a.w1a + (size_t)i * 8);
    // layer-0 weights as MFMA A fragments: lane (r, g) holds W0[16t + r][4ks + g]
    float w0f[6][3], b0f[6][4];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) w0f[t][ks] = a.w0[(16 * t + r) * 12 + 4 * ks + g];
        const float4 b = *reinterpret_cast<const float4*>(a.b0 + 16 * t + 4 * g);
        b0f[t][0] = b.x; b0f[t][1] = b.y; b0f[t][2] = b.z; b0f[t][3] = b.w;
    }
    int nmark = 0;
    auto mark = [&]() { if (a.trace && blockIdx.x == 0 && tid == 0 && nmark < 32) a.trace[nmark] = clock64(); ++nmark; };
    mark();
    // first row of pillar pb0 + tid of group g (the sentinel entry, tid == npil, is one past the last pillar's rows).  A group is ~8 us of
    // mostly dependent round trips, so the NEXT group's entries are requested before this group's point pass and wait in a register.
    auto startOf = [&](uint32_t g) -> uint32_t {
        if (g >= ngroups) return 0u;
        const uint32_t pb = g * PF_PB;
        const int np = P - pb < (uint32_t)PF_PB ? (int)(P - pb) : PF_PB;
        if (tid > np) return 0u;
        const uint32_t p = pb + (tid < np ? tid : np - 1);
        const uint32_t s = a.pidx[(size_t)p * a.T];
        return tid < np ? s : s + a.pcnt[p];
    };
    uint32_t myStart = startOf(blockIdx.x);
  for (uint32_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    __syncthreads();                                 // everyone is done with the previous group's tables
    mark();
    const uint32_t pb0 = grp * PF_PB;
    const int npil = P - pb0 < (uint32_t)PF_PB ? (int)(P - pb0) : PF_PB;

    for (int i = tid; i < (SPLIT ? 2 : 1) * PF_PB * SM_LD; i += 64 * PF_NW) sM[i] = (_Float16)0.f;
    for (int i = tid; i < PF_PB * SU_LD; i += 64 * PF_NW) sU[i] = 0u;
    if (tid <= npil) sStart[tid] = myStart;
    myStart = startOf(grp + gridDim.x);               // (in flight under this group's work)
    __syncthreads();
    // ---- point pass: one pillar at a time per wave, 16 points per MFMA tile (a short tile is padded with copies of the
    // pillar's first point: duplicates do not change a maximum).  Every tile belongs to ONE pillar, so the pillar maxima are
    // plain reductions -- ds_max atomics from 16 lanes that share a pillar serialise at ~100 cycles per instruction --
    //   x0^T tile  = W0 (A) x f (B)      transposed layout, lane = point: the k-permuted B... operand of layer 1 (see mlp.hip)
    //   x0   tile  = f (A) x W0 (B)      lane (r', g') = column r', points 4g'..4g'+3: max over points = 3 in-register max + 2 shuffles
    //   u    tile  = x0 (A) x W1a (B)    same layout; the chained fragment works as the A operand just as well
    float b0c[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) b0c[t] = a.b0[16 * t + r];
    mark();
    // ---- work units of this group: packed tiles of up to four small pillars first, then the larger pillars (1..3 tiles each)
    if (tid < kWave) {
        const bool valid = tid < npil;
        const uint32_t c = valid ? sStart[tid + 1] - sStart[tid] : 0u;
        const bool small = valid && a.pack && c <= 4u;
        const unsigned long long ms = __ballot(small), ml = __ballot(valid && !small);
        const unsigned long long below = (1ull << tid) - 1ull;
        if (small) sSmall[__popcll(ms & below)] = (uint32_t)tid;
        else if (valid) sLarge[__popcll(ml & below)] = (uint32_t)tid;
        if (tid == 0) { sNum[0] = (uint32_t)__popcll(ms); sNum[1] = (uint32_t)__popcll(ml); }
    }
    __syncthreads();
    const int nsmall = (int)sNum[0], nlarge = (int)sNum[1], npacked = (nsmall + 3) >> 2, nunits = npacked + nlarge;
    // tiles of unit u; row of the point lane r works on in tile (u, t) (padding = the pillar's first point: duplicates do not change a maximum)
    auto tilesOf = [&](int u) -> int {
        if (u < npacked) return 1;
        const uint32_t pl_ = sLarge[u - npacked];
        return (int)((sStart[pl_ + 1] - sStart[pl_] + 15u) >> 4);
    };
    auto rowOf = [&](int u, int t) -> uint32_t {
        if (u < npacked) {
            const int idx = 4 * u + (r >> 2);
            const uint32_t pl_ = sSmall[idx < nsmall ? idx : 4 * u];
            const uint32_t s_ = sStart[pl_], c_ = sStart[pl_ + 1] - s_, i_ = (uint32_t)(r & 3);
            return s_ + (i_ < c_ ? i_ : 0u);
        }
        const uint32_t pl_ = sLarge[u - npacked];
        const uint32_t s_ = sStart[pl_], e_ = sStart[pl_ + 1], row = s_ + 16u * (uint32_t)t + (uint32_t)r;
        return row < e_ ? row : s_;
    };
    auto loadRow = [&](uint32_t row, float (&fbv)[3]) {
        const float* fr = a.feat + (size_t)row * PF_IN;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) { const int k = 4 * ks + g; fbv[ks] = k < PF_IN ? fr[k] : 0.f; }
    };
    int u = wave, t = 0;
    bool have = u < nunits;
    float fb[3], fnext[3];
    if (have) loadRow(rowOf(u, 0), fnext);
    float mx0[6], mxu[12];
#pragma unroll
    for (int k = 0; k < 6; ++k) mx0[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) mxu[k] = -INFINITY;
    while (have) {
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) fb[ks] = fnext[ks];
        const int cu = u, nt = tilesOf(cu);
        const bool lastOfUnit = t + 1 >= nt;
        // the tile after this one: its point rows are requested before the current tile's MFMAs
        int nu = cu, ntile = t + 1;
        if (lastOfUnit) { nu = cu + PF_NW; ntile = 0; }
        const bool haveNext = nu < nunits;
        if (haveNext) loadRow(rowOf(nu, ntile), fnext);

        floatx4 x0[6];
        float m0[6], mu[12];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            x0[k] = floatx4{b0f[k][0], b0f[k][1], b0f[k][2], b0f[k][3]};
            floatx4 d0 = {b0c[k], b0c[k], b0c[k], b0c[k]};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                x0[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0f[k][ks], fb[ks], x0[k], 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[ks], w0f[k][ks], d0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) x0[k][i] = fmaxf(x0[k][i], 0.f);             // ReLU (:144)
            m0[k] = fmaxf(fmaxf(d0[0], d0[1]), fmaxf(d0[2], d0[3]));                 // this lane group's four points
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) mu[k] = -INFINITY;
        if (!(a.dbg & 4)) {
            half8 f1[3], f1l[3];
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) {
                _Float16 h[8], l[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v0 = x0[2 * s_][i], v1 = x0[2 * s_ + 1][i];
                    if constexpr (SPLIT) {
                        h[i] = (_Float16)fminf(v0, 65504.f); h[4 + i] = (_Float16)fminf(v1, 65504.f);      // (x0 >= 0 after the ReLU)
                        // (the residuals as fp32 values first; the hi parts come from the clamped, i.e. materialised, value: see attention.hip on hipcc's fused conversions)
                        float d0 = v0 - (float)h[i], d1 = v1 - (float)h[4 + i];
                        asm volatile("" : "+v"(d0), "+v"(d1));
                        l[i] = (_Float16)d0; l[4 + i] = (_Float16)d1;
                    } else {                 // the fp16 frame: round 2's instruction stream (no clamp, no residuals)
                        h[i] = (_Float16)v0; h[4 + i] = (_Float16)v1; l[i] = l[4 + i] = (_Float16)0.f;
                    }
                }
                f1[s_] = half8{h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
                f1l[s_] = half8{l[0], l[1], l[2], l[3], l[4], l[5], l[6], l[7]};
            }
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_) {
                    const half8 wh = *reinterpret_cast<const half8*>(&sW1[((s_ * 12 + k) * 64 + lane) * 8]);
                    if constexpr (SPLIT) {
                        const half8 wl = *reinterpret_cast<const half8*>(&sW1[(((3 + s_) * 12 + k) * 64 + lane) * 8]);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1[s_], wl, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1l[s_], wh, acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1[s_], wh, acc, 0, 0, 0);
                }
                mu[k] = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
            }
        }
        if (cu < npacked) {
            // packed tile: lane group g holds the maxima of ITS pillar (points 4g..4g+3): every lane writes column 16k + r of that row
            const int idx = 4 * cu + g;
            if (idx < nsmall) {
                const uint32_t pl_ = sSmall[idx];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const float mv = fmaxf(m0[k], 0.f);                                                                   // max(ReLU(.)) = max(0, max(.))
                    const _Float16 mh = (_Float16)fminf(mv, 65504.f);
                    sM[pl_ * SM_LD + 16 * k + r] = mh;
                    if constexpr (SPLIT) { float dm = mv - (float)mh; asm volatile("" : "+v"(dm)); sM[(PF_PB + pl_) * SM_LD + 16 * k + r] = (_Float16)dm; }
                }
#pragma unroll
                for (int k = 0; k < 12; ++k) sU[pl_ * SU_LD + 16 * k + r] = __float_as_uint(mu[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) mx0[k] = fmaxf(mx0[k], m0[k]);
#pragma unroll
            for (int k = 0; k < 12; ++k) mxu[k] = fmaxf(mxu[k], mu[k]);
            if (lastOfUnit) {
                // across the four lane groups (points 4g'..4g'+3 of every tile), then lane group 0 owns column 16k + r
                const uint32_t pl_ = sLarge[cu - npacked];
#pragma unroll
                for (int k = 0; k < 6; ++k) mx0[k] = maxOverLaneGroups(mx0[k]);
#pragma unroll
                for (int k = 0; k < 12; ++k) mxu[k] = maxOverLaneGroups(mxu[k]);
                if (g == 0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const _Float16 mh = (_Float16)fminf(mx0[k], 65504.f);
                        sM[pl_ * SM_LD + 16 * k + r] = mh;
                        if constexpr (SPLIT) { float dm = mx0[k] - (float)mh; asm volatile("" : "+v"(dm)); sM[(PF_PB + pl_) * SM_LD + 16 * k + r] = (_Float16)dm; }
                    }
#pragma unroll
                    for (int k = 0; k < 12; ++k) sU[pl_ * SU_LD + 16 * k + r] = __float_as_uint(mxu[k]);
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) mx0[k] = 0.f;
#pragma unroll
                for (int k = 0; k < 12; ++k) mxu[k] = -INFINITY;
            }
        }
        u = nu; t = ntile; have = haveNext;
    }
    mark();
    __syncthreads();
    mark();
    // ---- per-pillar half: t^T = W1b m^T (16 pillars x 96 -> 192); wave w owns column tiles w, w + 8 ----------------------------
    half8 mf[3], mfl[3];                             // B fragment: lane (r, g) holds m[pillar r][32s + 8g + j]
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        mf[s] = *reinterpret_cast<const half8*>(&sM[r * SM_LD + 32 * s + 8 * g]);
        if constexpr (SPLIT) mfl[s] = *reinterpret_cast<const half8*>(&sM[(PF_PB + r) * SM_LD + 32 * s + 8 * g]);
    }
    const bool pv = r < npil && !(a.dbg & 8);
#pragma unroll
    for (int tt = 0; tt < (12 + PF_NW - 1) / PF_NW; ++tt) {
        const int t = wave + tt * PF_NW;
        if (t >= 12) break;
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const half8 wh = *reinterpret_cast<const half8*>(a.w1b + ((size_t)(s * 12 + t) * 64 + lane) * 8);
            if constexpr (SPLIT) {
                const half8 wl = *reinterpret_cast<const half8*>(a.w1b + ((size_t)((3 + s) * 12 + t) * 64 + lane) * 8);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, mf[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, mfl[s], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, mf[s], acc, 0, 0, 0);
        }
        if (pv) {                                    // lane (r, g): pillar r, columns 16t + 4g + i
            const int col = 16 * t + 4 * g;
            const float4 b = *reinterpret_cast<const float4*>(a.b1 + col);
            const uint4 u = *reinterpret_cast<const uint4*>(&sU[r * SU_LD + col]);
            float4 v;
            v.x = fmaxf(__uint_as_float(u.x) + acc[0] + b.x, 0.f); v.y = fmaxf(__uint_as_float(u.y) + acc[1] + b.y, 0.f);
            v.z = fmaxf(__uint_as_float(u.z) + acc[2] + b.z, 0.f); v.w = fmaxf(__uint_as_float(u.w) + acc[3] + b.w, 0.f);
            *reinterpret_cast<float4*>(a.out + (size_t)(pb0 + r) * PF_C1 + col) = v;
            if (a.out16) {
                half4 h; h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
                *reinterpret_cast<half4*>(a.out16 + (size_t)(pb0 + r) * PF_C1 + col) = h;
            }
        }
    }
  }
}

static int pfnCUs() {
    static int n = 0;
    if (!n) { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); n = hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }
    return n;
}

static inline int pfnPermuteK(int p) {       // see mlp.hip: position p of a permuted weight row holds column k(p)
    const int s = p / 32, q = p % 32, g = q / 8, j = q % 8;
    return 32 * s + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
}

// fields: max_pillars_num, weight0 [96][10], bias0 [96], weight1 [192][192] (the folded FC1: columns 0..95 act on x0, 96..191 on
// max_pillar(x0)), bias1 [192].  Inputs: feat [1,Nk,10] f32, pidx [1,P,T] i32, pcnt [1,P,1] i32, pillar_num [1].
// Outputs: pillar features [1,P,192] f32 + fp16 copy.
class DsvtPillarFeatureNetPlugin : public Plugin {
public:
    int max_pillars_;
    int pack_ = 1;                                                  // pillars with <= 4 points share MFMA tiles (0: the round-1 one-pillar-per-tile layout)
    int split_ = 0;                                                 // optional field "split_precision": layer 1 on hi / lo fp16 operand pairs (fp32 grade), fp32 output only
    std::vector<float> w0_, b0_, w1_, b1_;
    float *w0_dev_ = nullptr, *b0_dev_ = nullptr, *b1_dev_ = nullptr; _Float16 *w1a_dev_ = nullptr, *w1b_dev_ = nullptr;
    bool ok_ = false;
    DsvtPillarFeatureNetPlugin(int mp, const float* w0, const float* b0, const float* w1, const float* b1, int split = 0)
        : max_pillars_(mp), split_(split), w0_(w0, w0 + PF_C0 * PF_IN), b0_(b0, b0 + PF_C0), w1_(w1, w1 + (size_t)PF_C1 * PF_C1), b1_(b1, b1 + PF_C1) {
        std::vector<float> w0p((size_t)PF_C0 * 12, 0.f);
        for (int n = 0; n < PF_C0; ++n) for (int k = 0; k < PF_IN; ++k) w0p[n * 12 + k] = w0_[n * PF_IN + k];
        const size_t img = (size_t)3 * 12 * 512;
        std::vector<_Float16> wa((split_ ? 2 : 1) * img), wb((split_ ? 2 : 1) * img);
        for (int s = 0; s < 3; ++s)
            for (int t = 0; t < 12; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const size_t d = (((size_t)s * 12 + t) * 64 + lane) * 8 + j;
                        const int n = 16 * t + (lane & 15), p = 32 * s + 8 * (lane >> 4) + j;
                        const float va = w1_[(size_t)n * PF_C1 + pfnPermuteK(p)], vb = w1_[(size_t)n * PF_C1 + PF_C0 + p];
                        wa[d] = (_Float16)va;                                                // x0 half, chained (k-permuted) operand
                        wb[d] = (_Float16)vb;                                                // max half, natural k
                        if (split_) { wa[img + d] = (_Float16)(va - (float)wa[d]); wb[img + d] = (_Float16)(vb - (float)wb[d]); }
                    }
        auto upF = [](const std::vector<float>& h, float** d) {
            return hipMalloc(d, sizeof(float) * h.size()) == hipSuccess && hipMemcpy(*d, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice) == hipSuccess;
        };
        auto upH = [](const std::vector<_Float16>& h, _Float16** d) {
            return hipMalloc(d, sizeof(_Float16) * h.size()) == hipSuccess && hipMemcpy(*d, h.data(), sizeof(_Float16) * h.size(), hipMemcpyHostToDevice) == hipSuccess;
        };
        ok_ = upF(w0p, &w0_dev_) && upF(b0_, &b0_dev_) && upF(b1_, &b1_dev_) && upH(wa, &w1a_dev_) && upH(wb, &w1b_dev_);
    }
    ~DsvtPillarFeatureNetPlugin() override {
        for (void* p : {(void*)w0_dev_, (void*)b0_dev_, (void*)b1_dev_, (void*)w1a_dev_, (void*)w1b_dev_}) if (p) (void)hipFree(p);
    }
    const char* type() const override { return "DsvtPillarFeatureNetPlugin"; }
    int nbOutputs() const override { return split_ ? 1 : 2; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i < 0 || i >= nbOutputs()) return -1;
        *out = dims3(in[0].d[0], max_pillars_, PF_C1); return 0;
    }
    int outputType(int i, const int32_t*, int) const override { return i == 0 ? DSVT_FLOAT : DSVT_HALF; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int nbIn, int) const override {
        if (io[pos].format != DSVT_FORMAT_LINEAR) return false;
        if (pos >= 1 && pos <= 3) return io[pos].type == DSVT_INT32;
        if (pos < nbIn) return io[pos].type == DSVT_FLOAT;
        return io[pos].type == (pos == nbIn ? DSVT_FLOAT : DSVT_HALF);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        if (!ok_) return static_cast<int>(hipErrorOutOfMemory);
        PfnArgs a{};
        static unsigned long long* tr = nullptr; static int tron = -1;
        if (tron < 0) { tron = ablateEnv("DSVT_PFN_TRACE", 0) ? 1 : 0; if (tron) (void)hipMallocManaged(&tr, 8 * 64); }
        a.trace = tr;
        static int dbg = -1; if (dbg < 0) dbg = ablateEnv("DSVT_PFN_DBG", 0);
        a.dbg = dbg;
        a.feat = static_cast<const float*>(in[0]); a.pidx = static_cast<const uint32_t*>(in[1]);
        a.T = inDesc ? inDesc[1].dims.d[inDesc[1].dims.nbDims - 1] : 48;
        a.pcnt = static_cast<const uint32_t*>(in[2]); a.pillar_num = static_cast<const uint32_t*>(in[3]); a.max_pillars = max_pillars_;
        a.w0 = w0_dev_; a.b0 = b0_dev_; a.w1a = w1a_dev_; a.w1b = w1b_dev_; a.b1 = b1_dev_;
        a.out = static_cast<float*>(out[0]); a.out16 = split_ ? nullptr : static_cast<_Float16*>(out[1]);
        a.pack = pack_;
        if (zeroFill) {
            DSVT_CHECK(hipMemsetAsync(out[0], 0, sizeof(float) * (size_t)max_pillars_ * PF_C1, stream));
            if (!split_) DSVT_CHECK(hipMemsetAsync(out[1], 0, sizeof(_Float16) * (size_t)max_pillars_ * PF_C1, stream));
        }
        int grid = (split_ ? 1 : 2) * pfnCUs(); if (grid > cdiv(max_pillars_, PF_PB)) grid = cdiv(max_pillars_, PF_PB);       // two resident workgroups per CU (three: 82 vs 79 us); split: 92 KB of LDS, one
        if (split_) hipLaunchKernelGGL(pfn_kernel<true>, dim3(grid), dim3(64 * PF_NW), 0, stream, a);
        else hipLaunchKernelGGL(pfn_kernel<false>, dim3(grid), dim3(64 * PF_NW), 0, stream, a);
        if (tron) { (void)hipStreamSynchronize(stream); fprintf(stderr, "[pfn trace wg0]"); for (int i = 1; i < 24; ++i) fprintf(stderr, " %lld", (long long)(tr[i] - tr[0])); fprintf(stderr, "\n"); }
        return lastError();
    }
    size_t nFloats() const { return w0_.size() + b0_.size() + w1_.size() + b1_.size(); }
    size_t serializationSize() const override { return (split_ ? 3 : 2) * sizeof(int) + sizeof(float) * nFloats(); }
    void serialize(void* buf) const override {
        char* d = static_cast<char*>(buf);
        wr<int>(d, max_pillars_);
        for (const std::vector<float>* v : {&w0_, &b0_, &w1_, &b1_}) { memcpy(d, v->data(), sizeof(float) * v->size()); d += sizeof(float) * v->size(); }
        wr<int>(d, pack_);
        if (split_) wr<int>(d, split_);
    }
    Plugin* clone() const override {
        DsvtPillarFeatureNetPlugin* c = new DsvtPillarFeatureNetPlugin(max_pillars_, w0_.data(), b0_.data(), w1_.data(), b1_.data(), split_);
        c->pack_ = pack_; return c;
    }
};
static Plugin* pfnCreate(const DsvtPluginFieldCollection* fc) {
    const int mp = fieldInt(fc, "max_pillars_num");
    struct Need { const char* name; int len; } need[] = {{"weight0", PF_C0 * PF_IN}, {"bias0", PF_C0}, {"weight1", PF_C1 * PF_C1}, {"bias1", PF_C1}};
    const float* p[4];
    for (int i = 0; i < 4; ++i) {
        const DsvtPluginField* f = findField(fc, need[i].name);
        if (!f || !f->data || f->length != need[i].len) return nullptr;
        p[i] = static_cast<const float*>(f->data);
    }
    if (mp <= 0) return nullptr;
    DsvtPillarFeatureNetPlugin* pl = new DsvtPillarFeatureNetPlugin(mp, p[0], p[1], p[2], p[3], fieldInt(fc, "split_precision", 0) != 0);
    pl->pack_ = fieldInt(fc, "pack_small_pillars", 1) != 0;
    return pl;
}
static Plugin* pfnDeser(const void* data, size_t len) {
    if (len < sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    const int mp = rd<int>(d);
    const size_t n = (size_t)PF_C0 * PF_IN + PF_C0 + (size_t)PF_C1 * PF_C1 + PF_C1;
    if (mp <= 0 || len < sizeof(int) + n * sizeof(float)) return nullptr;
    std::vector<float> all(n); memcpy(all.data(), d, n * sizeof(float));
    const float* q = all.data();
    int pack = 1, split = 0;
    const int extra = trailingInts(len, sizeof(int) + n * sizeof(float), 2);
    if (extra < 0) return nullptr;
    if (extra >= 1) { const char* t = d + n * sizeof(float); pack = rd<int>(t) != 0; if (extra >= 2) split = rd<int>(t) != 0; }
    DsvtPillarFeatureNetPlugin* pl = new DsvtPillarFeatureNetPlugin(mp, q, q + PF_C0 * PF_IN, q + PF_C0 * PF_IN + PF_C0, q + PF_C0 * PF_IN + PF_C0 + (size_t)PF_C1 * PF_C1, split);
    pl->pack_ = pack;
    return pl;
}
static Creator g_pfnCreator{"DsvtPillarFeatureNetPlugin",
    {{"max_pillars_num", DSVT_FIELD_INT32}, {"weight0", DSVT_FIELD_FLOAT32}, {"bias0", DSVT_FIELD_FLOAT32}, {"weight1", DSVT_FIELD_FLOAT32},
     {"bias1", DSVT_FIELD_FLOAT32}, {"pack_small_pillars", DSVT_FIELD_INT32}, {"split_precision", DSVT_FIELD_INT32}},
    pfnCreate, pfnDeser, {}, {}};
static Registrar g_pfnReg(&g_pfnCreator);

}  // namespace dsvt
