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
// The compact point ids of a pillar are consecutive (Points2Features' canonical order: pillar-major, then slot) and every 16-point MFMA tile belongs
// to one pillar or to four small ones, so the maxima are plain register reductions: nothing is atomic, in LDS or in global memory.  After the point
// pass of a group of up to 16 pillars its wave computes t = W1b m (16 x 96 x 192 on the matrix cores, W1b fragments straight from L2) and
// writes  vfeat_p = ReLU(U_p + t_p + b1)  as fp32 and fp16.  Who owns which pillars: see "Round 4" above pfn_kernel.
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
    unsigned long long* trace;         // ablation build: s_memtime stamps of wave 0 of workgroup 7 (tools/bench_pfn.py)
    int dbg;                           // timing ablations of the ablation build (wrong results): 1 no point gather, 2 no layer-0 MFMA, 4 no layer-1 MFMA, 8 no stores, 16 no per-pillar GEMM, 32 no m maxima
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
    const float w = __builtin_amdgcn_fmed3f(__uint_as_float(a[0]), __uint_as_float(a[1]), __builtin_inff());      // (= max: see vmax below)
    const uint32_t x = __float_as_uint(w);
    const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return __builtin_amdgcn_fmed3f(__uint_as_float(b[0]), __uint_as_float(b[1]), __builtin_inff());
}

// max(a, b) as ONE instruction: fmaxf() is llvm.maxnum, which hipcc lowers (IEEE mode, no fast-math) to v_max_f32 of the two operands CANONICALISED, i.e. three
// v_max_f32 -- the round-3 kernel's tile was 648 of them, and VALU issue, not the matrix pipe, is what bounds this kernel.  med3(a, b, +inf) = max(a, b) for
// every non-NaN pair (the operands here are MFMA sums of finite inputs)
__device__ __forceinline__ float vmax(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
__device__ __forceinline__ float vmax4(const floatx4& v) { return vmax(vmax(v[0], v[1]), vmax(v[2], v[3])); }

#ifndef PFN_WG_PER_CU
#define PFN_WG_PER_CU 2
#endif
constexpr int PF_GB = 8 * PF_PB;       // work units per group: a pillar weighs 8 + its points, so a group holds at most 16 pillars
constexpr int PF_NC = 1024;            // coarse samples of the work prefix in LDS
constexpr int PF_LDM = PF_C0 + 8;      // halves per row of a wave's m tile (208-byte rows: conflict-free b128 reads)

// Round 4: a WAVE owns a group of up to 16 consecutive pillars from its first point to its last store -- no workgroup barrier after the weights have
// landed, nothing shared between waves but read-only tables -- and the groups are cut by WORK, not by pillar count (see the kernel's first lines).
// Rounds 2-3 gave every 16 pillars to a four-wave workgroup: 277 / 459 us per four-frame launch (fp16 / split) with the matrix pipe idle 80 % of the
// time, which this round traced to (a) load imbalance -- 4 to 48 tiles per 16 pillars, the heaviest wave of the launch at three times the mean --,
// (b) VALU issue: fmaxf() cost three instructions (see vmax), 648 v_max_f32 per tile, and (c) per-group barriers and LDS round trips.
// Now 178 / 268 us (one frame: 58 / 84 us, from 77 / 140), bit-identical outputs.
//   * the group's table lives in registers: lane p holds the first row of pillar p (lane npil: one past the last), counts by a lane shift, the
//     small pillars (<= 4 points) take slots 0 .. ns - 1 in order, the larger ones ns .. 15 (ballot + popcount), and ds_permute turns
//     (start, count, pillar) into slot-indexed lanes that every later lookup reads with ds_bpermute;
//   * U = max_points(W1a x0) never leaves the registers: packed tile u, lane group g' holds slot 4 u + g' (12 values per tile, 48 in all), a larger
//     pillar's maxima are selected into its slot's lane group.  The per-pillar GEMM t = m W1b^T runs with m as the A operand and its rows in
//     the order rho = 4 (slot & 3) + (slot >> 2), so that D_t[i] of lane group g' IS slot 4 i + g': U + t + b1 is a register add;
//   * m = max_points(x0) comes from a second layer-0 product in the other orientation (lane = channel, registers = points), as before: taking it
//     from the x0^T tile instead (quad / row maxima on the DPP path) was built and costs more VALU issue than the 18 fp32 MFMAs cost matrix pipe;
//   * the finished rows leave straight from the registers (64-byte runs, completed to lines in L2 by the next instruction);
//   * the next group's search advances under the current group's tiles, its first tile's points are requested before the current group's epilogue.
// What bounds it now (s_memtime trace of one wave, tools/bench_pfn.py with DSVT_PFN_TRACE=1 on the ablation build): a tile is ~6500 cycles per wave where
// its MFMAs are 1728 (fp16; two waves per SIMD: 3456 per tile pair), a group's epilogue ~8000: dependent LDS / L2 round trips that two 233-register
// waves per SIMD cannot hide.  Three waves need <= 168 registers; U alone is 48.
// SPLIT (the fp32-grade mode): both halves of layer 1 with hi / lo fp16 operand pairs, three MFMAs per product (see linear.hip
// linear_split_rows_kernel); layer 0 is fp32 MFMA either way.  W1a hi + lo = 72 KB of LDS: one eight-wave workgroup per CU; fp16: two of four waves.
struct PfnTable { uint32_t start, cnt, pill; int ns, nl, npil; uint32_t pb0; };    // (start / cnt / pill: lane = slot)

template <bool SPLIT>
__global__ void __launch_bounds__(64 * (SPLIT ? 8 : 4), SPLIT ? 1 : PFN_WG_PER_CU)
pfn_kernel(PfnArgs a)
{
    constexpr int NW = SPLIT ? 8 : 4;
    constexpr int WBUF = (SPLIT ? 2 : 1) * PF_PB * PF_LDM * 2;          // bytes per wave: the m tile (SPLIT: hi rows, then lo rows)
    __shared__ __attribute__((aligned(16))) _Float16 sW1[(SPLIT ? 2 : 1) * 3 * 12 * 512];                              // 36 KB (SPLIT: w_hi image, then w_lo image)
    __shared__ __attribute__((aligned(16))) unsigned char sBuf[NW * WBUF];
    __shared__ __attribute__((aligned(16))) float sB0[PF_C0], sB1[PF_C1];
    __shared__ uint32_t sCoarse[PF_NC + 1];
    uint32_t P = *a.pillar_num; if (P > (uint32_t)a.max_pillars) P = a.max_pillars;
    if (P == 0) return;
    // Groups of EQUAL WORK, not of equal pillar count (round 4): the tiles of 16 consecutive pillars range from 4 (sixteen single points) to 48 (sixteen
    // full pillars next to the sensor), and with whole groups per wave the heaviest wave of a four-frame launch had 125 tiles against a mean of 41.
    // The work prefix needs no scan: the first row of pillar p IS the point prefix, so  Wt(p) = 8 p + start(p)  is a strictly increasing estimate of the
    // tile work before pillar p (a packed single ~ 1/4 tile = 9 units, a full 48-point pillar 3 tiles = 56), and group k is the pillars with Wt in
    // [128 k, 128 k + 128): at most 16 of them, 3.6 .. 7 tiles.  A wave finds its group's first pillar with a two-level 16-ary search of 1025 samples
    // of Wt in LDS (taken once per workgroup) and, on launches of more than ~47k pillars, one round of 64 probes; the entries it then loads to fill
    // its table show where the group ends.
    const uint32_t total = a.pidx[(size_t)(P - 1) * a.T] + a.pcnt[P - 1];                  // points of the launch
    const uint32_t wtEnd = 8u * P + total;
    const uint32_t ngroups = (wtEnd + PF_GB - 1) / PF_GB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    if (blockIdx.x * NW >= ngroups) return;
    for (int i = tid; i < (SPLIT ? 2 : 1) * 3 * 12 * 64; i += 64 * NW)
        *reinterpret_cast<uint4*>(&sW1[i * 8]) = *reinterpret_cast<const uint4*>(a.w1a + (size_t)i * 8);
    for (int i = tid; i < PF_C0; i += 64 * NW) sB0[i] = a.b0[i];
    for (int i = tid; i < PF_C1; i += 64 * NW) sB1[i] = a.b1[i];
    for (int i = tid; i <= PF_NC; i += 64 * NW) {
        const uint32_t q = (uint32_t)(((unsigned long long)i * P) / PF_NC);
        sCoarse[i] = i == 0 ? 0u : q < P ? 8u * q + a.pidx[(size_t)q * a.T] : wtEnd;
    }
    __syncthreads();                                                    // the only workgroup barrier
    uint32_t grp = blockIdx.x * NW + wave;
    const uint32_t stride = gridDim.x * NW;
    if (grp >= ngroups) return;
    _Float16* sMw = reinterpret_cast<_Float16*>(sBuf + wave * WBUF);
    // layer-0 weights as MFMA A fragments: lane (r, g) holds W0[16t + r][4ks + g]
    float w0f[6][3];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) w0f[t][ks] = a.w0[(16 * t + r) * 12 + 4 * ks + g];
    int nmark = 0;
    auto mark = [&]() {
        if constexpr (kAblate) {
            if (a.trace && blockIdx.x == 7 && wave == 0) { if (lane == 0 && nmark < 510) a.trace[nmark] = __builtin_amdgcn_s_memtime(); ++nmark; }
        }
    };
    // Wt(q); q == P: the end of the launch; beyond: nothing
    auto wtLoad = [&](uint32_t q) -> uint32_t {
        if (q < P) return 8u * q + a.pidx[(size_t)q * a.T];
        return q == P ? wtEnd : 0xffffffffu;
    };
    const bool probing = (P + PF_NC - 1) / PF_NC + PF_PB + 2 > 64;       // a coarse interval + the group do not fit the 64 lanes of one load
    // stage A of the search for group k: the coarse interval [q0, q1] with Wt(q0) <= 128 k < Wt(q1) from LDS; requests the probes (lane l: q0 + l step)
    // or, without probing, the table entries (step 1)
    auto stageA = [&](uint32_t k, uint32_t& base, uint32_t& step, uint32_t& v) {
        const uint32_t target = k * PF_GB;
        const unsigned long long m1 = __ballot(sCoarse[16 * lane] <= target);
        const int j = 63 - __clzll(m1);
        const unsigned long long m2 = __ballot(lane <= 16 && sCoarse[16 * j + (lane <= 16 ? lane : 16)] <= target);
        const int i = 16 * j + (63 - __clzll(m2));
        const uint32_t q0 = (uint32_t)(((unsigned long long)i * P) / PF_NC), q1 = (uint32_t)(((unsigned long long)(i + 1) * P) / PF_NC);
        base = q0;
        step = probing ? (q1 - q0 + 63) / 64 : 1u;
        if (step == 0) step = 1;
        v = wtLoad(q0 + lane * step);
    };
    // stage C (probing): narrows the interval to one probe step and requests the table entries
    auto stageC = [&](uint32_t k, uint32_t& base, uint32_t& step, uint32_t& v) {
        const uint32_t target = k * PF_GB;
        for (;;) {
            const int l = __popcll(__ballot(v <= target)) - 1;           // (lane 0 probes q0: Wt(q0) <= target)
            base += (uint32_t)l * step;
            if (step + PF_PB + 2 <= 64) { v = wtLoad(base + lane); step = 1; return; }
            step = (step + 63) / 64;
            v = wtLoad(base + lane * step);
        }
    };
    auto bperm = [&](uint32_t v, int src) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v); };
    // stage D: lane j holds Wt(base + j); the group's pillars are the lanes with Wt in [target, target + 128); lane p of the table = first row of
    // the group's pillar p, lane npil = one past the last pillar's rows
    auto makeTable = [&](uint32_t k, uint32_t base, uint32_t v, PfnTable& T) {
        const uint32_t target = k * PF_GB, q = base + lane;
        const unsigned long long mb = __ballot(q < P && v >= target && v - target < (uint32_t)PF_GB);
        T.npil = __popcll(mb);
        const int first = mb ? __builtin_ctzll(mb) : 0;
        T.pb0 = base + first;
        const uint32_t s = (uint32_t)__shfl((int)(v - 8u * q), lane + first, 64);
        const uint32_t sn = (uint32_t)__shfl_down((int)s, 1, 64);
        const bool valid = lane < T.npil;
        const uint32_t c = valid ? sn - s : 0u;
        const bool small = valid && a.pack && c <= 4u;
        const unsigned long long ms = __ballot(small), ml = __ballot(valid && !small);
        const unsigned long long below = (1ull << lane) - 1ull;
        T.ns = __popcll(ms); T.nl = __popcll(ml);
        const int slot = small ? __popcll(ms & below) : valid ? T.ns + __popcll(ml & below) : lane;      // (a permutation of the 64 lanes)
        T.start = (uint32_t)__builtin_amdgcn_ds_permute(slot << 2, (int)s);
        T.cnt = (uint32_t)__builtin_amdgcn_ds_permute(slot << 2, (int)c);
        T.pill = (uint32_t)__builtin_amdgcn_ds_permute(slot << 2, lane);
    };
    // work units of a group: packed tiles of up to four small pillars first (slots 4 u .. 4 u + 3), then the larger pillars (1..3 tiles each).
    // Row of the point MFMA row r works on in tile (u, t); padding = the pillar's first point: duplicates do not change a maximum
    auto rowOf = [&](const PfnTable& T, int u, int t) -> uint32_t {
        const int npk = (T.ns + 3) >> 2;
        if (u < npk) {
            int idx = 4 * u + (r >> 2); if (idx >= T.ns) idx = 4 * u;
            const uint32_t s_ = bperm(T.start, idx), c_ = bperm(T.cnt, idx), i_ = (uint32_t)(r & 3);
            return s_ + (i_ < c_ ? i_ : 0u);
        }
        const int sl = T.ns + (u - npk);
        const uint32_t s_ = bperm(T.start, sl), c_ = bperm(T.cnt, sl), row = s_ + 16u * (uint32_t)t + (uint32_t)r;
        return row < s_ + c_ ? row : s_;
    };
    auto loadRow = [&](uint32_t row, float (&fbv)[3]) {
        if (kAblate && (a.dbg & 1)) row = lane;
        const float* fr = a.feat + (size_t)row * PF_IN;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) { const int k = 4 * ks + g; fbv[ks] = k < PF_IN ? fr[k] : 0.f; }
    };
    PfnTable T, Tn;
    uint32_t nb, nstep, nv;
    stageA(grp, nb, nstep, nv);
    if (probing) stageC(grp, nb, nstep, nv);
    makeTable(grp, nb, nv, T);
    float fb[3], fnext[3];
    if (T.npil) loadRow(rowOf(T, 0, 0), fnext);
  for (;;) {
    // the next group's search advances under this group's tiles: stage A now, stage C at the second tile, its table and first points at the last
    const uint32_t gnext = grp + stride;
    const bool haveNextGroup = gnext < ngroups;
    if (haveNextGroup) stageA(gnext, nb, nstep, nv);
    bool probed = !probing;
    const int npacked = (T.ns + 3) >> 2, nunits = npacked + T.nl;
    if (nunits == 0) {                                                  // (no pillar in this work interval: a pillar of more than 120 points)
        if (!haveNextGroup) break;
        if (!probed) stageC(gnext, nb, nstep, nv);
        makeTable(gnext, nb, nv, T); grp = gnext;
        if (T.npil) loadRow(rowOf(T, 0, 0), fnext);
        continue;
    }
    float U[4][12];                                                     // U[u][k], lane group g': slot 4 u + g', column 16 k + r
    float mx0[6], mxu[12];                                              // running maxima of a larger pillar
#pragma unroll
    for (int k = 0; k < 6; ++k) mx0[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) mxu[k] = -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 12; ++k) U[u][k] = 0.f;
    // ---- point pass: 16 points per MFMA tile, every tile belongs to one pillar or to four small ones
    //   x0^T tile  = W0 (A) x f (B)      lane = point, rows = channels: the k-permuted B... operand of layer 1 (see mlp.hip); max over points = DPP
    //   u    tile  = x0 (A) x W1a (B)    lane (r', g') = column r', points 4g'..4g'+3: max over points = 3 in-register max (+ 2 swaps for a larger pillar)
    int u = 0, t = 0;
    for (;;) {
        mark();                                                        // [7 i] tile start
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) fb[ks] = fnext[ks];
        const int cu = u;
        int nt = 1, sl = 0;
        if (cu >= npacked) {
            sl = T.ns + (cu - npacked);
            nt = (int)((__builtin_amdgcn_readfirstlane((int)bperm(T.cnt, sl)) + 15) >> 4);
        }
        const bool lastOfUnit = t + 1 >= nt;
        // the tile after this one (the next group's first one after the last): its point rows are requested before the current tile's MFMAs
        int nu = cu, ntile = t + 1;
        if (lastOfUnit) { nu = cu + 1; ntile = 0; }
        const bool lastOfGroup = nu >= nunits;
        if (haveNextGroup && !probed && (lastOfGroup || cu + t > 0)) { stageC(gnext, nb, nstep, nv); probed = true; }
        if (!lastOfGroup) loadRow(rowOf(T, nu, ntile), fnext);
        else if (haveNextGroup) { makeTable(gnext, nb, nv, Tn); if (Tn.npil) loadRow(rowOf(Tn, 0, 0), fnext); }

        mark();                                                        // [7 i + 1] next rows requested
        floatx4 x0[6];
        float m0[6];                                                     // max over this lane group's four points of channel 16 k + r, past the ReLU
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            x0[k] = *reinterpret_cast<const floatx4*>(&sB0[16 * k + 4 * g]);
            const float bc = sB0[16 * k + r];
            floatx4 d0 = {bc, bc, bc, bc};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
                if (!kAblate || !(a.dbg & 2)) {
                    x0[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0f[k][ks], fb[ks], x0[k], 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[ks], w0f[k][ks], d0, 0, 0, 0);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) x0[k][i] = vmax(x0[k][i], 0.f);             // ReLU (:144)
            m0[k] = vmax(vmax4(d0), 0.f);                                         // max(ReLU(.)) = max(0, max(.))
        }
        mark();                                                        // [7 i + 2] layer 0 done
        // the layer-1 A operand (chained: see mlp.hip) ...
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
                } else {                 // the fp16 frame: no clamp, no residuals
                    h[i] = (_Float16)v0; h[4 + i] = (_Float16)v1; l[i] = l[4 + i] = (_Float16)0.f;
                }
            }
            f1[s_] = half8{h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
            f1l[s_] = half8{l[0], l[1], l[2], l[3], l[4], l[5], l[6], l[7]};
        }
        mark();                                                        // [7 i + 3] operand built
        // ... and m = max over the pillar's points of x0 from the SAME product in the other orientation (x0 = f (A) x W0 (B): lane (r', g') = channel r',
        // points 4g'..4g'+3): 18 more fp32 MFMAs on a matrix pipe that has the time, instead of 48 cross-lane VALU steps on a VALU that has not.
        // m row of a slot: GEMM row rho = 4 (slot & 3) + (slot >> 2)
        auto storeM = [&](int rho, const float (&mv)[6]) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const _Float16 mh = (_Float16)fminf(mv[k], 65504.f);
                sMw[rho * PF_LDM + 16 * k + r] = mh;
                if constexpr (SPLIT) { float dm = mv[k] - (float)mh; asm volatile("" : "+v"(dm)); sMw[(PF_PB + rho) * PF_LDM + 16 * k + r] = (_Float16)dm; }
            }
        };
        if (kAblate && (a.dbg & 32)) {} else
        if (cu < npacked) {
            // packed tile: lane group g holds the maxima of slot 4 cu + g
            if (4 * cu + g < T.ns) storeM(4 * g + cu, m0);
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) mx0[k] = vmax(mx0[k], m0[k]);
            if (lastOfUnit) {
#pragma unroll
                for (int k = 0; k < 6; ++k) mx0[k] = maxOverLaneGroups(mx0[k]);
                if (g == 0) storeM(4 * (sl & 3) + (sl >> 2), mx0);
#pragma unroll
                for (int k = 0; k < 6; ++k) mx0[k] = 0.f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        mark();                                                        // [7 i + 4] m stored
        // ---- layer 1: nine steps (four column tiles x one 32-channel slice), the W1a fragments of a step read while the step before runs
        // (left alone hipcc hoists the reads of all twelve tiles: 330 spilled registers)
        float mu[12];
        if (kAblate && (a.dbg & 4)) {
#pragma unroll
            for (int k = 0; k < 12; ++k) mu[k] = 0.f;
        } else {
            constexpr int NFW = SPLIT ? 8 : 4;
            half8 wq[2][NFW];
            auto loadW1 = [&](int step, half8 (&w)[NFW]) {
                const int kb = 4 * (step / 3), s_ = step % 3;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    w[kk] = *reinterpret_cast<const half8*>(&sW1[((s_ * 12 + kb + kk) * 64 + lane) * 8]);
                    if constexpr (SPLIT) w[4 + kk] = *reinterpret_cast<const half8*>(&sW1[(((3 + s_) * 12 + kb + kk) * 64 + lane) * 8]);
                }
            };
            loadW1(0, wq[0]);
            floatx4 acc[4];
#pragma unroll
            for (int step = 0; step < 9; ++step) {
                const int kb = 4 * (step / 3), s_ = step % 3;
                if (step + 1 < 9) loadW1(step + 1, wq[(step + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (s_ == 0) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc[kk] = floatx4{0.f, 0.f, 0.f, 0.f};
                }
                if constexpr (SPLIT) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc[kk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1[s_], wq[step & 1][4 + kk], acc[kk], 0, 0, 0);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc[kk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1l[s_], wq[step & 1][kk], acc[kk], 0, 0, 0);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc[kk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1[s_], wq[step & 1][kk], acc[kk], 0, 0, 0);
                if (s_ == 2) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) mu[kb + kk] = vmax4(acc[kk]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        mark();                                                        // [7 i + 5] layer 1 done
        if (cu < npacked) {
            // in the u tile lane group g' holds slot 4 cu + g'
#pragma unroll
            for (int uu = 0; uu < 4; ++uu)
                if (uu == cu) {
#pragma unroll
                    for (int k = 0; k < 12; ++k) U[uu][k] = mu[k];
                }
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) mxu[k] = vmax(mxu[k], mu[k]);
            if (lastOfUnit) {
                // across the four lane groups (points 4g'..4g'+3 of every tile), then the slot's lane group keeps column 16k + r
#pragma unroll
                for (int k = 0; k < 12; ++k) mxu[k] = maxOverLaneGroups(mxu[k]);
#pragma unroll
                for (int uu = 0; uu < 4; ++uu)
                    if (uu == (sl >> 2)) {
#pragma unroll
                        for (int k = 0; k < 12; ++k) U[uu][k] = g == (sl & 3) ? mxu[k] : U[uu][k];
                    }
#pragma unroll
                for (int k = 0; k < 12; ++k) mxu[k] = -INFINITY;
            }
        }
        mark();                                                        // [7 i + 6] maxima kept
        if (lastOfGroup) break;
        u = nu; t = ntile;
    }
    mark();                                                            // epilogue: [e] start, [e + 1] GEMM done, [e + 2] stores issued, then 4 x the same stamp (rows of 7)
    // ---- per-pillar half: t = m W1b^T (16 slots x 96 -> 192), m (A) rows in the order rho, W1b (B) fragments straight from L2 --------------------
    asm volatile("" ::: "memory");
    half8 mf[3], mfl[3];                             // A fragment: lane (r, g) holds m[row r][32s + 8g + j]
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        mf[s] = *reinterpret_cast<const half8*>(&sMw[r * PF_LDM + 32 * s + 8 * g]);
        if constexpr (SPLIT) mfl[s] = *reinterpret_cast<const half8*>(&sMw[(PF_PB + r) * PF_LDM + 32 * s + 8 * g]);
    }
    asm volatile("" ::: "memory");
    // (the fragments of column tiles tt + 1 .. tt + 3 are in flight under the MFMAs of tile tt -- an L2 round trip is ~800 cycles, a tile's MFMAs 50 to 150 --
    // and no more: left alone hipcc hoists all 36 / 72 loads)
    constexpr int NF = SPLIT ? 6 : 3, RD = SPLIT ? 3 : 4;         // (SPLIT: a fourth stage of six fragments spills)
    half8 wf[RD][NF];
    const char* w1bBytes = reinterpret_cast<const char*>(a.w1b);
    uint32_t laneOff = (uint32_t)lane << 4;
    asm volatile("" : "+v"(laneOff));                                   // (opaque per group: nothing to hoist)
    auto loadW = [&](int tt, half8 (&w)[NF]) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            // (uniform base + constant in scalar registers, the lane's 32-bit byte offset in ONE vector register: as 64-bit lane addresses hipcc
            // hoists all 36 / 72 of them out of the group loop and spills them)
            w[s] = *reinterpret_cast<const half8*>(w1bBytes + (s * 12 + tt) * 1024 + laneOff);
            if constexpr (SPLIT) w[3 + s] = *reinterpret_cast<const half8*>(w1bBytes + ((3 + s) * 12 + tt) * 1024 + laneOff);
        }
    };
#pragma unroll
    for (int tt = 0; tt < RD - 1; ++tt) loadW(tt, wf[tt]);
    if (!kAblate || !(a.dbg & 16))
#pragma unroll
    for (int tt = 0; tt < 12; ++tt) {
        if (tt + RD - 1 < 12) loadW(tt + RD - 1, wf[(tt + RD - 1) % RD]);
        __builtin_amdgcn_sched_barrier(0);
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if constexpr (SPLIT) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(mf[s], wf[tt % RD][3 + s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(mfl[s], wf[tt % RD][s], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(mf[s], wf[tt % RD][s], acc, 0, 0, 0);
        }
        const float b = sB1[16 * tt + r];              // lane (r, g'): column 16 tt + r of rows 4 g' + i = slots 4 i + g'
#pragma unroll
        for (int i = 0; i < 4; ++i) U[i][tt] = vmax(U[i][tt] + acc[i] + b, 0.f);
        __builtin_amdgcn_sched_barrier(0);
    }
    mark();
    // the rows leave straight from the registers: lane (r, g') stores column 16 k + r of slot 4 u + g' -- 64-byte runs that consecutive instructions complete
    // to whole lines in L2, 48 fire-and-forget stores with immediate offsets.  (Staging four slots at a time through LDS for 16-byte stores: three dependent LDS
    // round trips per round, 5400 cycles per group on the s_memtime trace against 6500 for a whole tile.)
    if (!kAblate || !(a.dbg & 8)) {
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            if (4 * uu >= T.npil) break;
            const int slot = 4 * uu + g;
            const uint32_t pill = bperm(T.pill, slot);
            const bool ok = slot < T.npil;
            const size_t o = (size_t)(T.pb0 + pill) * PF_C1 + r;
            if (ok) {
#pragma unroll
                for (int k = 0; k < 12; ++k) a.out[o + 16 * k] = U[uu][k];
            }
            if (a.out16) {                            // fp16 copy: the even lane of a pair stores both columns
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    const float nb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(U[uu][k]), 0x101, 0xf, 0xf, false));      // row_shl:1 = lane + 1
                    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
                    half2_t h; h[0] = (_Float16)U[uu][k]; h[1] = (_Float16)nb;
                    if (ok && !(r & 1)) *reinterpret_cast<half2_t*>(a.out16 + o + 16 * k) = h;
                }
            }
        }
    }
    mark(); mark(); mark(); mark(); mark();
    if (!haveNextGroup) break;
    T = Tn; grp = gnext;
  }
}

static int pfnCUs() { return deviceCUs(); }

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
            return dsvtMalloc(d, sizeof(float) * h.size()) == hipSuccess && hipMemcpy(*d, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice) == hipSuccess;
        };
        auto upH = [](const std::vector<_Float16>& h, _Float16** d) {
            return dsvtMalloc(d, sizeof(_Float16) * h.size()) == hipSuccess && hipMemcpy(*d, h.data(), sizeof(_Float16) * h.size(), hipMemcpyHostToDevice) == hipSuccess;
        };
        ok_ = upF(w0p, &w0_dev_) && upF(b0_, &b0_dev_) && upF(b1_, &b1_dev_) && upH(wa, &w1a_dev_) && upH(wb, &w1b_dev_);
    }
    ~DsvtPillarFeatureNetPlugin() override {
        for (void* p : {(void*)w0_dev_, (void*)b0_dev_, (void*)b1_dev_, (void*)w1a_dev_, (void*)w1b_dev_}) if (p) (void)dsvtFree(p);
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
        if (tron < 0) { tron = ablateEnv("DSVT_PFN_TRACE", 0) ? 1 : 0; if (tron) { (void)hipMallocManaged(&tr, 8 * 512); memset(tr, 0, 8 * 512); } }
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
        // one eight-wave workgroup per CU (split: 128 KB of LDS) or three of four waves (51 KB each); a wave takes every (grid x waves)-th group of 16 pillars
        const int nw = split_ ? 8 : 4;
        int grid = (split_ ? 1 : PFN_WG_PER_CU) * pfnCUs(); if (grid > cdiv(cdiv(max_pillars_, PF_PB), nw)) grid = cdiv(cdiv(max_pillars_, PF_PB), nw);
        if (split_) hipLaunchKernelGGL(pfn_kernel<true>, dim3(grid), dim3(64 * nw), 0, stream, a);
        else hipLaunchKernelGGL(pfn_kernel<false>, dim3(grid), dim3(64 * nw), 0, stream, a);
        if (tron) {                                                      // rows of seven stamps (100 MHz), relative to the first
            (void)hipStreamSynchronize(stream);
            for (int i = 0; i < 70 && tr[7 * i]; ++i) { fprintf(stderr, "[pfn trace]"); for (int j = 0; j < 7; ++j) fprintf(stderr, " %6lld", (long long)(tr[7 * i + j] - tr[0])); fprintf(stderr, "\n"); }
            memset(tr, 0, 8 * 512);
        }
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
