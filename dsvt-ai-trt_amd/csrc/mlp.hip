// mlp.hip -- DsvtEncoderMlpPlugin: everything a DSVT encoder layer does after the attention core,
// in ONE launch (fp16 MFMA operands, fp32 accumulate / LayerNorm):
//
//     s1 = LN1(att Wo^T + bo + x)                       src/dsvt-ai-trt.cpp:448, 669-676
//     h  = GELU(s1 W1^T + b1)                           :506, gelu.cu:208-209
//     s2 = LN2(s1 + h W2^T + b2)                        :525, 684-690
//     x' = LN3(s2 + x)      [ x' = LN4(x' + xb) on the 2nd layer of a block ]   :691-697, 750-756
//
// In the reference this is 3 FullyConnected layers, 4-5 ElementWise SUMs, 3-4 LayerNorm plugins (3 kernels
// each) and a GELU plugin: ~20 launches and ~0.6 GB of traffic per layer.  As three DsvtLinear launches it
// is ~180 MB; here s1 and h never leave the register file (~100 MB: att16 + x (+ xb) in, x' + x'16 out).
//
// How the chaining works.  Each GEMM computes the transposed tile D[n][m] (see linear.hip): afterwards lane
// (r, g) holds, for activation row m = r, the output columns n = 16t + 4g + i (t = 0..11, i = 0..3).  The next
// GEMM needs that row as its B operand: 8 k-values per lane and k-step.  The MFMA sums over k, so ANY bijection
// between (lane group g, element j) and k is fine as long as the weight fragment uses the same one.  Taking
// k-step s = tiles {2s, 2s+1}:  element j < 4 -> k = 32s + 4g + j,  j >= 4 -> k = 32s + 16 + 4g + (j-4)  makes the
// B fragment exactly the eight fp32 values the lane already holds (converted to fp16) -- no shuffle, no LDS.
// The host stores W1 and W2 with their k columns permuted the same way (dsvt::permuteK), so the W fragment is
// still one contiguous ds_read_b128.
//
// Workgroup = 128 rows (4 waves x 32).  FC1 is produced 64 columns at a time and each piece is consumed at once as a
// 64-wide K slab of FC2, so neither s1 nor h ever leaves the register file; the weights arrive as fifteen 24 KB stages
// (see encoder_mlp_stream_kernel below).
#include "plugin_base.h"
#include "device_utils.h"
#include <cstdio>
#include <cstdlib>

namespace dsvt {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int MC = 192, MF = 384;                 // d_model, FFN width (include/params.h:80-84)
constexpr int MNT = MC / 16;                      // 12 column tiles
constexpr int MNSTEP = MC / 32;                   // 6 k-steps per 192-wide slab
constexpr int MROWS = 128;
constexpr int kMlpTraceBlocks = 4096;    // workgroups the DSVT_MLP_TRACE buffer holds (64 stamps each); later workgroups do not stamp
constexpr int MLP_SMALL_MAX = 8192;        // rows beyond one tile per CU that are cut into small workgroups
constexpr int MLP_SW = 2;                  // waves of a small workgroup, 32 rows each (round 2, one frame on this kernel: two waves 53.6 vs one 50.1 us; round 3, the tail
                                           // after two whole rounds at four frames: 1.25 vs 1.27 ms per eight launches, three waves 1.27-1.30)

__device__ __forceinline__ float mlpGelu(float x) {
    // 0.5 (1 + tanh u) = 1 / (1 + exp(-2u)),  u = x (B + C x^2):  seven VALU operations (the constants carry the -2 log2(e) of
    // the exp2); exp2 -> inf gives 1/inf = 0 -> -0 for very negative x, exp2 -> 0 gives x: both limits are the right ones
    const float Bn = -2.0f * 1.4426950408889634f * 0.7978845608028654f, Cn = -2.0f * 1.4426950408889634f * 0.035677408136300125f;
    const float z = x * __builtin_fmaf(Cn, x * x, Bn);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
}
__device__ __forceinline__ float rowSum4m(float v) {
    // (not rows4Sum(): the function returns the same bits -- tools/ubench/rows4_check.hip -- but inside THIS kernel hipcc then contracts / orders the
    // LayerNorm arithmetic around it differently and 7 % of the outputs move by an ulp or two: the same 2.2e-6 / 1.5e-7 from float64, no faster, and the
    // committed box sweeps (profiles/r04_mx_box_sweep.txt) would no longer be this library's bits)
    v += __shfl_xor(v, 16, kWave); v += __shfl_xor(v, 32, kWave);
    return v;
}
// LayerNorm over the 192 values of a row spread as acc[12][4] over 4 lanes
__device__ __forceinline__ void mlpLayerNorm(floatx4 (&acc)[MNT], const float* gm, const float* bt, int g, float eps) {
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < MNT; ++t) sum += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
    const float mean = rowSum4m(sum) / MC;
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < MNT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = acc[t][i] - mean; sq += d * d; }
    const float inv = 1.0f / sqrtf(rowSum4m(sq) / MC + eps);
#pragma unroll
    for (int t = 0; t < MNT; ++t) {
        const float4 g4 = *reinterpret_cast<const float4*>(gm + t * 16 + 4 * g), b4 = *reinterpret_cast<const float4*>(bt + t * 16 + 4 * g);
        acc[t][0] = (acc[t][0] - mean) * inv * g4.x + b4.x; acc[t][1] = (acc[t][1] - mean) * inv * g4.y + b4.y;
        acc[t][2] = (acc[t][2] - mean) * inv * g4.z + b4.z; acc[t][3] = (acc[t][3] - mean) * inv * g4.w + b4.w;
    }
}
// C-layout values of tiles {2s, 2s+1} -> B fragment of k-step s (see header)
__device__ __forceinline__ void packFrags(const floatx4 (&acc)[MNT], half8 (&f)[MNSTEP]) {
#pragma unroll
    for (int s = 0; s < MNSTEP; ++s) {
        half8 h;
        h[0] = (_Float16)acc[2 * s][0]; h[1] = (_Float16)acc[2 * s][1]; h[2] = (_Float16)acc[2 * s][2]; h[3] = (_Float16)acc[2 * s][3];
        h[4] = (_Float16)acc[2 * s + 1][0]; h[5] = (_Float16)acc[2 * s + 1][1]; h[6] = (_Float16)acc[2 * s + 1][2]; h[7] = (_Float16)acc[2 * s + 1][3];
        f[s] = h;
    }
}

// split precision (round 3): v = hi + lo with hi = fp16(v) (saturated), lo = fp16(v - hi); see linear.hip linear_split_rows_kernel
__device__ __forceinline__ void splitHalf(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)__builtin_fminf(__builtin_fmaxf(v, -65504.f), 65504.f);
    lo = (_Float16)__builtin_fminf(__builtin_fmaxf(v - (float)hi, -65504.f), 65504.f);
}
struct MlpHiLo { half8 hi, lo; };
__device__ __forceinline__ MlpHiLo splitFrag8(const float (&v)[8]) {
    _Float16 h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) splitHalf(v[i], h[i], l[i]);
    MlpHiLo o;
    o.hi = half8{h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
    o.lo = half8{l[0], l[1], l[2], l[3], l[4], l[5], l[6], l[7]};
    return o;
}
// C-layout values of tiles {2s, 2s+1} -> hi / lo B fragments of k-step s
__device__ __forceinline__ void packFragsSplit(const floatx4 (&acc)[MNT], half8 (&fh)[MNSTEP], half8 (&fl)[MNSTEP]) {
#pragma unroll
    for (int s = 0; s < MNSTEP; ++s) {
        const float v[8] = {acc[2 * s][0], acc[2 * s][1], acc[2 * s][2], acc[2 * s][3], acc[2 * s + 1][0], acc[2 * s + 1][1], acc[2 * s + 1][2], acc[2 * s + 1][3]};
        const MlpHiLo o = splitFrag8(v);
        fh[s] = o.hi; fl[s] = o.lo;
    }
}

// -------------------------------------------------------------------------------------
// Weights streamed by LDS-DMA.  (The first version of this kernel staged each weight slab through registers: one 128-row
// tile per workgroup walking nine exposed "load slab -> ds_write -> barrier -> MFMA" steps, SQ_WAIT_ANY = 70 % of wave
// cycles, MFMA busy 6 %, 87 us.)  The host packs the three matrices as TEN stages of 36 fragment rows (1 KB each, MFMA A-operand order):
//   stages 0,1    Wo columns [96h, 96h+96) x 192 k               row (ks, t)  <- Wo[96h + 16t + r][32ks + 8g + j]
//   stage 2+2q    W1 rows [96q, 96q+96) x 192 permuted k         row (ks, t)  <- W1p[96q + 16t + r][32ks + 8g + j]
//   stage 3+2q    W2 (all 192 rows) x permuted k [96q, 96q+96)   row (sp, t)  <- W2p[16t + r][96q + 32sp + 8g + j]
// and a workgroup copies stage s+1 into the other half of a two-slot ring with global_load_lds while the MFMAs of
// stage s run; one s_waitcnt + raw s_barrier per stage.  Biases and LayerNorm-1 parameters the stream phase needs
// come through the same DMA queue into LDS (an ordinary global load with a DMA in flight makes hipcc wait vmcnt(0));
// x is loaded in the prologue straight into the out-proj accumulator (acc = x, + bo after the barrier) and re-read for
// LayerNorm 3 once nothing is in flight; no store is issued before the last stage has landed.
#ifndef MLP_SPLIT_PREFETCH
#define MLP_SPLIT_PREFETCH 1          // fragment pairs of step s + 1 read under the MFMAs of step s (split kernel, one row tile per wave)
#endif
#ifndef MLP_SPLIT_ELASTIC
#define MLP_SPLIT_ELASTIC 1            // ten-wave single-frame variant (spills ~60 registers and is still 18 us per layer faster than two rounds of eight-wave blocks)
#endif
#ifndef MLP_SPLIT_PQ
#define MLP_SPLIT_PQ 64            // FC1 columns per piece in the split-precision kernel.  32 (24 KB stages, 78.8 KB of LDS: TWO workgroups per CU) was built and
                                   // measured in round 4: correct, and SLOWER -- 337 vs 318 us per four-frame launch (twice the stages and barriers; the kernel is
                                   // bound by its matrix work, not by occupancy: without the correction-term MFMAs 237 / 251 us)
#endif
constexpr int MP_FLOATS = 1280;                    // bo | ln1_g | ln1_b | b1 (384) | b2 | pad  -> five 1 KB DMA rows
constexpr int MP_BO = 0, MP_G1 = 192, MP_B1LN = 384, MP_B1 = 576, MP_B2 = 960;

typedef __attribute__((address_space(1))) const void* mlp_gsrc_t;
typedef __attribute__((address_space(3))) void* mlp_ldst_t;

struct MlpStreamArgs {
    const _Float16* att; const float* att32;         // attention output rows: fp16 (default) or fp32 (split-precision kernel)
    const float* x; const float* xb;
    const _Float16* Wp;                              // MS_STAGES x 36 x 512 halfs
    const float* params;                             // MP_FLOATS
    const float* ln_g; const float* ln_b;            // [4][192]
    float* out; _Float16* out16;
    const uint32_t* count; int max_rows; float eps;
    unsigned long long* trace;                       // debugging: per-workgroup phase timestamps, or nullptr
    int ncu;                                         // CUs of the device (tile plan, see the kernel)
    int dbg;                                         // timing ablations (wrong results): 1 no LN1, 2 no GELU, 4 no final LNs, 8 no stores, 16 (SPLIT) no correction-term MFMAs, 64 no weight DMA
    int sel, thr;                                    // 0: run; 1: run only when the row count is > thr; 2: only when it is <= thr (see enqueue)
};

__device__ __forceinline__ void mlpStageBarrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// LayerNorm with gamma / beta in LDS
__device__ __forceinline__ void mlpLayerNormLds(floatx4 (&acc)[MNT], const float* gm, const float* bt, int g, float eps) {
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < MNT; ++t) sum += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
    const float mean = rowSum4m(sum) / MC;
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < MNT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = acc[t][i] - mean; sq += d * d; }
    const float inv = 1.0f / sqrtf(rowSum4m(sq) / MC + eps);
#pragma unroll
    for (int t = 0; t < MNT; ++t) {
        const float4 g4 = *reinterpret_cast<const float4*>(gm + t * 16 + 4 * g), b4 = *reinterpret_cast<const float4*>(bt + t * 16 + 4 * g);
        acc[t][0] = (acc[t][0] - mean) * inv * g4.x + b4.x; acc[t][1] = (acc[t][1] - mean) * inv * g4.y + b4.y;
        acc[t][2] = (acc[t][2] - mean) * inv * g4.z + b4.z; acc[t][3] = (acc[t][3] - mean) * inv * g4.w + b4.w;
    }
}

// MT 16-row tiles per wave, NW waves: 128 rows per workgroup either way.  <1, 8> needs 4 waves/SIMD (<= 128 VGPRs, spills);
// <2, 4> runs at 2 waves/SIMD with 256 VGPRs and halves the LDS reads per MFMA.
// PQ = FC1 columns per piece: 64 => fifteen 24-row stages (24 KB).  A stage's MFMAs (0.2-0.35 us) are much shorter than a
// DMA round trip (~2 us), so the ring has THREE slots and stage s+2 is requested when stage s starts: the end-of-stage wait
// is a counted vmcnt that retires stage s+1 and leaves s+2 in flight.
// RS = slots of the weight ring, D = RS - 1 = prefetch distance in stages.  Three slots (85 KB: two workgroups per CU) are the default.
// Six slots (159 KB, DSVT_MLP_RING=6: stage s + 5 requested when stage s starts, four stages in flight behind the counted wait) were
// built on the theory that the ~1.3 us a stage takes is DMA latency; measured (round 2, tools/trace_mlp.py + rocprofv3): 46.5 vs 47.6 us
// host-launched, 46.3 vs 43.4 us inside the frame graph -- no gain, so latency is not what a stage waits for.  The timing ablations say
// what is: without MFMAs the launch is as long as with them (45.9 us), without LayerNorms / GELU / stores / MFMAs it still takes 27 us.
// That floor is the memory system: every one of the 256 workgroups streams the same 360 KB of weights L2 -> LDS, 92 MB per launch on
// top of the 93 MB of activations, i.e. 185 MB in 44 us = 4.2 TB/s against the ~6.4 TB/s the LDS-DMA stream reaches chip-wide
// (MI355X_MICROARCH.md "ldsdma-fill").  Fewer re-streamed weight bytes per row (more rows per workgroup) is the remaining lever.
// SPLIT (round 3, the fp32-grade mode): att arrives as fp32, every GEMM operand is a hi / lo fp16 pair and every product three MFMAs
// (w_hi a_hi + w_lo a_hi + w_hi a_lo, fp32 accumulate); a stage holds its fragment rows twice, the SRH rows of w_hi followed by the
// same rows of w_lo (48 KB stages, three slots: 149 KB, one eight-wave workgroup per CU at <= 256 registers); s1 and h are split in
// registers where the fp16 kernel rounds them; only the fp32 result is written.
template <int MT, int NW, int PQ, int RS, bool SPLIT = false>
__global__ void __launch_bounds__(64 * NW, (SPLIT ? (NW == 10 || NW == 4 ? 1 : 2) : MT == 1 ? 3 : 2))
encoder_mlp_stream_kernel(MlpStreamArgs a)
{
    constexpr int PQT = PQ / 16, PQS = PQ / 32, SRH = 6 * PQT, SR = SPLIT ? 2 * SRH : SRH, SB = SR * 1024, NWO = MC / PQ, NPIECE = MF / PQ;
    constexpr int LO = SRH * 1024;                  // byte offset of a stage's w_lo rows
    constexpr int NST = NWO + 2 * NPIECE, NRW = SR / NW, D = RS - 1;
    // ln_g [4][192] | ln_b [4][192] of the final LayerNorms (six 1 KB DMA rows): with three slots they take the slot stage NST-3 leaves when
    // the last piece starts (no LDS of their own: 78,848 B, two workgroups per CU); the deeper ring has no free slot then, they get 6 KB
    constexpr bool LN_IN_RING = RS == 3;
    constexpr int LNP = LN_IN_RING ? 0 : 8 * MC * 4;
    constexpr bool ELASTIC = MT == 1 && NW == 10;   // 8 .. 10 waves of 16 rows are live, chosen from the row count (see below)
    static_assert(12 * PQS == SRH && (ELASTIC || (SR % NW == 0 && (NRW == 3 || NRW == 6 || NRW == 12))), "uniform request count per wave");
    __shared__ __attribute__((aligned(16))) unsigned char lds[RS * SB + MP_FLOATS * 4 + LNP];     // RS = 3: 78,848 B; RS = 6: 158,720 B (one workgroup per CU)
    const uint32_t cnt = *a.count;
    const int M = (int)(cnt < (uint32_t)a.max_rows ? cnt : (uint32_t)a.max_rows);
    if ((a.sel == 1 && M <= a.thr) || (a.sel == 2 && M > a.thr)) return;     // the other kernel of the pair takes this row count
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Tile plan.  One workgroup per CU takes 41-45 us whatever the row count; a CU that hosts two 128-row workgroups takes 58-62 us
    // (measured, tools/mlp_rows.py), so 269 tiles on 256 CUs cost as much as 512.
    //  * <2, 4> (four waves x 32 rows): when the rows just overflow one tile per CU, the overflow (<= MLP_SMALL_MAX rows) is cut into
    //    32-row workgroups of ONE wave (the other waves exit): they stream the same weights but share a CU with a full workgroup
    //    at a fraction of its LDS / MFMA load.
    //  * <1, 10> (elastic): the block has ten waves of 16 rows and 8, 9 or 10 of them are live, so that ncu workgroups cover the
    //    rows (16 x 9 x 256 = 36,864 >= 34,483); the others exit at once.  Two waves per SIMD (168 VGPRs) overlap one wave's GELU /
    //    LayerNorm / waits with the other's MFMAs.  Beyond 160 rows per CU: eight live waves, workgroups in rounds.
    int nwa = NW;                                    // live waves of this workgroup
    int m0;
    bool small = false;
    if (ELASTIC) {
        int need = (M + 16 * a.ncu - 1) / (16 * a.ncu);
        // beyond ten live waves per CU (two frames per launch: ~270 rows per CU): TWO workgroups per CU one after the other, sized so
        // that two rounds of ncu workgroups still cover the rows -- 69k rows = 480 workgroups of nine waves, not 540 of eight (a
        // third round for 28 workgroups).  (LDS -- 78.8 KB -- would let two share a CU, the registers do not: 84 + 84 accumulation
        // registers per wave = three waves per SIMD = twelve wave slots per CU.)
        if (need > NW && RS == 3) need = (M + 32 * a.ncu - 1) / (32 * a.ncu);
        nwa = need <= 8 ? 8 : need <= NW ? need : 8;
        if (a.dbg & 32) nwa = 8;
        if (wave >= nwa) return;
        m0 = blockIdx.x * 16 * nwa;
    } else {
        constexpr int WGR = 16 * MT * NW;                // rows of a full workgroup (128; 256 for <2, 8>)
        const int T = (M + WGR - 1) / WGR;
        // Rounds: two full workgroups share a CU, so 2 ncu of them are a round.  What is left after the whole rounds -- at most
        // MLP_SMALL_MAX rows -- runs as small workgroups: after a half round (the rows just overflow one tile per CU; round 2's rule), or,
        // round 3, after any number of whole rounds: four frames per launch are 1074 tiles = two rounds of 512 and 50 more, which as full
        // workgroups are a third round of one lonely workgroup per CU (44 us of the launch's 158).
        int FULL = T;
        if (!SPLIT && WGR == MROWS && !(a.dbg & 32)) {
            const int base = T / (2 * a.ncu) * (2 * a.ncu), rem = T - base;
            if (base > 0 && rem > 0 && M - base * WGR <= MLP_SMALL_MAX) FULL = base;
            else if (rem > a.ncu && M - (base + a.ncu) * WGR <= MLP_SMALL_MAX) FULL = base + a.ncu;
        }
        // split precision (one workgroup per CU: ncu of them are a round): four frames per launch are 1074 tiles = four rounds and 50 more, which as full
        // workgroups are a fifth round on 50 CUs (51 us of the launch's 278); as 200 two-wave workgroups they are a round on 200 CUs that is bound by
        // the weight stream alone
        if (SPLIT && WGR == MROWS && (!kAblate || !(a.dbg & 32))) {
            const int base = T / a.ncu * a.ncu, rem = T - base;
            if (base > 0 && rem > 0 && M - base * WGR <= MLP_SMALL_MAX) FULL = base;
        }
        small = (int)blockIdx.x >= FULL;
        m0 = blockIdx.x * WGR;
        if (small) {
            m0 = FULL * WGR + ((int)blockIdx.x - FULL) * 16 * MT * MLP_SW;
            if (wave >= MLP_SW) return;
            nwa = MLP_SW;
        }
    }
    if (m0 >= M) return;
    const int r = lane & 15, g = lane >> 4;
    int row[MT], rc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        row[mt] = m0 + wave * 16 * MT + mt * 16 + r;
        rc[mt] = row[mt] < M ? row[mt] : M - 1;
    }
    const int nreq = (SR - wave + nwa - 1) / nwa;    // weight rows this wave requests per stage (elastic: 3 or 2)
    const float* prm = reinterpret_cast<const float*>(lds + RS * SB);
    int nmark = 0;
    auto mark = [&]() { if (a.trace && tid == 0 && nmark < 64 && blockIdx.x < kMlpTraceBlocks) a.trace[blockIdx.x * 64 + nmark] = clock64(); ++nmark; };   // (the trace buffer holds kMlpTraceBlocks workgroups: a four-frame launch has more)
    mark();
    auto request = [&](int s) {
        if (small) {                                 // (MLP_SW waves share the SR rows)
#pragma unroll
            for (int j = 0; j < SR / MLP_SW; ++j) {
                const int rw = wave + j * MLP_SW;
                __builtin_amdgcn_global_load_lds((mlp_gsrc_t)(a.Wp + ((size_t)s * SR + rw) * 512 + lane * 8),
                                                 (mlp_ldst_t)(lds + (s % RS) * SB + rw * 1024), 16, 0, 0);
            }
            return;
        }
        if (ELASTIC) {
#pragma unroll
            for (int j = 0; j < (SR + 7) / 8; ++j) {         // (at least eight live waves)
                const int rw = wave + j * nwa;
                if (rw < SR)
                    __builtin_amdgcn_global_load_lds((mlp_gsrc_t)(a.Wp + ((size_t)s * SR + rw) * 512 + lane * 8),
                                                     (mlp_ldst_t)(lds + (s % RS) * SB + rw * 1024), 16, 0, 0);
            }
            return;
        }
        if (kAblate && (a.dbg & 64)) return;
#pragma unroll
        for (int j = 0; j < (SR + NW - 1) / NW; ++j) {
            const int rw = wave + j * NW;
            if (SR % NW == 0 || rw < SR)
                __builtin_amdgcn_global_load_lds((mlp_gsrc_t)(a.Wp + ((size_t)s * SR + rw) * 512 + lane * 8),
                                                 (mlp_ldst_t)(lds + (s % RS) * SB + rw * 1024), 16, 0, 0);
        }
    };
    // end of stage st: stage st + 1 has landed, the younger requests (stages st + 2 .. st + D, as far as they exist) stay in flight;
    // then the barrier.  Each wave counts its own requests: `per` per stage.
    const int per = small ? SR / MLP_SW : ELASTIC ? nreq : NRW;
    auto stageEnd = [&](int st) {
        const int ahead = NST - 2 - st;              // stages younger than st + 1 that exist
        const int keep = (ahead < D - 1 ? (ahead > 0 ? ahead : 0) : D - 1) * per;
        switch (keep) {
            case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); break;
            case 24: asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;      // (waiting for more than needed is always correct)
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    request(0);
#pragma unroll
    for (int j = 0; j < MP_FLOATS / 256; ++j) {
        const int pr = wave + j * nwa;
        if (pr < MP_FLOATS / 256)
            __builtin_amdgcn_global_load_lds((mlp_gsrc_t)(a.params + pr * 256 + lane * 4), (mlp_ldst_t)(lds + RS * SB + pr * 1024), 16, 0, 0);
    }
    // the parameters of the final LayerNorms travel by LDS-DMA too, so that the epilogue's only global loads are x / xb
    auto requestLnParams = [&](unsigned char* dst) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int pr = wave + j * nwa;
            if (pr < 6)
                __builtin_amdgcn_global_load_lds((mlp_gsrc_t)((pr < 3 ? a.ln_g + pr * 256 : a.ln_b + (pr - 3) * 256) + lane * 4),
                                                 (mlp_ldst_t)(dst + pr * 1024), 16, 0, 0);
        }
    };
    if (!LN_IN_RING) requestLnParams(lds + RS * SB + MP_FLOATS * 4);
#pragma unroll
    for (int d = 1; d < D; ++d) request(d);
    // ---- prologue: the att row as the out-proj B operand, x straight into the out-proj accumulator -----------------
    half8 fa[MT][MNSTEP], fal[SPLIT ? MT : 1][MNSTEP];
    floatx4 araw[SPLIT ? MT : 1][2 * MNSTEP];
    floatx4 acc[MT][MNT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if constexpr (SPLIT) {
#pragma unroll
            for (int s = 0; s < 2 * MNSTEP; ++s) araw[mt][s] = *reinterpret_cast<const floatx4*>(a.att32 + (size_t)rc[mt] * MC + (s >> 1) * 32 + g * 8 + (s & 1) * 4);
        } else {
#pragma unroll
            for (int s = 0; s < MNSTEP; ++s) fa[mt][s] = *reinterpret_cast<const half8*>(a.att + (size_t)rc[mt] * MC + s * 32 + g * 8);
        }
#pragma unroll
        for (int t = 0; t < MNT; ++t) {
            const float4 xv = *reinterpret_cast<const float4*>(a.x + (size_t)rc[mt] * MC + t * 16 + 4 * g);
            acc[mt][t] = floatx4{xv.x, xv.y, xv.z, xv.w};
        }
    }
    // hipcc's own wait for these ordinary loads belongs here, not behind the next stage's request (see linear.hip)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if constexpr (SPLIT) {
#pragma unroll
            for (int s = 0; s < 2 * MNSTEP; ++s) asm volatile("" :: "v"(araw[mt][s]));
        } else {
#pragma unroll
            for (int s = 0; s < MNSTEP; ++s) asm volatile("" :: "v"(fa[mt][s]));
        }
#pragma unroll
        for (int t = 0; t < MNT; ++t) asm volatile("" :: "v"(acc[mt][t]));
    }
    mark();
    mlpStageBarrier();                               // stages 0 .. D-1 + parameters landed
    mark();
    if constexpr (SPLIT) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < MNSTEP; ++s) {
                const floatx4 u = araw[mt][2 * s], w = araw[mt][2 * s + 1];
                const float v[8] = {u[0], u[1], u[2], u[3], w[0], w[1], w[2], w[3]};
                const MlpHiLo o = splitFrag8(v);
                fa[mt][s] = o.hi; fal[mt][s] = o.lo;
            }
    }
#pragma unroll
    for (int t = 0; t < MNT; ++t) {
        const float4 b = *reinterpret_cast<const float4*>(prm + MP_BO + t * 16 + 4 * g);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { acc[mt][t][0] += b.x; acc[mt][t][1] += b.y; acc[mt][t][2] += b.z; acc[mt][t][3] += b.w; }
    }
    const unsigned char* lbase = lds + lane * 16;
    // (split precision, one row tile per wave) one pass of NS steps over a stage in LDS: step s reads NT (w_hi, w_lo) fragment pairs and issues 3 NT MFMAs; the
    // pairs of step s + 1 are read BEFORE the MFMAs of step s (left to itself the step is "read eight fragments, wait ~300-500 cycles of LDS latency with
    // eight waves reading, 192 cycles of MFMAs" -- the same finding as in linear_split_resident_kernel)
    auto gemmSplit = [&](const unsigned char* slot, auto nsTag, auto ntTag, auto fragOff, auto opHi, auto opLo, auto accOf) {
        constexpr int NS = decltype(nsTag)::value, NT = decltype(ntTag)::value;
        half8 wq[2][2 * NT];
        auto ld = [&](int s_, half8 (&w)[2 * NT]) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                w[t] = *reinterpret_cast<const half8*>(slot + fragOff(s_, t));
                w[NT + t] = *reinterpret_cast<const half8*>(slot + LO + fragOff(s_, t));
            }
        };
        ld(0, wq[0]);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            if (s_ + 1 < NS) ld(s_ + 1, wq[(s_ + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const half8 oh = opHi(s_), ol = opLo(s_);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                floatx4& c = accOf(s_, t);
                if (!kAblate || !(a.dbg & 16)) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[s_ & 1][NT + t], oh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[s_ & 1][t], ol, c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[s_ & 1][t], oh, c, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // (one wave per SIMD, MT row tiles per wave -- the DSVT_MLP_SPLIT_VARIANT experiments) the same pass with NT fragment pairs per step feeding 3 NT MT MFMAs
    auto gemmSplitM = [&](const unsigned char* slot, auto nsTag, auto ntTag, auto fragOff, auto opHi, auto opLo, auto accOf) {
        constexpr int NS = decltype(nsTag)::value, NT = decltype(ntTag)::value;
        half8 wq[2][2 * NT];
        auto ld = [&](int s_, half8 (&w)[2 * NT]) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                w[t] = *reinterpret_cast<const half8*>(slot + fragOff(s_, t));
                w[NT + t] = *reinterpret_cast<const half8*>(slot + LO + fragOff(s_, t));
            }
        };
        ld(0, wq[0]);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            if (s_ + 1 < NS) ld(s_ + 1, wq[(s_ + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            // the three products of an accumulator are a dependent chain: with one wave per SIMD nobody fills its bubbles, so the NT MT accumulators of the step
            // take product 1, then product 2, then product 3 (the order of a row's sums does not change)
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        floatx4& c = accOf(s_, t, mt);
                        if (pr == 0 && (!kAblate || !(a.dbg & 16))) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[s_ & 1][NT + t], opHi(s_, mt), c, 0, 0, 0);
                        if (pr == 1 && (!kAblate || !(a.dbg & 16))) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[s_ & 1][t], opLo(s_, mt), c, 0, 0, 0);
                        if (pr == 2) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[s_ & 1][t], opHi(s_, mt), c, 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    constexpr bool PREFM = SPLIT && MT > 1 && NW == 4 && MLP_SPLIT_PREFETCH;
    constexpr int NTM = 2;                           // fragment pairs per step of gemmSplitM (two buffers of 2 NTM fragments: 32 registers)
    constexpr bool PREF = SPLIT && MT == 1 && NW == 8 && MLP_SPLIT_PREFETCH;          // (the ten-wave single-frame variant has 168 registers: 219 spilled with the second fragment set)
    using std::integral_constant;

    // ---- stages 0 .. NWO-1: out-proj, PQ columns per stage ---------------------------------------------------------------
#pragma unroll
    for (int h = 0; h < NWO; ++h) {
        request(h + D);                              // (NWO + D <= NST: the stages beyond the out-proj are W1 pieces / W2 slabs)
        const unsigned char* sl = lbase + (h % RS) * SB;
        if constexpr (PREF) {
            gemmSplit(sl, integral_constant<int, MNSTEP>{}, integral_constant<int, PQT>{}, [&](int ks, int t) { return (ks * PQT + t) * 1024; },
                      [&](int ks) { return fa[0][ks]; }, [&](int ks) { return fal[0][ks]; }, [&](int, int t) -> floatx4& { return acc[0][h * PQT + t]; });
        } else if constexpr (PREFM) {
            constexpr int TG = PQT / NTM;
            gemmSplitM(sl, integral_constant<int, MNSTEP * TG>{}, integral_constant<int, NTM>{}, [&](int s_, int t) { return ((s_ / TG) * PQT + (s_ % TG) * NTM + t) * 1024; },
                       [&](int s_, int mt) { return fa[mt][s_ / TG]; }, [&](int s_, int mt) { return fal[mt][s_ / TG]; },
                       [&](int s_, int t, int mt) -> floatx4& { return acc[mt][h * PQT + (s_ % TG) * NTM + t]; });
        } else
#pragma unroll
        for (int ks = 0; ks < MNSTEP; ++ks) {
#pragma unroll
            for (int t = 0; t < PQT; ++t) {
                const half8 wf = *reinterpret_cast<const half8*>(sl + (ks * PQT + t) * 1024);
                if (SPLIT && (!kAblate || !(a.dbg & 16))) {
                    const half8 wl = *reinterpret_cast<const half8*>(sl + LO + (ks * PQT + t) * 1024);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[mt][h * PQT + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, fa[mt][ks], acc[mt][h * PQT + t], 0, 0, 0);
                        acc[mt][h * PQT + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, fal[mt][ks], acc[mt][h * PQT + t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][h * PQT + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, fa[mt][ks], acc[mt][h * PQT + t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        mark();
        if (h + 1 < NWO) stageEnd(h);
    }
    half8 fs1[MT][MNSTEP], fs1l[SPLIT ? MT : 1][MNSTEP];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (!(a.dbg & 1)) mlpLayerNormLds(acc[mt], prm + MP_G1, prm + MP_B1LN, g, a.eps);         // s1 = LN1(att Wo^T + bo + x)
        if constexpr (SPLIT) packFragsSplit(acc[mt], fs1[mt], fs1l[mt]);
        else packFrags(acc[mt], fs1[mt]);                                       // s1 as the FC1 operand
    }
#pragma unroll
    for (int t = 0; t < MNT; ++t) {                                     // the FC2 accumulator starts at s1 + b2 (LN2's residual, fp32)
        const float4 b = *reinterpret_cast<const float4*>(prm + MP_B2 + t * 16 + 4 * g);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { acc[mt][t][0] += b.x; acc[mt][t][1] += b.y; acc[mt][t][2] += b.z; acc[mt][t][3] += b.w; }
    }
    mark();
    stageEnd(NWO - 1);                               // first W1 piece landed
    mark();

    // ---- FC1 in four 96-column pieces, each consumed at once as a 96-wide K slab of FC2 -------------------------------
    // W1 piece q = stage NWO + 2q, W2 slab q = stage NWO + 2q + 1
#pragma unroll 1
    for (int q = 0; q < NPIECE; ++q) {
        const int stA = NWO + 2 * q, stB = stA + 1;
        if (stA + D < NST) request(stA + D);
        else if (LN_IN_RING) requestLnParams(lds + (NST % RS) * SB);      // last piece: nothing left to stream, the slot stage NST-3 has just left is free
        const unsigned char* slotA = lbase + (stA % RS) * SB;
        floatx4 acc2[MT][PQT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < PQT; ++t) acc2[mt][t] = floatx4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PREF) {
            gemmSplit(slotA, integral_constant<int, MNSTEP>{}, integral_constant<int, PQT>{}, [&](int ks, int t) { return (ks * PQT + t) * 1024; },
                      [&](int ks) { return fs1[0][ks]; }, [&](int ks) { return fs1l[0][ks]; }, [&](int, int t) -> floatx4& { return acc2[0][t]; });
        } else if constexpr (PREFM) {
            constexpr int TG = PQT / NTM;
            gemmSplitM(slotA, integral_constant<int, MNSTEP * TG>{}, integral_constant<int, NTM>{}, [&](int s_, int t) { return ((s_ / TG) * PQT + (s_ % TG) * NTM + t) * 1024; },
                       [&](int s_, int mt) { return fs1[mt][s_ / TG]; }, [&](int s_, int mt) { return fs1l[mt][s_ / TG]; },
                       [&](int s_, int t, int mt) -> floatx4& { return acc2[mt][(s_ % TG) * NTM + t]; });
        } else
#pragma unroll
        for (int ks = 0; ks < MNSTEP; ++ks) {
#pragma unroll
            for (int t = 0; t < PQT; ++t) {
                const half8 wf = *reinterpret_cast<const half8*>(slotA + (ks * PQT + t) * 1024);
                if (SPLIT && (!kAblate || !(a.dbg & 16))) {
                    const half8 wl = *reinterpret_cast<const half8*>(slotA + LO + (ks * PQT + t) * 1024);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc2[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, fs1[mt][ks], acc2[mt][t], 0, 0, 0);
                        acc2[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, fs1l[mt][ks], acc2[mt][t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc2[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, fs1[mt][ks], acc2[mt][t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        half8 fh[MT][PQS], fhl[SPLIT ? MT : 1][PQS];
#pragma unroll
        for (int sp = 0; sp < PQS; ++sp) {
            const float4 b0 = *reinterpret_cast<const float4*>(prm + MP_B1 + q * PQ + (2 * sp) * 16 + 4 * g);
            const float4 b1v = *reinterpret_cast<const float4*>(prm + MP_B1 + q * PQ + (2 * sp + 1) * 16 + 4 * g);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                half8 h;
                if constexpr (SPLIT) {
                    const float v[8] = {mlpGelu(acc2[mt][2 * sp][0] + b0.x), mlpGelu(acc2[mt][2 * sp][1] + b0.y), mlpGelu(acc2[mt][2 * sp][2] + b0.z), mlpGelu(acc2[mt][2 * sp][3] + b0.w),
                                        mlpGelu(acc2[mt][2 * sp + 1][0] + b1v.x), mlpGelu(acc2[mt][2 * sp + 1][1] + b1v.y), mlpGelu(acc2[mt][2 * sp + 1][2] + b1v.z), mlpGelu(acc2[mt][2 * sp + 1][3] + b1v.w)};
                    const MlpHiLo o = splitFrag8(v);
                    h = o.hi; fhl[mt][sp] = o.lo;
                } else
                if (a.dbg & 2) {
                    h[0] = (_Float16)acc2[mt][2 * sp][0]; h[1] = (_Float16)acc2[mt][2 * sp][1]; h[2] = (_Float16)acc2[mt][2 * sp][2]; h[3] = (_Float16)acc2[mt][2 * sp][3];
                    h[4] = (_Float16)acc2[mt][2 * sp + 1][0]; h[5] = (_Float16)acc2[mt][2 * sp + 1][1]; h[6] = (_Float16)acc2[mt][2 * sp + 1][2]; h[7] = (_Float16)acc2[mt][2 * sp + 1][3];
                } else {
                h[0] = (_Float16)mlpGelu(acc2[mt][2 * sp][0] + b0.x); h[1] = (_Float16)mlpGelu(acc2[mt][2 * sp][1] + b0.y);
                h[2] = (_Float16)mlpGelu(acc2[mt][2 * sp][2] + b0.z); h[3] = (_Float16)mlpGelu(acc2[mt][2 * sp][3] + b0.w);
                h[4] = (_Float16)mlpGelu(acc2[mt][2 * sp + 1][0] + b1v.x); h[5] = (_Float16)mlpGelu(acc2[mt][2 * sp + 1][1] + b1v.y);
                h[6] = (_Float16)mlpGelu(acc2[mt][2 * sp + 1][2] + b1v.z); h[7] = (_Float16)mlpGelu(acc2[mt][2 * sp + 1][3] + b1v.w);
                }
                fh[mt][sp] = h;
            }
        }
        mark();
        stageEnd(stA);                               // W2 slab landed; everyone is done with the W1 piece
        if (kAblate) mark();
        if (stB + D < NST) request(stB + D);
        const unsigned char* slotB = lbase + (stB % RS) * SB;
        if constexpr (PREF) {                        // steps of four column tiles: (32-column slice sp of the piece, tile group)
            gemmSplit(slotB, integral_constant<int, PQS * 3>{}, integral_constant<int, 4>{}, [&](int s_, int t) { return ((s_ / 3) * 12 + (s_ % 3) * 4 + t) * 1024; },
                      [&](int s_) { return fh[0][s_ / 3]; }, [&](int s_) { return fhl[0][s_ / 3]; }, [&](int s_, int t) -> floatx4& { return acc[0][(s_ % 3) * 4 + t]; });
        } else if constexpr (PREFM) {
            constexpr int TG = MNT / NTM;
            gemmSplitM(slotB, integral_constant<int, PQS * TG>{}, integral_constant<int, NTM>{}, [&](int s_, int t) { return ((s_ / TG) * 12 + (s_ % TG) * NTM + t) * 1024; },
                       [&](int s_, int mt) { return fh[mt][s_ / TG]; }, [&](int s_, int mt) { return fhl[mt][s_ / TG]; },
                       [&](int s_, int t, int mt) -> floatx4& { return acc[mt][(s_ % TG) * NTM + t]; });
        } else
#pragma unroll
        for (int sp = 0; sp < PQS; ++sp) {
#pragma unroll
            for (int t = 0; t < MNT; ++t) {
                const half8 wf = *reinterpret_cast<const half8*>(slotB + (sp * 12 + t) * 1024);
                if (SPLIT && (!kAblate || !(a.dbg & 16))) {
                    const half8 wl = *reinterpret_cast<const half8*>(slotB + LO + (sp * 12 + t) * 1024);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, fh[mt][sp], acc[mt][t], 0, 0, 0);
                        acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, fhl[mt][sp], acc[mt][t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, fh[mt][sp], acc[mt][t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        mark();
        if (stB + 1 < NST) stageEnd(stB);            // next W1 piece landed; everyone is done with the W2 slab
        if (kAblate) mark();
    }
    mark();
    // ---- s2 = LN2(s1 + f + b2) (already summed); x' = LN3(s2 + x); [x' = LN4(x' + xb)]: nothing in flight any more ------
    const float* lnL = reinterpret_cast<const float*>(LN_IN_RING ? lds + (NST % RS) * SB : lds + RS * SB + MP_FLOATS * 4);        // ln_g [4][192] | ln_b [4][192]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int rcl = rc[mt];
        asm volatile("" : "+v"(rcl));                // (row addresses re-formed here, not carried through the streaming loop in spilled registers)
        if (!(a.dbg & 4)) {
        mlpLayerNormLds(acc[mt], lnL + MC, lnL + 4 * MC + MC, g, a.eps);
#pragma unroll
        for (int t = 0; t < MNT; ++t) {
            const float4 xv = *reinterpret_cast<const float4*>(a.x + (size_t)rcl * MC + t * 16 + 4 * g);
            acc[mt][t][0] += xv.x; acc[mt][t][1] += xv.y; acc[mt][t][2] += xv.z; acc[mt][t][3] += xv.w;
        }
        mlpLayerNormLds(acc[mt], lnL + 2 * MC, lnL + 4 * MC + 2 * MC, g, a.eps);
        }
        __builtin_amdgcn_sched_barrier(0);           // (x and xb rows of a tile together: 96 registers of loads)
        if (a.xb) {
#pragma unroll
            for (int t = 0; t < MNT; ++t) {
                const float4 xv = *reinterpret_cast<const float4*>(a.xb + (size_t)rcl * MC + t * 16 + 4 * g);
                acc[mt][t][0] += xv.x; acc[mt][t][1] += xv.y; acc[mt][t][2] += xv.z; acc[mt][t][3] += xv.w;
            }
            mlpLayerNormLds(acc[mt], lnL + 3 * MC, lnL + 4 * MC + 3 * MC, g, a.eps);
        }
        __builtin_amdgcn_sched_barrier(0);           // (keeps the second tile's loads from being hoisted over the first tile's arithmetic: spills)
    }
    // every load of the epilogue (LayerNorm parameters, x, xb of BOTH row tiles) is issued before the first store: gfx950 counts
    // loads and stores in one vmcnt, so a load after a store is awaited behind that store's write acknowledgement
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int rw = row[mt];
        asm volatile("" : "+v"(rw));                 // (the store addresses are formed here, not kept alive through the streaming loop)
        // wide stores (see linearEpilogue): after v_permlane16_swap of tiles t, t + 1 the lane of row r holds 8 consecutive columns
#pragma unroll
        for (int t = 0; t < MNT; t += 2) {
            floatx4 X = acc[mt][t], Y = acc[mt][t + 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(X[i]), __float_as_uint(Y[i]), false, false);
                X[i] = __uint_as_float(sw[0]); Y[i] = __uint_as_float(sw[1]);
            }
            if (rw < M && !(a.dbg & 8)) {
                const int col = t * 16 + (g & 1) * 16 + (g >> 1) * 8;
                float* o = a.out + (size_t)rw * MC + col;
                *reinterpret_cast<float4*>(o) = make_float4(X[0], X[1], X[2], X[3]);
                *reinterpret_cast<float4*>(o + 4) = make_float4(Y[0], Y[1], Y[2], Y[3]);
                if constexpr (!SPLIT) {
                    half8 h = {(_Float16)X[0], (_Float16)X[1], (_Float16)X[2], (_Float16)X[3], (_Float16)Y[0], (_Float16)Y[1], (_Float16)Y[2], (_Float16)Y[3]};
                    *reinterpret_cast<half8*>(a.out16 + (size_t)rw * MC + col) = h;
                }
            }
        }
    }
    if (a.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); mark(); }
}

// position p = 32s + 8g + j of a permuted weight row holds the column k(p) the chained B fragment carries there
static inline int permuteK(int p) {
    const int s = p / 32, q = p % 32, g = q / 8, j = q % 8;
    return 32 * s + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
}

class DsvtEncoderMlpPlugin : public Plugin {
public:
    int max_rows_, has_block_ln_; float eps_;
    int frames_ = 0;                                              // optional field "frames": frames whose rows one launch carries (0 = unknown: decided by the row count on the device)
    int split_ = 0;                                               // optional field "split_precision": fp32 att in, fp32 x' out, hi / lo fp16 operands (fp32 grade)
    std::vector<float> wo_, w1_, w2_, bo_, b1_, b2_, lg_, lb_;     // as given (natural order)
    float *lg_dev_ = nullptr, *lb_dev_ = nullptr;
    _Float16* wp_dev_ = nullptr; float* prm_dev_ = nullptr;       // stage image + LDS parameter block of the streamed kernel
    int pq_ = 64;                                                 // FC1 columns per piece of that image
    bool ok_ = false;
    DsvtEncoderMlpPlugin(int max_rows, int has_block_ln, float eps, const float* wo, const float* w1, const float* w2,
                         const float* bo, const float* b1, const float* b2, const float* lg, const float* lb, int split = 0)
        : max_rows_(max_rows), has_block_ln_(has_block_ln), eps_(eps), split_(split), wo_(wo, wo + MC * MC), w1_(w1, w1 + MF * MC),
          w2_(w2, w2 + MC * MF), bo_(bo, bo + MC), b1_(b1, b1 + MF), b2_(b2, b2 + MC),
          lg_(lg, lg + (3 + has_block_ln) * MC), lb_(lb, lb + (3 + has_block_ln) * MC) {
        lg_.resize(4 * MC, 1.f); lb_.resize(4 * MC, 0.f);
        if (split_) pq_ = MLP_SPLIT_PQ;
        auto upF = [](const std::vector<float>& src, float** d) {
            return dsvtMalloc(d, sizeof(float) * src.size()) == hipSuccess &&
                   hipMemcpy(*d, src.data(), sizeof(float) * src.size(), hipMemcpyHostToDevice) == hipSuccess;
        };
        ok_ = upF(lg_, &lg_dev_) && upF(lb_, &lb_dev_);
        if (!ok_) return;
        // stage image (see encoder_mlp_stream_kernel), for pq_ FC1 columns per piece
        // (split precision: a stage holds its SRH fragment rows twice, w_hi = fp16(w) then w_lo = fp16(w - w_hi))
        const int PQ = pq_, PQT = PQ / 16, PQS = PQ / 32, SRH = 6 * PQT, SR = split_ ? 2 * SRH : SRH, NWO = MC / PQ, NPIECE = MF / PQ;
        std::vector<_Float16> wp((size_t)(NWO + 2 * NPIECE) * SR * 512);
        // (ablate build, DSVT_MLP_LO8=1: w_lo rounded to what an e4m3 byte + a power-of-two scale per weight row would hold -- the accuracy half of
        // "stream w_lo at half rate", measured before any kernel is written for it: tools/mx_box_sweep.py)
        const bool lo8 = split_ && ablateEnv("DSVT_MLP_LO8", 0) != 0;
        auto rowScales = [&](const std::vector<float>& W, int rows, int cols) {
            std::vector<float> sc(rows, 1.f);
            for (int n = 0; n < rows; ++n) {
                float mx = 0.f;
                for (int k = 0; k < cols; ++k) { const float v = W[(size_t)n * cols + k]; mx = std::max(mx, std::fabs(v - (float)(_Float16)v)); }
                if (mx > 0.f) sc[n] = std::ldexp(1.f, (int)std::floor(std::log2(224.f / mx)));
            }
            return sc;
        };
        std::vector<float> sco, sc1, sc2;
        if (lo8) { sco = rowScales(wo_, MC, MC); sc1 = rowScales(w1_, MF, MC); sc2 = rowScales(w2_, MC, MF); }
        auto e4m3 = [](float x) {
            const float a = std::fabs(x);
            if (a == 0.f) return 0.f;
            int e; (void)std::frexp(a, &e);
            const int E = std::max(e - 1, -6);
            const float q = std::ldexp(1.f, E - 3);
            return std::copysign(std::min(std::nearbyint(a / q) * q, 448.f), x);
        };
        auto put = [&](int stage, int rowi, int lane, int j, float v, float sc = 0.f) {
            const _Float16 hi = (_Float16)v;
            wp[(((size_t)stage * SR + rowi) * 64 + lane) * 8 + j] = hi;
            float lo = v - (float)hi;
            if (lo8 && sc > 0.f) lo = e4m3(lo * sc) / sc;
            if (split_) wp[(((size_t)stage * SR + SRH + rowi) * 64 + lane) * 8 + j] = (_Float16)lo;
        };
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int r = lane & 15, g = lane >> 4;
                for (int h = 0; h < NWO; ++h)
                    for (int ks = 0; ks < 6; ++ks)
                        for (int t = 0; t < PQT; ++t)
                            put(h, ks * PQT + t, lane, j, wo_[(size_t)(PQ * h + 16 * t + r) * MC + 32 * ks + 8 * g + j], lo8 ? sco[PQ * h + 16 * t + r] : 0.f);
                for (int q = 0; q < NPIECE; ++q) {
                    for (int ks = 0; ks < 6; ++ks)
                        for (int t = 0; t < PQT; ++t)
                            put(NWO + 2 * q, ks * PQT + t, lane, j, w1_[(size_t)(PQ * q + 16 * t + r) * MC + permuteK(32 * ks + 8 * g + j)], lo8 ? sc1[PQ * q + 16 * t + r] : 0.f);
                    for (int sp = 0; sp < PQS; ++sp)
                        for (int t = 0; t < 12; ++t)
                            put(NWO + 2 * q + 1, sp * 12 + t, lane, j, w2_[(size_t)(16 * t + r) * MF + permuteK(PQ * q + 32 * sp + 8 * g + j)], lo8 ? sc2[16 * t + r] : 0.f);
                }
            }
        std::vector<float> prm(MP_FLOATS, 0.f);
        for (int i = 0; i < MC; ++i) { prm[MP_BO + i] = bo_[i]; prm[MP_G1 + i] = lg_[i]; prm[MP_B1LN + i] = lb_[i]; prm[MP_B2 + i] = b2_[i]; }
        for (int i = 0; i < MF; ++i) prm[MP_B1 + i] = b1_[i];
        ok_ = dsvtMalloc(&wp_dev_, sizeof(_Float16) * wp.size()) == hipSuccess &&
              hipMemcpy(wp_dev_, wp.data(), sizeof(_Float16) * wp.size(), hipMemcpyHostToDevice) == hipSuccess && upF(prm, &prm_dev_);
    }
    ~DsvtEncoderMlpPlugin() override {
        for (void* p : {(void*)lg_dev_, (void*)lb_dev_, (void*)wp_dev_, (void*)prm_dev_})
            if (p) (void)dsvtFree(p);
    }
    const char* type() const override { return "DsvtEncoderMlpPlugin"; }
    int nbOutputs() const override { return split_ ? 1 : 2; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i < 0 || i >= nbOutputs()) return -1;
        *out = dims3(in[0].d[0], max_rows_, MC); return 0;
    }
    int outputType(int i, const int32_t*, int) const override { return i == 0 ? DSVT_FLOAT : DSVT_HALF; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int nbIn, int) const override {
        if (io[pos].format != DSVT_FORMAT_LINEAR) return false;
        if (pos == 0) return io[pos].type == (split_ ? DSVT_FLOAT : DSVT_HALF);
        if (pos == 1) return io[pos].type == DSVT_INT32;
        if (pos < nbIn) return io[pos].type == DSVT_FLOAT;
        return io[pos].type == (pos == nbIn ? DSVT_FLOAT : DSVT_HALF);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    // inputs: att16 [1,P,192] half, count [1], x [1,P,192] f32 (, xb [1,P,192] f32)   outputs: x' f32, x' f16
    int enqueue(const DsvtPluginTensorDesc*, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        if (!ok_) return static_cast<int>(hipErrorOutOfMemory);
        MlpStreamArgs b{};
        b.att = split_ ? nullptr : static_cast<const _Float16*>(in[0]); b.att32 = split_ ? static_cast<const float*>(in[0]) : nullptr;
        b.count = static_cast<const uint32_t*>(in[1]);
        b.x = static_cast<const float*>(in[2]); b.xb = has_block_ln_ ? static_cast<const float*>(in[3]) : nullptr;
        b.Wp = wp_dev_; b.params = prm_dev_; b.ln_g = lg_dev_; b.ln_b = lb_dev_;
        b.out = static_cast<float*>(out[0]); b.out16 = split_ ? nullptr : static_cast<_Float16*>(out[1]); b.max_rows = max_rows_; b.eps = eps_;
        if (zeroFill) {
            DSVT_CHECK(hipMemsetAsync(out[0], 0, sizeof(float) * (size_t)max_rows_ * MC, stream));
            if (!split_) DSVT_CHECK(hipMemsetAsync(out[1], 0, sizeof(_Float16) * (size_t)max_rows_ * MC, stream));
        }
        const int ncu = deviceCUs();
        b.ncu = ncu;
        if (split_) {                  // one kernel for every row count: eight waves x 16 rows, one workgroup per CU (149 KB of LDS)
            static int sdbg = -1; if (sdbg < 0) sdbg = ablateEnv("DSVT_MLP_DBG", 0);
            b.dbg = sdbg;
#if MLP_SPLIT_ELASTIC
            // one frame per launch: ten-wave blocks of which 8 .. 10 waves are live, so that ONE workgroup per CU covers the rows (a 34.4k-row frame =
            // 256 workgroups of nine waves; fixed eight-wave workgroups are 269 = a second round of 13).  The 168-register budget of ten waves spills
            // and the variant still wins at one frame, 89.9 against 107.8 us per launch; at two frames it loses (174 against 166 us) and is not used
            if (frames_ > 0 ? frames_ <= 1 : max_rows_ <= 5 * ncu * MROWS / 4) {
                hipLaunchKernelGGL((encoder_mlp_stream_kernel<1, 10, MLP_SPLIT_PQ, 3, true>), dim3(cdiv(max_rows_, MROWS)), dim3(640), 0, stream, b);
                return lastError();
            }
#endif
            // Round 5, review item 5 ("32-row tiles to halve the fragment reads"): the eight-wave kernel holds 250 registers at ONE 16-row tile per wave, so more rows
            // per fragment read means one wave per SIMD at up to 512 registers.  Built as <3, 4> (four waves x 48 rows = 192 rows per weight stream: 256 + 252
            // registers, no spill; a third of the fragment reads, two thirds of the weight bytes per row), <2, 4> and <4, 4> (227 spilled registers), with the
            // fragment pairs of step s + 1 read under the MFMAs of step s and the three products of an accumulator issued a step apart (gemmSplitM).  Measured
            // (tools/mlp_split_variants.py, four frames = 137k rows per launch, us): <1, 8> shipped 250-256, <3, 4> 253-260, <2, 4> 288-290, <4, 4> 440.  Outputs agree with
            // the shipped kernel up to hipcc's fma contraction of the LayerNorm / GELU arithmetic (1.6 % of the values by one ulp; without gemmSplitM bit for bit).
            // Why no gain (s_memtime stamps of a <3, 4> workgroup, DSVT_MLP_TRACE): 188k cycles per 192 rows of which the MFMAs are 52k; the prologue (att + x rows
            // of every CU at once: 10 B / cycle / CU, the HBM burst) is 29k and the epilogue 31k with nothing to run under them (149 KB of LDS: one workgroup per
            // CU), and inside the streaming loop a stage of 216 MFMAs (3.5k cycles) takes 5.7k (FC1 + GELU + split) / 8.7k (FC2): with ONE wave per SIMD
            // nothing fills the GELU / split / LayerNorm VALU blocks, the barrier skew or the request issue.  The ablation ladder of both kernels (DSVT_MLP_DBG): no weight
            // stream -10 .. -17 us (the L2 -> LDS stream is NOT what bounds either), no correction MFMAs -30 (<1, 8>) / -55 (<3, 4>), no LayerNorm / GELU -14 / -23, no stores
            // -19 / -14, all of them 166 / 148 us left.  <1, 8> per 128 rows: 46k cycles of LDS fragment reads (8 waves x 737 KB at 128 B / cycle) + 35k of MFMAs
            // in 108k; <3, 4>: 23k + 52k in 188k -- the headroom is in <3, 4>, behind hand-interleaved VALU / MFMA streams (sched_group_barrier); not built this round.
            if constexpr (kAblate) {   // experiments (DSVT_MLP_SPLIT_VARIANT): one wave per SIMD at up to 512 registers, MT 16-row tiles per wave
                static int sv = -1; if (sv < 0) sv = ablateEnv("DSVT_MLP_SPLIT_VARIANT", 0);
                static unsigned long long* tr = nullptr; static int tron = -1;         // per-workgroup phase stamps (tools/mlp_split_variants.py)
                if (tron < 0) { tron = ablateEnv("DSVT_MLP_TRACE", 0) ? 1 : 0; if (tron) (void)hipMallocManaged(&tr, 8 * 64 * kMlpTraceBlocks); }
                b.trace = tr;
                struct Dump { hipStream_t st; unsigned long long* tr; int on; ~Dump() {
                    if (!on) return;
                    (void)hipStreamSynchronize(st);
                    for (int w : {0, 200, 700}) { fprintf(stderr, "[mlp trace wg%d]", w); for (int i = 1; i < 40; ++i) fprintf(stderr, " %lld", (long long)(tr[w * 64 + i] - tr[w * 64])); fprintf(stderr, "\n"); }
                } } dump{stream, tr, tron};
                if (sv == 0) { hipLaunchKernelGGL((encoder_mlp_stream_kernel<1, 8, MLP_SPLIT_PQ, 3, true>), dim3(cdiv(max_rows_, MROWS) + MLP_SMALL_MAX / (16 * MLP_SW)), dim3(512), 0, stream, b); return lastError(); }
                if (sv == 3) { hipLaunchKernelGGL((encoder_mlp_stream_kernel<3, 4, MLP_SPLIT_PQ, 3, true>), dim3(cdiv(max_rows_, 192)), dim3(256), 0, stream, b); return lastError(); }
                if (sv == 4) { hipLaunchKernelGGL((encoder_mlp_stream_kernel<4, 4, MLP_SPLIT_PQ, 3, true>), dim3(cdiv(max_rows_, 256)), dim3(256), 0, stream, b); return lastError(); }
                if (sv == 2) { hipLaunchKernelGGL((encoder_mlp_stream_kernel<2, 4, MLP_SPLIT_PQ, 3, true>), dim3(cdiv(max_rows_, 128) + MLP_SMALL_MAX / (32 * MLP_SW)), dim3(256), 0, stream, b); return lastError(); }
            }
            // (+ the two-wave workgroups of a last partial round: at most MLP_SMALL_MAX rows of 32)
            hipLaunchKernelGGL((encoder_mlp_stream_kernel<1, 8, MLP_SPLIT_PQ, 3, true>), dim3(cdiv(max_rows_, MROWS) + MLP_SMALL_MAX / (16 * MLP_SW)), dim3(512), 0, stream, b);
            return lastError();
        }
        // Kernel by row count.  Up to 2.5 x 128 rows per CU (one or two frames per launch): <1,10> elastic, ONE workgroup per CU at a time
        // (168 registers per wave).  Beyond (three or more frames per launch): <2,4>, four waves x 32 rows at 256 registers, of which TWO
        // share a CU (one wave of each per SIMD) and overlap each other's LayerNorm / GELU / store phases with MFMA + DMA -- a CU hosting
        // two takes 58-62 us for 256 rows where the elastic kernel takes 2 x 44: three frames 121 vs 147 us per launch, four 155 vs 186
        // (two frames: 539 workgroups = one full round of 512 and a 27-workgroup third of 43 us: 100 vs 89 us, hence the threshold).
        // The count lives on the device, so a plugin whose capacity spans both regimes launches BOTH kernels and the one whose regime it
        // is not returns at once (an empty launch: ~2 us).  DSVT_MLP_VARIANT forces one: 1 = <2,4> + small overflow workgroups,
        // 2 = <1,8>, 3 = <1,10> elastic, 4 = <2,8>.
        static int forced = -1;
        if (forced < 0) forced = ablateEnv("DSVT_MLP_VARIANT", 0);
        const int thr = 5 * ncu * MROWS / 2;
        if (!forced && frames_ > 0) return launchVariant(b, frames_ >= 3 ? 1 : 3, stream);       // the caller said which regime its launches are in (either kernel is correct for any count)
        if (!forced && max_rows_ > thr) {
            b.thr = thr;
            b.sel = 2; int rc = launchVariant(b, 3, stream); if (rc) return rc;
            b.sel = 1; return launchVariant(b, 1, stream);
        }
        return launchVariant(b, forced ? forced : 3, stream);
    }
    int launchVariant(MlpStreamArgs b, int variant, hipStream_t stream) const {
        const int ncu = b.ncu, over = max_rows_ - ncu * MROWS;
        const int srows = 16 * (variant == 2 ? 1 : 2) * MLP_SW;                 // rows of a small workgroup
        (void)over;
        const int gfull = cdiv(max_rows_, MROWS);                               // (covers the elastic variant too: >= 128 rows per workgroup)
        const dim3 grid(gfull + cdiv(MLP_SMALL_MAX, srows));                    // full workgroups of the whole rounds + the small ones of the remainder (the rest exit at once)
        static unsigned long long* tr = nullptr; static int tron = -1;         // tools/trace_mlp.py
        if (tron < 0) { tron = ablateEnv("DSVT_MLP_TRACE", 0) ? 1 : 0; if (tron) (void)hipMallocManaged(&tr, 8 * 64 * kMlpTraceBlocks); }
        b.trace = tr;
        static int dbg = -1; if (dbg < 0) dbg = ablateEnv("DSVT_MLP_DBG", 0);
        b.dbg = dbg;
        static int ring = -1;          // DSVT_MLP_RING=6: the six-slot ring (measured: no gain, see the kernel's header)
        if (ring < 0) ring = ablateEnv("DSVT_MLP_RING", 3);
        bool launched = false;
        if constexpr (kAblate) {       // the measured-and-dropped variants: ablation build only (DSVT_MLP_VARIANT / DSVT_MLP_RING)
            launched = true;
            if (variant == 3 && ring == 6) hipLaunchKernelGGL((encoder_mlp_stream_kernel<1, 10, 64, 6>), grid, dim3(640), 0, stream, b);
            else if (variant == 2) hipLaunchKernelGGL((encoder_mlp_stream_kernel<1, 8, 64, 3>), grid, dim3(512), 0, stream, b);
            else if (variant == 4) hipLaunchKernelGGL((encoder_mlp_stream_kernel<2, 8, 64, 3>), dim3(cdiv(max_rows_, 256)), dim3(512), 0, stream, b);   // experiment
            else launched = false;
            // (variant 4, an experiment: ONE ring for 256 rows, eight waves in lockstep: 188 vs 155 us per four-frame launch for two independent <2,4> workgroups per CU.
            // (The same with the upper four waves one stage behind the lower four, on a four-slot ring: 222 us.  What two independent workgroups
            // overlap is not stage against stage -- both kinds are 48 MFMAs per wave -- but one's prologue and epilogue, 40 % of a
            // workgroup's life, against the other's streaming loop: a lag of half a lifetime, which no ring in 160 KB can hold.))
        }
        if (launched) {}
        else if (variant == 3) hipLaunchKernelGGL((encoder_mlp_stream_kernel<1, 10, 64, 3>), grid, dim3(640), 0, stream, b);
        else hipLaunchKernelGGL((encoder_mlp_stream_kernel<2, 4, 64, 3>), grid, dim3(256), 0, stream, b);
        if (tron) {
            (void)hipStreamSynchronize(stream);
            for (int w : {0, 200}) { fprintf(stderr, "[mlp trace wg%d]", w); for (int i = 1; i < 40; ++i) fprintf(stderr, " %lld", (long long)(tr[w * 64 + i] - tr[w * 64])); fprintf(stderr, "\n"); }
        }
        return lastError();
    }
    size_t nFloats() const { return wo_.size() + w1_.size() + w2_.size() + bo_.size() + b1_.size() + b2_.size() + 2 * (size_t)(3 + has_block_ln_) * MC; }
    size_t serializationSize() const override { return 2 * sizeof(int) + sizeof(float) + sizeof(float) * nFloats() + ((frames_ || split_) ? sizeof(int) : 0) + (split_ ? sizeof(int) : 0); }
    void serialize(void* buf) const override {
        char* d = static_cast<char*>(buf);
        wr<int>(d, max_rows_); wr<int>(d, has_block_ln_); wr<float>(d, eps_);
        auto put = [&](const std::vector<float>& v, size_t n) { memcpy(d, v.data(), sizeof(float) * n); d += sizeof(float) * n; };
        put(wo_, wo_.size()); put(w1_, w1_.size()); put(w2_, w2_.size()); put(bo_, bo_.size()); put(b1_, b1_.size()); put(b2_, b2_.size());
        put(lg_, (size_t)(3 + has_block_ln_) * MC); put(lb_, (size_t)(3 + has_block_ln_) * MC);
        if (frames_ || split_) wr<int>(d, frames_);                  // (trailing, only when set: older blobs stay valid)
        if (split_) wr<int>(d, split_);
    }
    Plugin* clone() const override {
        auto* p = new DsvtEncoderMlpPlugin(max_rows_, has_block_ln_, eps_, wo_.data(), w1_.data(), w2_.data(), bo_.data(), b1_.data(), b2_.data(),
                                           lg_.data(), lb_.data(), split_);
        p->frames_ = frames_;
        return p;
    }
};

static Plugin* mlpCreate(const DsvtPluginFieldCollection* fc) {
    int max_rows = fieldInt(fc, "max_rows"), hb = fieldInt(fc, "has_block_norm");
    if (max_rows <= 0 || hb < 0 || hb > 1) return nullptr;
    struct Need { const char* name; int len; } need[] = {
        {"out_proj_weight", MC * MC}, {"linear1_weight", MF * MC}, {"linear2_weight", MC * MF}, {"out_proj_bias", MC},
        {"linear1_bias", MF}, {"linear2_bias", MC}, {"ln_weights", (3 + hb) * MC}, {"ln_bias", (3 + hb) * MC}};
    const float* p[8];
    for (int i = 0; i < 8; ++i) {
        const DsvtPluginField* f = findField(fc, need[i].name);
        if (!f || !f->data || f->length != need[i].len) return nullptr;
        p[i] = static_cast<const float*>(f->data);
    }
    auto* pl = new DsvtEncoderMlpPlugin(max_rows, hb, fieldFloat(fc, "ln_eps", 0.f), p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], fieldInt(fc, "split_precision", 0) != 0);
    const int fr = fieldInt(fc, "frames", 0);
    pl->frames_ = fr > 0 ? fr : 0;
    return pl;
}
static Plugin* mlpDeser(const void* data, size_t len) {
    if (len < 2 * sizeof(int) + sizeof(float)) return nullptr;
    const char* d = static_cast<const char*>(data);
    int max_rows = rd<int>(d), hb = rd<int>(d); float eps = rd<float>(d);
    if (max_rows <= 0 || hb < 0 || hb > 1) return nullptr;
    size_t n = (size_t)MC * MC + 2 * (size_t)MF * MC + MC + MF + MC + 2 * (size_t)(3 + hb) * MC;
    if (len < 2 * sizeof(int) + sizeof(float) + n * sizeof(float)) return nullptr;
    std::vector<float> all(n); memcpy(all.data(), d, n * sizeof(float));
    const float* q = all.data();
    const float* wo = q; q += MC * MC; const float* w1 = q; q += MF * MC; const float* w2 = q; q += MC * MF;
    const float* bo = q; q += MC; const float* b1 = q; q += MF; const float* b2 = q; q += MC;
    const float* lg = q; q += (3 + hb) * MC; const float* lb = q;
    const size_t base = 2 * sizeof(int) + sizeof(float) + n * sizeof(float);
    int fr = 0, sp = 0;
    const int extra = trailingInts(len, base, 2);
    if (extra < 0) return nullptr;
    if (extra >= 1) memcpy(&fr, d + n * sizeof(float), sizeof(int));
    if (extra >= 2) memcpy(&sp, d + n * sizeof(float) + sizeof(int), sizeof(int));
    auto* pl = new DsvtEncoderMlpPlugin(max_rows, hb, eps, wo, w1, w2, bo, b1, b2, lg, lb, sp != 0);
    pl->frames_ = fr > 0 ? fr : 0;
    return pl;
}
static Creator g_mlpCreator{"DsvtEncoderMlpPlugin",
    {{"max_rows", DSVT_FIELD_INT32}, {"has_block_norm", DSVT_FIELD_INT32}, {"ln_eps", DSVT_FIELD_FLOAT32},
     {"out_proj_weight", DSVT_FIELD_FLOAT32}, {"out_proj_bias", DSVT_FIELD_FLOAT32}, {"linear1_weight", DSVT_FIELD_FLOAT32},
     {"linear1_bias", DSVT_FIELD_FLOAT32}, {"linear2_weight", DSVT_FIELD_FLOAT32}, {"linear2_bias", DSVT_FIELD_FLOAT32},
     {"ln_weights", DSVT_FIELD_FLOAT32}, {"ln_bias", DSVT_FIELD_FLOAT32}, {"frames", DSVT_FIELD_INT32}, {"split_precision", DSVT_FIELD_INT32}},
    mlpCreate, mlpDeser, {}, {}};
static Registrar g_mlpReg(&g_mlpCreator);

}  // namespace dsvt
