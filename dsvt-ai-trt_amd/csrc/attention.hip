// attention.hip -- window multi-head attention over sets of 36 voxels, for gfx950.
//
//   MultiHeadAttentionPlugin   drop-in for the ~35 TensorRT layers built by multHeadAttention()
//                              (src/dsvt-ai-trt.cpp:288-458): inputs q,k,v [1,S,36,C] and the
//                              per-head key-padding mask [1,S,H,36] from GetSet output 3, output
//                              [1,S,36,C].  in-proj -> per-head softmax(QK^T + mask)V -> out-proj.
//   DsvtSetAttentionPlugin     fused GetValueByIndex + attention core + MapSetFeature2Voxel
//                              (plugins/src/getValueByIndex.cu:282-355, src/dsvt-ai-trt.cpp:352-417,
//                              plugins/src/mapSetFeature2voxel.cu:258-320) on Q/K/V that were projected
//                              per VOXEL ROW (DsvtLinearPlugin with add_cols): a set's 36 slots only
//                              repeat voxels, so projecting P rows instead of 36*S slots does the
//                              same arithmetic once per voxel.
//
// One 256-thread workgroup = one set x four heads; each wavefront owns one head.
//   * the 36 gathered Q/K/V row slices (4 heads x 24 ch) are staged in LDS with coalesced float4
//     reads (36 x 3 x 384 B); rows padded to 100 floats => conflict-free fragment reads;
//   * S^T = K Q^T on v_mfma_f32_16x16x4_f32, keys as rows: a lane then holds 12 keys of ONE
//     query column, the softmax needs 2 cross-lane steps (xor 16, 32) instead of 6;
//   * the S^T accumulator layout (lane group g <-> keys 4g..4g+3) is exactly the A-operand
//     layout of the PV product, so P never leaves registers;
//   * duplicate (masked) slots are not written back: their rows equal the first occurrence.
// The 36x36 score matrix (71 MB/layer in the reference) never exists in memory.
#include "plugin_base.h"
#include "device_utils.h"
#include "linear.h"
#include <cstdlib>

namespace dsvt {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int AL = 36;        // voxels per set (VOXEL_NUM_SET, include/params.h:70)
constexpr int ADH = 24;       // head dim (192 / 8)
constexpr int AHB = 4;        // heads per workgroup
constexpr int ALD = 100;      // LDS row stride in floats (96 + 4)

struct AttnArgs {
    const void* qkv; int qkv_ld;         // rows x [q(C) | k(C) | v(C)], fp32 or (IO16) fp16
    const uint32_t* inds;                // [S, 36] voxel row of each slot, or nullptr: row = set*36 + slot
    const float* mask; int mask_set_stride, mask_head_stride;
    const uint32_t* set_num; int max_sets;
    void* out; int out_ld;               // fp32 or (IO16) fp16
    int C, H;
    int dbg;                             // timing ablations of the fp16 kernel (wrong results): 1 no gathers, 2 no stores, 4 no arithmetic
    int split;                           // fp32 I/O on the split-precision kernel (hi / lo fp16 operand pairs) instead of v_mfma_f32_16x16x4_f32
};

// IO16: Q/K/V rows arrive as fp16 and the result is written as fp16; the arithmetic in between
// (fp32 LDS image, fp32 MFMA, fp32 softmax) is the same.
template <bool IO16>
__global__ void __launch_bounds__(256)
set_attention_kernel(AttnArgs a)
{
    __shared__ __attribute__((aligned(16))) float sQ[AL * ALD];
    __shared__ __attribute__((aligned(16))) float sK[AL * ALD];
    __shared__ __attribute__((aligned(16))) float sV[AL * ALD];
    __shared__ uint32_t sRow[AL];
    __shared__ float sMask[AHB][AL];

    const int nhb = a.H / AHB;
    const int set = blockIdx.x / nhb, hq = blockIdx.x % nhb;
    uint32_t S = *a.set_num; if (S > (uint32_t)a.max_sets) S = a.max_sets;
    if ((uint32_t)set >= S) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;

    if (tid < AL) sRow[tid] = a.inds ? a.inds[(size_t)set * AL + tid] : (uint32_t)(set * AL + tid);
    if (tid < AHB * AL) {
        int h = tid / AL, k = tid % AL;
        sMask[h][k] = a.mask[(size_t)set * a.mask_set_stride + (size_t)(hq * AHB + h) * a.mask_head_stride + k];
    }
    __syncthreads();
    // ---- stage the 36 gathered rows: 3 segments x 24 float4 each -----------------------------
    if (IO16) {
        typedef _Float16 half8v __attribute__((ext_vector_type(8)));
        for (int i = tid; i < AL * 3 * 12; i += 256) {
            int slot = i / 36, rem = i % 36, seg = rem / 12, c8 = (rem % 12) * 8;
            const _Float16* src = static_cast<const _Float16*>(a.qkv) + (size_t)sRow[slot] * a.qkv_ld + seg * a.C + hq * (AHB * ADH) + c8;
            const half8v v = *reinterpret_cast<const half8v*>(src);
            float* dst = (seg == 0 ? sQ : seg == 1 ? sK : sV) + slot * ALD + c8;
            *reinterpret_cast<float4*>(dst) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
        }
    } else {
        for (int i = tid; i < AL * 3 * 24; i += 256) {
            int slot = i / 72, rem = i % 72, seg = rem / 24, c4 = (rem % 24) * 4;
            const float* src = static_cast<const float*>(a.qkv) + (size_t)sRow[slot] * a.qkv_ld + seg * a.C + hq * (AHB * ADH) + c4;
            float4 v = *reinterpret_cast<const float4*>(src);
            float* dst = (seg == 0 ? sQ : seg == 1 ? sK : sV) + slot * ALD + c4;
            *reinterpret_cast<float4*>(dst) = v;
        }
    }
    __syncthreads();

    const int hoff = wave * ADH;
    // ---- S^T[key][query] = sum_d K[key][d] Q[query][d]; lane group g carries d = 6g..6g+5 ------
    floatx4 sc[3][3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < 3; ++u) sc[t][u] = floatx4{0.f, 0.f, 0.f, 0.f};
    {
        float kf[3][6], qf[3][6];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            int row = 16 * t + r; row = row < AL ? row : AL - 1;          // rows >= 36 are padding: clamp, mask later
            const float2* pk = reinterpret_cast<const float2*>(&sK[row * ALD + hoff + g * 6]);
            const float2* pq = reinterpret_cast<const float2*>(&sQ[row * ALD + hoff + g * 6]);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float2 kk = pk[j], qq = pq[j];
                kf[t][2 * j] = kk.x; kf[t][2 * j + 1] = kk.y; qf[t][2 * j] = qq.x; qf[t][2 * j + 1] = qq.y;
            }
        }
#pragma unroll
        for (int s = 0; s < 6; ++s)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    sc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[t][s], qf[u][s], sc[t][u], 0, 0, 0);
    }
    // ---- softmax over keys for each query column (lane holds keys 16t + 4g + i, query 16u + r) ----
    // logits + mask broadcast over queries (src/dsvt-ai-trt.cpp:412), softmax over the key axis (:414-415)
    float mk[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) { int key = 16 * t + 4 * g + i; mk[t][i] = key < AL ? sMask[wave][key] : 0.f; }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int key = 16 * t + 4 * g + i;
                float v = key < AL ? sc[t][u][i] + mk[t][i] : -INFINITY;
                sc[t][u][i] = v; mx = fmaxf(mx, v);
            }
        mx = rows4Max(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int key = 16 * t + 4 * g + i;
                float e = key < AL ? expf(sc[t][u][i] - mx) : 0.f;
                sc[t][u][i] = e; sum += e;
            }
        sum = rows4Sum(sum);
        float inv = 1.0f / sum;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) sc[t][u][i] *= inv;
    }
    // ---- O[query][d] = sum_key P[query][key] V[key][d]  (:417) --------------------------------
    floatx4 oc[3][2];
#pragma unroll
    for (int u = 0; u < 3; ++u) { oc[u][0] = floatx4{0.f, 0.f, 0.f, 0.f}; oc[u][1] = floatx4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int key = 16 * t + 4 * g + i; key = key < AL ? key : AL - 1;  // P is 0 for padded keys
            float v0 = sV[key * ALD + hoff + r];
            float v1 = (16 + r) < ADH ? sV[key * ALD + hoff + 16 + r] : 0.f;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                oc[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[t][u][i], v0, oc[u][0], 0, 0, 0);
                oc[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[t][u][i], v1, oc[u][1], 0, 0, 0);
            }
        }
    // ---- write back: lane holds queries 16u + 4g + i, channels 16dt + r -------------------------
    const int h = hq * AHB + wave;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int q = 16 * u + 4 * g + i;
            if (q >= AL) continue;
            if (a.inds && sMask[wave][q] < 0.f) continue;      // duplicate slot: the first occurrence writes the identical row
            if (IO16) {
                _Float16* dst = static_cast<_Float16*>(a.out) + (size_t)sRow[q] * a.out_ld + h * ADH;
                dst[r] = (_Float16)oc[u][0][i];
                if (16 + r < ADH) dst[16 + r] = (_Float16)oc[u][1][i];
            } else {
                float* dst = static_cast<float*>(a.out) + (size_t)sRow[q] * a.out_ld + h * ADH;
                dst[r] = oc[u][0][i];
                if (16 + r < ADH) dst[16 + r] = oc[u][1][i];
            }
        }
}


// -------------------------------------------------------------------------------------
// fp16 I/O on v_mfma_f32_16x16x32_f16.  The fp32 kernel above spends 35 % of its cycles in the matrix pipe
// (126 v_mfma_f32_16x16x4_f32 per head at 32 cycles each); with fp16 operands one K = 32 step covers the padded
// head dim (24 -> 32) of Q K^T and 32 keys of P V: 21 MFMAs of 16 cycles per head.
//   * Q, K rows are staged as fp16 (row stride 208 B = 13 x 16 B: conflict-free ds_read_b128 over 16 rows); lane
//     group g = 3 of a fragment is the zero padding d = 24..31.
//   * S^T accumulators (lane group g <-> keys 16t + 4g + i) become the A operand of P V after a conversion to fp16:
//     k-step 0 = tiles t = 0, 1, k-step 1 = tile t = 2 + zeros.  The MFMA sums over k, so V only has to be read with
//     the same (g, j) <-> key map: V is staged TRANSPOSED ([channel][key], 64 keys, pad zeroed) and a B fragment is two
//     8-byte reads (keys 4g..4g+3 of tile t).
//   * scores, softmax and the output accumulation stay fp32.
typedef _Float16 ahalf8 __attribute__((ext_vector_type(8)));
typedef _Float16 ahalf4 __attribute__((ext_vector_type(4)));
constexpr int AQL = 112;      // halfs per staged Q / K row: the 16-byte fragment reads (row r, lane group g < 3) are conflict-free under ds_read_b128's lane groups
                              // {0-3, 12-15, 20-27}, ... for a stride of 112 (enumerated; 104 -- rounds 1-5 -- costs two extra LDS cycles per read, 96 four, 128 twenty-eight)
constexpr int AVL = 40;       // halfs per staged V^T row (36 keys + 4: 80-byte rows, conflict-free 8-byte fragment reads; the reads of
                              // keys >= 40 -- lane groups g >= 2 of the third key tile -- run into the next row and are discarded)

// (waves-per-SIMD hint 6: 73 registers = six workgroups per CU; without it hipcc takes 76 + 36 accumulation registers = four)
__global__ void __launch_bounds__(256, 6)
set_attention_f16_kernel(AttnArgs a)
{
    __shared__ __attribute__((aligned(16))) _Float16 sQ[AL * AQL];
    __shared__ __attribute__((aligned(16))) _Float16 sK[AL * AQL];
    __shared__ __attribute__((aligned(16))) _Float16 sVt[AHB * ADH * AVL + 16];      // 23.4 KB in all: seven workgroups per CU
    __shared__ uint32_t sRow[AL];
    __shared__ float sMask[AHB][AL];

    const int nhb = a.H / AHB;
    const int set = blockIdx.x / nhb, hq = blockIdx.x % nhb;
    uint32_t S = *a.set_num; if (S > (uint32_t)a.max_sets) S = a.max_sets;
    if ((uint32_t)set >= S) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;

    // (the row-index / mask values the epilogue needs are loaded here and written to LDS after the staging loads are in flight)
    uint32_t myRow = (uint32_t)(set * AL + (tid < AL ? tid : 0)); float myMask = 0.f;
    if (a.inds && tid < AL) myRow = a.inds[(size_t)set * AL + tid];
    if (tid < AHB * AL) myMask = a.mask[(size_t)set * a.mask_set_stride + (size_t)(hq * AHB + tid / AL) * a.mask_head_stride + tid % AL];
    // ---- stage the 36 gathered rows: Q, K as rows, V transposed -------------------------------------
    // (every thread reads its slot's row index itself: no LDS round trip + barrier between the index and the row loads; the
    // key columns 36..63 of sVt stay unwritten -- their B fragments are zeroed in registers below)
    // All row indices first, then all rows, then the LDS writes: left as loops hipcc keeps "load index, wait, load row, wait, write"
    // per item -- ten dependent memory round trips per thread instead of two.
    // Q, K: item = (slot, Q | K, 8-channel chunk): consecutive lanes write consecutive 16-byte pieces of a row.
    // V transposed: item = (8-channel chunk, slot) with the SLOT fastest, so the eight 2-byte writes of a wave instruction land on
    // consecutive keys of one channel row (the chunk-fastest order hit two LDS banks with twelve lanes: SQ_LDS_BANK_CONFLICT was
    // 74 % of the kernel's LDS cycles)
    constexpr int NQK = (AL * 2 * 12 + 255) / 256, NV = (12 * AL + 255) / 256, NIT = NQK + NV;
    int slotOf[NIT], segOf[NIT], c8Of[NIT]; bool live[NIT];
#pragma unroll
    for (int k = 0; k < NQK; ++k) {
        const int i = tid + 256 * k; live[k] = i < AL * 2 * 12;
        const int ii = live[k] ? i : 0, rem = ii % 24;
        slotOf[k] = ii / 24; segOf[k] = rem / 12; c8Of[k] = (rem % 12) * 8;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = tid + 256 * k; live[NQK + k] = i < 12 * AL;
        const int ii = live[NQK + k] ? i : 0;
        slotOf[NQK + k] = ii % AL; segOf[NQK + k] = 2; c8Of[NQK + k] = (ii / AL) * 8;
    }
    uint32_t rowOf[NIT];
    if (a.inds) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) rowOf[k] = a.inds[(size_t)set * AL + slotOf[k]];
    } else {
#pragma unroll
        for (int k = 0; k < NIT; ++k) rowOf[k] = (uint32_t)(set * AL + slotOf[k]);
    }
    ahalf8 val[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k)
        val[k] = (a.dbg & 1) ? ahalf8{0, 0, 0, 0, 0, 0, 0, 0}
                             : *reinterpret_cast<const ahalf8*>(static_cast<const _Float16*>(a.qkv) + (size_t)rowOf[k] * a.qkv_ld + segOf[k] * a.C + hq * (AHB * ADH) + c8Of[k]);
    if (tid < AL) sRow[tid] = myRow;
    if (tid < AHB * AL) sMask[tid / AL][tid % AL] = myMask;
#pragma unroll
    for (int k = 0; k < NQK; ++k)
        if (live[k]) *reinterpret_cast<ahalf8*>(&(segOf[k] == 0 ? sQ : sK)[slotOf[k] * AQL + c8Of[k]]) = val[k];
#pragma unroll
    for (int k = NQK; k < NIT; ++k)
        if (live[k]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) sVt[(c8Of[k] + j) * AVL + slotOf[k]] = val[k][j];
        }
    __syncthreads();
    if (a.dbg & 4) return;

    const int hoff = wave * ADH;
    const ahalf8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // ---- S^T[key][query] = sum_d K[key][d] Q[query][d] ----------------------------------------------
    floatx4 sc[3][3];
    {
        ahalf8 kf[3], qf[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            int row = 16 * t + r; row = row < AL ? row : AL - 1;          // rows >= 36 are padding: clamp, mask later
            kf[t] = g < 3 ? *reinterpret_cast<const ahalf8*>(&sK[row * AQL + hoff + g * 8]) : zero8;
            qf[t] = g < 3 ? *reinterpret_cast<const ahalf8*>(&sQ[row * AQL + hoff + g * 8]) : zero8;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int u = 0; u < 3; ++u)
                sc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t], qf[u], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    // ---- softmax over keys for each query column (lane holds keys 16t + 4g + i, query 16u + r) ----
    // The ablations (DSVT_ATTN_DBG, tools/one_attn.py) put 21 of the launch's 33 us in this arithmetic, not in the gathers: seven
    // workgroups per CU share four SIMDs and expf alone expanded to 13 instructions x 36 scores per lane.  Here a score costs one add
    // (mask; -inf for the padded keys 36..47, so no select), one fma and one v_exp_f32 (exp2 of (s - max) * log2 e: arguments <= 0),
    // and the row sum one v_rcp_f32 -- both within 1 ulp, far below the fp16 rounding of P that follows.
    constexpr float kLog2e = 1.4426950408889634f;
    float mk[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) { int key = 16 * t + 4 * g + i; mk[t][i] = key < AL ? sMask[wave][key] : -INFINITY; }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float v = sc[t][u][i] + mk[t][i]; sc[t][u][i] = v; mx = fmaxf(mx, v); }
        mx = rows4Max(mx);
        const float nm = -mx * kLog2e;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[t][u][i], kLog2e, nm)); sc[t][u][i] = e; sum += e; }
        sum = rows4Sum(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) sc[t][u][i] *= inv;
    }
    // ---- O[query][d] = sum_key P[query][key] V[key][d] -------------------------------------------------
    ahalf8 vb[2][2];                     // [channel tile][k-step]
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const bool dv = 16 * dt + r < ADH;
        const _Float16* pv = &sVt[(hoff + (dv ? 16 * dt + r : 0)) * AVL + 4 * g];
        const ahalf4 v0 = *reinterpret_cast<const ahalf4*>(pv), v1 = *reinterpret_cast<const ahalf4*>(pv + 16), v2 = *reinterpret_cast<const ahalf4*>(pv + 32);
        ahalf8 b0 = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        ahalf8 b1 = {v2[0], v2[1], v2[2], v2[3], 0, 0, 0, 0};
        vb[dt][0] = dv ? b0 : zero8; vb[dt][1] = (dv && g == 0) ? b1 : zero8;      // keys 32 + 4g + i: only g = 0 is real (and staged)
    }
    floatx4 oc[3][2];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        ahalf8 p0, p1;
#pragma unroll
        for (int i = 0; i < 4; ++i) { p0[i] = (_Float16)sc[0][u][i]; p0[4 + i] = (_Float16)sc[1][u][i]; p1[i] = (_Float16)sc[2][u][i]; p1[4 + i] = (_Float16)0.f; }
#pragma unroll
        // O^T = V^T P^T (V^T as the A operand): the lane of query r then holds FOUR consecutive channels 16 dt + 4 g + i, an
        // 8-byte store, instead of one channel of four queries (24 two-byte stores per lane)
        for (int dt = 0; dt < 2; ++dt) {
            oc[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[dt][0], p0, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            oc[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[dt][1], p1, oc[u][dt], 0, 0, 0);
        }
    }
    // ---- write back: lane holds query 16u + r, channels 16dt + 4g + i -------------------------
    const int h = hq * AHB + wave;
    if (a.dbg & 2) return;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int q = 16 * u + r;
        if (q >= AL) continue;
        if (a.inds && sMask[wave][q] < 0.f) continue;      // duplicate slot: the first occurrence writes the identical row
        _Float16* dst = static_cast<_Float16*>(a.out) + (size_t)sRow[q] * a.out_ld + h * ADH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            if (16 * dt + 4 * g >= ADH) continue;
            ahalf4 o4 = {(_Float16)oc[u][dt][0], (_Float16)oc[u][dt][1], (_Float16)oc[u][dt][2], (_Float16)oc[u][dt][3]};
            *reinterpret_cast<ahalf4*>(dst + 16 * dt) = o4;
        }
    }
}

// -------------------------------------------------------------------------------------
// Split precision (round 3): fp32 Q / K / V rows in, fp32 rows out, the two products of the attention core at fp32 GRADE on the fp16
// matrix cores.  The fp32 kernel at the top of this file spends its time in 126 v_mfma_f32_16x16x4_f32 per head (32 cycles each) and
// in fp32 LDS images; here every operand is the pair hi = fp16(v), lo = fp16(v - hi) and a product is three 16x16x32 fp16 MFMAs:
//     S^T = K_hi Q_hi^T + K_lo Q_hi^T + K_hi Q_lo^T            (27 MFMAs per head)
//     O^T = V_hi^T P_hi^T + V_lo^T P_hi^T + V_hi^T P_lo^T      (36 MFMAs per head; P split in registers after the fp32 softmax)
// Layouts are set_attention_f16_kernel's (Q / K rows of 112 halfs, V transposed in 40-half rows), twice: 48.4 KB of LDS, three
// workgroups per CU.  Softmax in fp32 with v_exp_f32 / v_rcp_f32 (1 ulp each: fp32 grade).
__global__ void __launch_bounds__(256, 5)
set_attention_split_kernel(AttnArgs a)
{
    __shared__ __attribute__((aligned(16))) _Float16 sQ[2][AL * AQL];                 // [hi | lo]
    __shared__ __attribute__((aligned(16))) _Float16 sK[2][AL * AQL];
    // V^T lives where Q was (round 6, late): the Q / K fragments are in registers after one read, so V^T is written into Q's rows behind a barrier while the S^T products
    // run -- 32 KB of LDS instead of 48 KB, FIVE workgroups per CU instead of three (the kernel waits for its gathers: more sets in flight per CU;
    // four: 157 / 121 us per four-frame launch of the two window configurations against 173 / 130, tools/ab_attn_split.py)
    // Measured and dropped the same day: persistent workgroups (grid = workgroups per CU x CUs, the next item's slot indices and masks prefetched into second copies of the
    // LDS tables, so that only a workgroup's first item waits index -> rows): the loop keeps ~12 more registers alive -- at five workgroups per CU 28 spilled registers and
    // 233 / 181 us, at four (126 registers) 161 / 120 against 156.5 / 118 for this kernel: the other workgroups of the CU already cover the index wait.
    constexpr int SVT = AHB * ADH * AVL + 16;
    static_assert(2 * SVT <= 2 * AL * AQL, "V^T fits in Q's rows");
    _Float16 (*sVt)[SVT] = reinterpret_cast<_Float16 (*)[SVT]>(&sQ[0][0]);
    // the slots' row indices and the heads' key masks live in the 32 padding bytes of K's rows (columns 96 .. 111 of a 112-half row: no fragment read touches them):
    // 32,256 B of LDS per workgroup = FIVE workgroups per CU (160 KB)
    static_assert(AQL - AHB * ADH >= 2 * AHB && AQL % 2 == 0, "padding columns");
    auto sRow = [&](int q) -> uint32_t& { return *reinterpret_cast<uint32_t*>(&sK[1][q * AQL + AHB * ADH]); };
    auto sMask = [&](int w, int q) -> float& { return *reinterpret_cast<float*>(&sK[0][q * AQL + AHB * ADH + 2 * w]); };

    const int nhb = a.H / AHB;
    const int set = blockIdx.x / nhb, hq = blockIdx.x % nhb;
    uint32_t S = *a.set_num; if (S > (uint32_t)a.max_sets) S = a.max_sets;
    if ((uint32_t)set >= S) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;

    uint32_t myRow = (uint32_t)(set * AL + (tid < AL ? tid : 0)); float myMask = 0.f;
    if (a.inds && tid < AL) myRow = a.inds[(size_t)set * AL + tid];
    if (tid < AHB * AL) myMask = a.mask[(size_t)set * a.mask_set_stride + (size_t)(hq * AHB + tid / AL) * a.mask_head_stride + tid % AL];
    // ---- stage the 36 gathered fp32 rows as hi / lo fp16: Q, K as rows, V transposed (see set_attention_f16_kernel for the order of the
    // loads: all indices, all rows, then the LDS writes).  Item = (slot, Q | K, 4-channel chunk) / (4-channel chunk, slot) for V.
    // V items are (4-channel chunk, PAIR of slots): the two slots' values of a channel are one 4-byte LDS store.  (Until round 6 a V item was one slot and stored eight
    // single halfs: two lanes of every pair wrote the two halves of one dword -- 5.1 M bank-conflict cycles per four-frame launch, a third of the kernel's LDS-active cycles.)
    // Item -> thread mapping (round 6, late): a thread keeps ONE (Q | K, 4-channel chunk) and walks the slots in steps of five (240 threads x 8 steps cover 36 slots x 48
    // chunks), and ONE slot pair of V and walks the 4-channel chunks in steps of fourteen (252 threads x 2 steps cover 18 pairs x 24 chunks): the division / modulo decode
    // of an item happens once per thread instead of once per item (eleven items: the kernel issues 20 VALU instructions per MFMA and is bound by them).
    constexpr int QKT = 2 * 24, QKS = 256 / QKT, NQK = (AL + QKS - 1) / QKS;          // 48 chunks per slot, 5 slots per step, 8 steps
    constexpr int VPT = AL / 2, VCS = 256 / VPT, NVP = (24 + VCS - 1) / VCS, NIT = NQK + 2 * NVP;      // 18 pairs, 14 chunks per step, 2 steps
    static_assert(AL % 2 == 0, "slot pairs");
    int slotOf[NIT], segOf[NIT], c4Of[NIT]; bool live[NIT];
    {
        const int u = tid % QKT, s0 = tid / QKT;
#pragma unroll
        for (int k = 0; k < NQK; ++k) {
            const int slot = s0 + QKS * k;
            live[k] = tid < QKT * QKS && slot < AL;
            slotOf[k] = live[k] ? slot : 0; segOf[k] = u / 24; c4Of[k] = (u % 24) * 4;
        }
        const int sp = tid % VPT, c0 = tid / VPT;
#pragma unroll
        for (int k = 0; k < NVP; ++k) {
            const int c = c0 + VCS * k; const bool lv = tid < VPT * VCS && c < 24;
#pragma unroll
            for (int e = 0; e < 2; ++e) {                                    // items NQK + 2k, NQK + 2k + 1: the even and the odd slot of the pair
                live[NQK + 2 * k + e] = lv; slotOf[NQK + 2 * k + e] = 2 * sp + e; segOf[NQK + 2 * k + e] = 2; c4Of[NQK + 2 * k + e] = (lv ? c : 0) * 4;
            }
        }
    }
    uint32_t rowOf[NIT];
    if (a.inds) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) rowOf[k] = a.inds[(size_t)set * AL + slotOf[k]];
    } else {
#pragma unroll
        for (int k = 0; k < NIT; ++k) rowOf[k] = (uint32_t)(set * AL + slotOf[k]);
    }
    floatx4 val[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k)
        val[k] = *reinterpret_cast<const floatx4*>(static_cast<const float*>(a.qkv) + (size_t)rowOf[k] * a.qkv_ld + segOf[k] * a.C + hq * (AHB * ADH) + c4Of[k]);
    if (tid < AL) sRow(tid) = myRow;
    if (tid < AHB * AL) sMask(tid / AL, tid % AL) = myMask;
    auto split4 = [](const floatx4& v, _Float16 (&h)[4], _Float16 (&l)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = (_Float16)__builtin_fminf(__builtin_fmaxf(v[j], -65504.f), 65504.f);
            l[j] = (_Float16)__builtin_fminf(__builtin_fmaxf(v[j] - (float)h[j], -65504.f), 65504.f);
        }
    };
#pragma unroll
    for (int k = 0; k < NQK; ++k) {
        if (!live[k]) continue;
        _Float16 h[4], l[4];
        split4(val[k], h, l);
        _Float16* dh = (segOf[k] == 0 ? sQ[0] : sK[0]) + slotOf[k] * AQL + c4Of[k];
        _Float16* dl = (segOf[k] == 0 ? sQ[1] : sK[1]) + slotOf[k] * AQL + c4Of[k];
        *reinterpret_cast<ahalf4*>(dh) = ahalf4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<ahalf4*>(dl) = ahalf4{l[0], l[1], l[2], l[3]};
    }
    __syncthreads();

    const int hoff = wave * ADH;
    const ahalf8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // ---- S^T[key][query] = sum_d K[key][d] Q[query][d] ----------------------------------------------
    floatx4 sc[3][3];
    {
        ahalf8 kh[3], kl[3], qh[3], ql[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            int row = 16 * t + r; row = row < AL ? row : AL - 1;          // rows >= 36 are padding: clamp, mask later
            const int o = row * AQL + hoff + g * 8;
            kh[t] = g < 3 ? *reinterpret_cast<const ahalf8*>(&sK[0][o]) : zero8; kl[t] = g < 3 ? *reinterpret_cast<const ahalf8*>(&sK[1][o]) : zero8;
            qh[t] = g < 3 ? *reinterpret_cast<const ahalf8*>(&sQ[0][o]) : zero8; ql[t] = g < 3 ? *reinterpret_cast<const ahalf8*>(&sQ[1][o]) : zero8;
        }
        // every wave has its Q / K fragments: Q's rows become V^T
        __syncthreads();
#pragma unroll
    for (int k = 0; k < NVP; ++k) {
        const int e0 = NQK + 2 * k;
        if (!live[e0]) continue;
        _Float16 h0[4], l0[4], h1[4], l1[4];
        split4(val[e0], h0, l0); split4(val[e0 + 1], h1, l1);
        typedef _Float16 ahalf2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                    // (slotOf[e0] is even and AVL is even: 4-byte aligned)
            *reinterpret_cast<ahalf2*>(&sVt[0][(c4Of[e0] + j) * AVL + slotOf[e0]]) = ahalf2{h0[j], h1[j]};
            *reinterpret_cast<ahalf2*>(&sVt[1][(c4Of[e0] + j) * AVL + slotOf[e0]]) = ahalf2{l0[j], l1[j]};
        }
    }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                floatx4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[t], qh[u], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[t], ql[u], c, 0, 0, 0);
                sc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[t], qh[u], c, 0, 0, 0);
            }
    }
    // ---- softmax over keys for each query column (lane holds keys 16t + 4g + i, query 16u + r) ----
    constexpr float kLog2e = 1.4426950408889634f;
    float mk[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) { int key = 16 * t + 4 * g + i; mk[t][i] = key < AL ? sMask(wave, key) : -INFINITY; }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float v = sc[t][u][i] + mk[t][i]; sc[t][u][i] = v; mx = fmaxf(mx, v); }
        mx = rows4Max(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float e = __builtin_amdgcn_exp2f((sc[t][u][i] - mx) * kLog2e); sc[t][u][i] = e; sum += e; }
        sum = rows4Sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) sc[t][u][i] *= inv;
    }
#ifdef DSVT_ABLATE
    if (a.dbg == 99 || a.dbg == 98) {       // debugging: the probabilities (99) of keys 0..23 instead of the output channels
        const int h_ = hq * AHB + wave;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int q = 16 * u + r;
            if (q >= AL || (a.inds && sMask(wave, q) < 0.f)) continue;
            float* dst = static_cast<float*>(a.out) + (size_t)sRow(q) * a.out_ld + h_ * ADH;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int key = 16 * t + 4 * g + i; if (key < ADH) dst[key] = sc[t][u][i]; }
        }
        return;
    }
    if (a.dbg == 97 || a.dbg == 96) {       // debugging: the hi (97) / lo (96) parts of the probabilities of keys 0..23, as the PV product sees them
        const int h_ = hq * AHB + wave;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int q = 16 * u + r;
            if (q >= AL || (a.inds && sMask(wave, q) < 0.f)) continue;
            float* dst = static_cast<float*>(a.out) + (size_t)sRow(q) * a.out_ld + h_ * ADH;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int key = 16 * t + 4 * g + i;
                    const float p = sc[t][u][i]; const _Float16 hh = (_Float16)p; float d = p - (float)hh; asm volatile("" : "+v"(d)); const _Float16 ll = (_Float16)d;
                    if (key < ADH) dst[key] = a.dbg == 97 ? (float)hh : (float)ll;
                }
        }
        return;
    }
#endif
    // ---- O[query][d] = sum_key P[query][key] V[key][d] -------------------------------------------------
    __syncthreads();                     // V^T is staged
    ahalf8 vb[2][2][2];                  // [hi | lo][channel tile][k-step]
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const bool dv = 16 * dt + r < ADH;
            const _Float16* pv = &sVt[pl][(hoff + (dv ? 16 * dt + r : 0)) * AVL + 4 * g];
            const ahalf4 v0 = *reinterpret_cast<const ahalf4*>(pv), v1 = *reinterpret_cast<const ahalf4*>(pv + 16), v2 = *reinterpret_cast<const ahalf4*>(pv + 32);
            ahalf8 b0 = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            ahalf8 b1 = {v2[0], v2[1], v2[2], v2[3], 0, 0, 0, 0};
            vb[pl][dt][0] = dv ? b0 : zero8; vb[pl][dt][1] = (dv && g == 0) ? b1 : zero8;      // keys 32 + 4g + i: only g = 0 is real (and staged)
        }
    floatx4 oc[3][2];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        _Float16 ph[12], pl_[12];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {                                  // 0 <= p <= 1
                // p as a materialised fp32 value FIRST.  hipcc (-ffp-contract=fast, the HIP default) otherwise fuses "e * inv -> half" into
                // v_fma_mixlo_f16 for the RESIDUAL (one rounding of the exact product) while the hi FRAGMENT is v_cvt_pk_f16_f32 of the fp32-rounded
                // product (two roundings): in the rare double-rounding cases the two hi values differ by one fp16 ulp and the pair (hi, lo) is off by
                // 2^-14 of probability mass -- found as 15 outlier (row, head) pairs of 44,000 at 1e-4 (tools/dbg_attn_split*.py; the other split
                // helpers clamp before they convert, which materialises the value)
                float p = sc[t][u][i];
                asm volatile("" : "+v"(p));
                ph[4 * t + i] = (_Float16)p;
                pl_[4 * t + i] = (_Float16)(p - (float)ph[4 * t + i]);
            }
        const ahalf8 p0h = {ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7]}, p1h = {ph[8], ph[9], ph[10], ph[11], 0, 0, 0, 0};
        const ahalf8 p0l = {pl_[0], pl_[1], pl_[2], pl_[3], pl_[4], pl_[5], pl_[6], pl_[7]}, p1l = {pl_[8], pl_[9], pl_[10], pl_[11], 0, 0, 0, 0};
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            floatx4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[1][dt][0], p0h, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[1][dt][1], p1h, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[0][dt][0], p0l, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[0][dt][1], p1l, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[0][dt][0], p0h, c, 0, 0, 0);
            oc[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[0][dt][1], p1h, c, 0, 0, 0);
        }
    }
    // ---- write back: lane holds query 16u + r, channels 16dt + 4g + i -------------------------
    const int h = hq * AHB + wave;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int q = 16 * u + r;
        if (q >= AL) continue;
        if (a.inds && sMask(wave, q) < 0.f) continue;      // duplicate slot: the first occurrence writes the identical row
        float* dst = static_cast<float*>(a.out) + (size_t)sRow(q) * a.out_ld + h * ADH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            if (16 * dt + 4 * g >= ADH) continue;
            *reinterpret_cast<floatx4*>(dst + 16 * dt) = oc[u][dt];
        }
    }
}

static int launchAttention(const AttnArgs& a_, bool io16, hipStream_t stream) {
    static int dbg = -1;
    if (dbg < 0) dbg = ablateEnv("DSVT_ATTN_DBG", 0);
    AttnArgs a = a_; a.dbg = dbg;
    dim3 grid((unsigned)(a.max_sets * (a.H / AHB))), block(256);
    static int f16mma = -1;        // DSVT_ATTN_F32MMA=1: fp16 I/O on the fp32 matrix instructions (the previous kernel)
    if (f16mma < 0) f16mma = ablateEnv("DSVT_ATTN_F32MMA", 0) ? 0 : 1;
    if (io16 && f16mma) hipLaunchKernelGGL(set_attention_f16_kernel, grid, block, 0, stream, a);
    else if (io16) hipLaunchKernelGGL(set_attention_kernel<true>, grid, block, 0, stream, a);
    else if (a.split) hipLaunchKernelGGL(set_attention_split_kernel, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(set_attention_kernel<false>, grid, block, 0, stream, a);
    return lastError();
}

static bool f32Lin(const DsvtPluginTensorDesc& t) { return t.type == DSVT_FLOAT && t.format == DSVT_FORMAT_LINEAR; }
static bool i32Lin(const DsvtPluginTensorDesc& t) { return t.type == DSVT_INT32 && t.format == DSVT_FORMAT_LINEAR; }

// =====================================================================================
// MultiHeadAttentionPlugin
// =====================================================================================
class MultiHeadAttentionPlugin : public Plugin {
public:
    int max_win_num_, L_, C_, H_;
    std::vector<float> wi_, bi_, wo_, bo_;         // as in the .wts file (un-scaled)
    float *wi_dev_ = nullptr, *bi_dev_ = nullptr, *wo_dev_ = nullptr, *bo_dev_ = nullptr;
    bool ok_ = false;
    MultiHeadAttentionPlugin(int mw, int L, int C, int H, const float* wi, const float* bi, const float* wo, const float* bo)
        : max_win_num_(mw), L_(L), C_(C), H_(H), wi_(wi, wi + 3 * (size_t)C * C), bi_(bi, bi + 3 * C),
          wo_(wo, wo + (size_t)C * C), bo_(bo, bo + C) {
        // Q is divided by sqrt(head_dim) after the bias (src/dsvt-ai-trt.cpp:386-405); the constant
        // is folded into the Q rows of in_proj (rows 0..C-1, include/helper.h:369-433)
        std::vector<float> wis(wi_), bis(bi_);
        const float scale = sqrtf((float)(C / H));
        for (size_t i = 0; i < (size_t)C * C; ++i) wis[i] = wis[i] / scale;
        for (int i = 0; i < C; ++i) bis[i] = bis[i] / scale;
        auto up = [](const std::vector<float>& h, float** d) {
            if (dsvtMalloc(d, sizeof(float) * h.size()) != hipSuccess) return false;
            return hipMemcpy(*d, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice) == hipSuccess;
        };
        ok_ = up(wis, &wi_dev_) && up(bis, &bi_dev_) && up(wo_, &wo_dev_) && up(bo_, &bo_dev_);
    }
    ~MultiHeadAttentionPlugin() override { for (float* p : {wi_dev_, bi_dev_, wo_dev_, bo_dev_}) if (p) (void)dsvtFree(p); }
    const char* type() const override { return "MultiHeadAttentionPlugin"; }
    int nbOutputs() const override { return 1; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i != 0) return -1;
        *out = dims4(in[0].d[0], max_win_num_, L_, C_); return 0;                  // src/dsvt-ai-trt.cpp:455
    }
    int outputType(int, const int32_t* t, int) const override { return t[0]; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {
        return pos == 4 ? i32Lin(io[pos]) : pos >= 0 && pos <= 5 && f32Lin(io[pos]);
    }
    size_t rows() const { return (size_t)max_win_num_ * L_; }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override {
        return alignUp(sizeof(float) * rows() * 3 * C_) + alignUp(sizeof(float) * rows() * C_);
    }
    int enqueue(const DsvtPluginTensorDesc*, const DsvtPluginTensorDesc*, const void* const* in, void* const* out,
                void* workspace, hipStream_t stream) override {
        if (!ok_) return static_cast<int>(hipErrorOutOfMemory);
        WsCarver ws(workspace);
        float* qkv = ws.take<float>(rows() * 3 * C_);
        float* att = ws.take<float>(rows() * C_);
        const uint32_t* S = static_cast<const uint32_t*>(in[4]);
        for (int p = 0; p < 3; ++p) {                                              // :328-330 three FullyConnected
            LinearArgs la{};
            la.A = static_cast<const float*>(in[p]); la.W = wi_dev_ + (size_t)p * C_ * C_; la.bias = bi_dev_ + p * C_;
            la.out = qkv + p * C_; la.out_ld = 3 * C_; la.count = S; la.row_mult = L_; la.max_rows = (int)rows();
            la.K = C_; la.N = C_;
            int rc = launchLinearF32(la, stream); if (rc) return rc;
        }
        AttnArgs aa{};
        aa.qkv = qkv; aa.qkv_ld = 3 * C_; aa.inds = nullptr;
        aa.mask = static_cast<const float*>(in[3]); aa.mask_set_stride = H_ * L_; aa.mask_head_stride = L_;   // [S,H,36]
        aa.set_num = S; aa.max_sets = max_win_num_; aa.out = att; aa.out_ld = C_; aa.C = C_; aa.H = H_;
        int rc = launchAttention(aa, false, stream); if (rc) return rc;
        if (zeroFill) DSVT_CHECK(hipMemsetAsync(out[0], 0, sizeof(float) * rows() * C_, stream));
        LinearArgs lo{};                                                           // :448 out_proj
        lo.A = att; lo.W = wo_dev_; lo.bias = bo_dev_; lo.out = static_cast<float*>(out[0]); lo.out_ld = C_;
        lo.count = S; lo.row_mult = L_; lo.max_rows = (int)rows(); lo.K = C_; lo.N = C_;
        return launchLinearF32(lo, stream);
    }
    size_t serializationSize() const override { return 4 * sizeof(int) + sizeof(float) * (wi_.size() + bi_.size() + wo_.size() + bo_.size()); }
    void serialize(void* buf) const override {
        char* d = static_cast<char*>(buf);
        wr<int>(d, max_win_num_); wr<int>(d, L_); wr<int>(d, C_); wr<int>(d, H_);
        for (const std::vector<float>* v : {&wi_, &bi_, &wo_, &bo_}) { memcpy(d, v->data(), sizeof(float) * v->size()); d += sizeof(float) * v->size(); }
    }
    Plugin* clone() const override { return new MultiHeadAttentionPlugin(max_win_num_, L_, C_, H_, wi_.data(), bi_.data(), wo_.data(), bo_.data()); }
};
static bool attnShapeOk(int mw, int L, int C, int H) {
    return mw > 0 && L == AL && H > 0 && H % AHB == 0 && C == H * ADH;            // 36-voxel sets, 24-channel heads
}
static Plugin* mhaCreate(const DsvtPluginFieldCollection* fc) {
    int mw = fieldInt(fc, "max_win_num"), L = fieldInt(fc, "voxel_num_set"), C = fieldInt(fc, "channel_num"), H = fieldInt(fc, "num_heads");
    const DsvtPluginField* wi = findField(fc, "in_proj_weight"); const DsvtPluginField* bi = findField(fc, "in_proj_bias");
    const DsvtPluginField* wo = findField(fc, "out_proj_weight"); const DsvtPluginField* bo = findField(fc, "out_proj_bias");
    if (!attnShapeOk(mw, L, C, H) || !wi || !bi || !wo || !bo) return nullptr;
    if (wi->length != 3 * C * C || bi->length != 3 * C || wo->length != C * C || bo->length != C) return nullptr;
    return new MultiHeadAttentionPlugin(mw, L, C, H, static_cast<const float*>(wi->data), static_cast<const float*>(bi->data),
                                        static_cast<const float*>(wo->data), static_cast<const float*>(bo->data));
}
static Plugin* mhaDeser(const void* data, size_t len) {
    if (len < 4 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    int mw = rd<int>(d), L = rd<int>(d), C = rd<int>(d), H = rd<int>(d);
    if (!attnShapeOk(mw, L, C, H)) return nullptr;
    size_t need = 4 * (size_t)C * C + 4 * (size_t)C;
    if (len < 4 * sizeof(int) + need * sizeof(float)) return nullptr;
    std::vector<float> all(need); memcpy(all.data(), d, need * sizeof(float));
    const float* wi = all.data(); const float* bi = wi + 3 * (size_t)C * C; const float* wo = bi + 3 * C; const float* bo = wo + (size_t)C * C;
    return new MultiHeadAttentionPlugin(mw, L, C, H, wi, bi, wo, bo);
}
static Creator g_mhaCreator{"MultiHeadAttentionPlugin",
    {{"max_win_num", DSVT_FIELD_INT32}, {"voxel_num_set", DSVT_FIELD_INT32}, {"channel_num", DSVT_FIELD_INT32},
     {"num_heads", DSVT_FIELD_INT32}, {"in_proj_weight", DSVT_FIELD_FLOAT32}, {"in_proj_bias", DSVT_FIELD_FLOAT32},
     {"out_proj_weight", DSVT_FIELD_FLOAT32}, {"out_proj_bias", DSVT_FIELD_FLOAT32}},
    mhaCreate, mhaDeser, {}, {}};
static Registrar g_mhaReg(&g_mhaCreator);

// =====================================================================================
// DsvtSetAttentionPlugin: inputs qkv[1,P,3C], inds[1,2,S,36], mask[1,2,S,36], S[1] -> out[1,P,C]
// =====================================================================================
class DsvtSetAttentionPlugin : public Plugin {
public:
    int max_win_num_, L_, C_, H_, axis_id_, max_pillars_num_, io_half_;
    DsvtSetAttentionPlugin(int mw, int L, int C, int H, int axis, int mp, int io_half)
        : max_win_num_(mw), L_(L), C_(C), H_(H), axis_id_(axis), max_pillars_num_(mp), io_half_(io_half) {}
    const char* type() const override { return "DsvtSetAttentionPlugin"; }
    int nbOutputs() const override { return 1; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i != 0) return -1;
        *out = dims3(in[0].d[0], max_pillars_num_, C_); return 0;
    }
    int outputType(int, const int32_t* t, int) const override { return t[0]; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {
        if (pos == 1 || pos == 3) return i32Lin(io[pos]);
        if (pos == 2) return f32Lin(io[pos]);
        return (pos == 0 || pos == 4) && io[pos].format == DSVT_FORMAT_LINEAR && io[pos].type == (io_half_ == 1 ? DSVT_HALF : DSVT_FLOAT);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc*, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        if (zeroFill) DSVT_CHECK(hipMemsetAsync(out[0], 0, (io_half_ == 1 ? 2 : 4) * (size_t)max_pillars_num_ * C_, stream));   // mapSetFeature2voxel.cu:314
        AttnArgs aa{};
        aa.qkv = in[0]; aa.qkv_ld = 3 * C_;
        aa.inds = static_cast<const uint32_t*>(in[1]) + (size_t)axis_id_ * max_win_num_ * L_;    // getValueByIndex.cu:292
        // the reference feeds the axis-0 mask to both layers of a block (src/dsvt-ai-trt.cpp:658,708);
        // the two axes mask the same slots, so axis 0 is used here too
        aa.mask = static_cast<const float*>(in[2]); aa.mask_set_stride = L_; aa.mask_head_stride = 0;
        aa.set_num = static_cast<const uint32_t*>(in[3]); aa.max_sets = max_win_num_;
        aa.out = out[0]; aa.out_ld = C_; aa.C = C_; aa.H = H_; aa.split = io_half_ == 2;
        return launchAttention(aa, io_half_ == 1, stream);
    }
    size_t serializationSize() const override { return 7 * sizeof(int); }
    void serialize(void* buf) const override {
        char* d = static_cast<char*>(buf);
        wr<int>(d, max_win_num_); wr<int>(d, L_); wr<int>(d, C_); wr<int>(d, H_); wr<int>(d, axis_id_); wr<int>(d, max_pillars_num_); wr<int>(d, io_half_);
    }
    Plugin* clone() const override { return new DsvtSetAttentionPlugin(max_win_num_, L_, C_, H_, axis_id_, max_pillars_num_, io_half_); }
};
static Plugin* saNew(int mw, int L, int C, int H, int axis, int mp, int io_half) {
    return (attnShapeOk(mw, L, C, H) && (axis == 0 || axis == 1) && mp > 0 && io_half >= 0 && io_half <= 2)      // io_half 2: fp32 I/O, split-precision products
               ? new DsvtSetAttentionPlugin(mw, L, C, H, axis, mp, io_half) : nullptr;
}
static Plugin* saCreate(const DsvtPluginFieldCollection* fc) {
    return saNew(fieldInt(fc, "max_win_num"), fieldInt(fc, "voxel_num_set"), fieldInt(fc, "channel_num"), fieldInt(fc, "num_heads"),
                 fieldInt(fc, "axis_id"), fieldInt(fc, "max_pillars_num"), fieldInt(fc, "io_half", 0));
}
static Plugin* saDeser(const void* data, size_t len) {
    if (len < 7 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    int mw = rd<int>(d), L = rd<int>(d), C = rd<int>(d), H = rd<int>(d), axis = rd<int>(d), mp = rd<int>(d), ioh = rd<int>(d);
    return saNew(mw, L, C, H, axis, mp, ioh);
}
static Creator g_saCreator{"DsvtSetAttentionPlugin",
    {{"max_win_num", DSVT_FIELD_INT32}, {"voxel_num_set", DSVT_FIELD_INT32}, {"channel_num", DSVT_FIELD_INT32},
     {"num_heads", DSVT_FIELD_INT32}, {"axis_id", DSVT_FIELD_INT32}, {"max_pillars_num", DSVT_FIELD_INT32},
     {"io_half", DSVT_FIELD_INT32}},
    saCreate, saDeser, {}, {}};
static Registrar g_saReg(&g_saCreator);

}  // namespace dsvt
