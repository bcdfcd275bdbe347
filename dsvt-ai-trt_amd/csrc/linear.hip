// linear.hip -- DsvtLinearPlugin: y = epilogue(A' * W^T + b) on the matrix cores of gfx950.
//
// Where the reference calls TensorRT's addFullyConnected (src/dsvt-ai-trt.cpp:283, 328-330,
// 448, 476, 490, 506, 525) followed by separate elementwise / LayerNorm / GELU plugins, this
// op runs the FC on MFMA and fuses what surrounds it:
//   prologue  A' = A + A2 for the first `add_cols` output columns (q = k = x + pos, v = x:
//             plugins/src/getValueByIndex.cu:299-301, hoisted from set slots to voxel rows)
//   epilogue  + bias; ReLU (:144) or tanh-GELU (plugins/src/gelu.cu:208-209); then up to three
//             chained "add residual, LayerNorm" stages (plugins/src/layerNorm.cu:261-402 and the
//             ElementWise SUMs around it, src/dsvt-ai-trt.cpp:669-697, 750-756).
// Rows are limited by a device-side count (count * row_mult), like every reference plugin.
//
// Both kernels compute the TRANSPOSED product tile  D[n][m] = sum_k W[n][k] A[m][k]  (W fragment
// as the MFMA A operand, activation fragment as the B operand).  In the 16x16 C/D layout a lane
// then owns ONE activation row (m = lane & 15) and four CONSECUTIVE output columns
// (n = 16t + 4*(lane>>4) + i): bias / residual / gamma / beta / output are 16-byte vector accesses,
// and a LayerNorm row reduction is 12 in-register adds plus two shuffles (xor 16, 32).
//
//   linear_f32_kernel   v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate (parity mode).
//                       64 rows x 192 columns per workgroup, K streamed through LDS in 32-chunks,
//                       rows padded to 40 floats => conflict-free ds_read_b128 (enumerated).
//   linear_f16_kernel   v_mfma_f32_16x16x32_f16: operands rounded to fp16, fp32 accumulate and
//                       fp32 epilogue.  128 rows x 192 columns per workgroup; a 192-wide K slab of
//                       W sits in LDS ([192][208] halfs, 32-byte pad => conflict-free), activations
//                       go straight from global memory into the B fragment (a lane needs 8
//                       consecutive k of one row: 32 contiguous bytes of an fp32 row, 16 of an fp16
//                       row) with the loads of a whole slab issued up front.
#include "plugin_base.h"
#include "device_utils.h"
#include "linear.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace dsvt {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 192, BK = 32, LDS_LD = 40, NT = BN / 16;

__device__ __forceinline__ float geluFast(float x) {
    // tanh-GELU of the reference (gelu.cu:208-209) in fp32:  (0.5 + 0.5 tanh(u)) x  with
    // tanh(u) = 1 - 2 / (exp(2u) + 1)  =>  x * (1 - 1 / (exp(2u) + 1)).  exp overflow gives 1/inf = 0 -> x,
    // underflow gives 1/1 -> 0: both limits are the right ones.
    // 0.5 (1 + tanh u) = 1 / (1 + exp(-2u)),  u = x (B + C x^2):  seven VALU operations (the constants carry the -2 log2(e) of
    // the exp2); exp2 -> inf gives 1/inf = 0 -> -0 for very negative x, exp2 -> 0 gives x: both limits are the right ones
    const float Bn = -2.0f * 1.4426950408889634f * 0.7978845608028654f, Cn = -2.0f * 1.4426950408889634f * 0.035677408136300125f;
    const float z = x * __builtin_fmaf(Cn, x * x, Bn);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
}

// sum over the 4 lane groups that share an activation row (lanes differing in bits 4..5)
__device__ __forceinline__ float rowSum4(float v) {
    return rows4Sum(v);
}

// Epilogue shared by both kernels.  `acc[t]` = output columns n0 + 16t + 4g .. +3 of activation
// row `row` (one row per lane, g = lane >> 4).
template <bool WITH_LN = true>
__device__ __forceinline__ void linearEpilogue(floatx4 (&acc)[NT], const LinearArgs& a, int n0, int row, int g, int M, int N)
{
    const bool valid = row < M;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = n0 + t * 16 + 4 * g;
        if (col < N) {
            if (a.bias) {
                const float4 b = *reinterpret_cast<const float4*>(a.bias + col);
                acc[t][0] += b.x; acc[t][1] += b.y; acc[t][2] += b.z; acc[t][3] += b.w;
            }
            if (a.act == ACT_RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] = fmaxf(acc[t][i], 0.f);
            } else if (a.act == ACT_GELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] = geluFast(acc[t][i]);
            }
        }
    }
    for (int s = 0; WITH_LN && s < a.n_ln; ++s) {       // y = LayerNorm_s(y + res_s); N <= BN (checked on the host), n0 == 0
        const float* res = a.res[s];
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = t * 16 + 4 * g;
            if (col < N) {
                if (valid) {
                    const float4 rv = *reinterpret_cast<const float4*>(res + (size_t)row * N + col);
                    acc[t][0] += rv.x; acc[t][1] += rv.y; acc[t][2] += rv.z; acc[t][3] += rv.w;
                }
                sum += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
            }
        }
        const float mean = rowSum4(sum) / N;                                         // layerNorm.cu:304-308
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (t * 16 + 4 * g < N)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d = acc[t][i] - mean; sq += d * d; }
        const float inv = 1.0f / sqrtf(rowSum4(sq) / N + a.eps);                      // :333-337, :274
        const float* gm = a.gamma[s]; const float* bt = a.beta[s];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = t * 16 + 4 * g;
            if (col < N) {
                const float4 g4 = *reinterpret_cast<const float4*>(gm + col), b4 = *reinterpret_cast<const float4*>(bt + col);
                acc[t][0] = (acc[t][0] - mean) * inv * g4.x + b4.x;                  // :274-276
                acc[t][1] = (acc[t][1] - mean) * inv * g4.y + b4.y;
                acc[t][2] = (acc[t][2] - mean) * inv * g4.z + b4.z;
                acc[t][3] = (acc[t][3] - mean) * inv * g4.w + b4.w;
            }
        }
    }
    // wide stores: v_permlane16_swap exchanges the odd 16-lane rows of tile t with the even rows of tile t + 1, after which the lane of
    // row r holds EIGHT consecutive columns n0 + 16 t + 16 (g & 1) + 8 (g >> 1): 16-byte fp16 / 2 x 16-byte fp32 stores, half as many
    if ((N & 15) == 0 && (a.out_ld & 7) == 0 && !(NT & 1)) {
#pragma unroll
        for (int t = 0; t < NT; t += 2) {
            floatx4 X = acc[t], Y = acc[t + 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(X[i]), __float_as_uint(Y[i]), false, false);
                X[i] = __uint_as_float(sw[0]); Y[i] = __uint_as_float(sw[1]);
            }
            const int col = n0 + t * 16 + (g & 1) * 16 + (g >> 1) * 8;
            if (!valid || col >= N) continue;
            if (a.out) {
                float* o = a.out + (size_t)row * a.out_ld + col;
                *reinterpret_cast<float4*>(o) = make_float4(X[0], X[1], X[2], X[3]);
                *reinterpret_cast<float4*>(o + 4) = make_float4(Y[0], Y[1], Y[2], Y[3]);
            }
            if (a.out16) {
                half8 h = {(_Float16)X[0], (_Float16)X[1], (_Float16)X[2], (_Float16)X[3], (_Float16)Y[0], (_Float16)Y[1], (_Float16)Y[2], (_Float16)Y[3]};
                *reinterpret_cast<half8*>(a.out16 + (size_t)row * a.out_ld + col) = h;
            }
        }
        return;
    }
    if (!valid) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = n0 + t * 16 + 4 * g;
        if (col < N) {
            if (a.out) *reinterpret_cast<float4*>(a.out + (size_t)row * a.out_ld + col) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
            if (a.out16) {
                half4 h; h[0] = (_Float16)acc[t][0]; h[1] = (_Float16)acc[t][1]; h[2] = (_Float16)acc[t][2]; h[3] = (_Float16)acc[t][3];
                *reinterpret_cast<half4*>(a.out16 + (size_t)row * a.out_ld + col) = h;
            }
        }
    }
}

__device__ __forceinline__ int rowLimit(const LinearArgs& a) {
    const long long m = (long long)(*a.count) * a.row_mult;
    return (int)(m < a.max_rows ? m : a.max_rows);
}

// -------------------------------------------------------------------------------------
// fp32 MFMA
// -------------------------------------------------------------------------------------
template <bool VEC>
__global__ void __launch_bounds__(256)
linear_f32_kernel(LinearArgs a)
{
    __shared__ __attribute__((aligned(16))) float sA[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float sW[BN * LDS_LD];
    const int M = rowLimit(a);
    const int m0 = blockIdx.x * BM;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int K = a.K, N = a.N;
    const float* A = static_cast<const float*>(a.A);
    const float* A2 = static_cast<const float*>(a.A2);

    for (int n0 = 0; n0 < N; n0 += BN) {
        floatx4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
        const bool add = n0 < a.add_cols;
        const int ntiles = (N - n0 + 15) / 16 < NT ? (N - n0 + 15) / 16 : NT;

        for (int k0 = 0; k0 < K; k0 += BK) {
            __syncthreads();
            // ---- stage A (64 x 32) and W (192 x 32) into LDS ------------------------------
            for (int i = tid; i < BM * (BK / 4); i += 256) {
                int row = i >> 3, c4 = (i & 7) * 4, gr = m0 + row, k = k0 + c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gr < M) {
                    if (VEC) {
                        if (k < K) {
                            v = *reinterpret_cast<const float4*>(A + (size_t)gr * K + k);
                            if (add) {
                                float4 w = *reinterpret_cast<const float4*>(A2 + (size_t)gr * K + k);
                                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
                            }
                        }
                    } else {
                        float t[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            t[j] = (k + j < K) ? A[(size_t)gr * K + k + j] + (add ? A2[(size_t)gr * K + k + j] : 0.f) : 0.f;
                        v = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
                *reinterpret_cast<float4*>(&sA[row * LDS_LD + c4]) = v;
            }
            for (int i = tid; i < BN * (BK / 4); i += 256) {
                int n = i >> 3, c4 = (i & 7) * 4, gn = n0 + n, k = k0 + c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gn < N) {
                    if (VEC) {
                        if (k < K) v = *reinterpret_cast<const float4*>(a.W + (size_t)gn * K + k);
                    } else {
                        float t[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[j] = (k + j < K) ? a.W[(size_t)gn * K + k + j] : 0.f;
                        v = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
                *reinterpret_cast<float4*>(&sW[n * LDS_LD + c4]) = v;
            }
            __syncthreads();
            // ---- MFMA: lane (r, g) holds k = {4g..4g+3} and {16+4g..16+4g+3} of the chunk --
            const float* pa = &sA[(wave * 16 + r) * LDS_LD + g * 4];
            const float4 a0 = *reinterpret_cast<const float4*>(pa);
            const float4 a1 = *reinterpret_cast<const float4*>(pa + 16);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int t4 = 0; t4 < NT; t4 += 4) {
                if (t4 < ntiles) {
                    float bv[4][8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float* pb = &sW[((t4 + j) * 16 + r) * LDS_LD + g * 4];
                        const float4 b0 = *reinterpret_cast<const float4*>(pb);
                        const float4 b1 = *reinterpret_cast<const float4*>(pb + 16);
                        bv[j][0] = b0.x; bv[j][1] = b0.y; bv[j][2] = b0.z; bv[j][3] = b0.w;
                        bv[j][4] = b1.x; bv[j][5] = b1.y; bv[j][6] = b1.z; bv[j][7] = b1.w;
                    }
#pragma unroll
                    for (int s = 0; s < 8; ++s)
#pragma unroll
                        for (int j = 0; j < 4; ++j)      // D[n][m]: W fragment is the A operand
                            acc[t4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j][s], av[s], acc[t4 + j], 0, 0, 0);
                }
            }
        }
        linearEpilogue(acc, a, n0, m0 + wave * 16 + r, g, M, N);
    }
}

int launchLinearF32(const LinearArgs& a, hipStream_t stream) {
    if (a.a_half) return -3;
    dim3 grid(cdiv(a.max_rows, BM)), block(256);
    if (a.K % 4 == 0) hipLaunchKernelGGL(linear_f32_kernel<true>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(linear_f32_kernel<false>, grid, block, 0, stream, a);
    return lastError();
}

// -------------------------------------------------------------------------------------
// fp16 MFMA
// -------------------------------------------------------------------------------------
constexpr int BM16 = 128;         // rows per workgroup: 4 waves x 2 MFMA row tiles
constexpr int KS = 192;           // K slab held in LDS
constexpr int LDW = KS + 16;      // halfs per LDS row
constexpr int NSTEP = KS / 32;    // MFMA k-steps per slab

__device__ __forceinline__ half8 toHalf8(float4 x, float4 y) {
    half8 h;
    h[0] = (_Float16)x.x; h[1] = (_Float16)x.y; h[2] = (_Float16)x.z; h[3] = (_Float16)x.w;
    h[4] = (_Float16)y.x; h[5] = (_Float16)y.y; h[6] = (_Float16)y.z; h[7] = (_Float16)y.w;
    return h;
}

// B-operand fragment (8 consecutive k of one activation row) straight from global memory
template <bool AHALF, bool ADD>
__device__ __forceinline__ half8 loadFrag(const void* A, const void* A2, size_t off, size_t off2);
template <bool AHALF, bool ADD>
__device__ __forceinline__ half8 loadFrag(const void* A, const void* A2, size_t off) { return loadFrag<AHALF, ADD>(A, A2, off, off); }
template <bool AHALF, bool ADD>
__device__ __forceinline__ half8 loadFrag(const void* A, const void* A2, size_t off, size_t off2) {
    if (AHALF) {
        half8 v = *reinterpret_cast<const half8*>(static_cast<const _Float16*>(A) + off);
        if (ADD) v += *reinterpret_cast<const half8*>(static_cast<const _Float16*>(A2) + off2);
        return v;
    } else {
        const float* p = static_cast<const float*>(A) + off;
        float4 x = *reinterpret_cast<const float4*>(p), y = *reinterpret_cast<const float4*>(p + 4);
        if (ADD) {
            const float* q = static_cast<const float*>(A2) + off2;
            const float4 u = *reinterpret_cast<const float4*>(q), w = *reinterpret_cast<const float4*>(q + 4);
            x.x += u.x; x.y += u.y; x.z += u.z; x.w += u.w; y.x += w.x; y.y += w.y; y.z += w.z; y.w += w.w;
        }
        return toHalf8(x, y);
    }
}

// MT = 16-row MFMA tiles per wave, NW = waves per workgroup; rows per workgroup = 16 * MT * NW = 128.
// Instantiated as <1, 8>: 8 waves x 16 rows, <=128 VGPRs, 2 workgroups/CU = 4 waves/SIMD (measured 25 % faster than 4 waves x 32 rows
// on this register-staged kernel: more waves to hide the load -> barrier -> MFMA -> store chain of one tile per SIMD)
template <bool AHALF, int MT, int NW>
__global__ void __launch_bounds__(64 * NW, (MT == 1 ? 4 : 2))
linear_f16_kernel(LinearArgs a, const _Float16* __restrict__ Wh)
{
    __shared__ __attribute__((aligned(16))) _Float16 sWh[BN * LDW];      // 79,872 B: two workgroups per CU
    constexpr int NTHR = 64 * NW;
    const int M = rowLimit(a);
    const int m0 = blockIdx.x * BM16;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int K = a.K, N = a.N;
    int row[MT], rc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        row[mt] = m0 + wave * 16 * MT + mt * 16 + r;
        rc[mt] = row[mt] < M ? row[mt] : M - 1;                          // clamp loads; rows >= M are never stored
    }

    for (int n0 = 0; n0 < N; n0 += BN) {
        floatx4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[mt][t] = floatx4{0.f, 0.f, 0.f, 0.f};
        const bool add = n0 < a.add_cols;
        const int ntiles = (N - n0 + 15) / 16 < NT ? (N - n0 + 15) / 16 : NT;
        for (int ks = 0; ks < K; ks += KS) {
            // the activation fragments of the whole slab are requested first, so their HBM latency
            // overlaps the W slab's trip L2 -> VGPR -> LDS
            half8 f[MT][NSTEP];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const size_t o = (size_t)rc[mt] * K + ks + g * 8;
                if (a.pe_xy) {                   // operand row = ReLU(BN(FC(xy))) computed in registers (K_in = 2)
                    const float2 xy = *reinterpret_cast<const float2*>(a.pe_xy + (size_t)rc[mt] * 2);
#pragma unroll
                    for (int s = 0; s < NSTEP; ++s) {
                        const int k0 = ks + s * 32 + g * 8;
                        half8 v;
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const float4 w0 = *reinterpret_cast<const float4*>(a.pe_w0 + k0 + 4 * hh);
                            const float4 w1 = *reinterpret_cast<const float4*>(a.pe_w1 + k0 + 4 * hh);
                            const float4 bb = *reinterpret_cast<const float4*>(a.pe_b + k0 + 4 * hh);
                            v[4 * hh + 0] = (_Float16)fmaxf(fmaf(xy.x, w0.x, fmaf(xy.y, w1.x, bb.x)), 0.f);
                            v[4 * hh + 1] = (_Float16)fmaxf(fmaf(xy.x, w0.y, fmaf(xy.y, w1.y, bb.y)), 0.f);
                            v[4 * hh + 2] = (_Float16)fmaxf(fmaf(xy.x, w0.z, fmaf(xy.y, w1.z, bb.z)), 0.f);
                            v[4 * hh + 3] = (_Float16)fmaxf(fmaf(xy.x, w0.w, fmaf(xy.y, w1.w, bb.w)), 0.f);
                        }
                        f[mt][s] = v;
                    }
                } else if (add) {
#pragma unroll
                    for (int s = 0; s < NSTEP; ++s) f[mt][s] = loadFrag<AHALF, true>(a.A, a.A2, o + s * 32);
                } else {
#pragma unroll
                    for (int s = 0; s < NSTEP; ++s) f[mt][s] = loadFrag<AHALF, false>(a.A, nullptr, o + s * 32);
                }
            }
            __syncthreads();                                   // previous slab's MFMAs are done with sWh
            for (int i = tid; i < BN * (KS / 8); i += NTHR) {
                const int n = i / (KS / 8), c = i % (KS / 8);
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (n0 + n < N) v = *reinterpret_cast<const uint4*>(Wh + (size_t)(n0 + n) * K + ks + c * 8);
                *reinterpret_cast<uint4*>(&sWh[n * LDW + c * 8]) = v;
            }
            __syncthreads();
            const _Float16* pw = &sWh[r * LDW + g * 8];
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (t < ntiles) {
                        const half8 wf = *reinterpret_cast<const half8*>(pw + t * 16 * LDW + s * 32);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, f[mt][s], acc[mt][t], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) linearEpilogue(acc[mt], a, n0, row[mt], g, M, N);
    }
}


// -------------------------------------------------------------------------------------
// fp16 MFMA, weights streamed by LDS-DMA (K = 192, N a multiple of 192: every fp16 launch of the frame pipeline)
// -------------------------------------------------------------------------------------
// What bounds linear_f16_kernel above is not HBM or MFMA: a 35k-row problem gives every workgroup ONE 128-row tile,
// and that tile walks load W slab -> ds_write -> barrier -> MFMA -> next slab with every load exposed (SQ_WAIT_ANY
// = 73 % of wave cycles, MFMA busy 6 %).  Here the host packs W in MFMA-fragment order -- stage = 96 columns x 192 k =
// 36 rows of 1 KB, row (k-step ks, tile t) = [lane (r, g)][8 halfs] <- W[96 s + 16 t + r][32 ks + 8 g + j] -- so a stage
// is a linear 36 KB image: it is copied by global_load_lds (no staging registers, no ds_write) into a two-slot ring one
// stage ahead of the MFMAs, an A fragment is a lane-linear (conflict-free) ds_read_b128, and the only waits are one
// s_waitcnt + raw s_barrier.
constexpr int SROWS = 36, SBYTES = SROWS * 1024;

typedef __attribute__((address_space(1))) const void* glds_src_t;
typedef __attribute__((address_space(3))) void* glds_dst_t;

__device__ __forceinline__ void stageBarrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// AMODE 0: fp32 A, 1: fp16 A, 2: position-embedding prologue (A' computed from xy).  blockIdx.y = 192-column chunk
// (one workgroup = 128 rows x 192 columns = two stages; the QKV launch is a 270 x 3 grid whose chunks re-read their
// rows from L2): no store is ever in flight while a stage is awaited -- stores share the in-order vmcnt queue with the
// DMA, and a store's acknowledgement takes microseconds when every workgroup writes at once.
template <int AMODE, int MT, int NW>
__device__ __forceinline__ void linearStreamBody(const LinearArgs a, const _Float16* __restrict__ Wp)
{
    // two stage slots (+ 3 KB: position-embedding parameters w0 | w1 | b, 192 floats each, fetched by the same DMA queue)
    __shared__ __attribute__((aligned(16))) unsigned char ring[2 * SBYTES + (AMODE == 2 ? 3072 : 0)];      // <= 76,800 B: two workgroups per CU
    const unsigned long long t_start = a.trace ? clock64() : 0ull;
    const int M = rowLimit(a);
    const int m0 = blockIdx.x * BM16;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    auto mark = [&](int i) { if (a.trace && tid == 0) a.trace[wg * 8 + i] = clock64(); };
    if (a.trace && tid == 0) a.trace[wg * 8] = t_start;
    mark(1);
    const int N = a.N, n0 = blockIdx.y * BN;
    auto request = [&](int s) {                      // s = 0, 1: the two 96-column stages of this chunk
#pragma unroll
        for (int j = 0; j < (SROWS + NW - 1) / NW; ++j) {
            const int row = wave + j * NW;
            if (SROWS % NW == 0 || row < SROWS)
                __builtin_amdgcn_global_load_lds((glds_src_t)(Wp + ((size_t)(2 * blockIdx.y + s) * SROWS + row) * 512 + lane * 8),
                                                 (glds_dst_t)(ring + s * SBYTES + row * 1024), 16, 0, 0);
        }
    };
    request(0);
    if (AMODE == 2 && wave < 3) {                    // pe_w0 | pe_w1 | pe_b, 768 B each: one 1 KB request per array
        const float* src = (wave == 0 ? a.pe_w0 : wave == 1 ? a.pe_w1 : a.pe_b) + (lane < 48 ? lane * 4 : 0);
        __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(ring + 2 * SBYTES + wave * 1024), 16, 0, 0);
    }
    request(1);
    int row[MT], rc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        row[mt] = m0 + wave * 16 * MT + mt * 16 + r;
        rc[mt] = row[mt] < M ? row[mt] : M - 1;                          // clamp loads; rows >= M are never stored
    }
    // B-operand fragments of the whole K = 192 row (A + A2 for the chunks below add_cols)
    half8 f[MT][NSTEP];
    float2 xy[MT];
    if (AMODE == 2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xy[mt] = *reinterpret_cast<const float2*>(a.pe_xy + (size_t)rc[mt] * 2);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("" :: "v"(xy[mt].x), "v"(xy[mt].y));
    } else {
        const bool withA2 = n0 < a.add_cols;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const size_t o = (size_t)rc[mt] * KS + g * 8;
            if (withA2) {
                size_t o2 = o;
                if (a.a2_c2d) {                          // table row = window cell of this voxel
                    const int32_t* c = a.a2_c2d + (size_t)rc[mt] * 3;
                    o2 = (size_t)((c[0] * a.a2_wy + c[1]) * a.a2_wx + c[2]) * KS + g * 8;
                }
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) f[mt][s] = loadFrag<AMODE == 1, true>(a.A, a.A2, o + s * 32, o2 + s * 32);
            } else {
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) f[mt][s] = loadFrag<AMODE == 1, false>(a.A, nullptr, o + s * 32);
            }
        }
        // make hipcc place its wait for these ordinary loads HERE (with an LDS-DMA in flight it waits vmcnt(0) at the first
        // use of an ordinarily loaded register): both stages and the rows are then awaited together, once
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) asm volatile("" :: "v"(f[mt][s]));
    }
    mark(2);
    stageBarrier();                                  // both stages (and the row fragments) have landed
    mark(3);
    if (AMODE == 2) {                                // operand row = ReLU(BN(FC(xy))), K_in = 2, parameters from LDS
        const float* pw0 = reinterpret_cast<const float*>(ring + 2 * SBYTES), *pw1 = pw0 + 256, *pb = pw0 + 512;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const int k0 = s * 32 + g * 8;
                half8 v;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const float4 w0 = *reinterpret_cast<const float4*>(pw0 + k0 + 4 * hh);
                    const float4 w1 = *reinterpret_cast<const float4*>(pw1 + k0 + 4 * hh);
                    const float4 bb = *reinterpret_cast<const float4*>(pb + k0 + 4 * hh);
                    v[4 * hh + 0] = (_Float16)fmaxf(fmaf(xy[mt].x, w0.x, fmaf(xy[mt].y, w1.x, bb.x)), 0.f);
                    v[4 * hh + 1] = (_Float16)fmaxf(fmaf(xy[mt].x, w0.y, fmaf(xy[mt].y, w1.y, bb.y)), 0.f);
                    v[4 * hh + 2] = (_Float16)fmaxf(fmaf(xy[mt].x, w0.z, fmaf(xy[mt].y, w1.z, bb.z)), 0.f);
                    v[4 * hh + 3] = (_Float16)fmaxf(fmaf(xy[mt].x, w0.w, fmaf(xy[mt].y, w1.w, bb.w)), 0.f);
                }
                f[mt][s] = v;
            }
        __builtin_amdgcn_sched_barrier(0);
    }

    floatx4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = floatx4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* slot = ring + lane * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ks = 0; ks < NSTEP; ++ks)
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const half8 wf = *reinterpret_cast<const half8*>(slot + h * SBYTES + (ks * 6 + t) * 1024);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][6 * h + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, f[mt][ks], acc[mt][6 * h + t], 0, 0, 0);
                if (t == 5) __builtin_amdgcn_sched_barrier(0);     // bound the hoisting of fragment reads (register budget)
            }
    mark(6);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) linearEpilogue<true>(acc[mt], a, n0, row[mt], g, M, N);
    if (a.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); mark(7); }
}

template <int AMODE, int MT, int NW>
__global__ void __launch_bounds__(64 * NW, (MT == 1 ? 4 : 2))
linear_f16_stream_kernel(LinearArgs a, const _Float16* __restrict__ Wp)
{
    linearStreamBody<AMODE, MT, NW>(a, Wp);
}

// The eight position-embedding MLPs of a frame (one per encoder layer: src/dsvt-ai-trt.cpp:603-637 builds them per block
// and per layer) depend only on the window coordinates, not on the features: one launch, blockIdx.z = layer, instead of
// eight 270-workgroup launches spread between the attention layers.
struct PosEmbedLayer {
    const float* xy; const float* pe_w0; const float* pe_w1; const float* pe_b;     // first FC (K_in = 2, BN folded) + ReLU
    const _Float16* Wp; const float* bias;                                           // second FC (fragment-ordered), bias
    _Float16* out16;
};
constexpr int kMaxPosLayers = 8;
struct PosEmbedBatch { PosEmbedLayer layer[kMaxPosLayers]; };

__global__ void __launch_bounds__(512, 4)
posembed_batched_kernel(LinearArgs a, PosEmbedBatch b)
{
    // (a dynamic index into a by-value kernel argument would be copied to scratch; constant indices are scalar kernarg loads)
    const float* xy = b.layer[0].xy; const float* w0 = b.layer[0].pe_w0; const float* w1 = b.layer[0].pe_w1; const float* pb = b.layer[0].pe_b;
    const _Float16* Wp = b.layer[0].Wp; const float* bias = b.layer[0].bias; _Float16* out16 = b.layer[0].out16;
#pragma unroll
    for (int i = 1; i < kMaxPosLayers; ++i)
        if ((int)blockIdx.z == i) { xy = b.layer[i].xy; w0 = b.layer[i].pe_w0; w1 = b.layer[i].pe_w1; pb = b.layer[i].pe_b; Wp = b.layer[i].Wp; bias = b.layer[i].bias; out16 = b.layer[i].out16; }
    LinearArgs la{};
    la.count = a.count; la.row_mult = 1; la.max_rows = a.max_rows; la.K = KS; la.N = KS; la.out_ld = KS; la.act = ACT_NONE;
    la.pe_xy = xy; la.pe_w0 = w0; la.pe_w1 = w1; la.pe_b = pb; la.bias = bias; la.out16 = out16;
    linearStreamBody<2, 1, 8>(la, Wp);
}

// -------------------------------------------------------------------------------------
// All column chunks of a row tile in ONE workgroup (QKV: N = 576 = three 192-column chunks; fp16 A, fp16 output, bias only).
// The one-chunk-per-workgroup grid above reads the activation rows once per chunk and runs 810 workgroups in 1.58 rounds of
// two per CU; here a workgroup loads its rows once, keeps FOUR 36 KB weight stages in LDS (chunk c + 1 streams in while chunk
// c computes) and stores chunk c while chunk c + 1 computes.  The stores of a chunk are YOUNGER than the LDS-DMA requests the
// next chunk waits for, so the wait is a counted vmcnt that leaves exactly those stores in flight (one in-order counter, see
// DESIGN.md "one counter").  Rows per workgroup are elastic like the encoder MLP's: 8, 9 or 10 live waves of 16 rows so that one
// workgroup per CU covers the rows.
constexpr int RW_NW = 10;             // waves of the block (live: 8 .. 10)
__global__ void __launch_bounds__(64 * RW_NW, 3)
linear_f16_rows_kernel(LinearArgs a, const _Float16* __restrict__ Wp, int ncu)
{
    __shared__ __attribute__((aligned(16))) unsigned char ring[4 * SBYTES + 4096];   // four weight stages + the bias (<= 1024 floats): one workgroup per CU
    const int M = rowLimit(a);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    int need = (M + 16 * ncu - 1) / (16 * ncu);
    // beyond 160 rows per CU the workgroups run in rounds (151 KB of LDS: one per CU); sized for TWO full rounds when that suffices
    // (69k rows: 480 workgroups of nine waves = 2 rounds, not 540 of eight = 3)
    if (need > RW_NW) need = (M + 32 * ncu - 1) / (32 * ncu);
    const int nwa = need <= 8 ? 8 : need <= RW_NW ? need : 8;
    if (wave >= nwa) return;
    const int m0 = blockIdx.x * 16 * nwa;
    if (m0 >= M) return;
    const int NCH = a.N / BN;                                        // column chunks (two stages each)
    const int nreq = (SROWS - wave + nwa - 1) / nwa;                 // fragment rows this wave requests per stage (3 .. 5)
    auto request = [&](int st) {                                     // stage st (= 2 chunk + half) -> slot st % 4
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int row = wave + j * nwa;
            if (row < SROWS)
                __builtin_amdgcn_global_load_lds((glds_src_t)(Wp + ((size_t)st * SROWS + row) * 512 + lane * 8),
                                                 (glds_dst_t)(ring + (st & 3) * SBYTES + row * 1024), 16, 0, 0);
        }
    };
    auto waitKeep = [&](int keep) {                                  // wave-uniform; then the workgroup barrier
        switch (keep) {
            case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); break;
            case 14: asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory"); break;
            case 16: asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // the bias comes by LDS-DMA as well and is read back with inline-asm ds_read: an ordinary global load in the chunk loop would be
    // awaited with vmcnt(0) (behind the previous chunk's stores), and so would a compiler-visible ds_read of memory an LDS-DMA
    // may be writing
    if (wave < 4) {
        const int f0 = wave * 256 + lane * 4;
        const float* src = a.bias ? a.bias + (f0 + 3 < a.N ? f0 : 0) : reinterpret_cast<const float*>(Wp);       // (unused lanes: any valid address)
        __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(ring + 4 * SBYTES + wave * 1024), 16, 0, 0);
    }
    const uint32_t bias_lds = (uint32_t)(uintptr_t)(glds_dst_t)(ring + 4 * SBYTES);
    request(0); request(1);
    if (NCH > 1) { request(2); request(3); }
    const int row = m0 + wave * 16 + r, rc = row < M ? row : M - 1;
    const bool waveValid = m0 + wave * 16 < M;                       // (else this wave issues no store)
    const int nst = waveValid ? 6 : 0;                               // wide stores of one chunk per wave
    // operand fragments: x (+ A2 row, gathered through the window cell when a2_c2d is set) for the chunks below add_cols, x alone above
    half8 fx[NSTEP], fp[NSTEP];
    {
        const size_t o = (size_t)rc * KS + g * 8;
        size_t o2 = o;
        if (a.a2_c2d) { const int32_t* c = a.a2_c2d + (size_t)rc * 3; o2 = (size_t)((c[0] * a.a2_wy + c[1]) * a.a2_wx + c[2]) * KS + g * 8; }
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) fx[s] = *reinterpret_cast<const half8*>(static_cast<const _Float16*>(a.A) + o + s * 32);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s)
            fp[s] = a.add_cols > 0 ? *reinterpret_cast<const half8*>(static_cast<const _Float16*>(a.A2) + o2 + s * 32) : half8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) { asm volatile("" :: "v"(fx[s])); asm volatile("" :: "v"(fp[s])); }     // hipcc's wait for the row loads: here, once
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) fp[s] += fx[s];
    }
    // stages 0, 1 landed (2, 3 may stay in flight: they are younger)
    waitKeep(NCH > 1 ? 2 * nreq : 0);
    const unsigned char* slot = ring + lane * 16;
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        const bool add = c * BN < a.add_cols;
        floatx4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned char* sp = slot + ((2 * c + h) & 3) * SBYTES;
#pragma unroll
            for (int ks = 0; ks < NSTEP; ++ks) {
                const half8 f = add ? fp[ks] : fx[ks];
#pragma unroll
                for (int t = 0; t < 6; ++t)
                    acc[6 * h + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(sp + (ks * 6 + t) * 1024), f, acc[6 * h + t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // everyone is done with the two slots of chunk c: the stages of chunk c + 2 may overwrite them
        const bool more = c + 2 < NCH;
        if (c + 1 < NCH) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
            if (more) { request(2 * c + 4); request(2 * c + 5); }
        }
        // bias + wide fp16 stores of chunk c (issued AFTER the requests above: they stay the youngest entries of the counter)
        {
            const int n0 = c * BN;
#pragma unroll
            for (int t = 0; t < NT; t += 2) {
                floatx4 X = acc[t], Y = acc[t + 1];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(X[i]), __float_as_uint(Y[i]), false, false);
                    X[i] = __uint_as_float(sw[0]); Y[i] = __uint_as_float(sw[1]);
                }
                if (a.bias) {                                // the lane's eight columns after the swap are contiguous: two 16-byte LDS reads
                    const uint32_t ad = bias_lds + (uint32_t)(n0 + t * 16 + (g & 1) * 16 + (g >> 1) * 8) * 4u;
                    floatx4 b0, b1;
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b0), "=&v"(b1) : "v"(ad));
#pragma unroll
                    for (int i = 0; i < 4; ++i) { X[i] += b0[i]; Y[i] += b1[i]; }
                }
                if (row < M) {
                    half8 hv = {(_Float16)X[0], (_Float16)X[1], (_Float16)X[2], (_Float16)X[3], (_Float16)Y[0], (_Float16)Y[1], (_Float16)Y[2], (_Float16)Y[3]};
                    *reinterpret_cast<half8*>(a.out16 + (size_t)row * a.out_ld + n0 + t * 16 + (g & 1) * 16 + (g >> 1) * 8) = hv;
                }
            }
        }
        // the stages of chunk c + 1 have landed: everything older than [requests of chunk c + 2] [stores of chunk c] is retired
        if (c + 1 < NCH) waitKeep((more ? 2 * nreq : 0) + nst);
    }
}

// -------------------------------------------------------------------------------------
// QKV with the weights RESIDENT in LDS (round 2), for launches that carry three or more frames' rows.  The kernel above re-streams all
// 216 KB of W_qkv for every 128-160 rows, through one ten-wave workgroup per CU whose waves move in lockstep (load rows, three chunks,
// store).  Here a CU keeps HALF of the output columns' weights -- three 96-column stages, 108 KB -- in LDS for the whole launch
// (workgroups alternate between the two halves, so every activation row is read by two workgroups: 384 B more per row next to the
// 1152 B it produces) and its eight waves walk the 16-row tiles independently: no barrier after the prologue, the rows of a wave's
// next tile are loaded while it computes the current one (register ping-pong), the window-cell index of the tile after that (the
// position table's gather index) one tile earlier still.  Measured per launch, rows of 1 / 2 / 4 frames: 26.5 / 42.4 / 75.8 us
// against 24.7 / 44.0 / 85.4 for the streamed kernel (the 108 KB prologue of every CU costs what one frame's rows save); 265 MB
// of traffic in 76 us = 3.5 TB/s -- twelve waves at 168 registers: the same 76 us.
constexpr int RS_NW = 8, RS_SPT = 3;
template <bool TABLE>             // the A2 rows are gathered through the window cell (a2_c2d)
__global__ void __launch_bounds__(64 * RS_NW, 2)
linear_f16_resident_kernel(LinearArgs a, const _Float16* __restrict__ Wp)
{
    __shared__ __attribute__((aligned(16))) unsigned char ring[RS_SPT * SBYTES + 4096];   // three weight stages + the bias
    const int M = rowLimit(a);
    if (M <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int ntype = a.N / (96 * RS_SPT);
    const int type = (int)blockIdx.x % ntype, j = (int)blockIdx.x / ntype, nj = (int)gridDim.x / ntype;
    for (int row = wave; row < RS_SPT * SROWS; row += RS_NW)
        __builtin_amdgcn_global_load_lds((glds_src_t)(Wp + ((size_t)(type * RS_SPT) * SROWS + row) * 512 + lane * 8), (glds_dst_t)(ring + row * 1024), 16, 0, 0);
    if (wave < 4) {
        const int f0 = wave * 256 + lane * 4;
        const float* src = a.bias ? a.bias + (f0 + 3 < a.N ? f0 : 0) : reinterpret_cast<const float*>(Wp);       // (unused lanes: any valid address)
        __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(ring + RS_SPT * SBYTES + wave * 1024), 16, 0, 0);
    }
    const uint32_t bias_lds = (uint32_t)(uintptr_t)(glds_dst_t)(ring + RS_SPT * SBYTES);
    const int ntile = (M + 15) >> 4, step = nj * RS_NW;
    int tt = j * RS_NW + wave;
    // the window cell of this lane's row of tile t (the position table's gather index), as loaded: the multiply waits for the load,
    // so it is left to the tile that USES the index (one tile later).  Prefetches past the last tile re-read the last tile (no
    // branch around a load: a value merged from two paths is copied, and the copy waits for the load)
    const bool hasA2 = a.add_cols > 0;
    auto cellOf = [&](int t) -> int2 {
        t = t < ntile ? t : ntile - 1;
        const int row = t * 16 + r, rc = row < M ? row : M - 1;
        if (!TABLE) return make_int2(0, rc);
        const int32_t* c = a.a2_c2d + (size_t)rc * 3;
        return make_int2(c[0] * a.a2_wy + c[1], c[2]);      // (z * wy + y, x): a2_wy = 0 for the pillar model's 2-D tables
    };
    auto loadRows = [&](int t, int2 cell, half8 (&x)[NSTEP], half8 (&p)[NSTEP]) {
        t = t < ntile ? t : ntile - 1;
        const int row = t * 16 + r, rc = row < M ? row : M - 1;
        const size_t o = (size_t)rc * KS + g * 8, o2 = (size_t)((TABLE ? cell.x * a.a2_wx : 0) + cell.y) * KS + g * 8;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) x[s] = *reinterpret_cast<const half8*>(static_cast<const _Float16*>(a.A) + o + s * 32);
        if (hasA2) {
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) p[s] = *reinterpret_cast<const half8*>(static_cast<const _Float16*>(a.A2) + o2 + s * 32);
        } else {
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) p[s] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    const unsigned char* slot = ring + lane * 16;
    // one 16-row tile: fp += fx, three stages of 36 MFMAs, bias / fp16 / wide stores per stage
    auto tile = [&](int t, half8 (&fx)[NSTEP], half8 (&fp)[NSTEP]) {
        const int row = t * 16 + r;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) fp[s] += fx[s];
#pragma unroll
        for (int h = 0; h < RS_SPT; ++h) {
            const int n0 = (type * RS_SPT + h) * 96;
            const bool add = n0 < a.add_cols;
            floatx4 acc[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) acc[u] = floatx4{0.f, 0.f, 0.f, 0.f};
            const unsigned char* sp = slot + h * SBYTES;
#pragma unroll
            for (int ks = 0; ks < NSTEP; ++ks) {
                const half8 f = add ? fp[ks] : fx[ks];
#pragma unroll
                for (int u = 0; u < 6; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(sp + (ks * 6 + u) * 1024), f, acc[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 6; u += 2) {
                floatx4 X = acc[u], Y = acc[u + 1];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(X[i]), __float_as_uint(Y[i]), false, false);
                    X[i] = __uint_as_float(sw[0]); Y[i] = __uint_as_float(sw[1]);
                }
                const int col = n0 + u * 16 + (g & 1) * 16 + (g >> 1) * 8;       // the lane's eight columns after the swap are contiguous
                if (a.bias) {
                    const uint32_t ad = bias_lds + (uint32_t)col * 4u;
                    floatx4 b0, b1;
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b0), "=&v"(b1) : "v"(ad));
#pragma unroll
                    for (int i = 0; i < 4; ++i) { X[i] += b0[i]; Y[i] += b1[i]; }
                }
                if (row < M) {
                    half8 hv = {(_Float16)X[0], (_Float16)X[1], (_Float16)X[2], (_Float16)X[3], (_Float16)Y[0], (_Float16)Y[1], (_Float16)Y[2], (_Float16)Y[3]};
                    *reinterpret_cast<half8*>(a.out16 + (size_t)row * a.out_ld + col) = hv;
                }
            }
        }
    };
    half8 xa[NSTEP], pa[NSTEP], xb[NSTEP], pb[NSTEP];
    loadRows(tt, cellOf(tt), xa, pa);
    int2 cellN = cellOf(tt + step);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the weights (and the first rows) have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    while (tt < ntile) {
        const int tn = tt + step;
        loadRows(tn, cellN, xb, pb);
        const int2 cell2 = cellOf(tn + step);
        tile(tt, xa, pa);
        if (tn >= ntile) break;
        tt = tn + step;
        loadRows(tt, cell2, xa, pa);
        cellN = cellOf(tt + step);
        tile(tn, xb, pb);
    }
}

// -------------------------------------------------------------------------------------
// QKV at fp32 GRADE on the fp16 matrix cores (compute_type 2, "split precision"; round 3).  The reference's arithmetic is fp32
// (include/params.h:332 leaves USE_FP16 commented out; every plugin demands kFLOAT), and fp16 operands cannot meet the 1e-3 box bar
// (profiles/r02_f16_error_attribution.txt); v_mfma_f32_16x16x4_f32 meets it at 1/16 of the fp16 matrix rate.  Here every operand is the
// pair  hi = fp16(v), lo = fp16(v - hi)  (22 mantissa bits) and a product is three fp16 MFMAs with fp32 accumulation:
//     a w  ~=  a_hi w_hi + a_hi w_lo + a_lo w_hi                      (the dropped a_lo w_lo term is 2^-22 of the product)
// A, A2 (the position table) and the output are fp32 tensors in HBM; the split of the activations happens in registers, the weights
// are split by the host.  Structure = linear_f16_rows_kernel: a workgroup loads its rows once, keeps four 36 KB weight stages in LDS
// (a stage = 48 output columns: per k-step three fragment rows of w_hi followed by the same three of w_lo), a column chunk = two
// stages = 96 columns; the fp32 stores of chunk c drain under the MFMAs of chunk c + 1 (counted vmcnt, see above).
constexpr int SP_NW = 8;              // waves of 16 rows (<= 256 VGPRs: the row fragments alone are 96)
constexpr int SP_COLS = 48;           // output columns per stage

// hi / lo split of 8 consecutive k of one row (the two float4 halves), saturated so that |v| > 65504 gives hi = +-65504, not inf
struct HiLo8 { half8 hi, lo; };
__device__ __forceinline__ HiLo8 splitFrag(floatx4 x, floatx4 y) {
    const float v[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    _Float16 h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = (_Float16)__builtin_fminf(__builtin_fmaxf(v[i], -65504.f), 65504.f);
        l[i] = (_Float16)__builtin_fminf(__builtin_fmaxf(v[i] - (float)h[i], -65504.f), 65504.f);
    }
    HiLo8 o;
    o.hi = half8{h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
    o.lo = half8{l[0], l[1], l[2], l[3], l[4], l[5], l[6], l[7]};
    return o;
}

template <bool TABLE>             // the A2 rows are gathered through the window cell (a2_c2d)
__global__ void __launch_bounds__(64 * SP_NW, 2)
linear_split_rows_kernel(LinearArgs a, const _Float16* __restrict__ Wp)
{
    __shared__ __attribute__((aligned(16))) unsigned char ring[4 * SBYTES + 4096];   // four weight stages + the bias (<= 1024 floats): one workgroup per CU
    const int M = rowLimit(a);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 16 * SP_NW;
    if (m0 >= M) return;
    const int NCH = a.N / (2 * SP_COLS);                             // 96-column chunks (two stages each)
    const int nreq = (SROWS - wave + SP_NW - 1) / SP_NW;             // fragment rows this wave requests per stage (5 or 4)
    auto request = [&](int st) {                                     // stage st (= 2 chunk + half) -> slot st % 4
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int row = wave + j * SP_NW;
            if (row < SROWS)
                __builtin_amdgcn_global_load_lds((glds_src_t)(Wp + ((size_t)st * SROWS + row) * 512 + lane * 8),
                                                 (glds_dst_t)(ring + (st & 3) * SBYTES + row * 1024), 16, 0, 0);
        }
    };
    auto waitKeep = [&](int keep) {                                  // wave-uniform; then the workgroup barrier
        switch (keep) {
            case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory"); break;
            case 14: asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory"); break;
            case 16: asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;      // (waiting for more than needed is always correct)
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    if (wave < 4) {                                                  // the bias by LDS-DMA (see linear_f16_rows_kernel)
        const int f0 = wave * 256 + lane * 4;
        const float* src = a.bias ? a.bias + (f0 + 3 < a.N ? f0 : 0) : reinterpret_cast<const float*>(Wp);
        __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(ring + 4 * SBYTES + wave * 1024), 16, 0, 0);
    }
    const uint32_t bias_lds = (uint32_t)(uintptr_t)(glds_dst_t)(ring + 4 * SBYTES);
    request(0); request(1);
    if (NCH > 1) { request(2); request(3); }
    const int row = m0 + wave * 16 + r, rc = row < M ? row : M - 1;
    const bool waveValid = m0 + wave * 16 < M;                       // (else this wave issues no store)
    const int nst = waveValid ? 6 : 0;                               // 16-byte stores of one chunk per wave
    // operand fragments: x alone for the chunks at or above add_cols, x + A2 row (gathered through the window cell when TABLE) below
    half8 fxh[NSTEP], fxl[NSTEP], fph[NSTEP], fpl[NSTEP];
    {
        const float* px = static_cast<const float*>(a.A) + (size_t)rc * KS + g * 8;
        size_t o2 = (size_t)rc * KS + g * 8;
        if (TABLE) { const int32_t* c = a.a2_c2d + (size_t)rc * 3; o2 = (size_t)((c[0] * a.a2_wy + c[1]) * a.a2_wx + c[2]) * KS + g * 8; }
        const bool hasA2 = a.add_cols > 0;
        const float* pp = hasA2 ? static_cast<const float*>(a.A2) + o2 : px;
        floatx4 xv[2 * NSTEP], pv[2 * NSTEP];
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) { xv[2 * s] = *reinterpret_cast<const floatx4*>(px + s * 32); xv[2 * s + 1] = *reinterpret_cast<const floatx4*>(px + s * 32 + 4); }
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) { pv[2 * s] = *reinterpret_cast<const floatx4*>(pp + s * 32); pv[2 * s + 1] = *reinterpret_cast<const floatx4*>(pp + s * 32 + 4); }
#pragma unroll
        for (int s = 0; s < 2 * NSTEP; ++s) { asm volatile("" :: "v"(xv[s])); asm volatile("" :: "v"(pv[s])); }     // hipcc's wait for the row loads: here, once
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const HiLo8 fx = splitFrag(xv[2 * s], xv[2 * s + 1]);
            const HiLo8 fp = splitFrag(pv[2 * s] + xv[2 * s], pv[2 * s + 1] + xv[2 * s + 1]);      // q = k = x + pos in fp32 (getValueByIndex.cu:299-301)
            fxh[s] = fx.hi; fxl[s] = fx.lo; fph[s] = fp.hi; fpl[s] = fp.lo;
        }
    }
    waitKeep(NCH > 1 ? 2 * nreq : 0);                                // stages 0, 1 landed (2, 3 may stay in flight: they are younger)
    const unsigned char* slot = ring + lane * 16;
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        const bool add = c * 2 * SP_COLS < a.add_cols;
        floatx4 acc[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned char* sp = slot + ((2 * c + h) & 3) * SBYTES;
#pragma unroll
            for (int ks = 0; ks < NSTEP; ++ks) {
                const half8 fh = add ? fph[ks] : fxh[ks], fl = add ? fpl[ks] : fxl[ks];
                half8 wh[3], wl[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) { wh[t] = *reinterpret_cast<const half8*>(sp + (ks * 6 + t) * 1024); wl[t] = *reinterpret_cast<const half8*>(sp + (ks * 6 + 3 + t) * 1024); }
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[3 * h + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], fh, acc[3 * h + t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[3 * h + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], fl, acc[3 * h + t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[3 * h + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], fh, acc[3 * h + t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // everyone is done with the two slots of chunk c: the stages of chunk c + 2 may overwrite them
        const bool more = c + 2 < NCH;
        if (c + 1 < NCH) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
            if (more) { request(2 * c + 4); request(2 * c + 5); }
        }
        // bias + fp32 stores of chunk c (issued AFTER the requests above: they stay the youngest entries of the counter)
        {
            const int n0 = c * 2 * SP_COLS;
#pragma unroll
            for (int t = 0; t < 6; t += 2) {
                floatx4 X = acc[t], Y = acc[t + 1];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(X[i]), __float_as_uint(Y[i]), false, false);
                    X[i] = __uint_as_float(sw[0]); Y[i] = __uint_as_float(sw[1]);
                }
                const int col = n0 + t * 16 + (g & 1) * 16 + (g >> 1) * 8;       // the lane's eight columns after the swap are contiguous
                if (a.bias) {
                    const uint32_t ad = bias_lds + (uint32_t)col * 4u;
                    floatx4 b0, b1;
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b0), "=&v"(b1) : "v"(ad));
#pragma unroll
                    for (int i = 0; i < 4; ++i) { X[i] += b0[i]; Y[i] += b1[i]; }
                }
                if (row < M) {
                    float* o = a.out + (size_t)row * a.out_ld + col;
                    *reinterpret_cast<float4*>(o) = make_float4(X[0], X[1], X[2], X[3]);
                    *reinterpret_cast<float4*>(o + 4) = make_float4(Y[0], Y[1], Y[2], Y[3]);
                }
            }
        }
        // the stages of chunk c + 1 have landed: everything older than [requests of chunk c + 2] [stores of chunk c] is retired
        if (c + 1 < NCH) waitKeep((more ? 2 * nreq : 0) + nst);
    }
}

// -------------------------------------------------------------------------------------
// Split-precision QKV with the weights RESIDENT in LDS (round 4).  The kernel above streams all 442 KB of (w_hi, w_lo) through LDS for every
// 128 rows -- at the ~25 GB/s a CU's LDS-DMA path delivers that is 17.7 us per workgroup where its MFMAs are 10 us, 184 us per four-frame
// launch and, with 269 workgroups of a frame on 256 CUs, two rounds (76 us) for one frame.  Here a CU keeps a THIRD of the output columns --
// four 48-column stages of w_hi + w_lo, 144 KB -- for the whole launch (workgroups cycle through the three thirds, so every activation row is
// read by three workgroups of neighbouring CUs: from L2 after the first) and its eight waves walk the 16-row tiles independently, no barrier
// after the prologue, like linear_f16_resident_kernel.  A third lies wholly inside or outside the "x + position" columns (add_cols a multiple
// of 192), so a workgroup splits ONE operand per row tile (48 registers); the fp32 rows of the next tile are requested as soon as the current
// tile's are split (their registers are dead by then) and land under its 216 MFMAs.  133 us per four-frame launch, 51 (one frame, the pipeline's tables) against
// 184 / 76; DSVT_QKV_DBG ladder of the ablation build on tools/bench_qkv_split.py (dense position rows, 236 us): without stores 163, without the row loads 167,
// without MFMAs 207, nothing but the fragment reads and the loop 62 -- the fragment reads (144 KB per tile and wave, ~110 B per cycle and CU) are the floor
// under the 421 MB of fp32 traffic (3.2 TB/s mixed).  Walking PAIRS of row tiles per wave (k-step outermost, operands split one 32-channel slice at a time,
// 96 accumulator registers: every fragment feeds six MFMAs) was built and is bit-identical: 1.17 ms per eight launches against 1.07 at four frames (the 24
// stores of a pair leave in one burst), 0.39 against 0.41 at one -- not kept.
constexpr int RSS_NW = 8, RSS_SPT = 4;
template <bool TABLE>             // the A2 rows are gathered through the window cell (a2_c2d)
__global__ void __launch_bounds__(64 * RSS_NW, 1)
linear_split_resident_kernel(LinearArgs a, const _Float16* __restrict__ Wp, int dbg)      // dbg (ablation build): 1 no MFMA, 2 no stores, 4 rows not split, 8 rows loaded once
{
    __shared__ __attribute__((aligned(16))) unsigned char ring[RSS_SPT * SBYTES + 4096];   // four weight stages + the bias (<= 1024 floats)
    const int M = rowLimit(a);
    if (M <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    // the three workgroups of a row stream on ONE XCD (workgroups go to the eight XCDs round-robin, each with its own L2): the rows the first of them
    // fetches are L2 hits for the other two.  Per XCD the slots b / 8 = 0 .. 3 q - 1 form q streams; the slots left over form streams across XCDs
    constexpr int ntype = 3;
    const int q = (int)gridDim.x / 8 / ntype;                        // streams per XCD
    const int xcd = (int)blockIdx.x % 8, sl = (int)blockIdx.x / 8;
    const int nj = 8 * q + ((int)gridDim.x - 8 * q * ntype) / ntype;
    int type, j;
    if (sl < q * ntype) { type = sl % ntype; j = xcd * q + sl / ntype; }
    else {
        const int lo = (sl - q * ntype) * 8 + xcd;                    // leftover workgroups, in blockIdx order
        type = lo % ntype; j = 8 * q + lo / ntype;
        if (j >= nj) return;
    }
    for (int row = wave; row < RSS_SPT * SROWS; row += RSS_NW)
        __builtin_amdgcn_global_load_lds((glds_src_t)(Wp + ((size_t)(type * RSS_SPT) * SROWS + row) * 512 + lane * 8), (glds_dst_t)(ring + row * 1024), 16, 0, 0);
    if (wave < 4) {
        const int f0 = wave * 256 + lane * 4;
        const float* src = a.bias ? a.bias + (f0 + 3 < a.N ? f0 : 0) : reinterpret_cast<const float*>(Wp);       // (unused lanes: any valid address)
        __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(ring + RSS_SPT * SBYTES + wave * 1024), 16, 0, 0);
    }
    const uint32_t bias_lds = (uint32_t)(uintptr_t)(glds_dst_t)(ring + RSS_SPT * SBYTES);
    const int ntile = (M + 15) >> 4, step = nj * RSS_NW;
    int tt = j * RSS_NW + wave;
    const int nbase = type * RSS_SPT * SP_COLS;                        // first output column of this third
    const bool add = nbase < a.add_cols;                              // the operand is x + A2 row (q, k) or x alone (v)
    // the window cell of this lane's row of tile t (the position table's gather index); prefetches past the last tile re-read the last tile
    auto cellOf = [&](int t) -> int2 {
        t = t < ntile ? t : ntile - 1;
        const int row = t * 16 + r, rc = row < M ? row : M - 1;
        if (!TABLE) return make_int2(0, rc);
        const int32_t* c = a.a2_c2d + (size_t)rc * 3;
        return make_int2(c[0] * a.a2_wy + c[1], c[2]);      // (z * wy + y, x): a2_wy = 0 for the pillar model's 2-D tables
    };
    floatx4 xv[2 * NSTEP], pv[2 * NSTEP];
    auto loadRows = [&](int t, int2 cell) {
        t = t < ntile ? t : ntile - 1;
        const int row = t * 16 + r, rc = row < M ? row : M - 1;
        const float* px = static_cast<const float*>(a.A) + (size_t)rc * KS + g * 8;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) { xv[2 * s] = *reinterpret_cast<const floatx4*>(px + s * 32); xv[2 * s + 1] = *reinterpret_cast<const floatx4*>(px + s * 32 + 4); }
        if (add) {
            const float* pp = static_cast<const float*>(a.A2) + (size_t)((TABLE ? cell.x * a.a2_wx : 0) + cell.y) * KS + g * 8;
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) { pv[2 * s] = *reinterpret_cast<const floatx4*>(pp + s * 32); pv[2 * s + 1] = *reinterpret_cast<const floatx4*>(pp + s * 32 + 4); }
        }
    };
    const unsigned char* slot = ring + lane * 16;
    if (tt >= ntile) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); return; }
    loadRows(tt, cellOf(tt));
    int2 cellN = cellOf(tt + step);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the weights (and the first rows) have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    while (tt < ntile) {
        const int row = tt * 16 + r;
        half8 fh[NSTEP], fl[NSTEP];
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (kAblate && (dbg & 4)) { fh[s] = __builtin_bit_cast(half8, xv[2 * s]); fl[s] = __builtin_bit_cast(half8, xv[2 * s + 1]); continue; }
            const HiLo8 f = add ? splitFrag(pv[2 * s] + xv[2 * s], pv[2 * s + 1] + xv[2 * s + 1])      // q = k = x + pos in fp32 (getValueByIndex.cu:299-301)
                                : splitFrag(xv[2 * s], xv[2 * s + 1]);
            fh[s] = f.hi; fl[s] = f.lo;
        }
        // the next tile's rows into the registers just split (in flight under this tile's MFMAs), the gather index of the one after
        const int tn = tt + step;
        asm volatile("" ::: "memory");
        if (!kAblate || !(dbg & 8)) loadRows(tn, cellN);
        cellN = cellOf(tn + step);
        // 24 steps (stage h2 = 48 columns, k-step ks) of nine MFMAs; the six weight fragments of step i + 1 are read before the MFMAs of step i
        half8 wq[2][6];
        auto loadFr = [&](int i, half8 (&w)[6]) {
            const unsigned char* sp = slot + (i / NSTEP) * SBYTES + (i % NSTEP) * 6 * 1024;
#pragma unroll
            for (int t = 0; t < 6; ++t) w[t] = *reinterpret_cast<const half8*>(sp + t * 1024);
        };
        loadFr(0, wq[0]);
#pragma unroll
        for (int c = 0; c < RSS_SPT / 2; ++c) {                       // 96 columns at a time
            floatx4 acc[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int ks = 0; ks < NSTEP; ++ks) {
                    const int i = (2 * c + h) * NSTEP + ks;
                    if (i + 1 < RSS_SPT * NSTEP) loadFr(i + 1, wq[(i + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    const half8 (&w)[6] = wq[i & 1];
                    if (kAblate && (dbg & 1)) { acc[3 * h][0] += (float)w[0][0] + (float)w[5][7] + (float)fh[ks][0] + (float)fl[ks][1]; continue; }
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[3 * h + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[3 + t], fh[ks], acc[3 * h + t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[3 * h + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[t], fl[ks], acc[3 * h + t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[3 * h + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[t], fh[ks], acc[3 * h + t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const int n0 = nbase + c * 2 * SP_COLS;
#pragma unroll
            for (int t = 0; t < 6; t += 2) {
                floatx4 X = acc[t], Y = acc[t + 1];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(X[i]), __float_as_uint(Y[i]), false, false);
                    X[i] = __uint_as_float(sw[0]); Y[i] = __uint_as_float(sw[1]);
                }
                const int col = n0 + t * 16 + (g & 1) * 16 + (g >> 1) * 8;       // the lane's eight columns after the swap are contiguous
                if (a.bias) {
                    const floatx4 b0 = *reinterpret_cast<const floatx4*>(ring + RSS_SPT * SBYTES + col * 4), b1 = *reinterpret_cast<const floatx4*>(ring + RSS_SPT * SBYTES + col * 4 + 16);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { X[i] += b0[i]; Y[i] += b1[i]; }
                }
                if (row < M && (!kAblate || !(dbg & 2) || X[0] == 1.2345f)) {
                    float* o = a.out + (size_t)row * a.out_ld + col;
                    *reinterpret_cast<float4*>(o) = make_float4(X[0], X[1], X[2], X[3]);
                    *reinterpret_cast<float4*>(o + 4) = make_float4(Y[0], Y[1], Y[2], Y[3]);
                }
            }
        }
        tt = tn;
    }
    (void)bias_lds;
}

// stage image of the split kernel: stage s = output columns [48 s, 48 s + 48), row (ks, t) = w_hi of column tile t for t < 3, w_lo of tile t - 3 above
static std::vector<_Float16> packStagesSplit(const float* W, int N) {
    std::vector<_Float16> out((size_t)(N / SP_COLS) * SROWS * 512);
    for (int s = 0; s < N / SP_COLS; ++s)
        for (int ks = 0; ks < NSTEP; ++ks)
            for (int t = 0; t < 6; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const float w = W[(size_t)(SP_COLS * s + 16 * (t % 3) + (lane & 15)) * KS + 32 * ks + 8 * (lane >> 4) + j];
                        const _Float16 hi = (_Float16)w;
                        out[(((size_t)s * SROWS + ks * 6 + t) * 64 + lane) * 8 + j] = t < 3 ? hi : (_Float16)(w - (float)hi);
                    }
    return out;
}

static int launchLinearSplit(const LinearArgs& a, const _Float16* Wp, hipStream_t stream) {
    static int resident = -1;      // DSVT_LINEAR_RESIDENT=0: the streamed kernel for every shape
    if (resident < 0) resident = ablateEnv("DSVT_LINEAR_RESIDENT", 1);
    const int ncu = deviceCUs();
    // (the resident kernel forms row streams of three workgroups; a device -- or partition -- with fewer than three CUs would form none and
    // leave the output unwritten: it takes the streamed kernel)
    if (resident && ncu >= 3 && a.N == 3 * RSS_SPT * SP_COLS && (a.add_cols % (RSS_SPT * SP_COLS)) == 0) {
        const int ng = ncu;                                             // (the kernel forms row streams of three workgroups per XCD; at most two workgroups idle)
        static int dbg = -1; if (dbg < 0) dbg = ablateEnv("DSVT_QKV_DBG", 0);
        if (a.a2_c2d) hipLaunchKernelGGL(linear_split_resident_kernel<true>, dim3(ng), dim3(64 * RSS_NW), 0, stream, a, Wp, dbg);
        else hipLaunchKernelGGL(linear_split_resident_kernel<false>, dim3(ng), dim3(64 * RSS_NW), 0, stream, a, Wp, dbg);
        return lastError();
    }
    const dim3 grid(cdiv(a.max_rows, 16 * SP_NW)), block(64 * SP_NW);
    if (a.a2_c2d) hipLaunchKernelGGL(linear_split_rows_kernel<true>, grid, block, 0, stream, a, Wp);
    else hipLaunchKernelGGL(linear_split_rows_kernel<false>, grid, block, 0, stream, a, Wp);
    return lastError();
}

static int launchLinearF16Resident(const LinearArgs& a, const _Float16* Wp, hipStream_t stream) {
    const int ncu = deviceCUs();
    const int ntype = a.N / (96 * RS_SPT);
    if (a.a2_c2d) hipLaunchKernelGGL(linear_f16_resident_kernel<true>, dim3(ncu / ntype * ntype), dim3(64 * RS_NW), 0, stream, a, Wp);
    else hipLaunchKernelGGL(linear_f16_resident_kernel<false>, dim3(ncu / ntype * ntype), dim3(64 * RS_NW), 0, stream, a, Wp);
    return lastError();
}

int launchLinearF16Rows(const LinearArgs& a, const _Float16* Wp, hipStream_t stream) {
    const int ncu = deviceCUs();
    hipLaunchKernelGGL(linear_f16_rows_kernel, dim3(cdiv(a.max_rows, 128)), dim3(64 * RW_NW), 0, stream, a, Wp, ncu);
    return lastError();
}

int launchLinearF16Stream(const LinearArgs& a, const _Float16* Wp, hipStream_t stream) {
    if (a.K != KS || a.N % BN != 0 || (a.N > BN && a.n_ln > 0)) return -3;
    static int rowsOn = -1;        // DSVT_LINEAR_ROWS=0: one column chunk per workgroup for every layer
    if (rowsOn < 0) rowsOn = ablateEnv("DSVT_LINEAR_ROWS", 1);
    if (rowsOn && a.N > BN && a.a_half && !a.pe_xy && a.out16 && !a.out && a.act == ACT_NONE && a.n_ln == 0 && a.row_mult == 1 &&
        (a.add_cols % BN) == 0 && a.N <= 1024 && !a.trace) {
        static int resident = -1;  // DSVT_LINEAR_RESIDENT=0: the streamed whole-row kernel for every shape; 2: the resident one whatever the row capacity
        if (resident < 0) resident = ablateEnv("DSVT_LINEAR_RESIDENT", 1);
        if (resident && a.N == 96 * RS_SPT * 2 && (a.add_cols % 96) == 0 && (a.max_rows >= 3 * 65536 || resident == 2) && deviceCUs() >= a.N / (96 * RS_SPT)) return launchLinearF16Resident(a, Wp, stream);   // (fewer CUs than column types: the rows kernel, not an empty grid)
        return launchLinearF16Rows(a, Wp, stream);
    }
    dim3 grid(cdiv(a.max_rows, BM16), a.N / BN);
    const int amode = a.pe_xy ? 2 : a.a_half ? 1 : 0;
    static int mt2 = -1;           // DSVT_STREAM_MT=2: 4 waves x 32 rows (<= 256 VGPRs); default 8 waves x 16 rows (<= 128 VGPRs, 4 waves/SIMD):
    if (mt2 < 0) mt2 = ablateEnv("DSVT_STREAM_MT", 1);      // 1.5 % faster with two frames in flight, slower alone
    const bool wide = mt2 == 2;
#define DSVT_LS(AM) do { if constexpr (kAblate) { if (wide) { hipLaunchKernelGGL((linear_f16_stream_kernel<AM, 2, 4>), grid, dim3(256), 0, stream, a, Wp); break; } } \
                         hipLaunchKernelGGL((linear_f16_stream_kernel<AM, 1, 8>), grid, dim3(512), 0, stream, a, Wp); } while (0)
    if (amode == 2) DSVT_LS(2); else if (amode == 1) DSVT_LS(1); else DSVT_LS(0);
#undef DSVT_LS
    return lastError();
}

// fragment-ordered stage image of W [N][192] (N a multiple of 96)
static std::vector<_Float16> packStages(const float* W, int N) {
    std::vector<_Float16> out((size_t)N * KS);
    for (int s = 0; s < N / 96; ++s)
        for (int ks = 0; ks < NSTEP; ++ks)
            for (int t = 0; t < 6; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j)
                        out[(((size_t)s * SROWS + ks * 6 + t) * 64 + lane) * 8 + j] =
                            (_Float16)W[(size_t)(96 * s + 16 * t + (lane & 15)) * KS + 32 * ks + 8 * (lane >> 4) + j];
    return out;
}

int launchLinearF16(const LinearArgs& a, const _Float16* Wh, hipStream_t stream) {
    if (a.K % KS != 0) return -3;
    dim3 grid(cdiv(a.max_rows, BM16));
    if (a.a_half) hipLaunchKernelGGL((linear_f16_kernel<true, 1, 8>), grid, dim3(512), 0, stream, a, Wh);
    else hipLaunchKernelGGL((linear_f16_kernel<false, 1, 8>), grid, dim3(512), 0, stream, a, Wh);
    return lastError();
}

// -------------------------------------------------------------------------------------
// plugin
// -------------------------------------------------------------------------------------
enum { OUT_F32 = 0, OUT_F16 = 1, OUT_BOTH = 2 };

struct LinCfg {
    int max_rows, K, N, row_mult, act, add_cols, n_ln; float eps;
    int compute_type;      // 0: fp32 MFMA (exact fp32 products)   1: fp16 MFMA operands, fp32 accumulate   2: split-precision fp16 MFMA (hi + lo operands, fp32 grade)
    int input_half;        // A / A2 tensors are fp16 (needs compute_type 1)
    int output_mode;       // OUT_F32: one fp32 output; OUT_F16: one fp16 output; OUT_BOTH: fp32 + fp16 copy
    int a2_gather_wx;      // > 0: input 2 is a [cells, K] table and input 3 the [rows, 3] window coordinates (z, y, x); row m adds table row y * wx + x
    int a2_gather_wy;      // > 0 (optional field "add_gather_height", 3-D windows: BASELINE configs[4]): table row (z * wy + y) * wx + x
};

class DsvtLinearPlugin : public Plugin {
public:
    LinCfg c_;
    std::vector<float> w_, b_, g_, be_, pe_;       // pe_: [w0 (K) | w1 (K) | b (K)] of the fused K_in = 2 first FC, or empty
    float *w_dev_ = nullptr, *b_dev_ = nullptr, *g_dev_ = nullptr, *be_dev_ = nullptr, *pe_dev_ = nullptr;
    _Float16* wh_dev_ = nullptr;
    _Float16* wp_dev_ = nullptr;      // fragment-ordered stage image for the LDS-DMA kernel (K = 192, N % 192 == 0)
    _Float16* wps_dev_ = nullptr;     // hi | lo stage image of the split-precision kernel (compute_type 2)
    bool ok_ = false;
    bool useStream() const {
        static int v = -1;
        if (v < 0) v = ablateEnv("DSVT_LINEAR_STREAM", 1);      // 0: register-staged kernel everywhere (A/B runs)
        return v == 1 && useF16() && c_.K == KS && c_.N % BN == 0;
    }
    bool useF16() const { return c_.compute_type == 1 && c_.K % KS == 0; }
    DsvtLinearPlugin(const LinCfg& c, const float* w, const float* b, const float* g, const float* be,
                     const float* pe_w = nullptr, const float* pe_b = nullptr)
        : c_(c), w_(w, w + (size_t)c.N * c.K) {
        if (pe_w && pe_b) {                      // torch Linear(2 -> K).weight is [K][2]
            pe_.resize(3 * (size_t)c.K);
            for (int k = 0; k < c.K; ++k) { pe_[k] = pe_w[2 * k]; pe_[c.K + k] = pe_w[2 * k + 1]; pe_[2 * c.K + k] = pe_b[k]; }
        }
        if (b) b_.assign(b, b + c.N);
        if (c.n_ln) { g_.assign(g, g + (size_t)c.n_ln * c.N); be_.assign(be, be + (size_t)c.n_ln * c.N); }
        auto up = [](const std::vector<float>& h, float** d) {
            if (h.empty()) { *d = nullptr; return true; }
            if (dsvtMalloc(d, sizeof(float) * h.size()) != hipSuccess) return false;
            return hipMemcpy(*d, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice) == hipSuccess;
        };
        ok_ = up(w_, &w_dev_) && up(b_, &b_dev_) && up(g_, &g_dev_) && up(be_, &be_dev_) && up(pe_, &pe_dev_);
        if (ok_ && useF16()) {
            std::vector<_Float16> wh(w_.size());
            for (size_t i = 0; i < w_.size(); ++i) wh[i] = (_Float16)w_[i];
            ok_ = dsvtMalloc(&wh_dev_, sizeof(_Float16) * wh.size()) == hipSuccess &&
                  hipMemcpy(wh_dev_, wh.data(), sizeof(_Float16) * wh.size(), hipMemcpyHostToDevice) == hipSuccess;
        }
        if (ok_ && c_.compute_type == 2) {
            const std::vector<_Float16> wp = packStagesSplit(w_.data(), c_.N);
            ok_ = dsvtMalloc(&wps_dev_, sizeof(_Float16) * wp.size()) == hipSuccess &&
                  hipMemcpy(wps_dev_, wp.data(), sizeof(_Float16) * wp.size(), hipMemcpyHostToDevice) == hipSuccess;
        }
        if (ok_ && useStream()) {
            const std::vector<_Float16> wp = packStages(w_.data(), c_.N);
            ok_ = dsvtMalloc(&wp_dev_, sizeof(_Float16) * wp.size()) == hipSuccess &&
                  hipMemcpy(wp_dev_, wp.data(), sizeof(_Float16) * wp.size(), hipMemcpyHostToDevice) == hipSuccess;
        }
    }
    ~DsvtLinearPlugin() override {
        for (float* p : {w_dev_, b_dev_, g_dev_, be_dev_, pe_dev_}) if (p) (void)dsvtFree(p);
        if (wh_dev_) (void)dsvtFree(wh_dev_);
        if (wp_dev_) (void)dsvtFree(wp_dev_);
        if (wps_dev_) (void)dsvtFree(wps_dev_);
    }
    const char* type() const override { return "DsvtLinearPlugin"; }
    int nbOutputs() const override { return c_.output_mode == OUT_BOTH ? 2 : 1; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i < 0 || i >= nbOutputs()) return -1;
        *out = dims3(in[0].d[0], c_.max_rows, c_.N); return 0;
    }
    int outputType(int i, const int32_t*, int) const override {
        if (c_.output_mode == OUT_F16) return DSVT_HALF;
        return i == 0 ? DSVT_FLOAT : DSVT_HALF;
    }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int nbIn, int) const override {
        if (io[pos].format != DSVT_FORMAT_LINEAR) return false;
        if (pos == 1) return io[pos].type == DSVT_INT32;
        if (pos == 0 && !pe_.empty()) return io[pos].type == DSVT_FLOAT;        // the [rows, 2] xy tensor
        if (pos == 0 || (pos == 2 && c_.add_cols > 0)) return io[pos].type == (c_.input_half ? DSVT_HALF : DSVT_FLOAT);
        if (pos == 3 && c_.a2_gather_wx > 0) return io[pos].type == DSVT_INT32;
        if (pos < nbIn) return io[pos].type == DSVT_FLOAT;                      // residuals
        return io[pos].type == outputType(pos - nbIn, nullptr, 0);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    bool sharedInput(int index) const override { return c_.a2_gather_wx > 0 && index == 2; }      // the [cells, K] position table
    int enqueue(const DsvtPluginTensorDesc*, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        if (!ok_) return static_cast<int>(hipErrorOutOfMemory);
        LinearArgs a{};
        int idx = 0;
        a.A = in[idx++];
        if (!pe_.empty()) {
            a.pe_xy = static_cast<const float*>(a.A); a.A = nullptr;
            a.pe_w0 = pe_dev_; a.pe_w1 = pe_dev_ + c_.K; a.pe_b = pe_dev_ + 2 * (size_t)c_.K;
        }
        a.count = static_cast<const uint32_t*>(in[idx++]);
        a.A2 = c_.add_cols > 0 ? in[idx++] : nullptr;
        if (c_.a2_gather_wx > 0) { a.a2_c2d = static_cast<const int32_t*>(in[idx++]); a.a2_wx = c_.a2_gather_wx; a.a2_wy = c_.a2_gather_wy; }
        for (int s = 0; s < c_.n_ln; ++s) {
            a.res[s] = static_cast<const float*>(in[idx++]);
            a.gamma[s] = g_dev_ + (size_t)s * c_.N; a.beta[s] = be_dev_ + (size_t)s * c_.N;
        }
        a.W = w_dev_; a.bias = b_dev_;
        a.out = c_.output_mode == OUT_F16 ? nullptr : static_cast<float*>(out[0]);
        a.out16 = c_.output_mode == OUT_F16 ? static_cast<_Float16*>(out[0]) : c_.output_mode == OUT_BOTH ? static_cast<_Float16*>(out[1]) : nullptr;
        a.row_mult = c_.row_mult; a.max_rows = c_.max_rows; a.K = c_.K; a.N = c_.N; a.add_cols = c_.add_cols; a.act = c_.act;
        a.n_ln = c_.n_ln; a.eps = c_.eps; a.out_ld = c_.N; a.a_half = c_.input_half;
        if (zeroFill) {
            if (a.out) DSVT_CHECK(hipMemsetAsync(a.out, 0, sizeof(float) * (size_t)c_.max_rows * c_.N, stream));
            if (a.out16) DSVT_CHECK(hipMemsetAsync(a.out16, 0, sizeof(_Float16) * (size_t)c_.max_rows * c_.N, stream));
        }
        if (wps_dev_) return launchLinearSplit(a, wps_dev_, stream);
        if (c_.a2_gather_wx > 0 && !wp_dev_) return -4;           // the table gather lives in the streamed kernels only
        if (wp_dev_) {
            static unsigned long long* tr = nullptr; static int tron = -1;
            if (tron < 0) { tron = ablateEnv("DSVT_LINEAR_TRACE", 0) ? 1 : 0; if (tron) (void)hipMallocManaged(&tr, 8 * 8 * 4096); }
            a.trace = tr;
            const int rc = launchLinearF16Stream(a, wp_dev_, stream);
            if (tron) {
                (void)hipStreamSynchronize(stream);
                fprintf(stderr, "[linear trace N=%d pe=%d] wg0:", c_.N, (int)!pe_.empty());
                for (int i = 1; i < 8; ++i) fprintf(stderr, " %lld", (long long)(tr[i] - tr[0]));
                fprintf(stderr, " | wg200:"); for (int i = 1; i < 8; ++i) fprintf(stderr, " %lld", (long long)(tr[200 * 8 + i] - tr[200 * 8]));
                fprintf(stderr, " | start skew wg200-wg0 %lld\n", (long long)(tr[200 * 8] - tr[0]));
            }
            return rc;
        }
        return useF16() ? launchLinearF16(a, wh_dev_, stream) : launchLinearF32(a, stream);
    }
    size_t serializationSize() const override {
        return 13 * sizeof(int) + sizeof(float) + sizeof(float) * (w_.size() + b_.size() + g_.size() + be_.size() + pe_.size()) + (c_.a2_gather_wy > 0 ? sizeof(int) : 0);
    }
    void serialize(void* buf) const override {
        char* d = static_cast<char*>(buf);
        wr<int>(d, c_.max_rows); wr<int>(d, c_.K); wr<int>(d, c_.N); wr<int>(d, c_.row_mult); wr<int>(d, c_.act); wr<int>(d, c_.add_cols);
        wr<int>(d, c_.n_ln); wr<float>(d, c_.eps); wr<int>(d, b_.empty() ? 0 : 1); wr<int>(d, c_.compute_type);
        wr<int>(d, c_.input_half); wr<int>(d, c_.output_mode); wr<int>(d, pe_.empty() ? 0 : 1); wr<int>(d, c_.a2_gather_wx);
        for (const std::vector<float>* v : {&w_, &b_, &g_, &be_, &pe_}) { memcpy(d, v->data(), sizeof(float) * v->size()); d += sizeof(float) * v->size(); }
        if (c_.a2_gather_wy > 0) wr<int>(d, c_.a2_gather_wy);       // (trailing, only when set: older blobs stay valid)
    }
    Plugin* clone() const override {
        DsvtLinearPlugin* p = new DsvtLinearPlugin(c_, w_.data(), b_.empty() ? nullptr : b_.data(), g_.data(), be_.data());
        if (!pe_.empty()) { p->pe_ = pe_; p->ok_ = p->ok_ && dsvtMalloc(&p->pe_dev_, sizeof(float) * pe_.size()) == hipSuccess &&
                            hipMemcpy(p->pe_dev_, pe_.data(), sizeof(float) * pe_.size(), hipMemcpyHostToDevice) == hipSuccess; }
        return p;
    }
};

static Plugin* linNew(const LinCfg& c, const float* w, const float* b, const float* g, const float* be,
                      const float* pe_w = nullptr, const float* pe_b = nullptr) {
    if ((pe_w || pe_b) && !(pe_w && pe_b && c.compute_type == 1 && c.K % KS == 0 && c.add_cols == 0 && !c.input_half)) return nullptr;
    if (c.max_rows <= 0 || c.K <= 0 || c.N <= 0 || c.N % 4 != 0 || c.row_mult <= 0 || !w) return nullptr;
    if (c.compute_type < 0 || c.compute_type > 2 || c.output_mode < 0 || c.output_mode > 2) return nullptr;
    // split precision: the whole-row kernel only (K = 192, 96-column chunks, fp32 in / out, bias; the other epilogues live in DsvtEncoderMlpPlugin)
    if (c.compute_type == 2 && !(c.K == KS && c.N % (2 * SP_COLS) == 0 && c.N <= 1024 && !c.input_half && c.output_mode == OUT_F32 && c.n_ln == 0 &&
                                 c.act == ACT_NONE && c.row_mult == 1 && c.add_cols % (2 * SP_COLS) == 0 && !pe_w)) return nullptr;
    if (c.act < 0 || c.act > 2 || c.n_ln < 0 || c.n_ln > 3) return nullptr;
    if (c.n_ln > 0 && (c.N > BN || !g || !be)) return nullptr;             // a LayerNorm row must fit one tile
    if (c.add_cols < 0 || c.add_cols > c.N || (c.add_cols % BN != 0 && c.add_cols != c.N)) return nullptr;
    if (c.input_half && !(c.compute_type == 1 && c.K % KS == 0)) return nullptr;      // fp16 inputs only on the fp16 kernel
    if (c.a2_gather_wy < 0 || (c.a2_gather_wy > 0 && c.a2_gather_wx <= 0)) return nullptr;
    if (c.a2_gather_wx < 0 || (c.a2_gather_wx > 0 && !(c.add_cols > 0 && (c.input_half || c.compute_type == 2) && c.K == KS))) return nullptr;   // table gather: streamed fp16 / split kernels
    return new DsvtLinearPlugin(c, w, b, g, be, pe_w, pe_b);
}
static Plugin* linCreate(const DsvtPluginFieldCollection* fc) {
    const DsvtPluginField* w = findField(fc, "weight"); const DsvtPluginField* b = findField(fc, "bias");
    const DsvtPluginField* g = findField(fc, "ln_weights"); const DsvtPluginField* be = findField(fc, "ln_bias");
    LinCfg c{};
    c.max_rows = fieldInt(fc, "max_rows"); c.K = fieldInt(fc, "in_features"); c.N = fieldInt(fc, "out_features");
    c.row_mult = fieldInt(fc, "row_mult", 1); c.act = fieldInt(fc, "activation"); c.add_cols = fieldInt(fc, "add_cols");
    c.n_ln = fieldInt(fc, "num_layer_norms"); c.eps = fieldFloat(fc, "ln_eps", 0.f); c.compute_type = fieldInt(fc, "compute_type", 0);
    c.input_half = fieldInt(fc, "input_half", 0); c.output_mode = fieldInt(fc, "output_mode", 0);
    c.a2_gather_wx = fieldInt(fc, "add_gather_width", 0); c.a2_gather_wy = fieldInt(fc, "add_gather_height", 0);
    if (!w || !w->data || c.K <= 0 || c.N <= 0 || w->length != c.K * c.N) return nullptr;
    if (b && b->data && b->length != c.N) return nullptr;
    if (c.n_ln > 0 && (!g || !be || g->length != c.n_ln * c.N || be->length != c.n_ln * c.N)) return nullptr;
    const DsvtPluginField* pw = findField(fc, "pe_weight"); const DsvtPluginField* pb = findField(fc, "pe_bias");
    if ((pw && pw->data && pw->length != 2 * c.K) || (pb && pb->data && pb->length != c.K)) return nullptr;
    return linNew(c, static_cast<const float*>(w->data), (b && b->data) ? static_cast<const float*>(b->data) : nullptr,
                  g ? static_cast<const float*>(g->data) : nullptr, be ? static_cast<const float*>(be->data) : nullptr,
                  (pw && pw->data) ? static_cast<const float*>(pw->data) : nullptr, (pb && pb->data) ? static_cast<const float*>(pb->data) : nullptr);
}
static Plugin* linDeser(const void* data, size_t len) {
    if (len < 13 * sizeof(int) + sizeof(float)) return nullptr;
    const char* d = static_cast<const char*>(data);
    LinCfg c{};
    c.max_rows = rd<int>(d); c.K = rd<int>(d); c.N = rd<int>(d); c.row_mult = rd<int>(d); c.act = rd<int>(d); c.add_cols = rd<int>(d);
    c.n_ln = rd<int>(d); c.eps = rd<float>(d); int has_b = rd<int>(d); c.compute_type = rd<int>(d);
    c.input_half = rd<int>(d); c.output_mode = rd<int>(d); int has_pe = rd<int>(d); c.a2_gather_wx = rd<int>(d);
    if (c.K <= 0 || c.N <= 0 || c.n_ln < 0 || c.n_ln > 3) return nullptr;
    size_t need = (size_t)c.K * c.N + (has_b ? c.N : 0) + 2 * (size_t)c.n_ln * c.N + (has_pe ? 3 * (size_t)c.K : 0);
    if (len < 13 * sizeof(int) + sizeof(float) + need * sizeof(float)) return nullptr;
    std::vector<float> all(need);
    memcpy(all.data(), d, need * sizeof(float));
    const float* w = all.data(); const float* b = has_b ? w + (size_t)c.K * c.N : nullptr;
    const float* g = w + (size_t)c.K * c.N + (has_b ? c.N : 0); const float* be = g + (size_t)c.n_ln * c.N;
    {
        const size_t used = 13 * sizeof(int) + sizeof(float) + need * sizeof(float);
        const int extra = trailingInts(len, used, 1);
        if (extra < 0) return nullptr;
        if (extra >= 1) { const char* t = static_cast<const char*>(data) + used; c.a2_gather_wy = rd<int>(t); }
    }
    if (has_pe) {                                  // stored as [w0 | w1 | b]; rebuild the [K][2] weight the constructor expects
        const float* pe = be + (size_t)c.n_ln * c.N;
        std::vector<float> pw(2 * (size_t)c.K);
        for (int k = 0; k < c.K; ++k) { pw[2 * k] = pe[k]; pw[2 * k + 1] = pe[c.K + k]; }
        return linNew(c, w, b, g, be, pw.data(), pe + 2 * (size_t)c.K);
    }
    return linNew(c, w, b, g, be);
}
static Creator g_linCreator{"DsvtLinearPlugin",
    {{"max_rows", DSVT_FIELD_INT32}, {"in_features", DSVT_FIELD_INT32}, {"out_features", DSVT_FIELD_INT32},
     {"row_mult", DSVT_FIELD_INT32}, {"activation", DSVT_FIELD_INT32}, {"add_cols", DSVT_FIELD_INT32},
     {"num_layer_norms", DSVT_FIELD_INT32}, {"ln_eps", DSVT_FIELD_FLOAT32}, {"compute_type", DSVT_FIELD_INT32},
     {"input_half", DSVT_FIELD_INT32}, {"output_mode", DSVT_FIELD_INT32}, {"add_gather_width", DSVT_FIELD_INT32}, {"add_gather_height", DSVT_FIELD_INT32},
     {"weight", DSVT_FIELD_FLOAT32}, {"bias", DSVT_FIELD_FLOAT32}, {"ln_weights", DSVT_FIELD_FLOAT32}, {"ln_bias", DSVT_FIELD_FLOAT32},
     {"pe_weight", DSVT_FIELD_FLOAT32}, {"pe_bias", DSVT_FIELD_FLOAT32}},
    linCreate, linDeser, {}, {}};
static Registrar g_linReg(&g_linCreator);

// -------------------------------------------------------------------------------------
// DsvtPosEmbedPlugin: all position-embedding MLPs of a frame in one launch.
// fields: max_rows, num_layers, layer_input (int[num_layers]: which xy input each layer reads), pe_weight [L][192][2], pe_bias [L][192],
//         weight [L][192][192], bias [L][192].  Inputs: count [1], then the xy tensors [1,rows,2] f32.  Outputs: L tensors [1,rows,192] fp16.
class DsvtPosEmbedPlugin : public Plugin {
public:
    int max_rows_, L_;
    std::vector<int> src_;
    std::vector<float> pw_, pb_, w_, b_;
    float* pe_dev_ = nullptr; float* b_dev_ = nullptr; _Float16* wp_dev_ = nullptr;
    bool ok_ = false;
    DsvtPosEmbedPlugin(int max_rows, int L, const int* src, const float* pw, const float* pb, const float* w, const float* b)
        : max_rows_(max_rows), L_(L), src_(src, src + L), pw_(pw, pw + (size_t)L * KS * 2), pb_(pb, pb + (size_t)L * KS),
          w_(w, w + (size_t)L * KS * KS), b_(b, b + (size_t)L * KS) {
        std::vector<float> pe((size_t)L * 3 * KS);               // per layer: w0 | w1 | b
        for (int l = 0; l < L; ++l)
            for (int k = 0; k < KS; ++k) {
                pe[(size_t)l * 3 * KS + k] = pw_[((size_t)l * KS + k) * 2]; pe[(size_t)l * 3 * KS + KS + k] = pw_[((size_t)l * KS + k) * 2 + 1];
                pe[(size_t)l * 3 * KS + 2 * KS + k] = pb_[(size_t)l * KS + k];
            }
        std::vector<_Float16> wp;
        for (int l = 0; l < L; ++l) { const std::vector<_Float16> one = packStages(w_.data() + (size_t)l * KS * KS, KS); wp.insert(wp.end(), one.begin(), one.end()); }
        ok_ = dsvtMalloc(&pe_dev_, sizeof(float) * pe.size()) == hipSuccess && hipMemcpy(pe_dev_, pe.data(), sizeof(float) * pe.size(), hipMemcpyHostToDevice) == hipSuccess &&
              dsvtMalloc(&b_dev_, sizeof(float) * b_.size()) == hipSuccess && hipMemcpy(b_dev_, b_.data(), sizeof(float) * b_.size(), hipMemcpyHostToDevice) == hipSuccess &&
              dsvtMalloc(&wp_dev_, sizeof(_Float16) * wp.size()) == hipSuccess && hipMemcpy(wp_dev_, wp.data(), sizeof(_Float16) * wp.size(), hipMemcpyHostToDevice) == hipSuccess;
    }
    ~DsvtPosEmbedPlugin() override { if (pe_dev_) (void)dsvtFree(pe_dev_); if (b_dev_) (void)dsvtFree(b_dev_); if (wp_dev_) (void)dsvtFree(wp_dev_); }
    int nInputs() const { int m = 0; for (int v : src_) m = v > m ? v : m; return m + 2; }
    const char* type() const override { return "DsvtPosEmbedPlugin"; }
    int nbOutputs() const override { return L_; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i < 0 || i >= L_) return -1;
        *out = dims3(in[1].d[0], max_rows_, KS); return 0;
    }
    int outputType(int, const int32_t*, int) const override { return DSVT_HALF; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int nbIn, int) const override {
        if (io[pos].format != DSVT_FORMAT_LINEAR) return false;
        if (pos == 0) return io[pos].type == DSVT_INT32;
        return io[pos].type == (pos < nbIn ? DSVT_FLOAT : DSVT_HALF);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc*, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*, hipStream_t stream) override {
        if (!ok_) return static_cast<int>(hipErrorOutOfMemory);
        LinearArgs a{};
        a.count = static_cast<const uint32_t*>(in[0]);
        a.row_mult = 1; a.max_rows = max_rows_; a.K = KS; a.N = KS; a.out_ld = KS; a.act = ACT_NONE;
        PosEmbedBatch b{};
        for (int l = 0; l < L_; ++l) {
            PosEmbedLayer& Y = b.layer[l];
            Y.xy = static_cast<const float*>(in[1 + src_[l]]);
            Y.pe_w0 = pe_dev_ + (size_t)l * 3 * KS; Y.pe_w1 = Y.pe_w0 + KS; Y.pe_b = Y.pe_w0 + 2 * KS;
            Y.Wp = wp_dev_ + (size_t)l * KS * KS; Y.bias = b_dev_ + (size_t)l * KS; Y.out16 = static_cast<_Float16*>(out[l]);
            if (zeroFill) DSVT_CHECK(hipMemsetAsync(out[l], 0, sizeof(_Float16) * (size_t)max_rows_ * KS, stream));
        }
        hipLaunchKernelGGL(posembed_batched_kernel, dim3(cdiv(max_rows_, BM16), 1, L_), dim3(512), 0, stream, a, b);
        return lastError();
    }
    size_t serializationSize() const override { return (2 + L_) * sizeof(int) + sizeof(float) * (pw_.size() + pb_.size() + w_.size() + b_.size()); }
    void serialize(void* buf) const override {
        char* d = static_cast<char*>(buf);
        wr<int>(d, max_rows_); wr<int>(d, L_);
        for (int v : src_) wr<int>(d, v);
        for (const std::vector<float>* v : {&pw_, &pb_, &w_, &b_}) { memcpy(d, v->data(), sizeof(float) * v->size()); d += sizeof(float) * v->size(); }
    }
    Plugin* clone() const override { return new DsvtPosEmbedPlugin(max_rows_, L_, src_.data(), pw_.data(), pb_.data(), w_.data(), b_.data()); }
};
static Plugin* peNew(int max_rows, int L, const int* src, const float* pw, const float* pb, const float* w, const float* b) {
    if (max_rows <= 0 || L <= 0 || L > kMaxPosLayers) return nullptr;
    for (int l = 0; l < L; ++l) if (src[l] < 0 || src[l] > 7) return nullptr;
    return new DsvtPosEmbedPlugin(max_rows, L, src, pw, pb, w, b);
}
static Plugin* peCreate(const DsvtPluginFieldCollection* fc) {
    const int max_rows = fieldInt(fc, "max_rows"), L = fieldInt(fc, "num_layers");
    const DsvtPluginField* src = findField(fc, "layer_input"); const DsvtPluginField* pw = findField(fc, "pe_weight");
    const DsvtPluginField* pb = findField(fc, "pe_bias"); const DsvtPluginField* w = findField(fc, "weight"); const DsvtPluginField* b = findField(fc, "bias");
    if (L <= 0 || !src || !src->data || src->length != L || !pw || !pw->data || pw->length != L * KS * 2 || !pb || !pb->data || pb->length != L * KS ||
        !w || !w->data || w->length != L * KS * KS || !b || !b->data || b->length != L * KS) return nullptr;
    return peNew(max_rows, L, static_cast<const int*>(src->data), static_cast<const float*>(pw->data), static_cast<const float*>(pb->data),
                 static_cast<const float*>(w->data), static_cast<const float*>(b->data));
}
static Plugin* peDeser(const void* data, size_t len) {
    if (len < 2 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    const int max_rows = rd<int>(d), L = rd<int>(d);
    if (max_rows <= 0 || L <= 0 || L > kMaxPosLayers) return nullptr;
    const size_t nf = (size_t)L * (KS * 2 + KS + KS * KS + KS);
    if (len < (2 + L) * sizeof(int) + nf * sizeof(float)) return nullptr;
    std::vector<int> src(L); for (int l = 0; l < L; ++l) src[l] = rd<int>(d);
    std::vector<float> all(nf); memcpy(all.data(), d, nf * sizeof(float));
    const float* q = all.data();
    return peNew(max_rows, L, src.data(), q, q + (size_t)L * KS * 2, q + (size_t)L * KS * 3, q + (size_t)L * KS * 3 + (size_t)L * KS * KS);
}
static Creator g_peCreator{"DsvtPosEmbedPlugin",
    {{"max_rows", DSVT_FIELD_INT32}, {"num_layers", DSVT_FIELD_INT32}, {"layer_input", DSVT_FIELD_INT32}, {"pe_weight", DSVT_FIELD_FLOAT32},
     {"pe_bias", DSVT_FIELD_FLOAT32}, {"weight", DSVT_FIELD_FLOAT32}, {"bias", DSVT_FIELD_FLOAT32}},
    peCreate, peDeser, {}, {}};
static Registrar g_peReg(&g_peCreator);

}  // namespace dsvt
