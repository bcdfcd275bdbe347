// linear.h -- launch interface of the MFMA linear kernels (linear.hip), shared with attention.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dsvt {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

struct LinearArgs {
    const void* A;        // [rows, K] fp32, or fp16 when a_half
    const void* A2;       // added to A for output columns < add_cols (same dtype as A), or nullptr
    const float* W;       // [N, K] fp32 (fp32-MFMA kernel)
    const float* bias;    // [N] or nullptr
    const float* res[3]; const float* gamma[3]; const float* beta[3];   // chained "add residual, LayerNorm" stages
    float* out;           // [rows, out_ld] fp32, or nullptr
    _Float16* out16;      // [rows, out_ld] fp16 copy of the result, or nullptr
    const uint32_t* count;
    int row_mult, max_rows, K, N, add_cols, act, n_ln;
    int out_ld;           // row stride of out / out16 in elements (>= N); lets Q/K/V land in one [rows, 3C] buffer
    int a_half;           // A / A2 are fp16 (fp16-MFMA kernel only)
    // position-embedding prologue (fp16-MFMA kernel only): when pe_xy != nullptr the operand row is not read from A
    // but computed on the fly, A'[m][k] = relu(pe_xy[m][0] * pe_w0[k] + pe_xy[m][1] * pe_w1[k] + pe_b[k])  -- the first
    // FC + BN + ReLU of fullyConnectedBnLELU_fullyConnected (src/dsvt-ai-trt.cpp:461-492), K = 2
    const float* pe_xy; const float* pe_w0; const float* pe_w1; const float* pe_b;
    // gathered A2 (streamed fp16 kernel only): when a2_c2d != nullptr row m adds A2 row c2d[m][1] * a2_wx + c2d[m][2] (the window
    // cell of voxel m, WindowPartition output 4) instead of A2 row m: A2 is then a per-layer TABLE of position embeddings
    const int32_t* a2_c2d; int a2_wx;
    int a2_wy;            // 3-D windows: the table row is (c2d[m][0] * a2_wy + c2d[m][1]) * a2_wx + c2d[m][2]; 0 for the pillar model (z = 0)
    float eps;
    unsigned long long* trace;    // debugging: per-workgroup phase timestamps (s_memtime) of the streamed kernel, or nullptr
};

// y = epilogue(A' W^T + b) on v_mfma_f32_16x16x4_f32 (fp32 A only); returns 0 or a hipError_t
int launchLinearF32(const LinearArgs& a, hipStream_t stream);
// same on v_mfma_f32_16x16x32_f16 with fp16 weights Wh [N, K]; needs K % 192 == 0
int launchLinearF16(const LinearArgs& a, const _Float16* Wh, hipStream_t stream);

}  // namespace dsvt
