// linear.h -- launch interface of the MFMA linear kernel (linear.hip), shared with attention.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dsvt {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

struct LinearArgs {
    const float* A; const float* A2; const float* W; const float* bias;
    const float* res[3]; const float* gamma[3]; const float* beta[3];
    float* out;
    const uint32_t* count;
    int row_mult, max_rows, K, N, add_cols, act, n_ln;
    int out_ld;          // row stride of `out` in floats (>= N); lets Q/K/V land in one [rows, 3C] buffer
    float eps;
};

// enqueue y = epilogue(A' W^T + b) for up to max_rows rows; returns 0 or a hipError_t
int launchLinearF32(const LinearArgs& a, hipStream_t stream);

}  // namespace dsvt
