// conv.hip -- DsvtConv2dPlugin: NHWC fp16 implicit-GEMM convolution on v_mfma_f32_16x16x32_f16.
//
// SURVEY.md section 8(f)-1, the first row "next" to the hot path: the reference's BEV ResNet,
// deblocks and CenterHead (src/dsvt-ai-trt.cpp:1144-1468: addConvolutionNd / addDeconvolutionNd +
// addScale(BN) + ReLU + ElementWise SUM, helpers :149-246).  One kernel covers all of them:
//   * 3x3 / 1x1 convolution, stride 1 or 2, zero padding                      (convBnLELU, convBn)
//   * folded BatchNorm = per-channel bias in the epilogue                      (addBatchNorm2d :149-180)
//   * residual add + ReLU in the epilogue                                      (:1165-1166 ...)
//   * ConvTranspose with kernel == stride as a 1x1 convolution whose output-channel chunk picks
//     the (dy, dx) sub-pixel it writes (pixel shuffle in the store)            (deconvBnLELU :217-246)
//   * channel offset / stride on the output => the 3-way concat (:1363) is free.
//
// GEMM view: rows = output pixels, columns = output channels, K = taps x Cin.  It is the fp16
// linear kernel (linear.hip) with a gathered A operand: for tap (ky,kx) the B-operand fragment of
// a pixel is 8 consecutive channels of the input pixel at (y*s+ky-p, x*s+kx-p) -- 16 contiguous
// bytes of an NHWC row, loaded straight from global memory (zero outside the image) -- while the
// weights stream through LDS in [128 cout][KC cin] slabs, double buffered, rows padded by 16 halfs
// (conflict-free ds_read_b128, enumerated).  Transposed product tile D[cout][pixel]: a lane owns one
// pixel and four consecutive output channels, so bias / residual / store are vector accesses.
// Workgroup = 4 waves = 128 pixels x 128 output channels; wave = 32 pixels.
#include "plugin_base.h"
#include "device_utils.h"
#include "conv_args.h"
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace dsvt {

// bias / residual / ReLU / store of four consecutive output channels co..co+3 of output pixel opix
template <bool WITH_BIAS = true, bool SPL = false>
__device__ __forceinline__ void convStore(const ConvArgs& a, floatx4 acc, size_t opix, int co)
{
    if (co >= a.Cout) return;
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    const bool full = co + 3 < a.Cout;
    if (WITH_BIAS && a.bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (co + i < a.Cout) v[i] += a.bias[co + i];
    }
    if (a.res && full) {
        const half4 rv = *reinterpret_cast<const half4*>(a.res + opix * a.res_ld + co);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
        if (SPL && a.res_split && a.res_x8) {
            const unsigned w[1] = {*reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(a.res + opix * a.res_ld + 2 * a.res_split) + x8Offset(co))};
            float lo[4]; x8DecodeLo<4>(w, lo);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += lo[i];
        } else if (SPL && a.res_split) {
            const half4 rl = *reinterpret_cast<const half4*>(a.res + opix * a.res_ld + a.res_split + co);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += (float)rl[i];
        }
    }
    if (a.relu) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    if (a.out_f32) {
        float* o = static_cast<float*>(a.out) + opix * a.out_ld + a.out_coff + co;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (co + i < a.Cout) o[i] = v[i];
    } else if (SPL && a.split_out) {
        _Float16* o = static_cast<_Float16*>(a.out) + opix * a.out_ld + a.out_coff + co;
        half4 hi, lo;
        splitPlanes<4>(v, hi, lo);
        unsigned char* xp = reinterpret_cast<unsigned char*>(static_cast<_Float16*>(a.out) + opix * a.out_ld + 2 * a.split_out);
        if (full) {
            *reinterpret_cast<half4*>(o) = hi;
            if (a.x8_out != 2) *reinterpret_cast<half4*>(o + a.split_out) = lo;
            if (a.x8_out == 3) {}
            else if (a.x8_out) x8Store<4>(xp, a.out_coff + co, v, hi);
            else *reinterpret_cast<half4*>(o + 2 * a.split_out) = hi;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (co + i < a.Cout) {
                o[i] = hi[i];
                if (a.x8_out != 2) o[a.split_out + i] = lo[i];
                if (a.x8_out == 3) {}
                else if (a.x8_out) {
                    const int xo = x8Offset(a.out_coff + co + i);
                    xp[xo] = (unsigned char)packE4m3((v[i] - (float)hi[i]) * 2048.f, 0.f, 0.f, 0.f); xp[xo + 16] = (unsigned char)packE4m3(v[i], 0.f, 0.f, 0.f);
                } else o[2 * a.split_out + i] = hi[i];
            }
        }
    } else {
        _Float16* o = static_cast<_Float16*>(a.out) + opix * a.out_ld + a.out_coff + co;
        if (full) {
            half4 h; h[0] = (_Float16)v[0]; h[1] = (_Float16)v[1]; h[2] = (_Float16)v[2]; h[3] = (_Float16)v[3];
            *reinterpret_cast<half4*>(o) = h;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (co + i < a.Cout) o[i] = (_Float16)v[i];
        }
    }
}

// MT = 16-pixel MFMA tiles per wave, NW = waves per workgroup (128 pixels per workgroup either way)
template <int KC, int MT, int NW, bool SPL = false>
__global__ void __launch_bounds__(64 * NW, (MT == 1 ? 4 : 2))
conv_f16_kernel(ConvArgs a_)
{
    ConvArgs a = a_;
    {   // image blockIdx.z of a stack of a.nb images
        const size_t b = blockIdx.z, opx = (size_t)a.Ho * a.up * a.Wo * a.up;
        a.in += b * (size_t)a.H * a.W * a.Cin;
        if (a.res) a.res += b * opx * a.res_ld;
        a.out = a.out_f32 ? static_cast<void*>(static_cast<float*>(a.out) + b * opx * a.out_ld) : static_cast<void*>(static_cast<_Float16*>(a.out) + b * opx * a.out_ld);
    }
    constexpr int NTHR = 64 * NW;
    constexpr int KSTEPS = KC / 32, LDW = KC + 16, WPT = (CNB * KC / 8 + NTHR - 1) / NTHR;     // WPT: uint4 weight loads per thread per slab
    __shared__ __attribute__((aligned(16))) _Float16 sW[2][CNB * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int npix = a.Ho * a.Wo;
    const int n0 = blockIdx.y * CNB;
    const int nvalid = a.CoutRows - n0 < CNB ? a.CoutRows - n0 : CNB;
    const int ntiles = (nvalid + 15) / 16;
    const int Ktot = a.KH * a.KW * a.Cin;
    const int nck = a.Cin / KC, NS = a.KH * a.KW * nck;

    // XCD-aware tile order: the dispatcher places workgroup b on XCD b % 8 (observed, speed only); giving every
    // XCD a contiguous run of pixel tiles keeps the three input rows a 3x3 tap window re-reads in that XCD's L2
    const int per_xcd = gridDim.x / 8;                    // grid.x is rounded up to a multiple of 8 by the host
    const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (tile * CPX >= npix) return;
    int py[MT], px[MT]; bool pv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int p = tile * CPX + wave * 16 * MT + mt * 16 + r;
        pv[mt] = p < npix;
        p = pv[mt] ? p : npix - 1;
        py[mt] = p / a.Wo; px[mt] = p % a.Wo;
    }

    floatx4 acc[MT][CNT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < CNT; ++t) acc[mt][t] = floatx4{0.f, 0.f, 0.f, 0.f};

    auto loadW = [&](int s, uint4 (&wr)[WPT]) {
        const int tap = s / nck, cc = s - tap * nck;
        const _Float16* base = a.wt + (size_t)n0 * Ktot + (size_t)tap * a.Cin + cc * KC;
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int i = tid + j * NTHR, n = i / (KC / 8), c = i % (KC / 8);
            wr[j] = n < nvalid ? *reinterpret_cast<const uint4*>(base + (size_t)n * Ktot + c * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto storeW = [&](int buf, const uint4 (&wr)[WPT]) {
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int i = tid + j * NTHR, n = i / (KC / 8), c = i % (KC / 8);
            if (n < CNB) *reinterpret_cast<uint4*>(&sW[buf][n * LDW + c * 8]) = wr[j];
        }
    };
    auto loadA = [&](int s, half8 (&af)[MT][KSTEPS]) {
        const int tap = s / nck, cc = s - tap * nck;
        const int ky = tap / a.KW, kx = tap - ky * a.KW;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int yi = py[mt] * a.stride + ky - a.pad, xi = px[mt] * a.stride + kx - a.pad;
            const bool inb = yi >= 0 && yi < a.H && xi >= 0 && xi < a.W;
            const int c0 = (a.alias3 && cc * KC >= a.alias3) ? cc * KC - a.alias3 : cc * KC;     // (the phases of an x8 third plane read plane 0)
            const _Float16* src = a.in + ((size_t)(inb ? yi : 0) * a.W + (inb ? xi : 0)) * a.Cin + c0 + g * 8;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                half8 v = *reinterpret_cast<const half8*>(src + ks * 32);
                if (!inb) v = half8{0, 0, 0, 0, 0, 0, 0, 0};
                af[mt][ks] = v;
            }
        }
    };

    uint4 wr[WPT];
    half8 cur[MT][KSTEPS], nxt[MT][KSTEPS];
    // three fp16 products over an x8 third plane whose planes are ONE slab wide (the two stride-2 layers: C = KC = 128): the slab of the third plane
    // gathers the very rows the tap's first slab gathered, so those fragments are kept (16 registers) instead of fetched again -- a third of the
    // gathers of a kernel that waits for its gathers; the slab order, i.e. every sum, is unchanged
    const bool keepHi = a.alias3 == 2 * KC && nck == 3;
    half8 hiKeep[MT][KSTEPS];
    loadW(0, wr);
    loadA(0, cur);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) hiKeep[mt][ks] = cur[mt][ks];
    storeW(0, wr);
    __syncthreads();
    for (int s = 0; s < NS; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < NS;
        if (more) {                                                  // next slab's traffic is in flight during the MFMAs
            loadW(s + 1, wr);
            if (keepHi && (s + 1) % 3 == 2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int ks = 0; ks < KSTEPS; ++ks) nxt[mt][ks] = hiKeep[mt][ks];
            } else loadA(s + 1, nxt);
        }
        const _Float16* pw = &sW[buf][r * LDW + g * 8];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int t = 0; t < CNT; ++t)
                if (t < ntiles) {
                    const half8 wf = *reinterpret_cast<const half8*>(pw + t * 16 * LDW + ks * 32);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, cur[mt][ks], acc[mt][t], 0, 0, 0);
                }
        if (more) {
            storeW(buf ^ 1, wr);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    cur[mt][ks] = nxt[mt][ks];
                    if (keepHi && (s + 1) % 3 == 0) hiKeep[mt][ks] = nxt[mt][ks];
                }
        }
        __syncthreads();
    }

    // ---- epilogue: lane = pixel (mt, r), output rows n0 + 16t + 4g + i of the weight matrix -------
    const int sub = n0 / a.Cout;                         // (dy, dx) chunk of a pixel-shuffle deconvolution; 0 otherwise
    const int dy = sub / a.up, dx = sub - dy * a.up;
    const int cbase = n0 - sub * a.Cout;
    const int Wout = a.Wo * a.up;
    // the bias of every tile first (loads only): a load issued after a store is awaited behind the store's acknowledgement
    if (a.bias) {
#pragma unroll
        for (int t = 0; t < CNT; ++t) {
            const int co = cbase + t * 16 + 4 * g;
            if (t >= ntiles || co >= a.Cout) continue;
            float b4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) b4[i] = co + i < a.Cout ? a.bias[co + i] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[mt][t][i] += b4[i];
        }
    }
    if (a.wide && !a.res) {
        // wide stores (see the halo kernels): v_permlane16_swap of tiles t, t + 1 leaves eight consecutive channels per lane
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const size_t opix = pv[mt] ? (size_t)(py[mt] * a.up + dy) * Wout + (px[mt] * a.up + dx) : 0;
#pragma unroll
            for (int t = 0; t < CNT; t += 2) {
                floatx4 X = acc[mt][t], Y = acc[mt][t + 1];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(X[i]), __float_as_uint(Y[i]), false, false);
                    X[i] = __uint_as_float(sw[0]); Y[i] = __uint_as_float(sw[1]);
                }
                const int co = cbase + t * 16 + (g & 1) * 16 + (g >> 1) * 8;
                if (!pv[mt] || t >= ntiles || co >= a.Cout) continue;
                if constexpr (SPL) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { v[i] = a.relu ? fmaxf(X[i], 0.f) : X[i]; v[4 + i] = a.relu ? fmaxf(Y[i], 0.f) : Y[i]; }
                    storeHalf8<true>(a, v, opix, co);
                    continue;
                }
                half8 h;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    h[i] = (_Float16)(a.relu ? fmaxf(X[i], 0.f) : X[i]);
                    h[4 + i] = (_Float16)(a.relu ? fmaxf(Y[i], 0.f) : Y[i]);
                }
                *reinterpret_cast<half8*>(static_cast<_Float16*>(a.out) + opix * a.out_ld + a.out_coff + co) = h;
            }
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (!pv[mt]) continue;
        const size_t opix = (size_t)(py[mt] * a.up + dy) * Wout + (px[mt] * a.up + dx);
#pragma unroll
        for (int t = 0; t < CNT; ++t) {
            const int co = cbase + t * 16 + 4 * g;
            if (t >= ntiles || co >= a.Cout) continue;
            if constexpr (SPL) { convStore<false, true>(a, acc[mt][t], opix, co); continue; }      // (the bias is already in the accumulators)
            float v[4] = {acc[mt][t][0], acc[mt][t][1], acc[mt][t][2], acc[mt][t][3]};
            const bool full = co + 3 < a.Cout;
            if (a.res && full) {
                const half4 rv = *reinterpret_cast<const half4*>(a.res + opix * a.res_ld + co);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
            }
            if (a.relu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
            }
            if (a.out_f32) {
                float* o = static_cast<float*>(a.out) + opix * a.out_ld + a.out_coff + co;
#pragma unroll
                for (int i = 0; i < 4; ++i) if (co + i < a.Cout) o[i] = v[i];
            } else {
                _Float16* o = static_cast<_Float16*>(a.out) + opix * a.out_ld + a.out_coff + co;
                if (full) {
                    half4 h; h[0] = (_Float16)v[0]; h[1] = (_Float16)v[1]; h[2] = (_Float16)v[2]; h[3] = (_Float16)v[3];
                    *reinterpret_cast<half4*>(o) = h;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (co + i < a.Cout) o[i] = (_Float16)v[i];
                }
            }
        }
    }
}


// -------------------------------------------------------------------------------------
// Halo-tile kernel for the stride-1 layers (every 3x3 / 1x1 convolution of the BEV backbone and the
// CenterHead except the two stride-2 entries): the re-read factor of the kernel above (each input
// pixel fetched once per tap, each weight slab once per 128 pixels -- ~1 GB of L2->CU traffic for a
// 468x468x128 layer, waves parked in s_waitcnt 2/3 of the time) is what bounds it, not MFMA or LDS.
//
//   * workgroup = TH rows x 32 columns of output pixels x 128 output channels, TH waves; wave
//     (pg, cg) owns rows 2pg, 2pg+1 (four 16-pixel MFMA tiles) x 64 channels (four 16-channel
//     tiles): a 64 x 64 register tile, 16 MFMAs per 8 ds_read_b128.
//   * the (TH+2) x 34 x 64-channel input halo sits in LDS, 128 B per pixel, 16-byte chunk c of
//     pixel column hx stored at slot c ^ (hx & 7): the B fragment of tap (ky, kx) is one
//     conflict-free ds_read_b128 per 16-pixel tile; the row stride (40 pixels) is 0 mod 8 so the
//     swizzle term depends on the column only.  Each input pixel is fetched ~1.3x, not 9x.
//   * weights are packed by the host in MFMA-fragment order [k-step][16-channel tile][lane][8 halfs]:
//     a 64-wide K slab (one tap, one 64-channel chunk) is a linear 16 KB copy into LDS and an A
//     fragment is a lane-linear ds_read_b128.
//   * persistent grid (one workgroup per CU): a workgroup walks items = (pixel tile, 128-channel
//     chunk); the next halo (next 64-channel chunk, or the next item's first) and the next weight
//     slab are requested before the MFMAs of the current one and written to the other LDS buffer
//     after them -- one barrier per slab, no exposed prologue between items.
constexpr int HTW = 32;           // tile width in pixels
constexpr int HHS = 40;           // LDS halo row stride in pixels (>= HTW + 2, multiple of 8)

// Wide epilogue of one 16-pixel tile x two 16-channel tiles (t0, t0 + 1) of a wave: v_permlane16_swap exchanges
// the odd 16-lane rows of X with the even rows of Y, after which lane (r, g) holds EIGHT consecutive output
// channels of pixel r (first channel t0*16 + (g&1)*16 + (g>>1)*8): bias / residual / store are 16-byte accesses
// and the four lanes of a pixel cover a contiguous 64-byte segment.
template <bool WITH_BIAS = true, bool SPL = false>
__device__ __forceinline__ void convStoreWide(const ConvArgs& a, floatx4 X, floatx4 Y, bool valid, size_t opix, int co0, int g)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(X[i]), __float_as_uint(Y[i]), false, false);
        X[i] = __uint_as_float(sw[0]); Y[i] = __uint_as_float(sw[1]);
    }
    const int co = co0 + (g & 1) * 16 + (g >> 1) * 8;
    if (!valid || co >= a.Cout) return;
    float v[8] = {X[0], X[1], X[2], X[3], Y[0], Y[1], Y[2], Y[3]};
    if (WITH_BIAS && a.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(a.bias + co), b1 = *reinterpret_cast<const float4*>(a.bias + co + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if (a.res) {
        const half8 rv = *reinterpret_cast<const half8*>(a.res + opix * a.res_ld + co);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += (float)rv[i];
        if (SPL && a.res_split && a.res_x8) {
            const uint2 q = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(a.res + opix * a.res_ld + 2 * a.res_split) + x8Offset(co));
            const unsigned w[2] = {q.x, q.y};
            float lo[8]; x8DecodeLo<8>(w, lo);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += lo[i];
        } else if (SPL && a.res_split) {
            const half8 rl = *reinterpret_cast<const half8*>(a.res + opix * a.res_ld + a.res_split + co);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += (float)rl[i];
        }
    }
    if constexpr (SPL) {
        if (a.relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        storeHalf8<true>(a, v, opix, co);
        return;
    }
    half8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (_Float16)(a.relu ? fmaxf(v[i], 0.f) : v[i]);
    *reinterpret_cast<half8*>(static_cast<_Float16*>(a.out) + opix * a.out_ld + a.out_coff + co) = h;
}


// ---- the scaled MFMA of the fp16 + fp8 K loops (conv_wide_kernel<.., MX>, conv_halo_kernel<.., MX>): see the comment above conv_wide_kernel ----
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx8 __attribute__((ext_vector_type(8)));
// tile engine of the MX instantiations (tools/ab_conv.sh builds variants from -D flags; A/B numbers in profiles/README.md, round 4): four-step
// weight slabs (half the barriers: 7.70 vs 7.87 ms over the stage's layers), two slab buffers, four A fragments per fp16 batch, one per fp8 batch
#ifndef WIDE_TRICKLE4_NUM
#define WIDE_TRICKLE4_NUM 4
#endif
#ifndef WIDE_TRICKLE_CT4
#define WIDE_TRICKLE_CT4 1
#endif
#ifndef WIDE_TRICKLE1
#define WIDE_TRICKLE1 1
#endif
#ifndef WIDE_TRICKLE1_NUM
#define WIDE_TRICKLE1_NUM 4
#endif
#ifndef WIDE_SPS4_DEFAULT
#define WIDE_SPS4_DEFAULT 1
#endif
#ifndef MX_TRICKLE_MAIN
#define MX_TRICKLE_MAIN 1
#endif
#ifndef MX_TRICKLE_CROSS
#define MX_TRICKLE_CROSS 0
#endif
#ifndef MX_NWB
#define MX_NWB 2
#endif
#ifndef MX_NWB4
#define MX_NWB4 2
#endif
#ifndef MX_CH
#define MX_CH 4
#endif
#ifndef WIDE_PFB
#define WIDE_PFB 1
#endif
#ifndef MX_SPS
#define MX_SPS 4
#endif
#ifndef MX_CHX
#define MX_CHX 1
#endif
template <int OP>
__device__ __forceinline__ floatx4 mfmaX8(const intx8& A, const intx8& B, const floatx4& c, int scaleA) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 0, 0, OP, scaleA, 0, 127);          // fp8 x fp8; byte OP of scaleA; scale_b = 2^0
}
__device__ __forceinline__ floatx4 mfmaX8(int op, const intx8& A, const intx8& B, const floatx4& c, int scaleA) {      // (op is a constant after unrolling)
    switch (op & 3) { case 0: return mfmaX8<0>(A, B, c, scaleA); case 1: return mfmaX8<1>(A, B, c, scaleA); case 2: return mfmaX8<2>(A, B, c, scaleA); default: return mfmaX8<3>(A, B, c, scaleA); }
}


// CTW = 16-channel tiles per workgroup: 8 (128 output channels; wave (pg, cg) = 64 pixels x 64 channels) or 4 / 2
// (layers with <= 64 / <= 32 output channels: both waves of a row pair use the same channel tiles and split the four
// pixel tiles; a 16 KB weight slab then holds 2 / 4 taps, so the barrier cadence stays at 16 fragment rows)
// HB = halo buffers: 2 = the next halo streams in behind the weight slabs (one workgroup per CU);  1 = the halo is reloaded
// between phases (exposed, but the LDS footprint lets TWO 4-wave workgroups share a CU: their barriers and their
// LDS-read / MFMA phases are no longer in lockstep)
// MX (round 4, 1 x 1 layers with 128-channel chunks only): the input is a split tensor [hi | lo | x8] of C = Cin / 3 channels; C / 64 phases of the hi
// plane (two fp16 k-steps each, as before) are followed by C / 64 phases of the x8 plane -- 128 bytes per pixel again: two 32-channel groups of
// [lo8 0..15 | hi8 0..15 | lo8 16..31 | hi8 16..31] -- each ONE e4m3 K = 128 step: lane group g of the B operand reads chunks 4 (g >> 1) + (g & 1) and
// + 2 of its pixel (group g >> 1, lo8 for even g, hi8 for odd g; conflict-free under the slot = chunk ^ (hx & 7) swizzle, enumerated), the A operand two
// lane-linear 1 KB rows per channel tile with e4m3(2^e w_hi) / e4m3(2^(e + 11) w_lo) in the even / odd lane groups (DsvtConv2dPlugin::packMX1x1).
template <int TH, int KS, int CTW, int HB, bool SPL = false, bool MX = false>
__global__ void __launch_bounds__(64 * TH, 1)
conv_halo_kernel(ConvArgs a, const _Float16* __restrict__ Wp, const _Float16* __restrict__ zeros, int tilesX, int nitems, int nchunk, int dbg)
{
    static_assert(!MX || (KS == 1 && CTW == 8 && SPL), "the fp16 + fp8 loop of this kernel serves the 1 x 1 layers");
    constexpr bool NARROW = CTW < 8;
    constexpr int HS = KS == 1 ? 32 : HHS;                          // LDS halo row stride in pixels (multiple of 8)
    constexpr int HH = TH + KS - 1, HWU = HTW + KS - 1, T = KS * KS, PAD = KS / 2;
    constexpr int TPS = (8 / CTW) < T ? (8 / CTW) : T;              // taps per weight slab
    constexpr int NG = (T + TPS - 1) / TPS;                         // slabs per 64-channel phase
    constexpr int WROWS = TPS * 2 * CTW;                            // 1 KB fragment rows per slab (16, or 8 for a narrow 1x1)
    constexpr int CTP = CTW < 4 ? CTW : 4;                          // channel tiles per wave
    constexpr int NM = NARROW ? 2 : 4;                              // 16-pixel tiles per wave
    constexpr int HBYTES = HH * HS * 128, WBYTES = 16384;
    constexpr int NI = HH * HS / 8;                                 // 1 KB LDS-DMA instructions per halo (8 pixels x 128 B each)
    constexpr int HPW = (NI + TH - 1) / TH;                         // ... per wave
    constexpr int WPW = (WROWS + TH - 1) / TH;                      // weight rows per wave per slab
    static_assert(KS == 1 || HPW <= T, "one halo request per tap");
    // weight slabs in flight: NWB = 3 (slab s+2 requested when slab s starts, counted vmcnt at the slab end) is implemented and
    // measured 8 % SLOWER than one slab of lead on the 468x468 layers (117.7 vs 108.6 us): the slab-end wait is not what stalls
    constexpr int NWB = 2, LEAD = NWB - 1;
    // ONE shared object: halo[2] | wslab[2]  (a second __shared__ object makes hipcc drain the DMA queue before every ds_read)
    __shared__ __attribute__((aligned(16))) unsigned char smem[HB * HBYTES + NWB * WBYTES + 1024];      // ... | bias of the item's chunk
    constexpr int BIAS_OFF = HB * HBYTES + NWB * WBYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int pg = wave >> 1, cg = wave & 1;
    const int NPM = MX ? a.Cin / 192 : a.Cin >> 6;                    // 64-channel phases of fp16 k-steps (MX: of the hi plane; a.Cin = 3 C)
    const int NCC = MX ? 2 * NPM : NPM, NCT = NARROW ? CTW : nchunk * 8;
    const int m0 = NARROW ? 2 * cg : 0, cgc = NARROW ? 0 : cg;       // first pixel tile / channel group of this wave

    // halo requests of this lane: request j of the wave covers LDS pixels 8(wave + j TH) .. +7; lane = (pixel, 16-byte slot)
    int hpos[HPW];
#pragma unroll
    for (int j = 0; j < HPW; ++j) {
        const int lp = 8 * (wave + j * TH) + (lane >> 3), slot = lane & 7;
        const int hy = lp / HS, hx = lp - hy * HS;
        hpos[j] = (hy << 16) | (hx << 4) | (slot ^ (hx & 7));         // 16-byte channel chunk of the pixel stored in this slot
    }
    int goff[HPW];                                                    // element offset into the input, or -1: zeros
    const int perImg = nitems / a.nb;                                 // items of one image (items walk image after image)
    auto setup = [&](int yy, int xx, int bb) {
#pragma unroll
        for (int j = 0; j < HPW; ++j) {
            const int hx = (hpos[j] >> 4) & 0xfff;
            const int gy = yy - PAD + (hpos[j] >> 16), gx = xx - PAD + hx;
            const bool ok = hx < HWU && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            goff[j] = ok ? ((bb * a.H + gy) * a.W + gx) * a.Cin + (hpos[j] & 15) * 8 : -1;
        }
    };
    auto decode = [&](int it, int& yy, int& xx, int& ch, int& bb) {
        bb = it / perImg; it -= bb * perImg;
        ch = it % nchunk; const int t = it / nchunk;
        yy = (t / tilesX) * TH; xx = (t % tilesX) * HTW;
    };
    auto haloRequest = [&](int j, int cc, int hb) {                   // wave-uniform j: one LDS-DMA of 1 KB
        const int c0 = MX ? (cc >= NPM ? 128 * NPM + (cc - NPM) * 64 : cc * 64)                  // (MX: phase NPM + q = 128 bytes of the x8 plane, which starts at channel 2 C)
                          : (a.alias3 && cc * 64 >= a.alias3) ? cc * 64 - a.alias3 : cc * 64;        // (the phases of an x8 third plane read plane 0)
        const _Float16* src = goff[j] >= 0 ? a.in + goff[j] + c0 : zeros;
        __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(smem + hb * HBYTES + (wave + j * TH) * 1024), 16, 0, 0);
    };
    auto weightRequests = [&](int q0, int ch, int wb) {               // slab starting at k-step q0 (= 2 (cc T + tap))
#pragma unroll
        for (int j = 0; j < WPW; ++j) {
            const int u = wave + j * TH;
            if (WROWS % TH == 0 || u < WROWS) {
                size_t row = NARROW ? (size_t)q0 * CTW + u : (size_t)(q0 + (u >> 3)) * NCT + ch * 8 + (u & 7);
                if (MX && q0 >= 2 * NPM)                          // cross phase q0 / 2 - NPM: [NCT][2][1 KB] behind the fp16 rows
                    row = (size_t)2 * NPM * NCT + ((size_t)(q0 / 2 - NPM) * NCT + ch * 8 + (u >> 1)) * 2 + (u & 1);
                __builtin_amdgcn_global_load_lds((glds_src_t)(Wp + (row * 64 + lane) * 8), (glds_dst_t)(smem + HB * HBYTES + wb * WBYTES + u * 1024), 16, 0, 0);
            }
        }
    };

    int pbase[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) pbase[m] = ((2 * pg + ((m0 + m) >> 1)) * HS + ((m0 + m) & 1) * 16 + r) * 128;
    int swz[KS];
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) swz[kx] = (g ^ ((r + kx) & 7)) << 4;

    floatx4 acc[CTP][NM];
    auto tilesOf = [&](int ch) {                                      // valid 16-channel tiles of this wave in chunk ch
        int n = (a.CoutRows - ch * CNB + 15) / 16 - cgc * 4;
        return n < 0 ? 0 : n > CTP ? CTP : n;
    };
    // bias / residual / ReLU / store of a finished item (tile origin ey, ex; 128-channel chunk ech)
    auto epilogue = [&](int ey, int ex, int ech, int ebb) {
        const int n0 = ech * CNB, ctn = tilesOf(ech);
        const int sub = n0 / a.Cout, dy = sub / a.up, dx = sub - dy * a.up, cbase = n0 - sub * a.Cout;
        const int Wout = a.Wo * a.up;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int oy = ey + 2 * pg + ((m0 + m) >> 1), ox = ex + ((m0 + m) & 1) * 16 + r;
            const bool valid = oy < a.Ho && ox < a.Wo;
            const size_t opix = valid ? (size_t)((ebb * a.Ho + oy) * a.up + dy) * Wout + (ox * a.up + dx) : 0;
            if (a.wide) {
#pragma unroll
                for (int t0 = 0; t0 + 1 < CTP; t0 += 2)
                    if (t0 < ctn) convStoreWide<false, SPL>(a, acc[t0][m], acc[t0 + 1][m], valid, opix, cbase + cgc * 64 + t0 * 16, g);
            } else if (valid) {
#pragma unroll
                for (int ct = 0; ct < CTP; ++ct)
                    if (ct < ctn) convStore<false, SPL>(a, acc[ct][m], opix, cbase + cgc * 64 + ct * 16 + 4 * g);
            }
        }
    };

    // the bias of a chunk (<= 128 floats) is one more LDS-DMA piece and the accumulators start from it: the wide epilogue then has
    // no load between its stores (a load after a store is awaited with vmcnt(0), i.e. behind the store's write acknowledgement)
    const bool biasInit = a.bias != nullptr;                       // (the plugin pads the bias array with zeros to a whole float4)
    auto biasRequest = [&](int ch) {
        const int n0 = ch * CNB, sub = n0 / a.Cout, co = n0 - sub * a.Cout + lane * 4;
        const void* src = (biasInit && lane < 32 && co < a.Cout) ? static_cast<const void*>(a.bias + co) : static_cast<const void*>(zeros);
        if (MX && lane >= 32 && lane < 40) src = a.xscale + n0 + (lane - 32) * 16;       // the chunk's scale bytes: LDS bytes 512 .. 639
        __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(smem + BIAS_OFF), 16, 0, 0);
    };
    int xs = 0;                                                     // MX: scale bytes of this wave's four channel tiles (row r)

    int item = blockIdx.x;
    if (kAblate && (dbg & 16) && (gridDim.x & 7u) == 0u) item = (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3));      // XCD-aware item walk (measured, not kept): see conv_wide_kernel
    if (item >= nitems) return;
    int y0, x0, chunk, bimg;
    decode(item, y0, x0, chunk, bimg);
    setup(y0, x0, bimg);
    if (wave == 0) biasRequest(chunk);
#pragma unroll
    for (int j = 0; j < HPW; ++j)
        if (NI % TH == 0 || wave + j * TH < NI) haloRequest(j, 0, 0);
    weightRequests(0, chunk, 0);
    if (LEAD == 2) weightRequests(TPS * 2, chunk, 1);           // (NG >= 3 for the 3x3 variants: the second slab is in the same phase)
    slabBarrier(0);
    int hb = 0, wb = 0;
    const int wcnt = (WROWS - wave + TH - 1) / TH;              // weight requests this wave issues per slab
    int nreqPrev = 0;                                           // halo requests issued during the previous slab
    bool pending = false; int ey = 0, ex = 0, ech = 0, ebb = 0;  // finished item whose epilogue has not run yet
    for (;;) {
        const int ctn = tilesOf(chunk);
        int nitem = item, ny0 = y0, nx0 = x0, nch = chunk, nbimg = bimg;
        bool have_next = true;
        for (int cc = 0; cc < NCC; ++cc) {
            int ncc = cc + 1;
            if (ncc == NCC) {
                ncc = 0; nitem = item + gridDim.x; have_next = nitem < nitems;
                if (have_next) { decode(nitem, ny0, nx0, nch, nbimg); setup(ny0, nx0, nbimg); }
            }
            const bool haloNext = have_next && !(dbg & 1);
            int nreq = 0;                                        // halo requests issued after the current slab's weight requests
            bool wIssued = false;
#pragma unroll
            for (int tap = 0; tap < T; ++tap) {
                const int tl = tap % TPS, grp = tap / TPS;
                const bool lastTap = tap == T - 1, endOfSlab = tl == TPS - 1 || lastTap;
                if (tl == 0) {
                    if (tap == 0 && cc == 0) {
                        // epilogue of the previous item first: its stores are the oldest requests of this slab and have the
                        // whole MFMA block to complete
                        if (pending && !(dbg & 8)) epilogue(ey, ex, ech, ebb);
#pragma unroll
                        for (int ct = 0; ct < CTP; ++ct) {
                            const floatx4 b4 = *reinterpret_cast<const floatx4*>(smem + BIAS_OFF + (cgc * 64 + ct * 16 + 4 * g) * 4);      // zeros without a bias
#pragma unroll
                            for (int m = 0; m < NM; ++m) acc[ct][m] = b4;
                        }
                        if constexpr (MX) {
                            xs = 0;
#pragma unroll
                            for (int ct = 0; ct < CTP; ++ct) xs |= (int)smem[BIAS_OFF + 512 + (cgc * 4 + ct) * 16 + r] << (8 * ct);
                        }
                    }
                    // the slab LEAD ahead: (cc, grp + LEAD) or, past the end of this phase, (ncc, grp + LEAD - NG) of the next one
                    wIssued = false;
                    if (!(dbg & 2)) {
                        if (grp + LEAD < NG) { weightRequests((cc * T + (grp + LEAD) * TPS) * 2, chunk, (wb + LEAD) % NWB); wIssued = true; }
                        else if (have_next) {
                            weightRequests((ncc * T + (grp + LEAD - NG) * TPS) * 2, nch, (wb + LEAD) % NWB); wIssued = true;
                            if (ncc == 0 && grp + LEAD == NG && wave == 0) biasRequest(nch);      // first slab of the next item: its bias too
                        }
                    }
                    nreq = 0;
                }
                // the next halo is requested AFTER this slab's weight requests, one per tap: the slab-end wait (in-order
                // counter, vmcnt(nreq)) retires the weights and leaves the halo requests in flight across the barrier
                if (HB == 1) {
                } else if (KS == 1) {
                    if (haloNext) {
#pragma unroll
                        for (int j = 0; j < HPW; ++j)
                            if (NI % TH == 0 || wave + j * TH < NI) haloRequest(j, ncc, hb ^ 1);
                    }
                } else if (tap < HPW) {
                    if (haloNext && (NI % TH == 0 || wave + (tap < HPW ? tap : 0) * TH < NI)) {
                        haloRequest(tap < HPW ? tap : 0, ncc, hb ^ 1);
                        ++nreq;
                    }
                }
                const unsigned char* hbp = smem + hb * HBYTES;
                const unsigned char* wbp = smem + HB * HBYTES + wb * WBYTES + (lane << 4);
                const int ky = tap / KS, kx = tap - ky * KS;
                const int toff = (ky * HS + kx) * 128;
                const int sw = swz[kx];
                auto slab = [&](auto full) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        half8 A[CTP], B[NM];
#pragma unroll
                        for (int ct = 0; ct < CTP; ++ct) {
                            const int row = NARROW ? (tl * 2 + ks) * CTW + ct : ks * 8 + ct;
                            A[ct] = *reinterpret_cast<const half8*>(wbp + ((row + (NARROW ? 0 : cgc * 4)) << 10));
                        }
#pragma unroll
                        for (int m = 0; m < NM; ++m) B[m] = *reinterpret_cast<const half8*>(hbp + pbase[m] + toff + (sw ^ (ks << 6)));
#pragma unroll
                        for (int ct = 0; ct < CTP; ++ct)
                            if (decltype(full)::value || ct < ctn)
#pragma unroll
                                for (int m = 0; m < NM; ++m)
                                    acc[ct][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ct], B[m], acc[ct][m], 0, 0, 0);
                    }
                };
                auto slabX = [&]() {                                 // (MX) one cross step: 16 scaled MFMAs per wave
                    if constexpr (MX) {
                        intx8 A[CTP], B[NM];
                        const int c0 = 4 * (g >> 1) + (g & 1), s0 = (c0 ^ (r & 7)) << 4, s1 = ((c0 + 2) ^ (r & 7)) << 4;
#pragma unroll
                        for (int ct = 0; ct < CTP; ++ct) {
                            const unsigned char* wr = wbp + (((cgc * 4 + ct) * 2) << 10);
                            const intx4 lo = *reinterpret_cast<const intx4*>(wr), hi = *reinterpret_cast<const intx4*>(wr + 1024);
                            A[ct] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
#pragma unroll
                        for (int m = 0; m < NM; ++m) {
                            const intx4 lo = *reinterpret_cast<const intx4*>(hbp + pbase[m] + s0), hi = *reinterpret_cast<const intx4*>(hbp + pbase[m] + s1);
                            B[m] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
#pragma unroll
                        for (int ct = 0; ct < CTP; ++ct)
                            if (ct < ctn)
#pragma unroll
                                for (int m = 0; m < NM; ++m) acc[ct][m] = mfmaX8(ct, A[ct], B[m], acc[ct][m], xs);
                    }
                };
                if (dbg & 4) {} else if (MX && cc >= NPM) slabX(); else if (ctn == CTP) slab(std::true_type{}); else slab(std::false_type{});
                if (endOfSlab) {
                    // in issue order the queue holds: [W of the next slab] [halo reqs of the previous slab] [W issued at this slab's
                    // start] [halo reqs of this slab].  LEAD 1: retire everything but this slab's halo requests.  LEAD 2: retire the
                    // next slab's weights, i.e. leave the three younger groups in flight.  The last slab of a phase also publishes
                    // the next halo: its youngest request is at least two taps old by then, so only weights may stay in flight.
                    int keep;
                    if (LEAD == 1) keep = lastTap ? 0 : nreq;
                    else keep = (lastTap ? 0 : nreq + nreqPrev) + (wIssued ? wcnt : 0);
                    slabBarrier(keep);
                    nreqPrev = nreq;
                    wb = (wb + 1) % NWB;
                }
                if (HB == 1 && lastTap && haloNext) {            // everyone is done with the halo: reload it in place
#pragma unroll
                    for (int j = 0; j < HPW; ++j)
                        if (NI % TH == 0 || wave + j * TH < NI) haloRequest(j, ncc, 0);
                    slabBarrier(0);
                }
            }
            if (HB == 2) hb ^= 1;
        }
        pending = true; ey = y0; ex = x0; ech = chunk; ebb = bimg;
        if (!have_next) break;
        item = nitem; y0 = ny0; x0 = nx0; chunk = nch; bimg = nbimg;
    }
    if (!(dbg & 8)) epilogue(ey, ex, ech, ebb);
}

// -------------------------------------------------------------------------------------
// The 468x468 layers with 128-channel output chunks: 16 rows x 32 pixels x 128 channels per workgroup, wave w = rows 2w, 2w+1
// x ALL 128 channels (128 accumulator registers).  Against the 8-row kernel above this halves, per MFMA, the LDS-DMA pieces
// a wave issues (each costs 60-180 cycles of issue among MFMAs), the workgroup barriers, and cuts the fragment reads from 16
// to 12 per 32 MFMAs.  To keep two halo buffers in LDS the K loop walks 32-channel phases: a halo pixel is 64 B (four 16-byte
// chunks, chunk c of halo column hx in slot c ^ ((hx >> 1) & 2): conflict-free ds_read_b128 for all three kx), one
// (phase, tap) step is ONE k-step of the packed weights (8 fragment rows), a 16 KB weight slab = two steps = 64 MFMAs per
// wave per barrier.  Phases run continuously across taps and items: the halo of phase p+1 (or of the next item's phase 0)
// is requested during the first three slabs of phase p, the weights of slab s+1 when slab s starts; one vmcnt(0) + barrier
// per slab.  The slab loop is NOT unrolled (a fully unrolled body of 1152 MFMAs makes hipcc spill the accumulators).
// CT = 16-channel tiles per workgroup (8: 128-channel chunks; 4: 64-channel chunks, whose weight slab holds SPS = 4 steps:
// with nine-step phases and two halo buffers a slab cannot span more than four steps).
// NW = waves = half the tile rows.  <8, 8, 40> and <4, 8, 40> serve the 468x468 layers; <4, 4, 36> (8 rows x 32 pixels x 64
// channels, 4 waves, 78 KB of LDS: two independent workgroups per CU) serves the 234x234 and 117x117 layers, which have
// too few 16-row x 128-channel items for 256 CUs (117x117x256: 240 items of four waves instead of 120 of eight).
// HS = halo row stride in pixels (>= 34; HS * 64 B is a multiple of 256 B, so the bank pattern is that of one row).
#ifndef WIDE_REQ_V2
#define WIDE_REQ_V2 0
#endif
#ifndef WIDE_REQ_V3
#define WIDE_REQ_V3 0
#endif
template <int CT, int NW, int HS, int SPS, int RW>
struct WideCfg {
    static constexpr int ROWS = RW * NW, HH = ROWS + 2;
    static constexpr int NPC = (HH * HS * 64 + 1023) / 1024;       // LDS-DMA pieces (16 pixels each) per halo phase
    static constexpr int HBYTES = NPC * 1024, WBYTES = SPS * CT * 1024;
    static constexpr int PPW = (NPC + NW - 1) / NW;                // pieces per wave per phase
    static constexpr int RPW = (SPS * CT + NW - 1) / NW;            // weight rows per wave per slab (NW = 7: waves 0, 1 take three, the others two)
};

// NWB = weight slabs in LDS: 3 = the slab TWO ahead is requested when a slab starts and the slab-end wait leaves those requests
// in flight (one slab of MFMAs, 0.5-1 us, is shorter than an L2 -> LDS round trip under load); 2 where LDS must hold two workgroups.
// RW = tile rows per wave: 2 (64 pixels, four pixel tiles) or 1 (32 pixels, two pixel tiles: twice the waves on the same tile --
// for the small layers, where one wave per SIMD cannot hide its own LDS-DMA issue and wait time)
// MX (round 4; implies the split epilogue): the input is a split tensor [hi | lo | x8] of C = Cin / 3 real channels and the K loop is the fp32-grade
// product at HALF the matrix-pipe time of the [hi | lo | hi] walk: C / 32 "main" phases of the hi plane (nine fp16 k-steps each, as before) followed by
// C / 32 "cross" phases of the x8 plane (64 B per pixel too: the halo machinery does not change), each FIVE steps of
// v_mfma_scale_f32_16x16x128_f8f6f4: the K = 128 of step j are taps 2j, 2j + 1 x [lo8 | hi8] of the phase's 32 channels -- lane group g of the B
// operand reads tap 2j + (g >> 1), chunks (g & 1) and 2 + (g & 1) of its pixel (conflict-free with the same column swizzle: enumerated over the
// ds_read_b128 lane groups), lane group g of the A operand holds e4m3(2^e w_hi) (g even: pairs with lo8) or e4m3(2^(e + 11) w_lo) (g odd: pairs with
// hi8) of row n, and the row's scale byte 127 - 11 - e undoes both factors (tap 9 of the fifth step: zero weights).  Per 32 channels and tile pair:
// 9 fp16 + 5 fp8 MFMAs = 9 x 16 + 5 x 27 cycles of the pipe instead of 27 x 16.  A cross step's fragments are 2 KB per channel tile (two lane-linear
// 1 KB rows: bytes 0..15 and 16..31 of every lane), so a weight slab holds SPS / 2 cross steps.  Packed weights: DsvtConv2dPlugin::packMX.
// What bounds the slab loop (round 5, late; profiles/r05_conv_issue_bound.txt): the ALTERNATION of its two instruction kinds.  The inner loop of <8, 8, 36, 4, 2, 2, SPL> is 830 instructions per
// wave and slab -- 128 MFMAs in eight back-to-back blocks of 16 (256 matrix-pipe cycles each), and between them eight blocks of ~88 others (308 VALU, 335 SALU of slab schedule and addresses,
// 48 ds_read_b128, 32 branches, 11 LDS-DMA requests; no scratch access inside the loop).  The sched_barrier fences and the branches around every request keep hipcc from placing the others
// UNDER the MFMAs, so a wave alone needs 8 x (256 + ~290) = 4390 cycles per slab (s_memtime stamps, tools/trace_conv_split.py: exactly the wave that has priority), its partner 5580, + 370
// for the requests to land, + 160 at the barrier: 6100 where the pipe needs 4096.  DMA wait and barrier skew are 6 % of an item, the epilogue 8 %, the 27 slabs 70 %.  Variants measured on
// this (switches in the source, off): WIDE_REQ_V2 = 1 (wave index in an SGPR, branch-free validity test: 763 instructions) -1 % on the dense stage (-3.5 % without a residual, +3 % with);
// WIDE_REQ_V3 = 1 (request arithmetic from LDS tables + buffer descriptor: 727) -1 .. -1.8 %; no "step < NSTEP" guard around the MFMA batches: hipcc spills INSIDE the loop (22 scratch
// accesses per slab) and a layer takes 1088 instead of 731 us.  Fewer instructions buy little; they have to issue under the MFMAs of the same wave, i.e. the slab body as ONE straight-line
// region (descriptors precomputed in LDS, predicated requests) behind sched_group_barrier pipelines or in assembly: DESIGN.md section 6.
template <int CT, int NW, int HS, int SPS = (CT == 8 ? 2 : 4), int NWB = 3, int RW = 2, bool TR = false, bool SPL = false, bool MX = false>
__global__ void __launch_bounds__(64 * NW, (NW * (RW == 1 ? 1 : 2) <= 8 && (NW == 4 || RW == 1)) ? 2 : 1)
conv_wide_kernel(ConvArgs a, const _Float16* __restrict__ Wp, const _Float16* __restrict__ zeros, int tilesX, int nitems, int nchunk, int dbg)
{
    static_assert(!MX || SPL, "the MX K loop reads a split tensor and writes one");
    using C = WideCfg<CT, NW, HS, SPS, RW>;
    constexpr int XPS = SPS / 2;                                  // cross steps per weight slab
    constexpr int NM = 2 * RW;                                    // 16-pixel tiles per wave
    static_assert(SPS == 2 || SPS == 4, "nine-step phases and two halo buffers: a slab spans at most four steps");
    constexpr int WT_HS = HS, WT_HBYTES = C::HBYTES, WT_WBYTES = C::WBYTES, WT_NPC = C::NPC, WT_ROWS = C::ROWS, PPW = C::PPW;
    constexpr int PPS = SPS == 2 ? (PPW + 2) / 3 : PPW;           // halo pieces a wave requests per slab
    constexpr int NRS = (PPW + PPS - 1) / PPS;                    // ... over this many slabs
    constexpr int CH = CT > 4 ? (MX ? MX_CH : 4) : CT;  // A fragments read per batch
    constexpr int LEAD = NWB - 1;
    constexpr bool TRICKLE_K = (CT == 8 || (WIDE_TRICKLE_CT4 && !MX && LEAD == 1)) && NW == 8 && (LEAD >= 2 || (WIDE_TRICKLE1 && !MX));          // requests spread over the slab (see the slab loop)
    // WIDE_REQ_V3 (round 5, late): a halo request's per-lane arithmetic from two kernel-invariant LDS tables -- byte offset of (piece, lane)'s 16 bytes within the tile
    // [NPC][64] u32, (hy, hx) of its pixel [NPC][16] u16 -- and the request itself as buffer_load_dwordx4 ... lds through ONE descriptor over the input tensor with the
    // offset of an invalid lane set out of range (the hardware returns zeros): ~12 vector instructions per request instead of ~45 in three exec-mask branches
    constexpr int TBL_BYTES = WT_NPC * 64 * 4 + WT_NPC * 16 * 2;
    constexpr bool REQ3 = WIDE_REQ_V3 != 0 && !MX && 2 * WT_HBYTES + NWB * WT_WBYTES + 1024 + TBL_BYTES <= 160 * 1024;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WT_HBYTES + NWB * WT_WBYTES + 1024 + (REQ3 ? TBL_BYTES : 0)];      // halo[2] | wslab[NWB] | bias (| request tables)
    constexpr int BIAS_OFF = 2 * WT_HBYTES + NWB * WT_WBYTES;
    constexpr int TOFF_OFF = BIAS_OFF + 1024, THYX_OFF = TOFF_OFF + WT_NPC * 64 * 4;
#if WIDE_REQ_V2 == 1
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, g = lane >> 4;     // (wave index in an SGPR: the request bookkeeping is scalar)
#else
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
#endif
    const int NPM = MX ? a.Cin / 96 : a.Cin >> 5;                 // 32-channel phases of fp16 k-steps (MX: of the hi plane; a.Cin = 3 C)
    const int NP = MX ? 2 * NPM : NPM;                            // halo phases (even)
    const int NSTEP = NPM * 9, NSA = (NSTEP + SPS - 1) / SPS;     // (SPS = 4: the last fp16 slab may be partial; the packed weights end in a zero slab)
    const int NSLAB = NSA + (MX ? (5 * NPM + XPS - 1) / XPS : 0);
    const int NCT = a.CoutRows <= 64 ? 4 : (a.CoutRows + 127) / 128 * 8;      // 16-channel tiles per k-step of the packed weights
    // (Round 5 built a "split-native" phase walk for the three-product head -- per 32 channels q the phases (hi_q, w_hi), (hi_q AGAIN, w_lo) reading the halo the
    // first one left in LDS, (lo_q, w_hi): a third of the halo requests gone, same weights / MFMAs / fragment reads -- correct (tests pass) and SLOWER:
    // 11.15 vs 10.58 ms of convolutions per four-frame forward, 276 vs 285 frames/s.  Like round 4's re-ordered head-output kernel: a phase without halo
    // requests does not run faster, the bytes of the aliased plane -- an L2 hit -- were not what the kernel waits for.  The plain plane-order walk stays.)
    // (Also round 5: the CT = 8 instantiations hold 256 registers with 34-46 spilled, and the hoisted lane shares of the trickled requests are reloaded from scratch
    // inside the slab loop -- each reload waits with vmcnt(0), i.e. for every request in flight.  A loop without any scratch access (wave index in an SGPR, lane shares
    // recomputed per request) was built: same results, 4-5 % SLOWER on the three-product 128-channel layers, neutral on the fp16 frame:
    // profiles/r05_conv_dead_ends.txt.  The waits are not what this kernel loses time to.)
    const int perImg = nitems / a.nb;                             // items of one image (items walk image after image)
    auto decode = [&](int it, int& yy, int& xx, int& ch, int& bb) {
        bb = it / perImg; it -= bb * perImg;
        ch = it % nchunk; const int t = it / nchunk;
        yy = (t / tilesX) * WT_ROWS; xx = (t % tilesX) * HTW;
    };
    // piece pc of the halo of phase ph at tile origin (yy, xx) of image bb -> buffer hb
    const __amdgpu_buffer_rsrc_t inRsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.in), 0, (int)((size_t)a.nb * a.H * a.W * a.Cin * 2), 0x00020000);
    if constexpr (REQ3) {
        // this wave's pieces only: a wave reads back what it wrote (no workgroup barrier needed)
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pc = wave + NW * i;
            if (pc < WT_NPC) {
                const int lp = pc * 16 + (lane >> 2), hy = lp / WT_HS, hx = lp - hy * WT_HS;
                const int chunk = (lane & 3) ^ ((hx >> 1) & 2);
                const bool in = hx < HTW + 2 && hy < C::HH;
                // (padding pixels of the 36-pixel rows / beyond the last halo row are never read by a fragment: they fetch the tile's first pixel)
                *reinterpret_cast<uint32_t*>(smem + TOFF_OFF + (pc * 64 + lane) * 4) = in ? (uint32_t)(((hy * a.W + hx) * a.Cin + chunk * 8) * 2) : 0u;
                if ((lane & 3) == 0) *reinterpret_cast<unsigned short*>(smem + THYX_OFF + lp * 2) = in ? (unsigned short)(hy << 8 | hx) : (unsigned short)0x0101;
            }
        }
    }
    auto haloRequest = [&](int pc, int yy, int xx, int ph, int hb, int bb) {
        if constexpr (REQ3) {
            const uint32_t toff = *reinterpret_cast<const uint32_t*>(smem + TOFF_OFF + (pc * 64 + lane) * 4);
            const uint32_t hyx = *reinterpret_cast<const unsigned short*>(smem + THYX_OFF + (pc * 16 + (lane >> 2)) * 2);
            const int gy = yy - 1 + (int)(hyx >> 8), gx = xx - 1 + (int)(hyx & 255u);
            const unsigned ok = (unsigned)((unsigned)gy < (unsigned)a.H) & (unsigned)((unsigned)gx < (unsigned)a.W);
            const int coff = (a.alias3 && ph * 32 >= a.alias3) ? ph * 32 - a.alias3 : ph * 32;
            const uint32_t sbase = (uint32_t)((((bb * a.H + yy - 1) * a.W + xx - 1) * a.Cin + coff) * 2);          // (wraps for yy = 0 / xx = 0: the sum below is exact mod 2^32 for every valid lane)
            const uint32_t voff = ok ? toff + sbase : 0xFFFFFFF0u;                                           // (>= num_records: the tensor holds < 2^31 elements, checked by the plugin)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(inRsrc, (glds_dst_t)(smem + hb * WT_HBYTES + pc * 1024), 16, (int)voff, 0, 0, 0);
            return;
        }
        const int lp = pc * 16 + (lane >> 2), hy = lp / WT_HS, hx = lp - hy * WT_HS;
        const int chunk = (lane & 3) ^ ((hx >> 1) & 2);
        const int gy = yy - 1 + hy, gx = xx - 1 + hx;
        const int coff = MX ? (ph >= NPM ? 64 * NPM + (ph - NPM) * 32 : ph * 32)             // (MX: phase NPM + q = 64 bytes of the x8 plane, which starts at channel 2 C)
                            : (a.alias3 && ph * 32 >= a.alias3) ? ph * 32 - a.alias3 : ph * 32;   // (three fp16 products over an x8 third plane: its phases read plane 0)
#if WIDE_REQ_V2
        // branch-free: the address of every lane is formed (never dereferenced when the pixel is outside the image or the halo), then selected.  With the short-circuit
        // form hipcc wrapped the address arithmetic in three nested exec-mask branches per request, each with an s_waitcnt lgkmcnt(0) that also drains the fragment reads
        const unsigned ok = (unsigned)(hx < HTW + 2) & (unsigned)(hy < C::HH) & (unsigned)((unsigned)gy < (unsigned)a.H) & (unsigned)((unsigned)gx < (unsigned)a.W);
        const long off = (long)((bb * a.H + gy) * a.W + gx) * a.Cin + (coff + chunk * 8);
        const _Float16* src = ok ? a.in + off : zeros;
#else
        const bool ok = hx < HTW + 2 && hy < C::HH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const _Float16* src = ok ? a.in + (size_t)((bb * a.H + gy) * a.W + gx) * a.Cin + coff + chunk * 8 : zeros;
#endif
        __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(smem + hb * WT_HBYTES + pc * 1024), 16, 0, 0);
    };
    // the 16 fragment rows of slab sl (steps SPS sl ...) of chunk ch -> buffer wb
    auto weightRequest = [&](int sl, int ch, int wb, int j) {
        const int u = wave + NW * j;
        if ((SPS * CT) % NW != 0 && u >= SPS * CT) return;
        size_t row;
        if (MX) {         // packMX: [9 NPM fp16 steps][NCT][1 KB] | [5 NPM cross steps][NCT][2][1 KB]
            if (sl < NSA) row = (size_t)(SPS * sl + u / CT) * NCT + ch * CT + u % CT;
            else row = (size_t)NSTEP * NCT + ((size_t)(XPS * (sl - NSA) + u / (2 * CT)) * NCT + ch * CT + (u % (2 * CT)) / 2) * 2 + (u & 1);
        } else {
            const int step = SPS * sl + u / CT, ph = step / 9, tap = step - 9 * ph;
            row = (size_t)(2 * ((ph >> 1) * 9 + tap) + (ph & 1)) * NCT + ch * CT + u % CT;
        }
        __builtin_amdgcn_global_load_lds((glds_src_t)(Wp + (row * 64 + lane) * 8), (glds_dst_t)(smem + 2 * WT_HBYTES + wb * WT_WBYTES + u * 1024), 16, 0, 0);
    };
    auto weightRequests = [&](int sl, int ch, int wb) {
#pragma unroll
        for (int j = 0; j < C::RPW; ++j) weightRequest(sl, ch, wb, j);
    };

    // the chunk's bias (CT * 16 floats) travels as one more LDS-DMA piece and the accumulators start from it: the epilogue of a
    // layer without a residual then holds no load at all (a load issued after a store is awaited with vmcnt(0): gfx950 counts
    // loads and stores in one counter and hipcc cannot count across the two kinds)
    const bool biasInit = a.bias != nullptr;                       // (the plugin pads the bias array with zeros to a whole float4)
    auto biasRequest = [&](int ch) {
        const int n0 = ch * CT * 16, sub = n0 / a.Cout, co = n0 - sub * a.Cout + lane * 4;
        const void* src = (biasInit && lane < CT * 4 && co < a.Cout) ? static_cast<const void*>(a.bias + co) : static_cast<const void*>(zeros);
        if (MX && lane >= 32 && lane < 32 + CT) src = a.xscale + n0 + (lane - 32) * 16;      // the chunk's scale bytes (the table is padded to whole chunks): LDS bytes 512 ..
        __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(smem + BIAS_OFF), 16, 0, 0);
    };

    // An XCD-aware item walk, measured and NOT kept (round 5; ablation build, DSVT_CONV_DBG=16).  Workgroups go to the eight XCDs round-robin (workgroup b -> XCD b % 8) and
    // consecutive items are the channel chunks of one tile, then its right neighbour -- the same halo pixels -- so with item = b the chunks of a tile (five for the 64 -> 320
    // head stems) sit on different XCDs and every L2 fetches the tile's halo itself.  With XCD x walking items x G / 8 .. (x + 1) G / 8 - 1 of every round of G, FETCH_SIZE of
    // the 64-channel-chunk launches (shared 384 -> 64, 64 -> 320 stems) HALVES (1438 -> 754 MB as counted, per four-frame launch; the 128-channel layers 147 -> 138), the
    // dense stage alone does not move (9.42-9.46 against 9.42-9.45 ms, tools/conv_layers.py) and the two-stream frame gets SLOWER: 296.4 / 296.5 against 300.5 / 301.0
    // frames/s (tools/two_stream_fps.py, four alternating runs in one box).  These kernels do not wait for their L2 misses, and 32 neighbouring items per XCD start and
    // end their phases together.  The plain walk stays.
    int item = blockIdx.x;
    if (kAblate && (dbg & 16) && (gridDim.x & 7u) == 0u) item = (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    if (item >= nitems) return;
    int nmark = 0;
    const bool tracing = TR && a.trace != nullptr && (wave == 0 || wave == NW / 2);      // (TR: the instrumented instantiation, tools/trace_conv.py)
    auto mark = [&]() {
        if constexpr (TR) {
            if (tracing) { if (lane == 0 && nmark < CONV_TRACE_N) a.trace[(size_t)(blockIdx.x * 2 + (wave != 0)) * CONV_TRACE_N + nmark] = clock64(); ++nmark; }
        }
    };
    int y0, x0, chunk, bimg;
    decode(item, y0, x0, chunk, bimg);
    mark();
    if (wave == 0) biasRequest(chunk);
#pragma unroll
    for (int i = 0; i < PPW; ++i)
        if (wave + NW * i < WT_NPC) haloRequest(wave + NW * i, y0, x0, 0, 0, bimg);
    // PFA (NWB = 4: the slab THREE ahead is requested when a slab starts): the weights of slab s + 1 are published one slab early, so the first A
    // fragments of the next slab are read BEFORE the barrier that ends the current one, like its first B fragments (whose halo phase is
    // always published by then: requested in the first NRS slabs of the phase before) -- the first MFMAs after a barrier wait for no LDS read
    constexpr bool PFB = WIDE_PFB != 0, PFA = PFB && LEAD >= 3;
    weightRequests(0, chunk, 0);
    if (PFA) weightRequests(1, chunk, 1);
    slabBarrier(0);
    if (LEAD >= 2) weightRequests(PFA ? 2 : 1, chunk, PFA ? 2 : 1);      // (NSLAB >= 5; retired by the first slab-end wait)

    const int wreq = (SPS * CT - wave + NW - 1) / NW;             // weight requests THIS wave issues per slab (what its counted wait leaves in flight)
    const int pb = ((RW * wave) * WT_HS + r) * 64;                // this lane's pixel of pixel tile 0, tap (0, 0)
    const int aoff = lane << 4;
    int wb = 0;
    floatx4 acc[CT][NM];
    // fragment buffers (double buffered by hand inside a slab; the first fragments of a slab are loaded at the end of the slab before)
    half8 Bf[2][NM], Af[2][CH];
    constexpr int CHX = MX_CHX;                                   // (MX) channel tiles per batch of a cross step
    intx8 Bx[MX ? NM : 1], Ax[2][CHX];
    // B fragments of fp16 step t of the item (t = 9 NPM: step 0 of the next item -- phase 0, buffer 0: NPM is even)
    auto loadB = [&](int t, half8 (&B)[NM]) {
        const int ph = t / 9, tap = t - 9 * ph, ky = tap / 3, kx = tap - 3 * ky;
        const unsigned char* hbp = smem + (ph & 1) * WT_HBYTES + (ky * WT_HS + kx) * 64 + pb + ((g ^ (((r + kx) >> 1) & 2)) << 4);
#pragma unroll
        for (int m = 0; m < NM; ++m) B[m] = *reinterpret_cast<const half8*>(hbp + ((m >> 1) * WT_HS + (m & 1) * 16) * 64);
    };
    // A fragments of channel tiles c0 .. of step u of the slab in weight buffer wbuf
    auto loadA = [&](int wbuf, int u, int c0, half8 (&A)[CH]) {
        const unsigned char* wbp = smem + 2 * WT_HBYTES + wbuf * WT_WBYTES + (u * CT + c0) * 1024 + aoff;
#pragma unroll
        for (int ct = 0; ct < CH; ++ct) A[ct] = *reinterpret_cast<const half8*>(wbp + ct * 1024);
    };
    // (MX) B fragments of cross step x = (phase x / 5, tap pair x % 5) of the item
    auto loadBx = [&](int x) {
        const int q = x / 5, pr = x - 5 * q;
        int tap = 2 * pr + (g >> 1); tap = tap > 8 ? 8 : tap;                                 // (tap 9: zero weights)
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;                                    // tap / 3 for 0 .. 8
        const int slot = (g & 1) ^ (((r + kx) >> 1) & 2);
        const unsigned char* hbp = smem + ((NPM + q) & 1) * WT_HBYTES + (ky * WT_HS + kx) * 64 + pb;
#pragma unroll
        for (int m = 0; m < (MX ? NM : 1); ++m) {
            const unsigned char* pp = hbp + ((m >> 1) * WT_HS + (m & 1) * 16) * 64;
            const intx4 lo = *reinterpret_cast<const intx4*>(pp + (slot << 4)), hi = *reinterpret_cast<const intx4*>(pp + ((slot ^ 2) << 4));
            Bx[m] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    auto loadAx = [&](int wbuf, int u, int c0, intx8 (&A)[CHX]) {
        const unsigned char* wbp = smem + 2 * WT_HBYTES + wbuf * WT_WBYTES + (u * CT + c0) * 2048 + aoff;
#pragma unroll
        for (int ct = 0; ct < CHX; ++ct) {
            const intx4 lo = *reinterpret_cast<const intx4*>(wbp + ct * 2048), hi = *reinterpret_cast<const intx4*>(wbp + ct * 2048 + 1024);
            A[ct] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    for (;;) {
        int nitem = item + gridDim.x, ny0 = 0, nx0 = 0, nch = 0, nbimg = 0;
        const bool have_next = nitem < nitems;
        if (have_next) decode(nitem, ny0, nx0, nch, nbimg);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const floatx4 b4 = *reinterpret_cast<const floatx4*>(smem + BIAS_OFF + (ct * 16 + 4 * g) * 4);      // zeros without a bias
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[ct][m] = b4;
        }
        if (PFB) loadB(0, Bf[0]);                                    // the item's first fragments (its first slabs were published while the item before ran)
        if (PFA) loadA(wb, 0, 0, Af[0]);
        int xs[(CT + 3) / 4] = {};                                   // MX: scale byte of row (ct, r) in byte ct & 3 of xs[ct >> 2]
        if constexpr (MX) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) xs[ct >> 2] |= (int)smem[BIAS_OFF + 512 + ct * 16 + r] << (8 * (ct & 3));
        }
        // one slab; CROSS (compile time): a slab of the MX loop's fp8 steps (s >= NSA).  Two loops over this body, not one loop with a branch:
        // with both bodies in one loop hipcc spilled 141 registers of the accumulators into the MFMA stream
        // NEXT (compile time): what follows this slab -- 0: a slab of the same kind, 1: the first cross slab, 2: the end of the item (no
        // prefetch: the epilogue comes first).  The last slab of each loop is peeled so that no fragment buffer is conditionally written
        // inside a loop (hipcc then keeps it alive around the whole loop: 619 spilled registers).
        auto slab = [&](const int s, auto crossTag, auto nextTag) {
            constexpr bool CROSS = decltype(crossTag)::value;
            constexpr int NEXT = decltype(nextTag)::value;
            constexpr bool TRICKLE = TRICKLE_K && (!MX || (CROSS ? MX_TRICKLE_CROSS : MX_TRICKLE_MAIN));
            // halo of phase P (the phase after the one this slab starts in): its buffer is free once phase P - 2 has ended, i.e.
            // from slab s0 = ceil(9 (P - 1) / SPS) on, and the first slab that touches phase P is floor(9 P / SPS) > s0 + NRS - 1.
            // Weights of the slab LEAD ahead; the halo requests of a slab come BEFORE its weight requests: with LEAD = 2 the slab-end
            // wait keeps exactly the weight requests in flight.  In the 128-channel kernel (TRICKLE) the requests of a slab are not
            // issued in one go: a wave that issues is held while the CU's request path works through its queue (s_memtime stamps,
            // tools/trace_conv.py: 300 cycles per request when all eight waves issue at the slab start, 600-1350 cycles per slab
            // with the matrix pipe idle), so request j goes out after the MFMAs of batch j * NB / NREQ, when the wave would wait
            // for the pipe anyway: 88 -> 84 us (128 -> 128 channels), 123 -> 115 us (192 -> 128).  (Neutral to harmful on the
            // 64-channel variants, whose slabs hold four steps: 384 -> 64 channels 125 -> 144 us.)
            // Round 5 (late): with four-step slabs in TWO buffers (LEAD = 1: the next slab's weights must have landed at this slab's end) the requests are spread over the
            // FIRST HALF of the slab's batches (WIDE_TRICKLE1_NUM / WIDE_TRICKLE4_NUM eighths): three-product dense stage 9.72 -> 9.40 ms per four frames (128 -> 128 at
            // 468 x 468 760 -> 733 us, shared 384 -> 64 1150 -> 1085, 64 -> 320 stems 1080 -> 1045); spread over 5/8 already loses, over 3/4 or the whole slab the wait is exposed
            // (922 us / 1245 us).  The fp8 loops keep their requests at the slab start.
            int hP = (SPS * s) / 9 + 1, hk = s - (9 * (hP - 1) + SPS - 1) / SPS;
            if constexpr (CROSS) {                                    // cross phase q0 = five steps: the same rule on its own step count
                const int q0 = (XPS * (s - NSA)) / 5;
                hP = NPM + q0 + 1; hk = s - NSA - (5 * q0 + XPS - 1) / XPS;
            }
            const bool hIn = hP < NP, hOn = hk < NRS && (hIn || have_next) && !(dbg & 1);
            const int hyy = hIn ? y0 : ny0, hxx = hIn ? x0 : nx0, hph = hIn ? hP : 0, hbb = hIn ? bimg : nbimg;
            const int wt = s + LEAD, wbt = (wb + LEAD) % NWB;
            const bool wIn = wt < NSLAB, wIssued = (wIn || have_next) && !(dbg & 2);
            const int wsl = wIn ? wt : wt - NSLAB, wch = wIn ? chunk : nch;
            auto request = [&](int j) {                                // j: 0 .. NREQ - 1 (compile-time after unrolling)
                const int jh = j, jw = j - PPS;
                if (jh >= 0 && jh < PPS) {
                    const int pc = wave + NW * (hk * PPS + jh);
                    if (hOn && hk * PPS + jh < PPW && pc < WT_NPC) haloRequest(pc, hyy, hxx, hph, hP & 1, hbb);
                }
                if (jw >= 0 && jw < C::RPW && wIssued) {
                    if (jw == 0 && !wIn && wt == NSLAB && wave == 0) biasRequest(nch);   // (before the weights: retired with the older requests)
                    weightRequest(wsl, wch, wbt, jw);
                }
            };
            constexpr int NREQ = PPS + C::RPW;
            mark();                                                  // [5 s + 1] slab start
            if constexpr (!TRICKLE) {                                // all requests of the slab at its start
                if (hOn) {
#pragma unroll
                    for (int i = 0; i < PPS; ++i) {
                        const int pc = wave + NW * (hk * PPS + i);
                        if (hk * PPS + i < PPW && pc < WT_NPC) haloRequest(pc, hyy, hxx, hph, hP & 1, hbb);
                    }
                }
                if (wIssued) {
                    if (!wIn && wt == NSLAB && wave == 0) biasRequest(nch);   // (before the weights: retired with the older requests)
                    weightRequests(wsl, wch, wbt);
                }
            }
            mark();                                                  // [5 s + 2] requests issued
            // fragments are double buffered by hand: the reads of batch b + 1 (CH channel tiles x 4 pixel tiles = 4 CH MFMAs) are
            // issued BEFORE the MFMAs of batch b (left to itself hipcc emits read, s_waitcnt lgkmcnt(0), 8 MFMAs, read, ...)
            if constexpr (CROSS) {
                {
                    // a slab of cross steps: step x = (phase q, tap pair pr); batches of CHX channel tiles x NM pixel tiles; the B fragments of a step
                    // stay in ONE buffer (32 registers), the A fragments are double buffered
                    constexpr int BPX = CT / CHX, NBX = XPS * BPX;
                    static_assert(NBX % 2 == 0, "the A buffer of the next slab's first batch");
                    const int x0s = XPS * (s - NSA), wn = (wb + 1) % NWB;
                    if (!PFB) loadBx(x0s);
                    if (!PFA) loadAx(wb, 0, 0, Ax[0]);
#pragma unroll
                    for (int b = 0; b < NBX; ++b) {
                        const int u = b / BPX, c0 = (b % BPX) * CHX;
                        if (b + 1 < NBX) loadAx(wb, (b + 1) / BPX, ((b + 1) % BPX) * CHX, Ax[(b + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (x0s + u < 5 * NPM) {
#pragma unroll
                            for (int ct = 0; ct < CHX; ++ct)
#pragma unroll
                                for (int m = 0; m < NM; ++m)
                                    acc[c0 + ct][m] = mfmaX8(c0 + ct, Ax[b & 1][ct], Bx[m], acc[c0 + ct][m], xs[(c0 + ct) >> 2]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (b + 1 < NBX && (b + 1) % BPX == 0) loadBx(x0s + (b + 1) / BPX);
                        if (b + 1 == NBX && NEXT == 0 && PFB) { loadBx(x0s + XPS); if (PFA) loadAx(wn, 0, 0, Ax[0]); }      // the next slab's first fragments (the MFMAs above are still draining)
                        if (TRICKLE) {
#pragma unroll
                            for (int j = 0; j < NREQ; ++j)
                                if (j * NBX / NREQ == b) request(j);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            } else {
                constexpr int BPS = CT / CH, NB = SPS * BPS;          // batches per step / per slab
                static_assert(NB % 2 == 0, "the A buffer of the next slab's first batch");
                const int wn = (wb + 1) % NWB;
                if (!PFB) loadB(SPS * s, Bf[0]);
                if (!PFA) loadA(wb, 0, 0, Af[0]);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int u = b / BPS, c0 = (b % BPS) * CH;
                    if (b + 1 < NB) {
                        const int un = (b + 1) / BPS, cn = ((b + 1) % BPS) * CH;
                        if (cn == 0) loadB(SPS * s + un, Bf[un & 1]);
                        loadA(wb, un, cn, Af[(b + 1) & 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (SPS == 2 || SPS * s + u < NSTEP) {
#pragma unroll
                        for (int ct = 0; ct < CH; ++ct)
#pragma unroll
                            for (int m = 0; m < NM; ++m)
                                acc[c0 + ct][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Af[b & 1][ct], Bf[u & 1][m], acc[c0 + ct][m], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (b + 1 == NB && PFB) {                        // the next slab's first fragments (the MFMAs above are still draining)
                        if constexpr (NEXT == 1) { loadBx(0); if (PFA) loadAx(wn, 0, 0, Ax[0]); }
                        else if constexpr (NEXT == 0) { loadB(SPS * (s + 1), Bf[0]); if (PFA) loadA(wn, 0, 0, Af[0]); }
                    }
                    if (TRICKLE) {
#pragma unroll
                        for (int j = 0; j < NREQ; ++j)
                            if (j * (LEAD >= 2 ? NB : NB * (CT == 8 ? WIDE_TRICKLE1_NUM : WIDE_TRICKLE4_NUM) / 8) / NREQ == b) request(j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (TR && a.trace) {                                     // (wave-uniform; the wait and the barrier stamped apart)
                mark();                                              // [5 s + 3] MFMAs issued
                slabWait(LEAD >= 2 && wIssued ? wreq : 0);
                mark();                                              // [5 s + 4] own requests landed
                __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
                mark();                                              // [5 s + 5] barrier passed
            } else slabBarrier(LEAD >= 2 && wIssued ? wreq : 0);
            wb = (wb + 1) % NWB;
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
#pragma unroll 1
        for (int s = 0; s < NSA - 1; ++s) slab(s, std::false_type{}, I0{});
        if constexpr (MX) {
            slab(NSA - 1, std::false_type{}, I1{});
#pragma unroll 1
            for (int s = NSA; s < NSLAB - 1; ++s) slab(s, std::true_type{}, I0{});
            slab(NSLAB - 1, std::true_type{}, I2{});
        } else slab(NSA - 1, std::false_type{}, I2{});
        mark();
        // residual / ReLU / store (the bias is in the accumulators)
        if (!(dbg & 8)) {
            const int n0 = chunk * CT * 16;
            int ctn = (a.CoutRows - n0 + 15) / 16; ctn = ctn > CT ? CT : ctn;
            const int sub = n0 / a.Cout, dy = sub / a.up, dx = sub - dy * a.up, cbase = n0 - sub * a.Cout;
            const int Wout = a.Wo * a.up;
            if (a.wide) {
                constexpr int TP = CT / 2, NBLK = NM * TP;            // blocks b = (pixel tile m, channel-tile pair tp)
                const int cg8 = (g & 1) * 16 + (g >> 1) * 8;
                bool valid[NM]; size_t opix[NM];
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int oy = y0 + RW * wave + (m >> 1), ox = x0 + (m & 1) * 16 + r;
                    valid[m] = oy < a.Ho && ox < a.Wo;
                    opix[m] = valid[m] ? (size_t)((bimg * a.Ho + oy) * a.up + dy) * Wout + (ox * a.up + dx) : 0;
                }
                // The residual rows of EIGHT blocks at a time first (32 registers: the fragment buffers have just left them; all sixteen at
                // once spill), without a branch around any load (a lane with nothing to add reads the tensor's first bytes): written as
                // "load, add, store" per block, hipcc kept that order and awaited every load with vmcnt(0) -- sixteen dependent round trips
                // per item, each behind the previous block's store: a residual input cost a 468 x 468 layer 70 us of 340
                // (tools/conv_sequence.py).
                constexpr int RB = NBLK < 8 ? NBLK : 8;
                const bool hasRes = a.res != nullptr;
                if constexpr (SPL) {
                    // split precision: the residual is hi + lo (two planes), the result leaves as three planes; TWO blocks at a time (with four, the
                    // residual loads plus the hi / lo planes in flight spill 30 registers of the main loop's accumulators; with two, 10)
                    // (Round 5 software-pipelined this loop -- round k + 1's residual rows requested before round k is added and stored --: no gain on the
                    // residual layers (953 / 947 vs 964 / 950 us at 468 x 468, tools/conv_layers.py) and 34 spilled registers that cost the layers WITHOUT a residual
                    // 5 %.  The 160 us a residual costs such a layer are not eight exposed latencies, they are 448 MB read by all CUs at once at item end.)
                    constexpr int RS = RB / 4 > 0 ? RB / 4 : 1;
                    const bool splitRes = hasRes && a.res_split != 0, resX8 = splitRes && a.res_x8 != 0;
#pragma unroll
                    for (int b0 = 0; b0 < NBLK; b0 += RS) {
                        half8 rh[RS], rl[RS];
                        if (hasRes) {
#pragma unroll
                            for (int j = 0; j < RS; ++j) {
                                const int b = b0 + j, m = b / TP, tp = b % TP, co = cbase + tp * 32 + cg8;
                                const bool ok = valid[m] && co < a.Cout && 2 * tp < ctn;
                                rh[j] = *reinterpret_cast<const half8*>(a.res + (ok ? opix[m] * a.res_ld + co : 0));
                                if (resX8) {           // 8 lo8 bytes of the x8 plane into the first half of rl[j]
                                    const uint2 q = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(a.res + (ok ? opix[m] * a.res_ld + 2 * a.res_split : 0)) + (ok ? x8Offset(co) : 0));
                                    rl[j] = __builtin_bit_cast(half8, make_uint4(q.x, q.y, 0u, 0u));
                                } else rl[j] = *reinterpret_cast<const half8*>(a.res + ((ok && splitRes) ? opix[m] * a.res_ld + a.res_split + co : 0));
                            }
                        }
#pragma unroll
                        for (int j = 0; j < RS; ++j) {
                            const int b = b0 + j, m = b / TP, tp = b % TP, co = cbase + tp * 32 + cg8;
                            if (2 * tp >= ctn) continue;
                            floatx4 X = acc[2 * tp][m], Y = acc[2 * tp + 1][m];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(X[i]), __float_as_uint(Y[i]), false, false);
                                X[i] = __uint_as_float(sw[0]); Y[i] = __uint_as_float(sw[1]);
                            }
                            if (!valid[m] || co >= a.Cout) continue;
                            float v[8] = {X[0], X[1], X[2], X[3], Y[0], Y[1], Y[2], Y[3]};
                            if (hasRes && resX8) {
                                const uint4 q = __builtin_bit_cast(uint4, rl[j]);
                                const unsigned w[2] = {q.x, q.y};
                                float lo[8]; x8DecodeLo<8>(w, lo);
#pragma unroll
                                for (int i = 0; i < 8; ++i) v[i] += (float)rh[j][i] + lo[i];
                            } else if (hasRes) {
#pragma unroll
                                for (int i = 0; i < 8; ++i) v[i] += splitRes ? (float)rh[j][i] + (float)rl[j][i] : (float)rh[j][i];
                            }
                            if (a.relu) {
#pragma unroll
                                for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
                            }
                            storeHalf8<true>(a, v, opix[m], co);
                        }
                    }
                } else
#pragma unroll
                for (int b0 = 0; b0 < NBLK; b0 += RB) {
                    half8 rv[RB];
                    if (hasRes) {
#pragma unroll
                        for (int j = 0; j < RB; ++j) {
                            const int b = b0 + j, m = b / TP, tp = b % TP, co = cbase + tp * 32 + cg8;
                            const bool ok = valid[m] && co < a.Cout && 2 * tp < ctn;
                            rv[j] = *reinterpret_cast<const half8*>(a.res + (ok ? opix[m] * a.res_ld + co : 0));
                        }
                    }
#pragma unroll
                    for (int j = 0; j < RB; ++j) {
                        const int b = b0 + j, m = b / TP, tp = b % TP, co = cbase + tp * 32 + cg8;
                        if (2 * tp >= ctn) continue;
                        floatx4 X = acc[2 * tp][m], Y = acc[2 * tp + 1][m];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(X[i]), __float_as_uint(Y[i]), false, false);
                            X[i] = __uint_as_float(sw[0]); Y[i] = __uint_as_float(sw[1]);
                        }
                        if (!valid[m] || co >= a.Cout) continue;
                        float v[8] = {X[0], X[1], X[2], X[3], Y[0], Y[1], Y[2], Y[3]};
                        if (hasRes) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] += (float)rv[j][i];
                        }
                        half8 h;
#pragma unroll
                        for (int i = 0; i < 8; ++i) h[i] = (_Float16)(a.relu ? fmaxf(v[i], 0.f) : v[i]);
                        *reinterpret_cast<half8*>(static_cast<_Float16*>(a.out) + opix[m] * a.out_ld + a.out_coff + co) = h;
                    }
                }
            } else {                                                  // (not a layer of this network)
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int oy = y0 + RW * wave + (m >> 1), ox = x0 + (m & 1) * 16 + r;
                    if (!(oy < a.Ho && ox < a.Wo)) continue;
                    const size_t opix = (size_t)((bimg * a.Ho + oy) * a.up + dy) * Wout + (ox * a.up + dx);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        if (ct < ctn) convStore<false, SPL>(a, acc[ct][m], opix, cbase + ct * 16 + 4 * g);
                }
            }
        }
        mark();
        if (!have_next) break;
        // (the epilogue's loads and stores retire at the first slab end of the next item: nothing else is in flight)
        item = nitem; y0 = ny0; x0 = nx0; chunk = nch; bimg = nbimg;
    }
}

// -------------------------------------------------------------------------------------
// Narrow 3 x 3 layers whose weights are BLOCK-DIAGONAL over 64-channel phases (round 2): the CenterHead's five output
// convolutions (64 -> 2 / 1 / 3 / 2 / 10, src/dsvt-ai-trt.cpp:1378-1468) arrive as ONE 320 -> 18 layer whose output channel n reads
// only the 64 input channels of its head.  The halo kernel computes it densely: per 8-row x 32-pixel item 218 KB of halo (five phases)
// and 184 KB of weights through LDS, 1.4 GB of LDS-DMA per four-frame launch at the ~21 GB/s per CU that path delivers: 271 us.
// Here a workgroup belongs to ONE phase (= head) for the whole launch: its 18 fragment rows of weights (one 16-channel tile x 9 taps
// x 2 k-steps, 18 KB) are resident, and per item only that phase's halo (51 KB with the row padding) streams, double buffered -- 0.9 GB per launch.
// A zero weight contributes an exact +0 to an fp32 sum, so the outputs are the dense kernel's bits.
__global__ void __launch_bounds__(512, 1)
conv3x3_grouped_narrow_kernel(ConvArgs a, const _Float16* __restrict__ Wg, const int* __restrict__ chanTab, const _Float16* __restrict__ zeros,
                              int tilesX, int tilesY, int NCC)
{
    constexpr int TH = 8, HS = HHS, HH = TH + 2, HWU = HTW + 2, HBYTES = HH * HS * 128, NI = HH * HS / 8, HPW = (NI + TH - 1) / TH;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * HBYTES + 18 * 1024];            // halo[2] | the phase's weights
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int pg = wave >> 1, m0 = 2 * (wave & 1);                      // rows 2 pg, 2 pg + 1; pixel tiles m0, m0 + 1 of the row pair's four
    const int cc = (int)blockIdx.x % NCC, j = (int)blockIdx.x / NCC, nj = ((int)gridDim.x - cc + NCC - 1) / NCC;
    const int ntile = a.nb * tilesY * tilesX;
    if (j >= ntile) return;
    for (int u = wave; u < 18; u += TH)
        __builtin_amdgcn_global_load_lds((glds_src_t)(Wg + (((size_t)cc * 18 + u) * 64 + lane) * 8), (glds_dst_t)(smem + 2 * HBYTES + u * 1024), 16, 0, 0);
    int ch[4]; float bs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ch[i] = chanTab[cc * 16 + 4 * g + i];
        bs[i] = (ch[i] >= 0 && a.bias) ? a.bias[ch[i]] : 0.f;
    }
    int hpos[HPW], goff[HPW];
#pragma unroll
    for (int q = 0; q < HPW; ++q) {
        const int lp = 8 * (wave + q * TH) + (lane >> 3), slot = lane & 7;
        const int hy = lp / HS, hx = lp - hy * HS;
        hpos[q] = (hy << 16) | (hx << 4) | (slot ^ (hx & 7));
    }
    auto decode = [&](int t, int& yy, int& xx, int& bb) {
        const int per = tilesY * tilesX;
        bb = t / per; t -= bb * per;
        yy = (t / tilesX) * TH; xx = (t % tilesX) * HTW;
    };
    auto setup = [&](int yy, int xx, int bb) {
#pragma unroll
        for (int q = 0; q < HPW; ++q) {
            const int hx = (hpos[q] >> 4) & 0xfff;
            const int gy = yy - 1 + (hpos[q] >> 16), gx = xx - 1 + hx;
            const bool ok = hx < HWU && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            goff[q] = ok ? ((bb * a.H + gy) * a.W + gx) * a.Cin + cc * 64 + (hpos[q] & 15) * 8 : -1;
        }
    };
    auto haloRequests = [&](int hb) {
#pragma unroll
        for (int q = 0; q < HPW; ++q)
            if (NI % TH == 0 || wave + q * TH < NI) {
                const _Float16* src = goff[q] >= 0 ? a.in + goff[q] : zeros;
                __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(smem + hb * HBYTES + (wave + q * TH) * 1024), 16, 0, 0);
            }
    };
    int pbase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) pbase[m] = ((2 * pg + ((m0 + m) >> 1)) * HS + ((m0 + m) & 1) * 16 + r) * 128;
    int swz[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) swz[kx] = (g ^ ((r + kx) & 7)) << 4;
    const unsigned char* wres = smem + 2 * HBYTES + (lane << 4);

    int t = j, y0, x0, bimg;
    decode(t, y0, x0, bimg);
    setup(y0, x0, bimg);
    haloRequests(0);
    slabBarrier(0);
    int hb = 0;
    for (;;) {
        const int tn = t + nj;
        const bool have_next = tn < ntile;
        int ny0 = 0, nx0 = 0, nb_ = 0;
        if (have_next) { decode(tn, ny0, nx0, nb_); setup(ny0, nx0, nb_); haloRequests(hb ^ 1); }
        floatx4 acc[2] = {floatx4{bs[0], bs[1], bs[2], bs[3]}, floatx4{bs[0], bs[1], bs[2], bs[3]}};      // (the sums start from the bias, like the dense kernel's)
        const unsigned char* hbp = smem + hb * HBYTES;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky, toff = (ky * HS + kx) * 128, sw = swz[kx];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const half8 A = *reinterpret_cast<const half8*>(wres + ((tap * 2 + ks) << 10));
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const half8 B = *reinterpret_cast<const half8*>(hbp + pbase[m] + toff + (sw ^ (ks << 6)));
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc[m], 0, 0, 0);
                }
            }
        }
        if (have_next) slabBarrier(0);                               // the next halo has landed; everyone is done with this one
        // (the stores after the wait: they drain under the next item's MFMAs instead of being awaited with the halo.  The same loop with the
        // halo staged through registers, one or two items ahead with counted waits: 200 / 206 us against 197 -- the floor of this tiling is
        // its halo traffic, 10 x 34 pixels read per 8 x 32 written)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int oy = y0 + 2 * pg + ((m0 + m) >> 1), ox = x0 + ((m0 + m) & 1) * 16 + r;
            if (!(oy < a.Ho && ox < a.Wo)) continue;
            const size_t opix = (size_t)(bimg * a.Ho + oy) * a.Wo + ox;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (ch[i] < 0) continue;
                float v = acc[m][i];
                if (a.relu) v = fmaxf(v, 0.f);
                if (a.out_f32) static_cast<float*>(a.out)[opix * a.out_ld + a.out_coff + ch[i]] = v;
                else static_cast<_Float16*>(a.out)[opix * a.out_ld + a.out_coff + ch[i]] = (_Float16)v;
            }
        }
        if (!have_next) break;
        hb ^= 1; t = tn; y0 = ny0; x0 = nx0; bimg = nb_;
    }
}

// The same layer at fp32 grade (round 4): three fp16 products over a split input [hi | lo | x8] of C real channels (a.Cin = 3 C; the weights are the
// host's [w_hi | w_hi | w_lo] rows).  On the dense halo kernel the launch walks fifteen phases -- the third plane's re-read plane 0 -- of which every
// output channel needs three: 2.27 GB fetched and 685 us per four frames (one frame 212 us).  Here a workgroup keeps ONE head's w_hi AND w_lo tiles
// (36 KB) and per item streams that head's hi halo (buffer 0) and lo halo (buffer 1), 102 KB: the hi halo serves hi x w_hi and hi x w_lo, then the lo
// halo lo x w_hi, and each buffer is refilled for the next item as soon as its products are done (the hi request leaves under the lo products, the
// lo request under the next item's hi products).  Summation order per output: (hi w_hi, hi w_lo, lo w_hi) by tap -- the dense kernel's is (hi w_hi,
// lo w_hi, hi w_lo): the same products, fp32 rounding apart (2e-6 of scale).
__global__ void __launch_bounds__(512, 1)
conv3x3_grouped_narrow_split_kernel(ConvArgs a, const _Float16* __restrict__ Wg, const int* __restrict__ chanTab, const _Float16* __restrict__ zeros,
                                    int tilesX, int tilesY, int NCC)
{
    constexpr int TH = 8, HS = HHS, HH = TH + 2, HWU = HTW + 2, HBYTES = HH * HS * 128, NI = HH * HS / 8, HPW = (NI + TH - 1) / TH;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * HBYTES + 36 * 1024];            // hi halo | lo halo | the head's w_hi, w_lo tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int pg = wave >> 1, m0 = 2 * (wave & 1);                      // rows 2 pg, 2 pg + 1; pixel tiles m0, m0 + 1 of the row pair's four
    const int cc = (int)blockIdx.x % NCC, j = (int)blockIdx.x / NCC, nj = ((int)gridDim.x - cc + NCC - 1) / NCC;
    const int ntile = a.nb * tilesY * tilesX;
    if (j >= ntile) return;
    const int C = a.Cin / 3;                                           // real channels: plane p starts at channel p C
    for (int u = wave; u < 36; u += TH)
        __builtin_amdgcn_global_load_lds((glds_src_t)(Wg + (((size_t)cc * 36 + u) * 64 + lane) * 8), (glds_dst_t)(smem + 2 * HBYTES + u * 1024), 16, 0, 0);
    int ch[4]; float bs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ch[i] = chanTab[cc * 16 + 4 * g + i];
        bs[i] = (ch[i] >= 0 && a.bias) ? a.bias[ch[i]] : 0.f;
    }
    int hpos[HPW], goff[HPW];
#pragma unroll
    for (int q = 0; q < HPW; ++q) {
        const int lp = 8 * (wave + q * TH) + (lane >> 3), slot = lane & 7;
        const int hy = lp / HS, hx = lp - hy * HS;
        hpos[q] = (hy << 16) | (hx << 4) | (slot ^ (hx & 7));
    }
    auto decode = [&](int t, int& yy, int& xx, int& bb) {
        const int per = tilesY * tilesX;
        bb = t / per; t -= bb * per;
        yy = (t / tilesX) * TH; xx = (t % tilesX) * HTW;
    };
    auto setup = [&](int yy, int xx, int bb) {
#pragma unroll
        for (int q = 0; q < HPW; ++q) {
            const int hx = (hpos[q] >> 4) & 0xfff;
            const int gy = yy - 1 + (hpos[q] >> 16), gx = xx - 1 + hx;
            const bool ok = hx < HWU && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            goff[q] = ok ? ((bb * a.H + gy) * a.W + gx) * a.Cin + cc * 64 + (hpos[q] & 15) * 8 : -1;
        }
    };
    auto haloRequests = [&](int plane) {                               // plane 0 (hi) -> buffer 0, plane 1 (lo) -> buffer 1; of the item `goff` describes
#pragma unroll
        for (int q = 0; q < HPW; ++q)
            if (NI % TH == 0 || wave + q * TH < NI) {
                const _Float16* src = goff[q] >= 0 ? a.in + goff[q] + plane * C : zeros;
                __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(smem + plane * HBYTES + (wave + q * TH) * 1024), 16, 0, 0);
            }
    };
    int pbase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) pbase[m] = ((2 * pg + ((m0 + m) >> 1)) * HS + ((m0 + m) & 1) * 16 + r) * 128;
    int swz[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) swz[kx] = (g ^ ((r + kx) & 7)) << 4;
    const unsigned char* wres = smem + 2 * HBYTES + (lane << 4);
    // products of one halo buffer with one 18-row weight tile
    auto products = [&](const unsigned char* hbp, const unsigned char* wt, floatx4 (&acc)[2]) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky, toff = (ky * HS + kx) * 128, sw = swz[kx];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const half8 A = *reinterpret_cast<const half8*>(wt + ((tap * 2 + ks) << 10));
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const half8 B = *reinterpret_cast<const half8*>(hbp + pbase[m] + toff + (sw ^ (ks << 6)));
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc[m], 0, 0, 0);
                }
            }
        }
    };

    int t = j, y0, x0, bimg;
    decode(t, y0, x0, bimg);
    setup(y0, x0, bimg);
    haloRequests(0); haloRequests(1);
    slabBarrier(0);
    for (;;) {
        const int tn = t + nj;
        const bool have_next = tn < ntile;
        int ny0 = 0, nx0 = 0, nb_ = 0;
        floatx4 acc[2] = {floatx4{bs[0], bs[1], bs[2], bs[3]}, floatx4{bs[0], bs[1], bs[2], bs[3]}};      // (the sums start from the bias, like the dense kernel's)
        products(smem, wres, acc);                                    // hi x w_hi
        products(smem, wres + 18 * 1024, acc);                        // hi x w_lo
        // everyone is done with the hi halo: the next item's goes out now, under the lo products (the lo halo of THIS item has landed: it was
        // requested an item ago -- or with the hi halo, for the first item -- and is awaited here)
        slabBarrier(0);
        if (have_next) { decode(tn, ny0, nx0, nb_); setup(ny0, nx0, nb_); haloRequests(0); }
        products(smem + HBYTES, wres, acc);                           // lo x w_hi
        // the stores, then (after everyone is done with the lo halo) the next item's lo request; its hi halo is awaited with the barrier below
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int oy = y0 + 2 * pg + ((m0 + m) >> 1), ox = x0 + ((m0 + m) & 1) * 16 + r;
            if (!(oy < a.Ho && ox < a.Wo)) continue;
            const size_t opix = (size_t)(bimg * a.Ho + oy) * a.Wo + ox;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (ch[i] < 0) continue;
                float v = acc[m][i];
                if (a.relu) v = fmaxf(v, 0.f);
                if (a.out_f32) static_cast<float*>(a.out)[opix * a.out_ld + a.out_coff + ch[i]] = v;
                else static_cast<_Float16*>(a.out)[opix * a.out_ld + a.out_coff + ch[i]] = (_Float16)v;
            }
        }
        if (!have_next) break;
        // (the hi halo of the next item has landed -- vmcnt(0) also waits for this wave's stores; then everyone is done with the lo halo)
        slabBarrier(0);
        haloRequests(1);
        t = tn; y0 = ny0; x0 = nx0; bimg = nb_;
    }
}

// -------------------------------------------------------------------------------------
// 3 x 3 stride-1 layers with 64 INPUT channels and a multiple of 64 output channels (round 2): the CenterHead's five stems as one
// 64 -> 320 layer.  On conv_wide_kernel<8, 4, 36, 2, 2> an 8-row x 32-pixel x 128-channel item streams 147 KB of weights beside 46 KB of
// halo: 2 GB of LDS-DMA per four-frame launch, which is what bounds it (415-439 us; its MFMAs would take ~215).  Here a workgroup owns
// 64 output channels for the whole launch -- 72 fragment rows of weights (four channel tiles x 9 taps x 2 k-steps, 72 KB) resident --
// and walks 8-row x 32-pixel tiles of which only the halo moves: 0.9 GB per launch.  LDS holds ONE halo buffer beside the weights
// (51 + 72 KB); the next item's halo waits in seven staging registers per lane, requested before this item's 144 MFMAs per wave and
// written to LDS after them (through registers a one-item lead costs no second buffer; measured equal to a DMA double buffer on the
// narrow kernel above).  Wave (pg, half) = rows 2 pg, 2 pg + 1 x two pixel tiles x all four channel tiles (0.75 fragment reads per MFMA).
// Round 3: SIXTEEN-row items (18 x 34 halo pixels of 128 bytes = 78 KB beside the 72 KB of weights: 151 KB), wave w = rows 2w, 2w + 1 = FOUR pixel
// tiles x all four channel tiles: 4 + 4 fragment reads per 16 MFMAs = 0.5 per MFMA.  With eight-row items (two pixel tiles per wave, 0.75 reads
// per MFMA) the kernel sat exactly on the LDS port -- a 1 KB fragment read occupies it for 8 cycles, as long as an MFMA occupies one of the four
// matrix pipes, so r reads per MFMA cap the CU at 0.25 / r of its MFMA peak: 0.33 x 2.5 = 0.83 PFLOP/s, measured 0.80.
constexpr int C64_TH = 16, C64_HS = 34;
__global__ void __launch_bounds__(512, 1)
conv3x3_c64_resident_kernel(ConvArgs a, const _Float16* __restrict__ Wp, int NCT, const _Float16* __restrict__ zeros, int tilesX, int tilesY, int NTY)
{
    constexpr int TH = C64_TH, NWV = 8, HS = C64_HS, HH = TH + 2, HWU = HTW + 2, NI = (HH * HS + 7) / 8, HBYTES = NI * 1024, HPW = (NI + NWV - 1) / NWV;
    __shared__ __attribute__((aligned(16))) unsigned char smem[HBYTES + 72 * 1024 + 1024];        // halo | weights [k-step][4 tiles] | bias (64 floats)
    constexpr int W_OFF = HBYTES, BIAS_OFF = HBYTES + 72 * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int ty = (int)blockIdx.x % NTY, j = (int)blockIdx.x / NTY, nj = ((int)gridDim.x - ty + NTY - 1) / NTY;
    const int ntile = a.nb * tilesY * tilesX;
    if (j >= ntile) return;
    // the halo image's rows [q = tap * 2 + ks][16-channel tile ct of NCT]: this type's four tiles of every k-step
    for (int u = wave; u < 72; u += NWV) {
        const int q = u >> 2, ct = u & 3;
        __builtin_amdgcn_global_load_lds((glds_src_t)(Wp + (((size_t)q * NCT + ty * 4 + ct) * 64 + lane) * 8), (glds_dst_t)(smem + W_OFF + u * 1024), 16, 0, 0);
    }
    if (wave == 0) {                                                 // this type's 64 bias values (any valid address without a bias: unused)
        const float* src = a.bias ? a.bias + ((ty * 64) % a.Cout) + (lane < 16 ? lane * 4 : 0) : reinterpret_cast<const float*>(Wp);
        __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(smem + BIAS_OFF), 16, 0, 0);
    }
    int hpos[HPW];
#pragma unroll
    for (int q = 0; q < HPW; ++q) {
        const int lp = 8 * (wave + q * NWV) + (lane >> 3), slot = lane & 7;
        const int hy = lp / HS, hx = lp - hy * HS;
        hpos[q] = (hy << 16) | (hx << 4) | (slot ^ (hx & 7));
    }
    auto decode = [&](int t, int& yy, int& xx, int& bb) {
        const int per = tilesY * tilesX;
        bb = t / per; t -= bb * per;
        yy = (t / tilesX) * TH; xx = (t % tilesX) * HTW;
    };
    half8 stage[HPW];
    auto haloLoad = [&](int t) {                                     // tile t (clamped: no branch around a load) -> staging registers
        int yy, xx, bb;
        decode(t < ntile ? t : ntile - 1, yy, xx, bb);
        int goff[HPW];
#pragma unroll
        for (int q = 0; q < HPW; ++q) {
            const int hy = hpos[q] >> 16, hx = (hpos[q] >> 4) & 0xfff;
            const int gy = yy - 1 + hy, gx = xx - 1 + hx;
            const bool ok = hy < HH && hx < HWU && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            goff[q] = ok ? ((bb * a.H + gy) * a.W + gx) * a.Cin + (hpos[q] & 15) * 8 : -1;
        }
#pragma unroll
        for (int q = 0; q < HPW; ++q)
            if (NI % NWV == 0 || wave + q * NWV < NI) stage[q] = *reinterpret_cast<const half8*>(goff[q] >= 0 ? a.in + goff[q] : zeros);
    };
    auto haloWrite = [&]() {
#pragma unroll
        for (int q = 0; q < HPW; ++q)
            if (NI % NWV == 0 || wave + q * NWV < NI) *reinterpret_cast<half8*>(smem + (wave + q * NWV) * 1024 + lane * 16) = stage[q];
    };
    int pbase[4];                                                    // pixel tile m of this wave: row 2 wave + (m >> 1), columns 16 (m & 1) ...
#pragma unroll
    for (int m = 0; m < 4; ++m) pbase[m] = ((2 * wave + (m >> 1)) * HS + (m & 1) * 16 + r) * 128;
    int swz[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) swz[kx] = (g ^ ((r + kx) & 7)) << 4;
    const unsigned char* wres = smem + W_OFF + (lane << 4);
    const int n0 = ty * 64, sub = n0 / a.Cout, cbase = n0 - sub * a.Cout;       // (no pixel shuffle on these layers: sub = 0)

    int t = j;
    haloLoad(t);
    haloWrite();
    slabBarrier(0);                                                  // halo of the first item, weights and bias
    for (;;) {
        const int tn = t + nj;
        const bool have_next = tn < ntile;
        haloLoad(tn);                                                // in flight under this item's MFMAs
        int y0, x0, bimg;
        decode(t, y0, x0, bimg);
        floatx4 acc[4][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const floatx4 b4 = *reinterpret_cast<const floatx4*>(smem + BIAS_OFF + (ct * 16 + 4 * g) * 4);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[ct][m] = a.bias ? b4 : floatx4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky, toff = (ky * HS + kx) * 128, sw = swz[kx];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 A[4], B[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) A[ct] = *reinterpret_cast<const half8*>(wres + (((tap * 2 + ks) * 4 + ct) << 10));
#pragma unroll
                for (int m = 0; m < 4; ++m) B[m] = *reinterpret_cast<const half8*>(smem + pbase[m] + toff + (sw ^ (ks << 6)));
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[ct][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ct], B[m], acc[ct][m], 0, 0, 0);
            }
        }
        __syncthreads();                                             // everyone is done with this halo
        if (have_next) haloWrite();
        __syncthreads();
        // (the stores after the barriers: they drain under the next item's MFMAs)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int oy = y0 + 2 * wave + (m >> 1), ox = x0 + (m & 1) * 16 + r;
            const bool valid = oy < a.Ho && ox < a.Wo;
            const size_t opix = valid ? (size_t)(bimg * a.Ho + oy) * a.Wo + ox : 0;
            convStoreWide<false>(a, acc[0][m], acc[1][m], valid, opix, cbase, g);
            convStoreWide<false>(a, acc[2][m], acc[3][m], valid, opix, cbase + 32, g);
        }
        if (!have_next) break;
        t = tn;
    }
}

// -------------------------------------------------------------------------------------
// 1 x 1 stride-1 layers (the shortcut of the first block and the three deblocks = 1 x 1 + pixel shuffle) as a streaming GEMM
// with the weights RESIDENT in LDS (round 2).  The halo kernel above treats a 1 x 1 layer like a 3 x 3 one: per 8-row x 32-pixel x
// 128-channel item it streams the input tile and the weight slabs through LDS -- for the 256 -> 16 x 128 deblock that is the same
// 140 KB of input sixteen times (once per 128-column chunk) plus 64 KB of weights per item, 737 MB of LDS-DMA per four-frame
// launch at the ~25 GB/s a CU's request path delivers: 141 us for 252 MB of HBM traffic.  Here a CU keeps one column GROUP of the
// weights (up to 16 column tiles x Cin: <= 128 KB, DMA'd once from the halo kernel's fragment image) and its waves walk
// 16-pixel tiles of the flat pixel list independently, like linear_f16_resident_kernel: rows straight from global into B
// fragments (next tile's rows in flight while this one computes), no barrier after the prologue; a 128-column stage of a group is
// one sub-pixel (dy, dx) of the pixel shuffle, so the epilogue is the wide one (bias, ReLU, fp16, 16-byte stores).
// Needs Cout == 128, Cin in {128, 192, 256}, fp16 output with 16-byte alignment, no residual.
constexpr int C1_NW = 8;
// MT = 16-pixel tiles a wave works on together.  One tile: every MFMA reads its own 1 KB weight fragment from LDS (8 cycles of the CU's LDS
// port for an 8-cycle MFMA -- the 256 -> 16 x 128 deblock ran at half the rate of either).  Two tiles: a fragment feeds two MFMAs (the rows of
// both tiles in registers: 2 x 2 x KSTEPS x 4 registers, which is why it is the ONE-workgroup-per-CU configuration, KSTEPS = 8 -- 129 KB of weights).
template <int KSTEPS, bool STRIDED, int MT = 1>       // STRIDED: stride 2 (the shortcuts of the second and third stage): output pixel (y, x) reads input pixel (2y, 2x)
__global__ void __launch_bounds__(64 * C1_NW, MT == 1 ? 2 : 1)
conv1x1_resident_kernel(ConvArgs a, const _Float16* __restrict__ Wp, int NCT, int ngroup, int CTG)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[KSTEPS * 16 * 1024 + 1024];        // [k-step][column tile of the group] | bias
    constexpr int BIAS_OFF = KSTEPS * 16 * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int type = (int)blockIdx.x % ngroup, j = (int)blockIdx.x / ngroup, nj = (int)gridDim.x / ngroup;
    for (int u = wave; u < KSTEPS * CTG; u += C1_NW) {
        const int q = u / CTG, t = u - q * CTG;
        __builtin_amdgcn_global_load_lds((glds_src_t)(Wp + (((size_t)q * NCT + type * CTG + t) * 64 + lane) * 8), (glds_dst_t)(smem + u * 1024), 16, 0, 0);
    }
    if (wave == 0) {                                                 // Cout <= 256 floats (without a bias: any valid address, not used below)
        const float* src = a.bias ? a.bias + (lane * 4 < a.Cout ? lane * 4 : 0) : reinterpret_cast<const float*>(Wp);
        __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(smem + BIAS_OFF), 16, 0, 0);
    }
    const int HW = a.Ho * a.Wo, NPIX = a.nb * HW, ntile = (NPIX + 16 * MT - 1) / (16 * MT), step = nj * C1_NW;      // OUTPUT pixels (before the pixel shuffle), MT x 16 per wave step
    const float invHW = 1.0f / (float)HW, invW = 1.0f / (float)a.Wo;
    // output pixel -> (image, y, x): float quotients corrected by one step (exact for these sizes; belt and braces)
    auto split = [&](int pc, int& b, int& y, int& xq) {
        b = (int)(((float)pc + 0.5f) * invHW); b -= (b * HW > pc); b += ((b + 1) * HW <= pc);
        const int rem = pc - b * HW;
        y = (int)(((float)rem + 0.5f) * invW); y -= (y * a.Wo > rem); y += ((y + 1) * a.Wo <= rem);
        xq = rem - y * a.Wo;
    };
    auto loadRows = [&](int t, half8 (&x)[MT][KSTEPS]) {
        t = t < ntile ? t : ntile - 1;                               // (past the end: the last tile again, no branch around a load)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int p = (t * MT + m) * 16 + r, pc = p < NPIX ? p : NPIX - 1;
            size_t ipix = (size_t)pc;
            if (STRIDED) { int b, y, xq; split(pc, b, y, xq); ipix = (size_t)(b * a.H + y * a.stride) * a.W + xq * a.stride; }
            const _Float16* src = a.in + ipix * a.Cin + g * 8;
#pragma unroll
            for (int q = 0; q < KSTEPS; ++q) x[m][q] = *reinterpret_cast<const half8*>(src + q * 32);
        }
    };
    const unsigned char* slot = smem + lane * 16;
    const int nstage = CTG >> 3;                                     // 128-column stages of this group
    const int Wout = a.Wo * a.up;
    int sdy[2], sdx[2], scb[2];                                      // per stage of this group (<= 2): sub-pixel of the shuffle, first channel
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const int n0 = (type * nstage + st) * 128, sub = n0 / a.Cout;
        sdy[st] = sub / a.up; sdx[st] = sub - sdy[st] * a.up; scb[st] = n0 - sub * a.Cout;
    }
    auto tile = [&](int t, const half8 (&x)[MT][KSTEPS]) {
        bool valid[MT]; int b[MT], y[MT], xq[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int p = (t * MT + m) * 16 + r;
            valid[m] = p < NPIX;
            split(valid[m] ? p : 0, b[m], y[m], xq[m]);
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (st >= nstage) break;
            const int dy = sdy[st], dx = sdx[st], cbase = scb[st];
            const unsigned char* sp = slot + (st * 8) * 1024;
            constexpr int UB = 8 / MT;                               // column tiles per accumulator block (the rows of MT pixel tiles fill the registers)
#pragma unroll
            for (int u0 = 0; u0 < 8; u0 += UB) {
                floatx4 acc[MT][UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const floatx4 b4 = *reinterpret_cast<const floatx4*>(smem + BIAS_OFF + (cbase + (u0 + u) * 16 + 4 * g) * 4);
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][u] = a.bias ? b4 : floatx4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int q = 0; q < KSTEPS; ++q)
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const half8 wf = *reinterpret_cast<const half8*>(sp + (size_t)(q * CTG + u0 + u) * 1024);
#pragma unroll
                        for (int m = 0; m < MT; ++m) acc[m][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, x[m][q], acc[m][u], 0, 0, 0);
                    }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const size_t opix = (size_t)((b[m] * a.Ho + y[m]) * a.up + dy) * Wout + (xq[m] * a.up + dx);
#pragma unroll
                    for (int u = 0; u < UB; u += 2) convStoreWide<false>(a, acc[m][u], acc[m][u + 1], valid[m], opix, cbase + (u0 + u) * 16, g);
                }
            }
        }
    };
    int tt = j * C1_NW + wave;
    if constexpr (MT == 1) {
        half8 xa[MT][KSTEPS], xb[MT][KSTEPS];
        loadRows(tt, xa);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the weights (and the first rows) have landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        while (tt < ntile) {
            const int tn = tt + step;
            loadRows(tn, xb);
            tile(tt, xa);
            if (tn >= ntile) break;
            tt = tn + step;
            loadRows(tt, xa);
            tile(tn, xb);
        }
    } else {
        // two pixel tiles: their rows are 2 x KSTEPS x 4 registers, a second set does not fit -- the wave that shares the SIMD covers the load
        half8 xa[MT][KSTEPS];
        loadRows(tt, xa);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        while (tt < ntile) {
            tile(tt, xa);
            tt += step;
            if (tt < ntile) loadRows(tt, xa);
        }
    }
}

// The same for the fp32-grade stage (round 4): the 1 x 1 stride-1 layers on the fp16 + fp8 K loop (packMX1x1 image, conv_halo_kernel<.., MX>'s arithmetic
// in its order: C / 32 fp16 k-steps of the hi plane, then C / 64 e4m3 K = 128 steps of the x8 plane -- the same bits) with ONE 128-column stage resident per
// CU: its fp16 rows (C / 32 KB per column tile) and cross rows (C / 32 KB again), C / 2 KB in all.  The halo kernel streamed, per 8 x 32-pixel x 128-column
// item, the input tile (hi AND x8: 1 KB per pixel at C = 256) and 137 KB of weights through LDS: the 256 -> 16 x 128 deblock fetched the same 262 KB of
// input sixteen times, 6 MB of LDS-DMA per CU and launch, 352 us per four frames for 0.5 GB of HBM traffic.  Here the waves walk 16-pixel tiles with their
// rows (hi fragments + the two 16-byte x8 chunks per 64 channels) straight from global, next tile in flight under the current one; the workgroups of a
// column group's sixteen (four) siblings that read the same pixels sit on one XCD.
// KQ = fp16 k-steps per PASS, NH = passes over a pixel tile (C = 32 KQ NH; accumulators kept between the passes of a tile).  A pass needs its rows in
// registers -- 4 KQ for the hi fragments + 4 KQ for the x8 chunks -- and the NEXT pass's rows in flight, beside 32 accumulator and 64 fragment registers and the
// split epilogue: <4, 1> for C = 128, <2, 3> for C = 192, <4, 2> for C = 256.  The one-pass forms of the larger layers were built first: <6, 1> with one set of
// rows (a second spills): 256 us for the 192 -> 128 shortcut against 225; <8, 1>: 77-107 spilled registers in every arrangement tried (plain, or the two planes
// taking turns), 490-690 us for the 256 -> 16 x 128 deblock against the halo kernel's 352 and the two-pass form's 215.  (Several passes sum per pass
// [main, cross]: the same products as the halo kernel's [all main, all cross] in another fp32 order; <4, 1> is the halo kernel's order.)
template <int KQ, int NH>
__global__ void __launch_bounds__(64 * C1_NW, 1)
conv1x1_resident_mx_kernel(ConvArgs a, const _Float16* __restrict__ Wp, int NCT, int ngroup)
{
    constexpr int NPM = KQ / 2, CTG = 8, KQT = KQ * NH, NPT = NPM * NH;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KQT * CTG * 1024 + 1024];     // [fp16 k-step][tile] | [cross phase][tile][2] | bias (512 B) + scale bytes (128 B)
    constexpr int CROSS_OFF = KQT * CTG * 1024, BIAS_OFF = 2 * KQT * CTG * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    // workgroup -> (column group, row stream): the ngroup workgroups of a stream on one XCD (workgroups go to the XCDs round-robin)
    const int nx = (int)gridDim.x / 8, xcd = (int)blockIdx.x % 8, sl = (int)blockIdx.x / 8;        // (the grid is a multiple of 8 ngroup)
    const int type = sl % ngroup, j = xcd * (nx / ngroup) + sl / ngroup, nj = (int)gridDim.x / ngroup;
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(Wp);
    const size_t mainRows = (size_t)KQT * NCT;
    for (int u = wave; u < KQT * CTG; u += C1_NW) {                   // fp16 rows: [q][NCT] -> [q][CTG]
        const int q = u / CTG, t = u - q * CTG;
        __builtin_amdgcn_global_load_lds((glds_src_t)(Wb + (((size_t)q * NCT + type * CTG + t) * 64 + lane) * 16), (glds_dst_t)(smem + u * 1024), 16, 0, 0);
    }
    for (int u = wave; u < NPT * CTG * 2; u += C1_NW) {               // cross rows: [ph][NCT][2] -> [ph][CTG][2]
        const int ph = u / (CTG * 2), t2 = u - ph * CTG * 2;
        __builtin_amdgcn_global_load_lds((glds_src_t)(Wb + ((mainRows + ((size_t)ph * NCT + type * CTG) * 2 + t2) * 64 + lane) * 16), (glds_dst_t)(smem + CROSS_OFF + u * 1024), 16, 0, 0);
    }
    const int n0 = type * 128, sub = n0 / a.Cout, dy = sub / a.up, dx = sub - dy * a.up, cbase = n0 - sub * a.Cout;
    if (wave == 0) {                                                 // the stage's bias (128 floats) and scale bytes (128)
        const void* src = lane < 32 ? (a.bias ? static_cast<const void*>(a.bias + cbase + lane * 4) : static_cast<const void*>(Wp))
                                    : static_cast<const void*>(a.xscale + n0 + ((lane - 32) & 7) * 16);
        __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(smem + BIAS_OFF), 16, 0, 0);
    }
    const int HW = a.Ho * a.Wo, NPIX = a.nb * HW, ntile = (NPIX + 15) / 16, step = nj * C1_NW;      // OUTPUT pixels before the pixel shuffle = input pixels
    const float invHW = 1.0f / (float)HW, invW = 1.0f / (float)a.Wo;
    auto split = [&](int pc, int& b, int& y, int& xq) {
        b = (int)(((float)pc + 0.5f) * invHW); b -= (b * HW > pc); b += ((b + 1) * HW <= pc);
        const int rem = pc - b * HW;
        y = (int)(((float)rem + 0.5f) * invW); y -= (y * a.Wo > rem); y += ((y + 1) * a.Wo <= rem);
        xq = rem - y * a.Wo;
    };
    const int C = a.Cin / 3;
    const int tt0 = j * C1_NW + wave;
    const int nunit = tt0 < ntile ? ((ntile - tt0 + step - 1) / step) * NH : 0;      // (tile, pass) units of this wave: unit v = tile tt0 + (v / NH) step, pass v % NH
    struct Rows { half8 h[KQ]; intx8 x[NPM]; };
    auto loadRows = [&](int v, Rows& w) {
        v = v < nunit ? v : nunit - 1;                               // (past the end: the last unit again, no branch around a load)
        const int t = tt0 + (v / NH) * step, hp = v % NH;
        const int p = t * 16 + r, pc = p < NPIX ? p : NPIX - 1;
        const _Float16* src = a.in + (size_t)pc * a.Cin;
#pragma unroll
        for (int q = 0; q < KQ; ++q) w.h[q] = *reinterpret_cast<const half8*>(src + (hp * KQ + q) * 32 + g * 8);
        // x8 plane: per 64 channels 128 bytes = two 32-channel groups of [lo8 0..15 | hi8 0..15 | lo8 16..31 | hi8 16..31]; lane group g: chunks 4 (g >> 1) + (g & 1), + 2
        const unsigned char* xs_ = reinterpret_cast<const unsigned char*>(src + 2 * C) + hp * NPM * 128 + (4 * (g >> 1) + (g & 1)) * 16;
#pragma unroll
        for (int ph = 0; ph < NPM; ++ph) {
            const intx4 lo = *reinterpret_cast<const intx4*>(xs_ + ph * 128), hi = *reinterpret_cast<const intx4*>(xs_ + ph * 128 + 32);
            w.x[ph] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    const int Wout = a.Wo * a.up;
    int xsc[2] = {0, 0};                                             // scale byte of row (tile u, r) in byte u & 3 of xsc[u >> 2]
    floatx4 acc[CTG];
    auto unit = [&](int v, const Rows& w) {
        const int t = tt0 + (v / NH) * step, hp = NH == 1 ? 0 : v % NH;
        // (LDS addresses from an opaque copy of the lane offset: as loop invariants hipcc hoists the eight bias vectors -- 32 registers -- and every
        // fragment address out of the tile loop and spills them)
        int lo16 = lane * 16, g16 = g * 16;
        asm volatile("" : "+v"(lo16), "+v"(g16));
        const unsigned char* slot = smem + lo16;
        if (hp == 0) {
#pragma unroll
            for (int u = 0; u < CTG; ++u) {
                const floatx4 b4 = *reinterpret_cast<const floatx4*>(smem + BIAS_OFF + u * 64 + g16);
                acc[u] = a.bias ? b4 : floatx4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // KQ fp16 steps (eight 1 KB fragments) then 2 NPM half-steps of the fp8 phases (four column tiles x 2 KB): eight 16-byte reads per step, those of
        // step s + 1 issued before the MFMAs of step s (left alone hipcc hoists the reads of the whole tile: 500 spilled registers)
        constexpr int NSTEP_ = KQ + 2 * NPM;
        intx4 fb[2][8];
        auto loadStep = [&](int s_, intx4 (&f)[8]) {
            if (s_ < KQ) {
#pragma unroll
                for (int u = 0; u < 8; ++u) f[u] = *reinterpret_cast<const intx4*>(slot + ((hp * KQ + s_) * CTG + u) * 1024);
            } else {
                const int ph = hp * NPM + ((s_ - KQ) >> 1), u0 = ((s_ - KQ) & 1) * 4;
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] = *reinterpret_cast<const intx4*>(slot + CROSS_OFF + (((ph * CTG + u0) * 2) + i) * 1024);
            }
        };
        loadStep(0, fb[0]);
#pragma unroll
        for (int s_ = 0; s_ < NSTEP_; ++s_) {
            if (s_ + 1 < NSTEP_) loadStep(s_ + 1, fb[(s_ + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const intx4 (&f)[8] = fb[s_ & 1];
            if (s_ < KQ) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, f[u]), w.h[s_ < KQ ? s_ : 0], acc[u], 0, 0, 0);
            } else {
                const int ph = (s_ - KQ) >> 1, u0 = ((s_ - KQ) & 1) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const intx8 A = __builtin_shufflevector(f[2 * i], f[2 * i + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[u0 + i] = mfmaX8(u0 + i, A, w.x[ph < NPM ? ph : 0], acc[u0 + i], xsc[(u0 + i) >> 2]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (hp == NH - 1) {
            const int p = t * 16 + r;
            const bool valid = p < NPIX;
            int b, y, xq;
            split(valid ? p : 0, b, y, xq);
            const size_t opix = (size_t)((b * a.Ho + y) * a.up + dy) * Wout + (xq * a.up + dx);
#pragma unroll
            for (int u = 0; u < CTG; u += 2) {
                convStoreWide<false, true>(a, acc[u], acc[u + 1], valid, opix, cbase + u * 16, g);
                __builtin_amdgcn_sched_barrier(0);                      // (one pair of tiles at a time)
            }
        }
    };
    Rows xa;
    if (nunit > 0) loadRows(0, xa);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the weights (and the first rows) have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < CTG; ++u) xsc[u >> 2] |= (int)smem[BIAS_OFF + 512 + u * 16 + r] << (8 * (u & 3));
    if constexpr (KQ <= 4) {                                          // (two sets of rows in flight: 2 x 32 registers)
        Rows xb;
        for (int v = 0; v < nunit; v += 2) {
            loadRows(v + 1, xb);
            unit(v, xa);
            if (v + 1 >= nunit) break;
            loadRows(v + 2, xa);
            unit(v + 1, xb);
        }
    } else {                                                          // (one set of rows: kept for A/B builds of the one-pass forms)
        for (int v = 0; v < nunit; ++v) {
            unit(v, xa);
            if (v + 1 < nunit) loadRows(v + 1, xa);
        }
    }
}

// The 1 x 1 stride-1 layers of the fp32-grade stage on THREE fp16 products (round 5: the default head), same scheme: one 128-column stage of w_hi AND w_lo
// resident per CU (C / 2 KB), the waves walk 16-pixel tiles with their hi and lo rows straight from global (plane 0 and plane 1 of the [hi | lo | -] triple:
// the third plane is never read), per k-step 8 w_hi fragments -> 16 MFMAs (w_hi x hi, w_hi x lo), then 8 w_lo fragments -> 8 MFMAs (w_lo x hi).  The halo
// kernel walked 3 C / 64 phases per 8 x 32-pixel x 128-column item -- the hi plane through LDS twice, 8.4 C bytes of weights per column for every item --:
// 318-418 us per four-frame launch on these byte-bound layers (1.45 ms for the four of them).  Weights come from the halo kernel's fragment image of the
// host's [w_hi | w_hi | w_lo] rows: k-step q of plane 0 = w_hi, k-step 2 C / 32 + q = w_lo.  Summation order per k-step (hi w_hi, lo w_hi, hi w_lo) instead of
// per plane: 2^-22-grade either way (tests/test_conv_gpu.py::test_split_precision_1x1_resident).
template <int KQ, int NH>
__global__ void __launch_bounds__(64 * C1_NW, 1)
conv1x1_resident_split_kernel(ConvArgs a, const _Float16* __restrict__ Wp, int NCT, int ngroup)
{
    constexpr int CTG = 8, KQT = KQ * NH;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KQT * CTG * 1024 + 1024];     // w_hi [k-step][tile] | w_lo [k-step][tile] | bias (512 B)
    constexpr int LO_OFF = KQT * CTG * 1024, BIAS_OFF = 2 * KQT * CTG * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    // workgroup -> (column group, row stream): the ngroup workgroups of a stream on one XCD (workgroups go to the XCDs round-robin)
    const int nx = (int)gridDim.x / 8, xcd = (int)blockIdx.x % 8, sl = (int)blockIdx.x / 8;        // (the grid is a multiple of 8 ngroup)
    const int type = sl % ngroup, j = xcd * (nx / ngroup) + sl / ngroup, nj = (int)gridDim.x / ngroup;
    for (int u = wave; u < 2 * KQT * CTG; u += C1_NW) {               // rows [q][NCT] of plane 0 (w_hi) and of plane 2 (w_lo) -> [q][CTG]
        const int pl = u / (KQT * CTG), v = u - pl * KQT * CTG, q = v / CTG, t = v - q * CTG;
        __builtin_amdgcn_global_load_lds((glds_src_t)(Wp + (((size_t)(pl * 2 * KQT + q) * NCT + type * CTG + t) * 64 + lane) * 8), (glds_dst_t)(smem + u * 1024), 16, 0, 0);
    }
    const int n0 = type * 128, sub = n0 / a.Cout, dy = sub / a.up, dx = sub - dy * a.up, cbase = n0 - sub * a.Cout;
    if (wave == 0) {                                                 // the stage's bias (128 floats)
        const void* src = (a.bias && lane < 32) ? static_cast<const void*>(a.bias + cbase + lane * 4) : static_cast<const void*>(Wp);
        __builtin_amdgcn_global_load_lds((glds_src_t)src, (glds_dst_t)(smem + BIAS_OFF), 16, 0, 0);
    }
    const int HW = a.Ho * a.Wo, NPIX = a.nb * HW, ntile = (NPIX + 15) / 16, step = nj * C1_NW;      // OUTPUT pixels before the pixel shuffle = input pixels
    const float invHW = 1.0f / (float)HW, invW = 1.0f / (float)a.Wo;
    auto split = [&](int pc, int& b, int& y, int& xq) {
        b = (int)(((float)pc + 0.5f) * invHW); b -= (b * HW > pc); b += ((b + 1) * HW <= pc);
        const int rem = pc - b * HW;
        y = (int)(((float)rem + 0.5f) * invW); y -= (y * a.Wo > rem); y += ((y + 1) * a.Wo <= rem);
        xq = rem - y * a.Wo;
    };
    const int C = a.Cin / 3;
    const int tt0 = j * C1_NW + wave;
    const int nunit = tt0 < ntile ? ((ntile - tt0 + step - 1) / step) * NH : 0;      // (tile, pass) units of this wave: unit v = tile tt0 + (v / NH) step, pass v % NH
    struct Rows { half8 h[KQ]; half8 l[KQ]; };
    auto loadRows = [&](int v, Rows& w) {
        v = v < nunit ? v : nunit - 1;                               // (past the end: the last unit again, no branch around a load)
        const int t = tt0 + (v / NH) * step, hp = v % NH;
        const int p = t * 16 + r, pc = p < NPIX ? p : NPIX - 1;
        const _Float16* src = a.in + (size_t)pc * a.Cin + hp * KQ * 32 + g * 8;
#pragma unroll
        for (int q = 0; q < KQ; ++q) { w.h[q] = *reinterpret_cast<const half8*>(src + q * 32); w.l[q] = *reinterpret_cast<const half8*>(src + C + q * 32); }
    };
    const int Wout = a.Wo * a.up;
    floatx4 acc[CTG];
    auto unit = [&](int v, const Rows& w) {
        const int t = tt0 + (v / NH) * step, hp = NH == 1 ? 0 : v % NH;
        // (LDS addresses from an opaque copy of the lane offset: see conv1x1_resident_mx_kernel)
        int lo16 = lane * 16, g16 = g * 16;
        asm volatile("" : "+v"(lo16), "+v"(g16));
        const unsigned char* slot = smem + lo16;
        if (hp == 0) {
#pragma unroll
            for (int u = 0; u < CTG; ++u) {
                const floatx4 b4 = *reinterpret_cast<const floatx4*>(smem + BIAS_OFF + u * 64 + g16);
                acc[u] = a.bias ? b4 : floatx4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // 2 KQ half-steps: even = the eight w_hi fragments of k-step s (sixteen MFMAs: x hi, x lo), odd = the eight w_lo fragments (eight MFMAs: x hi);
        // the reads of half-step i + 1 are issued before the MFMAs of half-step i
        constexpr int NHS = 2 * KQ;
        intx4 fb[2][8];
        auto loadHalf = [&](int i, intx4 (&f)[8]) {
            const int s_ = i >> 1, off = (i & 1) ? LO_OFF : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) f[u] = *reinterpret_cast<const intx4*>(slot + off + ((hp * KQ + s_) * CTG + u) * 1024);
        };
        loadHalf(0, fb[0]);
#pragma unroll
        for (int i = 0; i < NHS; ++i) {
            if (i + 1 < NHS) loadHalf(i + 1, fb[(i + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const intx4 (&f)[8] = fb[i & 1];
            const int s_ = i >> 1;
            if ((i & 1) == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, f[u]), w.h[s_], acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, f[u]), w.l[s_], acc[u], 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, f[u]), w.h[s_], acc[u], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (hp == NH - 1) {
            const int p = t * 16 + r;
            const bool valid = p < NPIX;
            int b, y, xq;
            split(valid ? p : 0, b, y, xq);
            const size_t opix = (size_t)((b * a.Ho + y) * a.up + dy) * Wout + (xq * a.up + dx);
#pragma unroll
            for (int u = 0; u < CTG; u += 2) {
                convStoreWide<false, true>(a, acc[u], acc[u + 1], valid, opix, cbase + u * 16, g);
                __builtin_amdgcn_sched_barrier(0);                      // (one pair of tiles at a time)
            }
        }
    };
    Rows xa;
    if (nunit > 0) loadRows(0, xa);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the weights (and the first rows) have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    Rows xb;                                                          // (two sets of rows in flight: 2 x 8 KQ registers)
    for (int v = 0; v < nunit; v += 2) {
        loadRows(v + 1, xb);
        unit(v, xa);
        if (v + 1 >= nunit) break;
        loadRows(v + 2, xa);
        unit(v + 1, xb);
    }
}

static int numCUs();
static bool conv1x1ResidentShape(int KH, int KW, int stride, int pad, int Cin, int Cout, int rows) {
    static int on = -1;            // DSVT_CONV_1X1_RESIDENT=0: the halo / gather kernels for the 1 x 1 layers too
    if (on < 0) on = ablateEnv("DSVT_CONV_1X1_RESIDENT", 1);
    return on && KH == 1 && KW == 1 && (stride == 1 || stride == 2) && pad == 0 && (Cout == 128 || Cout == 256) && rows % 128 == 0 &&
           (Cin == 128 || Cin == 192 || Cin == 256);
}
static bool conv1x1ResidentEligible(const ConvArgs& a) {
    return conv1x1ResidentShape(a.KH, a.KW, a.stride, a.pad, a.Cin, a.Cout, a.CoutRows) && a.wide && !a.res && !a.out_f32 && !a.split_out && (a.stride == 1 || a.up == 1);
}

static bool conv3x3C64Eligible(const ConvArgs& a) {
    static int on = -1;            // DSVT_CONV_C64_RESIDENT=0: conv_wide_kernel for the 64-input-channel 3 x 3 layers too
    if (on < 0) on = ablateEnv("DSVT_CONV_C64_RESIDENT", 1);
    return on && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.up == 1 && a.Cin == 64 && a.CoutRows == a.Cout && a.Cout % 64 == 0 &&
           a.wide && !a.res && !a.out_f32 && !a.split_out;
}

static int launchConv1x1Resident(const ConvArgs& a, const _Float16* Wp, hipStream_t stream) {
    const int NCT = cdiv(a.CoutRows, CNB) * 8;                      // fragment rows per k-step of the halo image
    const int CTG = NCT < 16 ? NCT : 16, ngroup = NCT / CTG;        // (NCT is 8 or a multiple of 16 for these layers)
    if (NCT % CTG != 0) return -3;
    const dim3 grid(numCUs() / ngroup * ngroup), block(64 * C1_NW);
#define DSVT_C1(K_) do { if (a.stride == 2) hipLaunchKernelGGL((conv1x1_resident_kernel<K_, true>), grid, block, 0, stream, a, Wp, NCT, ngroup, CTG); \
                         else hipLaunchKernelGGL((conv1x1_resident_kernel<K_, false>), grid, block, 0, stream, a, Wp, NCT, ngroup, CTG); } while (0)
    if (a.Cin == 128) DSVT_C1(4); else if (a.Cin == 192) DSVT_C1(6);
    else if (a.stride == 2) hipLaunchKernelGGL((conv1x1_resident_kernel<8, true>), grid, block, 0, stream, a, Wp, NCT, ngroup, CTG);
    else hipLaunchKernelGGL((conv1x1_resident_kernel<8, false, 2>), grid, block, 0, stream, a, Wp, NCT, ngroup, CTG);       // (129 KB of weights: one workgroup per CU either way)
#undef DSVT_C1
    return lastError();
}

// 16-channel tiles per workgroup of the halo kernel (and of its packed weights)
static int haloChannelTiles(int coutRows) { return coutRows <= 32 ? 2 : coutRows <= 64 ? 4 : 8; }

static int numCUs() { return deviceCUs(); }

// Experiments that stay documented but are not product code (round 3: the ablation / tuning switches that used to be read from the
// environment inside these launchers are gone; measurements in profiles/README.md): 4-row halo tiles (5-10 % faster alone, 4 % slower
// with two frames in flight), balanced persistent grids (225 x 2 items instead of 256 workgroups: no gain), 14-row items on seven
// waves for the 468 x 468 layers (510 items = two full rounds: no gain), 40-pixel halo rows, four-wave 8-row tiles, a third weight slab
// in the halo kernel (8 % slower).
static int launchConvHalo(const ConvArgs& a, const _Float16* Wp, const _Float16* zeros, hipStream_t stream) {
    const int tilesX = cdiv(a.Wo, HTW), nchunk = cdiv(a.CoutRows, CNB);
    const int NBI = a.nb;                                       // images: every item count below is per image x NBI
    const bool spl = a.split_out != 0 || a.res_split != 0;      // split-precision epilogue (its own instantiations: see ConvArgs)
    static int ncuOverride = -1;                                // (ablation build: DSVT_CONV_NCU = persistent workgroups per launch)
    if (ncuOverride < 0) ncuOverride = ablateEnv("DSVT_CONV_NCU", 0);
    const int ncu = ncuOverride > 0 ? ncuOverride : numCUs();
    static int dbg = -1;                                        // timing ablations (wrong results): the -DDSVT_ABLATE build only, constant 0 in the product
    if (dbg < 0) dbg = ablateEnv("DSVT_CONV_DBG", 0);
    // 16-row tiles when they fill the CUs, else 8-row x 64-channel tiles (two workgroups per CU).  (Two channel
    // tiles: 32 MFMAs per slab cannot hide the halo stream, 86.7 vs 85.6 us on the 320 -> 18 head layer: the 8-row kernel below.)
    const int ctWide = haloChannelTiles(a.CoutRows);
#define DSVT_WIDE(GRID_, BLOCK_, NITEMS_, NCHUNK_, ...) do { \
        if (spl) hipLaunchKernelGGL((conv_wide_kernel<__VA_ARGS__, false, true>), dim3(GRID_), dim3(BLOCK_), 0, stream, a, Wp, zeros, tilesX, NITEMS_, NCHUNK_, dbg); \
        else hipLaunchKernelGGL((conv_wide_kernel<__VA_ARGS__, false, false>), dim3(GRID_), dim3(BLOCK_), 0, stream, a, Wp, zeros, tilesX, NITEMS_, NCHUNK_, dbg); \
        return lastError(); } while (0)
#define DSVT_WIDE_MX(GRID_, BLOCK_, NITEMS_, NCHUNK_, ...) do { \
        hipLaunchKernelGGL((conv_wide_kernel<__VA_ARGS__, false, true, true>), dim3(GRID_), dim3(BLOCK_), 0, stream, a, Wp, zeros, tilesX, NITEMS_, NCHUNK_, dbg); \
        return lastError(); } while (0)
    if (a.xscale && a.KH == 1) {                                // 1 x 1 layers on the fp16 + fp8 K loop (packMX1x1 image)
        if (ctWide != 8) return -3;
        static int res1 = -1;      // DSVT_CONV_1X1_RESIDENT=0: the halo kernel
        if (res1 < 0) res1 = ablateEnv("DSVT_CONV_1X1_RESIDENT", 1);
        const int C = a.Cin / 3, ngroup = a.CoutRows / 128;
        if (res1 && a.Cout == 128 && a.CoutRows % 128 == 0 && (C == 128 || C == 192 || C == 256) && a.wide && !a.res && !a.out_f32 && a.split_out && a.stride == 1 &&
            (ngroup == 1 || ngroup == 4 || ngroup == 16) && ncu % (8 * ngroup) == 0) {
            const int NCT = cdiv(a.CoutRows, CNB) * 8;
            if (C == 128) hipLaunchKernelGGL((conv1x1_resident_mx_kernel<4, 1>), dim3(ncu), dim3(64 * C1_NW), 0, stream, a, Wp, NCT, ngroup);
            else if (C == 192) hipLaunchKernelGGL((conv1x1_resident_mx_kernel<2, 3>), dim3(ncu), dim3(64 * C1_NW), 0, stream, a, Wp, NCT, ngroup);
            else hipLaunchKernelGGL((conv1x1_resident_mx_kernel<4, 2>), dim3(ncu), dim3(64 * C1_NW), 0, stream, a, Wp, NCT, ngroup);
            return lastError();
        }
        const int nit = cdiv(a.Ho, 8) * tilesX * nchunk * NBI;
        hipLaunchKernelGGL((conv_halo_kernel<8, 1, 8, 2, true, true>), dim3(nit < ncu ? nit : ncu), dim3(512), 0, stream, a, Wp, zeros, tilesX, nit, nchunk, dbg);
        return lastError();
    }
    if (!a.xscale && a.alias3 && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0) {      // 1 x 1 stride-1 layers on three fp16 products over [hi | lo | -]
        static int res1s = -1;     // DSVT_CONV_1X1_RESIDENT=0: the halo kernel
        if (res1s < 0) res1s = ablateEnv("DSVT_CONV_1X1_RESIDENT", 1);
        const int C = a.Cin / 3, ngroup = a.CoutRows / 128;
        if (res1s && ctWide == 8 && a.Cout == 128 && a.CoutRows % 128 == 0 && (C == 128 || C == 192 || C == 256) && a.wide && !a.res && !a.out_f32 && a.split_out &&
            (ngroup == 1 || ngroup == 4 || ngroup == 16) && ncu % (8 * ngroup) == 0) {
            const int NCT = cdiv(a.CoutRows, CNB) * 8;
            if (C == 128) hipLaunchKernelGGL((conv1x1_resident_split_kernel<4, 1>), dim3(ncu), dim3(64 * C1_NW), 0, stream, a, Wp, NCT, ngroup);
            else if (C == 192) hipLaunchKernelGGL((conv1x1_resident_split_kernel<2, 3>), dim3(ncu), dim3(64 * C1_NW), 0, stream, a, Wp, NCT, ngroup);
            else hipLaunchKernelGGL((conv1x1_resident_split_kernel<4, 2>), dim3(ncu), dim3(64 * C1_NW), 0, stream, a, Wp, NCT, ngroup);
            return lastError();
        }
    }
    if (a.xscale) {                                             // split input [hi | lo | x8] on the fp16 + fp8 K loop (packMX image); same tile choices as below
        if (a.KH != 3 || ctWide < 4) return -3;
        const int nwide = cdiv(a.Ho, 16) * tilesX * nchunk * NBI;
        if (nwide >= ncu) {
            if (ctWide == 8) DSVT_WIDE_MX(ncu, 512, nwide, nchunk, 8, 8, 36, MX_SPS, MX_NWB, 2);
            else DSVT_WIDE_MX(ncu, 512, nwide, nchunk, 4, 8, 40, 4, MX_NWB4, 2);
        }
        const int nch64 = cdiv(a.CoutRows, 64), nsmall = cdiv(a.Ho, 8) * tilesX * nch64 * NBI;
        const int n16 = cdiv(a.Ho, 16) * tilesX * nch64 * NBI;
        if (n16 * 10 >= ncu * 9 && n16 <= ncu) DSVT_WIDE_MX(n16, 512, n16, nch64, 4, 8, 40, 4, MX_NWB4, 2);
        DSVT_WIDE_MX(nsmall < 2 * ncu ? nsmall : 2 * ncu, 512, nsmall, nch64, 4, 8, 36, 4, 2, 1);
    }
#undef DSVT_WIDE_MX
    if (a.KH == 3 && ctWide >= 4) {
        const int nwide = cdiv(a.Ho, 16) * tilesX * nchunk * NBI;
        // round 6: launches too small for the 128-channel / 24-row items run the ky-row slab loop on the 64-channel items chosen below (DSVT_CONV_ROWS_SMALL=0: round 5's kernels)
        auto rowsSmall = [&](const ConvArgs& q) {
            static int on = -1; if (on < 0) on = ablateEnv("DSVT_CONV_ROWS", 1) && ablateEnv("DSVT_CONV_ROWS_SMALL", 1);
            return on && spl && q.variant != 1 && !q.trace && convRowsSmallEligible(q);
        };
        // short K (the 64 -> 320 head stems: 9 slabs per item, the epilogue weighs as much as the MFMAs): 8-row x 128-channel tiles on
        // four waves, two independent workgroups per CU, 2655 items: 116-122 vs 130 us (no gain on the K >= 1152 layers)
        if (ctWide == 8 && a.Cin <= 64) {
            const int n8 = cdiv(a.Ho, 8) * tilesX * nchunk * NBI, grid = n8 < 2 * ncu ? n8 : 2 * ncu;
            DSVT_WIDE(grid, 256, n8, nchunk, 8, 4, 36, 2, 2, 2);
        }
        {   // three-product layers whose channel count leaves half of the last 128-channel chunk empty (the 64 -> 320 head stems: 2.5 chunks -> a sixth of the
            // MFMAs wasted, and a short K -- six phases -- that streams 432 KB of weights per item): 64-channel chunks on 24-row items (see the 64-channel rule below)
            static int hct4 = -1; if (hct4 < 0) hct4 = ablateEnv("DSVT_CONV_HEADS_CT4", 1);
            const int nch64h = cdiv(a.CoutRows, 64), n24h = cdiv(a.Ho, 24) * tilesX * nch64h * NBI;
            if (hct4 && spl && ctWide == 8 && a.CoutRows % 128 == 64 && n24h >= ncu) {
                static int rows64h = -1; if (rows64h < 0) rows64h = ablateEnv("DSVT_CONV_ROWS", 1);           // round 6: the same items on conv_rows_kernel<4, 3> (conv_rows.hip)
                if (rows64h && a.variant != 1 && !a.trace && convRows64Eligible(a, ncu)) return launchConvRows64(a, Wp, ncu, stream);
                hipLaunchKernelGGL((conv_wide_kernel<4, 8, 36, 4, 2, 3, false, true>), dim3(ncu), dim3(512), 0, stream, a, Wp, zeros, tilesX, n24h, nch64h, dbg);
                return lastError();
            }
        }
        if (nwide >= ncu) {
            // halo row stride 36 pixels, not 40: 134,144 B of LDS instead of 142,336 (room for the OTHER frame's 23 KB attention workgroups)
            // Round-3 experiments, measured and dropped (one wave per SIMD loses both times -- a lone wave cannot hide its own LDS-DMA issue,
            // fragment-read latency and barrier waits, whatever it saves in fragment reads):
            //  * conv_wide_kernel<8, 4, 36, 2, 3, 4>: four waves of FOUR rows, 128 pixels x 128 channels = 256 accumulator registers per wave,
            //    16 fragment reads per 64 MFMAs instead of 12 per 32 -- hipcc spills 111-138 of the 512 registers; a 468 x 468 128 -> 128 layer
            //    takes 425 us per four frames against 278;
            //  * a split-native variant for the fp32-grade mode (per (32-channel group, tap) step the hi AND lo halo planes and the w_hi AND
            //    w_lo rows in LDS, three MFMAs per fragment pair: a third less halo / weight traffic and 0.25 fragment reads per MFMA instead
            //    of 0.375; LDS only holds 8-row tiles then, four waves of two rows): correct (boxes within 1e-5) and 364 registers without a
            //    spill, but 860-1054 us per 468 x 468 128 -> 128 layer against 831-954 for the plain kernel walking the [hi | lo | hi] triple.
            if (ctWide == 8) {
                // round 6: the ky-row-slab kernel (conv_rows.hip) for the three-product layers -- same tile, same MFMA order, bit-identical results
                static int rowsOn = -1; if (rowsOn < 0) rowsOn = ablateEnv("DSVT_CONV_ROWS", 1);
                if (rowsOn && spl && a.variant != 1 && convRowsEligible(a, ncu)) return launchConvRows(a, Wp, ncu, stream);
                // round 5, late: FOUR-step slabs in two buffers instead of two-step slabs in three -- half the workgroup barriers and counted waits per item, requests
                // at the slab start (no trickle: LEAD = 1).  Four frames per launch: three-product 128 -> 128 at 468 x 468 780 -> 760 us, with a residual 940 -> 877, dense
                // stage 9.95 -> 9.67 ms; fp16 frame 315 -> 308 / 367 -> 347 us, dense stage 4.00 -> 3.91 ms; one frame 2.98 -> 2.93 ms (WIDE_SPS4_DEFAULT / DSVT_CONV_SPS4=0: the two-step kernel)
                static int sps4 = -1; if (sps4 < 0) sps4 = ablateEnv("DSVT_CONV_SPS4", WIDE_SPS4_DEFAULT);
                if constexpr (kAblate) {     // the instrumented instantiation of the three-product kernel (DSVT_CONV_TRACE=1, tools/trace_conv_split.py)
                    if (a.trace && spl && sps4) { hipLaunchKernelGGL((conv_wide_kernel<8, 8, 36, 4, 2, 2, true, true>), dim3(ncu), dim3(512), 0, stream, a, Wp, zeros, tilesX, nwide, nchunk, dbg); return lastError(); }
                }
                if (sps4) DSVT_WIDE(ncu, 512, nwide, nchunk, 8, 8, 36, 4, 2, 2);
                DSVT_WIDE(ncu, 512, nwide, nchunk, 8, 8, 36, 2, 3, 2);
            } else {
                // 64 output channels (the shared 384 -> 64 head convolution): a 16-row item's accumulators are half a wave's budget (64 of 128 registers), so
                // THREE rows per wave -- 24 x 32-pixel items, 96 accumulator registers, 26 x 36-pixel halo phases (59 KB each) beside two 16 KB weight slabs:
                // the weight stream per output pixel falls by a third and the halo overhead from 18 / 16 to 26 / 24 (round 5; DSVT_CONV_RW3=0: 16-row items)
                // Measured (tools/bench_conv_mx.py, 468 x 468 384 -> 64, three products): four frames 1367 vs 1516 us (1200 items = 4.7 rounds of 256 against
                // 1800 = 7.03), ONE frame 497 vs 397 us (300 items = a second round for 44 of them) -- so the item height follows the round count: a 24-row
                // row costs ~0.92 of a 16-row one.  Split-precision instantiation only (the fp16 one spills 42 registers at three rows per wave).
                static int rw3 = -1; if (rw3 < 0) rw3 = ablateEnv("DSVT_CONV_RW3", 1);
                const int n24 = cdiv(a.Ho, 24) * tilesX * nchunk * NBI;
                if (rw3 && spl && n24 >= ncu && cdiv(n24, ncu) * 24 * 92 < cdiv(nwide, ncu) * 16 * 100) {
                    static int rows64 = -1; if (rows64 < 0) rows64 = ablateEnv("DSVT_CONV_ROWS", 1);         // round 6: the same items on conv_rows_kernel<4, 3>
                    if (rows64 && a.variant != 1 && !a.trace && convRows64Eligible(a, ncu)) return launchConvRows64(a, Wp, ncu, stream);
                    hipLaunchKernelGGL((conv_wide_kernel<4, 8, 36, 4, 2, 3, false, true>), dim3(ncu), dim3(512), 0, stream, a, Wp, zeros, tilesX, n24, nchunk, dbg);
                    return lastError();
                }
                if (rowsSmall(a)) return launchConvRowsSmall(a, Wp, 2, ncu, stream);                              // round 6: the same items on conv_rows_kernel<4, 2>
                DSVT_WIDE(ncu, 512, nwide, nchunk, 4, 8, 40, 4, 2, 2);
            }
        }
        const int nch64 = cdiv(a.CoutRows, 64), nsmall = cdiv(a.Ho, 8) * tilesX * nch64 * NBI;
        // 16-row x 64-channel items on eight waves when they nearly fill the CUs (234x234x128: 240 items, one per CU, 40 LDS-DMA
        // pieces per wave and item instead of 59)
        const int n16 = cdiv(a.Ho, 16) * tilesX * nch64 * NBI;
        if (n16 * 10 >= ncu * 9 && n16 <= ncu) {
            if (rowsSmall(a)) return launchConvRowsSmall(a, Wp, 2, ncu, stream);                                  // round 6: conv_rows_kernel<4, 2>
            DSVT_WIDE(n16, 512, n16, nch64, 4, 8, 40, 4, 2, 2);
        }
        if (rowsSmall(a)) return launchConvRowsSmall(a, Wp, 1, ncu, stream);                                      // round 6: conv_rows_kernel<4, 1>
        // 8 rows x 32 pixels x 64 channels: eight waves of ONE row each (117x117x256: 31.4 us; four waves of two rows: 35.6 us --
        // one wave per SIMD cannot hide its own LDS-DMA issue and wait time)
        DSVT_WIDE(nsmall < 2 * ncu ? nsmall : 2 * ncu, 512, nsmall, nch64, 4, 8, 36, 4, 2, 1);
    }
#undef DSVT_WIDE
    // the first halo-tile kernel: 1 x 1 layers the resident-weights kernel does not take, 3 x 3 layers with <= 32 output channels
    constexpr int th = 8;
    const int nitems = cdiv(a.Ho, th) * tilesX * nchunk * NBI;
    const int grid = nitems < ncu ? nitems : ncu;
#define DSVT_HALO(KS_, CTW_) do { \
        if (spl) hipLaunchKernelGGL((conv_halo_kernel<8, KS_, CTW_, 2, true>), dim3(grid), dim3(64 * th), 0, stream, a, Wp, zeros, tilesX, nitems, nchunk, dbg); \
        else hipLaunchKernelGGL((conv_halo_kernel<8, KS_, CTW_, 2, false>), dim3(grid), dim3(64 * th), 0, stream, a, Wp, zeros, tilesX, nitems, nchunk, dbg); } while (0)
    const int ctw = haloChannelTiles(a.CoutRows);
    if (a.KH == 3) { if (ctw == 8) DSVT_HALO(3, 8); else if (ctw == 4) DSVT_HALO(3, 4); else DSVT_HALO(3, 2); }
    else           { if (ctw == 8) DSVT_HALO(1, 8); else if (ctw == 4) DSVT_HALO(1, 4); else DSVT_HALO(1, 2); }
#undef DSVT_HALO
    return lastError();
}

static int launchConv(const ConvArgs& a, int KC, hipStream_t stream) {
    dim3 grid((unsigned)(cdiv(cdiv(a.Ho * a.Wo, CPX), 8) * 8), (unsigned)cdiv(a.CoutRows, CNB), (unsigned)a.nb);
    const bool spl = a.split_out != 0 || a.res_split != 0;
#define DSVT_GATHER(KC_) do { if (spl) hipLaunchKernelGGL((conv_f16_kernel<KC_, 1, 8, true>), grid, dim3(512), 0, stream, a); \
                              else hipLaunchKernelGGL((conv_f16_kernel<KC_, 1, 8, false>), grid, dim3(512), 0, stream, a); } while (0)
    if (KC == 128) DSVT_GATHER(128); else if (KC == 96) DSVT_GATHER(96); else if (KC == 64) DSVT_GATHER(64); else return -3;
#undef DSVT_GATHER
    return lastError();
}

// -------------------------------------------------------------------------------------
struct ConvCfg {
    int H, W, Cin, Cout, KH, KW, stride, pad, up, relu, has_res, out_ld, out_coff, out_f32;
    int split_out, split_res;      // split precision (fields "split_output" / "split_residual"): the output / the residual is an fp16 [hi | lo | hi] triple (see ConvArgs); split_out = 2: [hi | lo | x8], 3: [hi | - | x8], 4: [hi | lo | -] (a tensor that is only ever a residual)
    int split_in;                  // field "split_input": the input is a split tensor [hi | lo | x8] (Cin = 3 C).  1: the weight rows are the host's [w_hi | w_hi | w_lo] and the phases
                                   // of the third plane read plane 0; 2: the weight rows are the REAL fp32 rows [R][9][C] and the layer runs on conv_wide_kernel<.., MX>
};

// OCP e4m3 byte of f: round to nearest even, saturated at +-448 (what packE4m3 does on the device)
static unsigned char e4m3Encode(float f) {
    if (f != f) return 0x7f;
    const unsigned char sgn = std::signbit(f) ? 0x80 : 0;
    const float v = std::fabs(f);
    if (v >= 448.f) return sgn | 0x7e;
    if (v < std::ldexp(1.f, -6)) return sgn | (unsigned char)std::nearbyint(v * 512.f);      // subnormals, step 2^-9 (8 = the first normal)
    int e; const float m = std::frexp(v, &e);                     // v = m 2^e, m in [0.5, 1)
    float q = std::nearbyint(m * 16.f);                           // 8 .. 16
    if (q == 16.f) { q = 8.f; ++e; }
    const int E = e - 1 + 7;
    if (E > 15 || (E == 15 && q > 14.f)) return sgn | 0x7e;
    return sgn | (unsigned char)((E << 3) | ((int)q - 8));
}

class DsvtConv2dPlugin : public Plugin {
public:
    ConvCfg c_;
    std::vector<float> w_, b_;           // w_: [up*up*Cout][KH*KW][Cin]
    _Float16* w_dev_ = nullptr; float* b_dev_ = nullptr;
    _Float16* wp_dev_ = nullptr;         // fragment-packed copy for the halo kernel (stride-1 layers)
    _Float16* zeros_dev_ = nullptr;      // 256 zero bytes: LDS-DMA source of out-of-image halo pixels
    _Float16* wg_dev_ = nullptr; int* chan_dev_ = nullptr; int groups_ = 0; bool groups_split_ = false;      // block-diagonal narrow 3 x 3 layer: per-phase weight tiles + channel table (split: w_hi and w_lo tiles per group of a split input)
    unsigned char* xscale_dev_ = nullptr;                                        // split_in == 2: scale byte per weight row (wp_dev_ holds the packMX image)
    bool ok_ = false;
    int variant_ = 0;                    // field "kernel_variant" (a test / A-B knob, not serialized: a deserialized plugin takes the launcher's choice)
    int cinW() const { return c_.split_in == 2 ? c_.Cin / 3 : c_.Cin; }         // channels per tap of a weight row
    bool haloEligible() const {
        static int on = -1;
        if (on < 0) on = ablateEnv("DSVT_CONV_HALO", 1);
        return on && c_.stride == 1 && c_.KH == c_.KW && (c_.KH == 1 || c_.KH == 3) && c_.pad == c_.KH / 2 && c_.Cin % 64 == 0;
    }
    int Ho() const { return (c_.H + 2 * c_.pad - c_.KH) / c_.stride + 1; }
    int Wo() const { return (c_.W + 2 * c_.pad - c_.KW) / c_.stride + 1; }
    int rows() const { return c_.up * c_.up * c_.Cout; }
    int KC() const {
        { const int k = ablateEnv("DSVT_CONV_KC", 0); if (k > 0 && c_.Cin % k == 0) return k; }   // tuning knob (ablation build only)
        // large images: 64-channel slabs (150 VGPRs, 40 KB LDS => 3 waves/SIMD) hide the per-slab barrier better than
        // 128-channel ones (measured +10 % on the 468x468 layers); small images prefer fewer, fatter slabs
        if (Ho() * Wo() >= 100000 && c_.Cin % 64 == 0) return 64;
        return c_.Cin % 128 == 0 ? 128 : c_.Cin % 96 == 0 ? 96 : 64;
    }
    // split_in == 2: fragment image of the fp16 + fp8 K loop (conv_wide_kernel<.., MX>) from the real fp32 rows [R][9][C]:
    //   [fp16 step t = (phase t / 9, tap t % 9)][16-channel tile][lane (r, g)][8 halfs] <- fp16(W[tile 16 + r][tap][32 phase + 8 g + j])
    //   [cross step x = (phase x / 5, tap pair x % 5)][tile][h][lane (r, g)][16 bytes]  <- e4m3 of W_hi 2^e (g even) or W_lo 2^(e + 11) (g odd) at
    //       [tile 16 + r][tap 2 (x % 5) + (g >> 1)][32 phase + 16 h + j], zero for tap 9;  e = the row's exponent that brings max |w| under 448
    // and the scale byte 127 - 11 - e per row (one shared power of two for both kinds of fp8 block: a_lo8 carries 2^11, w_lo8 carries 2^(e + 11)).
    void packMX() {
        const ConvCfg& c = c_;
        const int C = c.Cin / 3, NPM = C / 32, R = rows();
        const int ctw = haloChannelTiles(R), NCT = ctw < 8 ? ctw : cdiv(R, CNB) * 8;
        const size_t mainRows = (size_t)9 * NPM * NCT, crossRows = (size_t)5 * NPM * NCT * 2;
        std::vector<unsigned char> img((mainRows + crossRows + (size_t)4 * NCT) * 1024, 0);          // (+ slack: a partial four-step slab is requested whole)
        std::vector<unsigned char> sc((size_t)NCT * 16, 127 - 11);
        std::vector<int> ex(R, 0);
        for (int n = 0; n < R; ++n) {
            float mx = 0.f;
            for (size_t i = 0; i < (size_t)9 * C; ++i) mx = std::fmax(mx, std::fabs(w_[(size_t)n * 9 * C + i]));
            int e = mx > 0.f ? (int)std::floor(std::log2(448.f / mx)) : 0;
            e = e < -60 ? -60 : e > 60 ? 60 : e;
            ex[n] = e; sc[n] = (unsigned char)(127 - 11 - e);
        }
        _Float16* img16 = reinterpret_cast<_Float16*>(img.data());
        for (int t = 0; t < 9 * NPM; ++t)
            for (int tile = 0; tile < NCT; ++tile)
                for (int lane = 0; lane < 64; ++lane) {
                    const int n = tile * 16 + (lane & 15), ph = t / 9, tap = t % 9;
                    if (n >= R) continue;
                    const size_t dst = (((size_t)t * NCT + tile) * 64 + lane) * 8, src = ((size_t)n * 9 + tap) * C + ph * 32 + (lane >> 4) * 8;
                    for (int j = 0; j < 8; ++j) img16[dst + j] = (_Float16)w_[src + j];
                }
        for (int x = 0; x < 5 * NPM; ++x)
            for (int tile = 0; tile < NCT; ++tile)
                for (int h = 0; h < 2; ++h)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int n = tile * 16 + (lane & 15), g = lane >> 4, q = x / 5, tap = 2 * (x % 5) + (g >> 1);
                        if (n >= R || tap > 8) continue;
                        const size_t dst = (mainRows + ((size_t)x * NCT + tile) * 2 + h) * 1024 + (size_t)lane * 16, src = ((size_t)n * 9 + tap) * C + q * 32 + h * 16;
                        for (int j = 0; j < 16; ++j) {
                            const float wv = w_[src + j], hi = (float)(_Float16)wv;
                            img[dst + j] = (g & 1) ? e4m3Encode(std::ldexp(wv - hi, ex[n] + 11)) : e4m3Encode(std::ldexp(hi, ex[n]));
                        }
                    }
        ok_ = dsvtMalloc(&wp_dev_, img.size()) == hipSuccess && hipMemcpy(wp_dev_, img.data(), img.size(), hipMemcpyHostToDevice) == hipSuccess &&
              dsvtMalloc(&xscale_dev_, sc.size()) == hipSuccess && hipMemcpy(xscale_dev_, sc.data(), sc.size(), hipMemcpyHostToDevice) == hipSuccess &&
              dsvtMalloc(&zeros_dev_, 256) == hipSuccess && hipMemset(zeros_dev_, 0, 256) == hipSuccess;
    }
    // the same for a 1 x 1 layer on conv_halo_kernel<.., MX> (64-channel phases): [fp16 k-step q = 2 phase + ks][tile][lane (r, g)][8 halfs] <-
    // fp16(W[tile 16 + r][64 phase + 32 ks + 8 g + j]), then [cross phase][tile][h][lane (r, g)][16 bytes] <- e4m3 of W_hi 2^e (g even) or W_lo 2^(e + 11)
    // (g odd) at channel 64 phase + 32 (g >> 1) + 16 h + j
    void packMX1x1() {
        const ConvCfg& c = c_;
        const int C = c.Cin / 3, NPM = C / 64, R = rows(), NCT = cdiv(R, CNB) * 8;
        const size_t mainRows = (size_t)2 * NPM * NCT, crossRows = (size_t)NPM * NCT * 2;
        std::vector<unsigned char> img((mainRows + crossRows + (size_t)2 * NCT) * 1024, 0);
        std::vector<unsigned char> sc((size_t)NCT * 16, 127 - 11);
        std::vector<int> ex(R, 0);
        for (int n = 0; n < R; ++n) {
            float mx = 0.f;
            for (int i = 0; i < C; ++i) mx = std::fmax(mx, std::fabs(w_[(size_t)n * C + i]));
            int e = mx > 0.f ? (int)std::floor(std::log2(448.f / mx)) : 0;
            e = e < -60 ? -60 : e > 60 ? 60 : e;
            ex[n] = e; sc[n] = (unsigned char)(127 - 11 - e);
        }
        _Float16* img16 = reinterpret_cast<_Float16*>(img.data());
        for (int q = 0; q < 2 * NPM; ++q)
            for (int tile = 0; tile < NCT; ++tile)
                for (int lane = 0; lane < 64; ++lane) {
                    const int n = tile * 16 + (lane & 15);
                    if (n >= R) continue;
                    const size_t dst = (((size_t)q * NCT + tile) * 64 + lane) * 8, src = (size_t)n * C + (q >> 1) * 64 + (q & 1) * 32 + (lane >> 4) * 8;
                    for (int j = 0; j < 8; ++j) img16[dst + j] = (_Float16)w_[src + j];
                }
        for (int ph = 0; ph < NPM; ++ph)
            for (int tile = 0; tile < NCT; ++tile)
                for (int h = 0; h < 2; ++h)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int n = tile * 16 + (lane & 15), g = lane >> 4;
                        if (n >= R) continue;
                        const size_t dst = (mainRows + ((size_t)ph * NCT + tile) * 2 + h) * 1024 + (size_t)lane * 16, src = (size_t)n * C + ph * 64 + (g >> 1) * 32 + h * 16;
                        for (int j = 0; j < 16; ++j) {
                            const float wv = w_[src + j], hi = (float)(_Float16)wv;
                            img[dst + j] = (g & 1) ? e4m3Encode(std::ldexp(wv - hi, ex[n] + 11)) : e4m3Encode(std::ldexp(hi, ex[n]));
                        }
                    }
        ok_ = dsvtMalloc(&wp_dev_, img.size()) == hipSuccess && hipMemcpy(wp_dev_, img.data(), img.size(), hipMemcpyHostToDevice) == hipSuccess &&
              dsvtMalloc(&xscale_dev_, sc.size()) == hipSuccess && hipMemcpy(xscale_dev_, sc.data(), sc.size(), hipMemcpyHostToDevice) == hipSuccess &&
              dsvtMalloc(&zeros_dev_, 256) == hipSuccess && hipMemset(zeros_dev_, 0, 256) == hipSuccess;
    }
    DsvtConv2dPlugin(const ConvCfg& c, const float* w, const float* b) : c_(c) {
        const size_t nw = (size_t)rows() * c.KH * c.KW * cinW();
        w_.assign(w, w + nw);
        if (b) b_.assign(b, b + c.Cout);
        if (c.split_in == 2) {
            ok_ = true;
            if (b) {
                const size_t nb = ((size_t)c.Cout + 3) / 4 * 4;
                ok_ = dsvtMalloc(&b_dev_, sizeof(float) * nb) == hipSuccess && hipMemset(b_dev_, 0, sizeof(float) * nb) == hipSuccess &&
                      hipMemcpy(b_dev_, b_.data(), sizeof(float) * c.Cout, hipMemcpyHostToDevice) == hipSuccess;
            }
            if (ok_) { if (c.KH == 1) packMX1x1(); else packMX(); }
            return;
        }
        std::vector<_Float16> wh(nw);
        for (size_t i = 0; i < nw; ++i) wh[i] = (_Float16)w_[i];
        ok_ = dsvtMalloc(&w_dev_, sizeof(_Float16) * nw) == hipSuccess &&
              hipMemcpy(w_dev_, wh.data(), sizeof(_Float16) * nw, hipMemcpyHostToDevice) == hipSuccess;
        if (ok_ && b) {                      // zero padded to whole float4s: the halo kernels fetch the bias as 16-byte LDS-DMA lanes
            const size_t nb = ((size_t)c.Cout + 3) / 4 * 4;
            ok_ = dsvtMalloc(&b_dev_, sizeof(float) * nb) == hipSuccess && hipMemset(b_dev_, 0, sizeof(float) * nb) == hipSuccess &&
                  hipMemcpy(b_dev_, b_.data(), sizeof(float) * c.Cout, hipMemcpyHostToDevice) == hipSuccess;
        }
        if (ok_ && (haloEligible() || conv1x1ResidentShape(c.KH, c.KW, c.stride, c.pad, c.Cin, c.Cout, rows()))) {
            // [k-step q = (cc * taps + tap) * 2 + ks][16-channel tile ct][lane (r, g)][8] <- W[ct*16 + r][tap][cc*64 + ks*32 + g*8 + j];
            // one 16 KB slab of zero padding at the end: the last slab of a narrow layer is requested whole
            const int T = c.KH * c.KW, NCC = c.Cin / 64, R = rows(), ctw = haloChannelTiles(R);
            const int NCT = ctw < 8 ? ctw : cdiv(R, CNB) * 8;
            std::vector<_Float16> wp((size_t)NCC * T * 2 * NCT * 512 + (size_t)(NCT > 4 ? 4 * NCT : 16) * 512, (_Float16)0.f);   // (a partial four-step slab: two k-steps)
            for (int cc = 0; cc < NCC; ++cc)
                for (int tap = 0; tap < T; ++tap)
                    for (int ks = 0; ks < 2; ++ks)
                        for (int ct = 0; ct < NCT; ++ct)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int n = ct * 16 + (lane & 15);
                                if (n >= R) continue;
                                const size_t dst = ((((size_t)(cc * T + tap) * 2 + ks) * NCT + ct) * 64 + lane) * 8;
                                const size_t src = ((size_t)n * T + tap) * c.Cin + cc * 64 + ks * 32 + (lane >> 4) * 8;
                                for (int j = 0; j < 8; ++j) wp[dst + j] = wh[src + j];
                            }
            ok_ = dsvtMalloc(&wp_dev_, sizeof(_Float16) * wp.size()) == hipSuccess &&
                  hipMemcpy(wp_dev_, wp.data(), sizeof(_Float16) * wp.size(), hipMemcpyHostToDevice) == hipSuccess &&
                  dsvtMalloc(&zeros_dev_, 256) == hipSuccess && hipMemset(zeros_dev_, 0, 256) == hipSuccess;
        }
        if (ok_ && zeros_dev_) packGrouped(wh);
    }
    // A narrow 3 x 3 stride-1 layer every output channel of which reads ONE 64-channel phase of the input (the CenterHead's five output
    // convolutions folded into one block-diagonal layer): per phase one 16-channel tile of fragment rows [tap][k-step] and the table of the
    // tile's real channels.  Anything else (a channel that reads two phases, more than 16 channels on a phase) stays on the dense kernels.
    void packGrouped(const std::vector<_Float16>& wh) {
        static int on = -1;        // DSVT_CONV_GROUPED=0: the dense halo kernel for block-diagonal layers too
        if (on < 0) on = ablateEnv("DSVT_CONV_GROUPED", 1);
        const ConvCfg& c = c_;
        if (!on || c.KH != 3 || c.KW != 3 || c.stride != 1 || c.pad != 1 || c.up != 1 || c.has_res || c.Cin % 64 != 0 || c.Cin < 128 || c.Cout > 64) return;
        if (c.split_in == 1) { packGroupedSplit(wh); return; }
        const int NCC = c.Cin / 64;
        std::vector<std::vector<int>> chans(NCC);
        for (int n = 0; n < c.Cout; ++n) {
            int phase = -1;
            for (int tap = 0; tap < 9; ++tap)
                for (int k = 0; k < c.Cin; ++k)
                    if (w_[((size_t)n * 9 + tap) * c.Cin + k] != 0.f) {
                        if (phase >= 0 && phase != k / 64) return;                 // reads two phases: dense
                        phase = k / 64;
                    }
            chans[phase < 0 ? 0 : phase].push_back(n);
        }
        for (auto& v : chans) if (v.size() > 16) return;
        std::vector<_Float16> wg((size_t)NCC * 18 * 512, (_Float16)0.f);
        std::vector<int> tab((size_t)NCC * 16, -1);
        for (int cc = 0; cc < NCC; ++cc)
            for (size_t i = 0; i < chans[cc].size(); ++i) {
                const int n = chans[cc][i];
                tab[cc * 16 + i] = n;
                for (int tap = 0; tap < 9; ++tap)
                    for (int ks = 0; ks < 2; ++ks)
                        for (int gq = 0; gq < 4; ++gq)
                            for (int jq = 0; jq < 8; ++jq)
                                wg[(((size_t)cc * 18 + tap * 2 + ks) * 64 + gq * 16 + i) * 8 + jq] = wh[((size_t)n * 9 + tap) * c.Cin + cc * 64 + ks * 32 + gq * 8 + jq];
            }
        if (dsvtMalloc(&wg_dev_, sizeof(_Float16) * wg.size()) != hipSuccess || dsvtMalloc(&chan_dev_, sizeof(int) * tab.size()) != hipSuccess ||
            hipMemcpy(wg_dev_, wg.data(), sizeof(_Float16) * wg.size(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(chan_dev_, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice) != hipSuccess) { ok_ = false; return; }
        groups_ = NCC;
    }
    // the same for three fp16 products over a split input [hi | lo | x8] (split_in = 1: rows [w_hi | w_hi | w_lo] over Cin = 3 C): every output channel
    // reads ONE 64-channel group of each plane, the first two blocks of its row are equal; per group 18 fragment rows of w_hi and 18 of w_lo
    void packGroupedSplit(const std::vector<_Float16>& wh) {
        const ConvCfg& c = c_;
        if (c.Cin % 192 != 0 || c.split_out != 0) return;
        const int C = c.Cin / 3, NQ = C / 64;
        std::vector<std::vector<int>> chans(NQ);
        for (int n = 0; n < c.Cout; ++n) {
            int phase = -1;
            for (int tap = 0; tap < 9; ++tap)
                for (int k = 0; k < c.Cin; ++k) {
                    const float v = w_[((size_t)n * 9 + tap) * c.Cin + k];
                    if (k < C && v != w_[((size_t)n * 9 + tap) * c.Cin + C + k]) return;      // (plane 1 must pair with the weights of plane 0)
                    if (v != 0.f) {
                        if (phase >= 0 && phase != (k % C) / 64) return;           // reads two groups: dense
                        phase = (k % C) / 64;
                    }
                }
            chans[phase < 0 ? 0 : phase].push_back(n);
        }
        for (auto& v : chans) if (v.size() > 16) return;
        std::vector<_Float16> wg((size_t)NQ * 36 * 512, (_Float16)0.f);
        std::vector<int> tab((size_t)NQ * 16, -1);
        for (int cc = 0; cc < NQ; ++cc)
            for (size_t i = 0; i < chans[cc].size(); ++i) {
                const int n = chans[cc][i];
                tab[cc * 16 + i] = n;
                for (int pl = 0; pl < 2; ++pl)                                      // w_hi (block 0), w_lo (block 2)
                    for (int tap = 0; tap < 9; ++tap)
                        for (int ks = 0; ks < 2; ++ks)
                            for (int gq = 0; gq < 4; ++gq)
                                for (int jq = 0; jq < 8; ++jq)
                                    wg[(((size_t)cc * 36 + pl * 18 + tap * 2 + ks) * 64 + gq * 16 + i) * 8 + jq] =
                                        wh[((size_t)n * 9 + tap) * c.Cin + pl * 2 * C + cc * 64 + ks * 32 + gq * 8 + jq];
            }
        if (dsvtMalloc(&wg_dev_, sizeof(_Float16) * wg.size()) != hipSuccess || dsvtMalloc(&chan_dev_, sizeof(int) * tab.size()) != hipSuccess ||
            hipMemcpy(wg_dev_, wg.data(), sizeof(_Float16) * wg.size(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(chan_dev_, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice) != hipSuccess) { ok_ = false; return; }
        groups_ = NQ; groups_split_ = true;
    }
    ~DsvtConv2dPlugin() override {
        if (w_dev_) (void)dsvtFree(w_dev_); if (b_dev_) (void)dsvtFree(b_dev_); if (wp_dev_) (void)dsvtFree(wp_dev_); if (zeros_dev_) (void)dsvtFree(zeros_dev_); if (wg_dev_) (void)dsvtFree(wg_dev_); if (chan_dev_) (void)dsvtFree(chan_dev_);
        if (xscale_dev_) (void)dsvtFree(xscale_dev_);
    }
    const char* type() const override { return "DsvtConv2dPlugin"; }
    bool handlesBatch() const override { return true; }          // a stack of images is ONE launch: the persistent kernels walk image after image
    int nbOutputs() const override { return 1; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i != 0) return -1;
        *out = dims4(in[0].d[0], Ho() * c_.up, Wo() * c_.up, c_.out_ld); return 0;
    }
    int outputType(int, const int32_t*, int) const override { return c_.out_f32 ? DSVT_FLOAT : DSVT_HALF; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int nbIn, int) const override {
        if (io[pos].format != DSVT_FORMAT_LINEAR) return false;
        return pos < nbIn ? io[pos].type == DSVT_HALF : io[pos].type == (c_.out_f32 ? DSVT_FLOAT : DSVT_HALF);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        if (!ok_) return static_cast<int>(hipErrorOutOfMemory);
        ConvArgs a{};
        a.in = static_cast<const _Float16*>(in[0]); a.H = c_.H; a.W = c_.W; a.Cin = c_.Cin;
        a.wt = w_dev_; a.bias = b_dev_;
        a.res = c_.has_res ? static_cast<const _Float16*>(in[1]) : nullptr;
        a.res_ld = (c_.has_res && inDesc) ? inDesc[1].dims.d[inDesc[1].dims.nbDims - 1] : c_.Cout;
        a.out = out[0]; a.out_ld = c_.out_ld; a.out_coff = c_.out_coff; a.out_f32 = c_.out_f32;
        a.Ho = Ho(); a.Wo = Wo(); a.CoutRows = rows(); a.Cout = c_.Cout;
        a.KH = c_.KH; a.KW = c_.KW; a.stride = c_.stride; a.pad = c_.pad; a.up = c_.up; a.relu = c_.relu;
        a.split_out = c_.split_out ? c_.out_ld / 3 : 0;
        a.res_split = (c_.has_res && c_.split_res) ? a.res_ld / 3 : 0;
        a.res_x8 = c_.split_res == 2;
        a.x8_out = c_.split_out == 2 ? 1 : c_.split_out == 3 ? 2 : c_.split_out == 4 ? 3 : 0;
        a.alias3 = c_.split_in == 1 ? c_.Cin / 3 * 2 : 0;
        a.xscale = c_.split_in == 2 ? xscale_dev_ : nullptr;
        a.variant = variant_;
        a.wide = !c_.out_f32 && c_.Cout % 16 == 0 && c_.out_ld % 8 == 0 && c_.out_coff % 8 == 0 && (!c_.has_res || a.res_ld % 8 == 0) &&
                 a.split_out % 8 == 0 && a.res_split % 8 == 0;
        a.nb = (inDesc && inDesc[0].dims.nbDims == 4 && inDesc[0].dims.d[0] > 1) ? inDesc[0].dims.d[0] : 1;
        if ((long)a.nb * c_.H * c_.W * c_.Cin >= (1l << 31)) return -2;                // the halo kernel addresses the input with 32-bit element offsets
        static int tron = -1;                                                          // tools/trace_conv.py
        if (tron < 0) tron = ablateEnv("DSVT_CONV_TRACE", 0);
        if (tron && wp_dev_ && haloEligible()) {
            static unsigned long long* tr = nullptr;
            const size_t n = (size_t)4096 * 2 * CONV_TRACE_N;
            if (!tr && dsvtMalloc(&tr, n * 8) != hipSuccess) return -3;
            (void)hipMemsetAsync(tr, 0, n * 8, stream);
            a.trace = tr;
            const int rc = launchConvHalo(a, wp_dev_, zeros_dev_, stream);
            (void)hipStreamSynchronize(stream);
            std::vector<unsigned long long> h(n);
            (void)hipMemcpy(h.data(), tr, n * 8, hipMemcpyDeviceToHost);
            for (int wg : {0, 1, 131, 255})
                for (int hw = 0; hw < 2; ++hw) {
                    const unsigned long long* t = h.data() + (size_t)(wg * 2 + hw) * CONV_TRACE_N;
                    fprintf(stderr, "[conv trace wg%d wave%s t0=%llu]", wg, hw ? "N/2" : "0", t[0] - h[0]);
                    for (int i = 1; i < CONV_TRACE_N && t[i]; ++i) fprintf(stderr, " %lld", (long long)(t[i] - t[0]));
                    fprintf(stderr, "\n");
                }
            return rc;
        }
        if (c_.split_in == 2) return launchConvHalo(a, wp_dev_, zeros_dev_, stream);
        if (groups_ > 0 && !a.split_out) {
            const int tilesX = cdiv(a.Wo, HTW), tilesY = cdiv(a.Ho, 8);
            if (groups_split_) hipLaunchKernelGGL(conv3x3_grouped_narrow_split_kernel, dim3(numCUs()), dim3(512), 0, stream, a, wg_dev_, chan_dev_, zeros_dev_, tilesX, tilesY, groups_);
            else hipLaunchKernelGGL(conv3x3_grouped_narrow_kernel, dim3(numCUs()), dim3(512), 0, stream, a, wg_dev_, chan_dev_, zeros_dev_, tilesX, tilesY, groups_);
            return lastError();
        }
        if (wp_dev_ && conv1x1ResidentEligible(a)) return launchConv1x1Resident(a, wp_dev_, stream);
        if (wp_dev_ && conv3x3C64Eligible(a)) {
            const int tilesX = cdiv(a.Wo, HTW), tilesY = cdiv(a.Ho, C64_TH);
            const int ctw = haloChannelTiles(a.CoutRows), NCT = ctw < 8 ? ctw : cdiv(a.CoutRows, CNB) * 8;      // column tiles per k-step of the halo image
            hipLaunchKernelGGL(conv3x3_c64_resident_kernel, dim3(numCUs()), dim3(512), 0, stream, a, wp_dev_, NCT, zeros_dev_, tilesX, tilesY, a.CoutRows / 64);
            return lastError();
        }
        if (wp_dev_ && haloEligible()) return launchConvHalo(a, wp_dev_, zeros_dev_, stream);
        return launchConv(a, KC(), stream);
    }
    static constexpr int kTrailerMagic = 0x34585644;       // "DVX4": the four-int trailer {split_out, split_res, split_in, magic}
    size_t serializationSize() const override {
        return 14 * sizeof(int) + sizeof(int) + sizeof(float) * (w_.size() + b_.size()) + (c_.split_in ? 4 * sizeof(int) : (c_.split_out || c_.split_res) ? 2 * sizeof(int) : 0);
    }
    void serialize(void* buf) const override {
        char* d = static_cast<char*>(buf);
        const int* ci = reinterpret_cast<const int*>(&c_);
        for (int i = 0; i < 14; ++i) wr<int>(d, ci[i]);
        wr<int>(d, b_.empty() ? 0 : 1);
        memcpy(d, w_.data(), sizeof(float) * w_.size()); d += sizeof(float) * w_.size();
        memcpy(d, b_.data(), sizeof(float) * b_.size()); d += sizeof(float) * b_.size();
        // trailing, only when set (older blobs stay valid); deserialize accepts exactly the three lengths: none, two ints, four ints ending in the magic word
        if (c_.split_in) { wr<int>(d, c_.split_out); wr<int>(d, c_.split_res); wr<int>(d, c_.split_in); wr<int>(d, kTrailerMagic); }
        else if (c_.split_out || c_.split_res) { wr<int>(d, c_.split_out); wr<int>(d, c_.split_res); }
    }
    Plugin* clone() const override { auto* p = new DsvtConv2dPlugin(c_, w_.data(), b_.empty() ? nullptr : b_.data()); p->variant_ = variant_; return p; }
};

static Plugin* convNew(const ConvCfg& c, const float* w, const float* b) {
    if (c.H <= 0 || c.W <= 0 || c.Cin <= 0 || c.Cin % 32 != 0 || c.Cout <= 0 || c.KH <= 0 || c.KW <= 0 || !w) return nullptr;
    if (c.stride < 1 || c.pad < 0 || c.up < 1 || c.out_ld < c.out_coff + c.Cout) return nullptr;
    if (c.Cin % 64 != 0) return nullptr;                               // K slabs are 64 / 96 / 128 channels wide
    if (c.up > 1 && (c.KH != 1 || c.KW != 1 || c.stride != 1 || c.Cout % CNB != 0)) return nullptr;   // pixel-shuffle chunks are whole workgroup columns
    if (!c.out_f32 && (c.out_ld % 4 != 0 || c.out_coff % 4 != 0)) return nullptr;
    if (c.split_out && (c.out_f32 || c.out_ld % 12 != 0 || c.out_ld / 3 < c.out_coff + c.Cout)) return nullptr;      // three planes of out_ld / 3 channels
    if (c.split_out < 0 || c.split_out > 4 || c.split_in < 0 || c.split_in > 2) return nullptr;
    if (c.split_out >= 2 && ((c.out_ld / 3) % 32 != 0 || c.out_coff % 8 != 0)) return nullptr;       // the x8 plane is laid out in 32-channel groups
    if ((c.split_res && !c.has_res) || c.split_res < 0 || c.split_res > 2) return nullptr;       // (2: the residual triple has no lo plane, its lo part comes from the x8 plane)
    if (c.split_in && c.Cin % 192 != 0) return nullptr;                                              // three planes of whole 64-channel phases
    // split_in 2: the layers the fp16 + fp8 K loops serve -- 3 x 3 stride 1 with more than 32 output channels (conv_wide_kernel) and 1 x 1 stride 1 with
    // whole 128-row weight chunks, pixel shuffle included (conv_halo_kernel)
    if (c.split_in == 2 && !((c.KH == 3 && c.KW == 3 && c.stride == 1 && c.pad == 1 && c.up == 1 && c.Cout > 32) ||
                             (c.KH == 1 && c.KW == 1 && c.stride == 1 && c.pad == 0 && (c.up * c.up * c.Cout) % CNB == 0 && c.Cin % 192 == 0))) return nullptr;
    return new DsvtConv2dPlugin(c, w, b);
}
static Plugin* convCreate(const DsvtPluginFieldCollection* fc) {
    ConvCfg c{};
    c.H = fieldInt(fc, "in_height"); c.W = fieldInt(fc, "in_width"); c.Cin = fieldInt(fc, "in_channels"); c.Cout = fieldInt(fc, "out_channels");
    c.KH = c.KW = fieldInt(fc, "kernel_size", 1); c.stride = fieldInt(fc, "stride", 1); c.pad = fieldInt(fc, "padding", 0);
    c.up = fieldInt(fc, "pixel_shuffle", 1); c.relu = fieldInt(fc, "relu", 0); c.has_res = fieldInt(fc, "has_residual", 0);
    c.out_ld = fieldInt(fc, "out_channel_stride", c.Cout); c.out_coff = fieldInt(fc, "out_channel_offset", 0); c.out_f32 = fieldInt(fc, "out_f32", 0);
    c.split_out = fieldInt(fc, "split_output", 0); c.split_res = fieldInt(fc, "split_residual", 0); c.split_in = fieldInt(fc, "split_input", 0);
    const DsvtPluginField* w = findField(fc, "weight"); const DsvtPluginField* b = findField(fc, "bias");
    if (!w || !w->data || c.Cin <= 0 || c.Cout <= 0 || c.up < 1) return nullptr;
    if ((long)w->length != (long)c.up * c.up * c.Cout * c.KH * c.KW * (c.split_in == 2 ? c.Cin / 3 : c.Cin)) return nullptr;
    if (b && b->data && b->length != c.Cout) return nullptr;
    Plugin* p = convNew(c, static_cast<const float*>(w->data), (b && b->data) ? static_cast<const float*>(b->data) : nullptr);
    if (p) static_cast<DsvtConv2dPlugin*>(p)->variant_ = fieldInt(fc, "kernel_variant", 0);
    return p;
}
static Plugin* convDeser(const void* data, size_t len) {
    if (len < 15 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    ConvCfg c{}; int* ci = reinterpret_cast<int*>(&c);
    for (int i = 0; i < 14; ++i) ci[i] = rd<int>(d);
    int has_b = rd<int>(d);
    if (c.Cin <= 0 || c.Cout <= 0 || c.up < 1 || c.KH <= 0 || c.KW <= 0) return nullptr;
    // the blob's length decides which trailer it carries: none, {split_out, split_res}, or {split_out, split_res, split_in, magic}; with
    // split_in == 2 the weight rows hold Cin / 3 channels per tap.  Any other length is refused (trailing bytes are never guessed at).
    const size_t nwFull = (size_t)c.up * c.up * c.Cout * c.KH * c.KW * c.Cin, nbias = has_b ? c.Cout : 0;
    size_t nw = nwFull;
    const size_t base = 15 * sizeof(int) + sizeof(float) * nbias;
    if (len == base + sizeof(float) * nwFull) {
    } else if (len == base + sizeof(float) * nwFull + 2 * sizeof(int)) {
        const char* t = static_cast<const char*>(data) + len - 2 * sizeof(int); c.split_out = rd<int>(t); c.split_res = rd<int>(t);
    } else {
        if (len < base + 4 * sizeof(int)) return nullptr;
        const char* t = static_cast<const char*>(data) + len - 4 * sizeof(int);
        c.split_out = rd<int>(t); c.split_res = rd<int>(t); c.split_in = rd<int>(t);
        if (rd<int>(t) != DsvtConv2dPlugin::kTrailerMagic || c.split_in < 1 || c.split_in > 2) return nullptr;
        nw = c.split_in == 2 ? nwFull / 3 : nwFull;
        if (len != base + sizeof(float) * nw + 4 * sizeof(int)) return nullptr;
    }
    std::vector<float> w(nw), b(nbias);
    memcpy(w.data(), d, sizeof(float) * nw); if (has_b) memcpy(b.data(), d + sizeof(float) * nw, sizeof(float) * c.Cout);
    return convNew(c, w.data(), has_b ? b.data() : nullptr);
}
static Creator g_convCreator{"DsvtConv2dPlugin",
    {{"in_height", DSVT_FIELD_INT32}, {"in_width", DSVT_FIELD_INT32}, {"in_channels", DSVT_FIELD_INT32}, {"out_channels", DSVT_FIELD_INT32},
     {"kernel_size", DSVT_FIELD_INT32}, {"stride", DSVT_FIELD_INT32}, {"padding", DSVT_FIELD_INT32}, {"pixel_shuffle", DSVT_FIELD_INT32},
     {"relu", DSVT_FIELD_INT32}, {"has_residual", DSVT_FIELD_INT32}, {"out_channel_stride", DSVT_FIELD_INT32},
     {"out_channel_offset", DSVT_FIELD_INT32}, {"out_f32", DSVT_FIELD_INT32}, {"split_output", DSVT_FIELD_INT32}, {"split_residual", DSVT_FIELD_INT32},
     {"split_input", DSVT_FIELD_INT32}, {"kernel_variant", DSVT_FIELD_INT32},
     {"weight", DSVT_FIELD_FLOAT32}, {"bias", DSVT_FIELD_FLOAT32}},
    convCreate, convDeser, {}, {}};
static Registrar g_convReg(&g_convCreator);


// =====================================================================================
// DsvtSplitHalfPlugin -- fp32 activations as the operand of an fp32-GRADE convolution on the fp16 matrix cores.
// fp32 mode of the frame (the mode whose boxes meet the 1e-3 bar) used to run its BEV stage on vendor fp32 convolutions.  Instead:
// a = hi + lo with hi = fp16(a), lo = fp16(a - hi) (|a - hi - lo| <= 2^-22 |a|), w = w_hi + w_lo likewise, and
//     conv(a, w) = conv(hi, w_hi) + conv(lo, w_hi) + conv(hi, w_lo)      (+ lo * w_lo ~ 2^-22, dropped)
// which is ONE DsvtConv2dPlugin launch over 3 Cin input channels [hi | lo | hi] with weight rows [w_hi | w_hi | w_lo] (built by the
// host: plugin.split_weight_rows), fp32 accumulation, fp32 output.  This op is the glue between two such convolutions:
//     y  = relu?(x (+ residual))           fp32 [.., C]     (output 0; the residual stream of the ResNet blocks stays fp32)
//     y3 = [fp16(y) | fp16(y - fp16(y)) | fp16(y)]          fp16 [.., 3C]   (output 1: the next convolution's input)
// Inputs: x [1,H,W,C] f32 (, residual [1,H,W,C] f32).  Reference semantics restated: convBnLELU / convBn + ElementWise SUM + ReLU
// of src/dsvt-ai-trt.cpp:149-246, 1144-1364 in fp32.
// =====================================================================================
__global__ void __launch_bounds__(256)
split_half_kernel(const float4* __restrict__ x, const float4* __restrict__ res, size_t groups, int C4, int relu,
                  float4* __restrict__ y, half4* __restrict__ y3)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;        // one group of four channels of one pixel
    if (i >= groups) return;
    float4 v = x[i];
    if (res) { const float4 r = res[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    y[i] = v;
    half4 hi, lo;
    hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
    lo[0] = (_Float16)(v.x - (float)hi[0]); lo[1] = (_Float16)(v.y - (float)hi[1]);
    lo[2] = (_Float16)(v.z - (float)hi[2]); lo[3] = (_Float16)(v.w - (float)hi[3]);
    const size_t pix = i / (size_t)C4, c4 = i % (size_t)C4;
    half4* o = y3 + pix * 3 * (size_t)C4 + c4;
    o[0] = hi; o[C4] = lo; o[2 * (size_t)C4] = hi;
}

class DsvtSplitHalfPlugin : public Plugin {
public:
    int C_, relu_, has_res_;
    DsvtSplitHalfPlugin(int C, int relu, int has_res) : C_(C), relu_(relu), has_res_(has_res) {}
    const char* type() const override { return "DsvtSplitHalfPlugin"; }
    int nbOutputs() const override { return 2; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i < 0 || i > 1 || in[0].nbDims < 2) return -1;
        *out = in[0];
        if (i == 1) out->d[out->nbDims - 1] = 3 * C_;
        return 0;
    }
    int outputType(int i, const int32_t*, int) const override { return i == 0 ? DSVT_FLOAT : DSVT_HALF; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int nbIn, int) const override {
        if (io[pos].format != DSVT_FORMAT_LINEAR) return false;
        return pos <= nbIn ? io[pos].type == DSVT_FLOAT : io[pos].type == DSVT_HALF;
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        if (!inDesc || inDesc[0].dims.nbDims < 2 || inDesc[0].dims.d[inDesc[0].dims.nbDims - 1] != C_) return -2;
        size_t pix = 1;
        for (int k = 0; k + 1 < inDesc[0].dims.nbDims; ++k) pix *= (size_t)inDesc[0].dims.d[k];
        const size_t groups = pix * (size_t)(C_ / 4);
        if (groups == 0) return 0;
        hipLaunchKernelGGL(split_half_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, stream, static_cast<const float4*>(in[0]),
                           has_res_ ? static_cast<const float4*>(in[1]) : nullptr, groups, C_ / 4, relu_, static_cast<float4*>(out[0]),
                           static_cast<half4*>(out[1]));
        return lastError();
    }
    size_t serializationSize() const override { return 3 * sizeof(int); }
    void serialize(void* b) const override { char* d = static_cast<char*>(b); wr<int>(d, C_); wr<int>(d, relu_); wr<int>(d, has_res_); }
    Plugin* clone() const override { return new DsvtSplitHalfPlugin(C_, relu_, has_res_); }
};
static Plugin* shNew(int C, int relu, int hr) { return (C > 0 && C % 4 == 0) ? new DsvtSplitHalfPlugin(C, relu != 0, hr != 0) : nullptr; }
static Plugin* shCreate(const DsvtPluginFieldCollection* fc) { return shNew(fieldInt(fc, "channel_num"), fieldInt(fc, "relu", 0), fieldInt(fc, "has_residual", 0)); }
static Plugin* shDeser(const void* data, size_t len) {
    if (len < 3 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    const int C = rd<int>(d), relu = rd<int>(d), hr = rd<int>(d);
    return shNew(C, relu, hr);
}
static Creator g_shCreator{"DsvtSplitHalfPlugin", {{"channel_num", DSVT_FIELD_INT32}, {"relu", DSVT_FIELD_INT32}, {"has_residual", DSVT_FIELD_INT32}},
                           shCreate, shDeser, {}, {}};
static Registrar g_shReg(&g_shCreator);

}  // namespace dsvt
