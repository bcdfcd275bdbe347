// conv_rows.hip -- conv_rows_kernel: the 3 x 3 stride-1 layers with 128-channel output chunks (the BEV ResNet blocks, src/dsvt-ai-trt.cpp:1144-1351) in the
// fp32-grade frame, with a slab loop the matrix pipe can be kept busy under.
//
// Round 5 measured what conv_wide_kernel<8, 8, 36, 4, 2, 2, SPL> (39 % of the frame in one symbol) loses time to (profiles/r05_conv_issue_bound.txt): its slab --
// four (phase, tap) steps = 128 MFMAs per wave -- carries ~700 OTHER instructions per wave (request addresses of ~45 instructions each, the schedule of which
// halo phase / weight slab goes where, exec-mask branches around every request), alternating in blocks with the MFMAs; an in-order wave cannot issue them under
// its own MFMAs and the two waves of a SIMD overlap each other's blocks only by chance: 6100 cycles per slab where the matrix pipe needs 4096.  Variants that
// shortened the other blocks by 8-14 % bought 1-3 %.  This kernel removes the other instructions instead of shortening them, by choosing the slab so that its
// schedule is a compile-time constant:
//
//   * same tile as conv_wide_kernel<8, 8, ..>: 16 rows x 32 pixels x 128 channels per workgroup, eight waves of two rows, 128 accumulator registers, the K loop
//     walks 32-channel phases of the [hi | lo | hi] input (a halo pixel = 64 B of LDS, chunk c of halo column hx in slot c ^ ((hx >> 1) & 2)), the packed
//     weights of DsvtConv2dPlugin::packHalo unchanged, the (phase, tap) steps in the same order: every accumulator sees the same sequence of MFMAs as in
//     conv_wide_kernel, so the results are BIT-IDENTICAL (tests/test_conv_gpu.py::test_rows_kernel_equals_wide_kernel);
//   * a slab = the THREE taps of one kernel row ky of one phase (96 MFMAs per wave, 24 KB of weights): a phase is exactly three slabs, slab s lives in weight
//     buffer s mod 3 = ky and halo buffer = phase parity, so with the loop body = two phases = six slabs, every LDS address of every fragment read is a
//     per-lane base register + an immediate, and every slab issues the same requests: the weights of slab s + 2 (three 1 KB rows per wave) and, in the
//     ky = 0 / 1 slabs, three / two of the wave's five halo pieces of the NEXT phase (34-pixel halo rows: 18 x 34 pixels = 39 pieces, five per wave);
//   * a request = s_mov m0 + one buffer_load_dwordx4 ... lds: the per-lane byte offsets of the wave's five halo pieces (validity folded in: a lane outside
//     the image carries an offset beyond num_records and the hardware writes the zeros) are computed ONCE PER ITEM into five registers, the phase / plane
//     and the weight row travel in the scalar offset; no branch, no per-request address arithmetic;
//   * no conditional request: the last phase of an item requests the first phase / slabs of the NEXT item (a workgroup without a next item re-requests its own:
//     the bytes land in buffers nobody reads again).
// What is left per slab and wave beside the 96 MFMAs: 36 ds_read_b128, 3-6 requests, a dozen scalar instructions, one s_waitcnt + s_barrier.
#include "plugin_base.h"
#include "device_utils.h"
#include "conv_args.h"
#include <type_traits>

namespace dsvt {

constexpr int RK_TW = 32;                          // tile width in pixels
constexpr int RK_HS = RK_TW + 2;                   // halo row stride in pixels (a fragment read touches ONE halo row: its bank pattern does not depend on the stride)
constexpr int RK_NW = 8;
// CT = 16-channel tiles per workgroup, RW = tile rows per wave.  <8, 2>: 16 rows x 32 pixels x 128 channels (128 accumulator registers; halo 18 x 34 pixels = 39 pieces
// of 1 KB, five per wave; 24 KB weight slabs) -- the BEV ResNet layers.  <4, 3>: 24 rows x 32 pixels x 64 channels (96 accumulator registers; halo 26 x 34 = 56 pieces,
// seven per wave; 12 KB weight slabs: twelve rows, so waves 0-3 request two and waves 4-7 one) -- the layers with 64-channel chunks (the shared 384 -> 64 head
// convolution, the 64 -> 320 head stems), whose item height round 5 chose for the weight stream per output pixel (conv_wide_kernel<4, 8, 36, 4, 2, 3>).
template <int CT, int RW>
struct RowsCfg {
    static constexpr int ROWS = 8 * RW, HH = ROWS + 2, NM = 2 * RW;
    static constexpr int PPW = (HH * RK_HS + 16 * RK_NW - 1) / (16 * RK_NW);   // halo pieces (1 KB = 16 pixels) per wave and phase
    static constexpr int NPC = RK_NW * PPW, HBYTES = NPC * 1024;               // (pieces beyond the halo are padding: their lanes carry out-of-range offsets)
    static constexpr int WBYTES = 3 * CT * 1024;                               // three taps x CT 16-channel tiles
    static constexpr int WOFF = 2 * HBYTES, BIAS = WOFF + 3 * WBYTES, SMEM = BIAS + 1024;
    static constexpr int BPS = CT / 4, NB = 3 * BPS;                           // batches (four channel tiles x NM pixel tiles of MFMAs) per step / per slab
    static constexpr int RPW = (3 * CT + RK_NW - 1) / RK_NW;                   // weight-request slots per wave and slab
    static constexpr bool WUNI = (3 * CT) % RK_NW == 0;                        // every wave fills every slot
    static constexpr bool BIGH = HBYTES + (2 * RK_HS + 2) * 64 + ((RW - 1) * RK_HS + 16) * 64 + 64 > 65535;   // the second halo buffer lies beyond a 16-bit offset: a base per buffer
    static_assert(CT % 4 == 0 && SMEM <= 160 * 1024, "LDS budget");
};
constexpr uint32_t RK_OOB = 0xFFFFFFF0u;           // byte offset of a lane whose halo pixel lies outside the image (>= num_records: the request writes zeros)

// RK_HALO_IN_KY0 = 1: all five halo pieces of the next phase are requested in the phase's FIRST slab and its slab-end wait leaves them in flight (they have two
// slabs to land: a halo line that misses the L2 takes ~900 cycles, a slab of MFMAs ~3000 per SIMD); 0: three in the first slab, two in the second, each
// waited for at its own slab end
#ifndef RK_HALO_IN_KY0
#define RK_HALO_IN_KY0 0
#endif
#ifndef RK_STAGGER
#define RK_STAGGER 0
#endif
#ifndef RK_SETPRIO
#define RK_SETPRIO 0
#endif
#ifndef RK_SKEW
#define RK_SKEW 10
#endif
// TR: the instrumented instantiation (ablation build, DSVT_CONV_TRACE=1, tools/trace_conv_rows.py): s_memtime stamps of waves 0 and 4 -- per slab [start, MFMAs
// issued, own requests landed, barrier passed], per item [K loop done, epilogue done]
// ABL: timing ablations (instantiated in the -DDSVT_ABLATE build only, DSVT_CONV_DBG; wrong results): 1 = no halo requests, 2 = no weight requests, 4 = no epilogue,
// 8 = no fragment reads
// EPI: the epilogue's flavour as a compile-time constant (the generic SPL epilogue with every residual / output-plane variant unrolled sixteen times is 35 KB of
// code, more than half of the 64 KB instruction cache): 0 = generic; 1 = no residual, output [hi | lo | -] (split_output = 4); 2 = [hi | lo] residual, output [hi | lo | -]
template <int CT, int RW, bool SPL, bool TR = false, int ABL = 0, int EPI = 0>
__global__ void __launch_bounds__(64 * RK_NW, 1)
conv_rows_kernel(ConvArgs a, const _Float16* __restrict__ Wp, int tilesX, int nitems, int nchunk, int yBase)
{
    using C = RowsCfg<CT, RW>;
    constexpr int RK_ROWS = C::ROWS, RK_HH = C::HH, NM = C::NM, RK_PPW = C::PPW, RK_HBYTES = C::HBYTES, RK_WBYTES = C::WBYTES, RK_WOFF = C::WOFF, RK_BIAS = C::BIAS;
    constexpr int BPS = C::BPS, NB = C::NB, RPW = C::RPW;
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::SMEM];          // halo[2] | wslab[3] | bias
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, g = lane >> 4;
    const bool upper = wave >= RK_NW / 2;                                          // (RK_STAGGER) the second wave of its SIMD
    const int NP = a.Cin >> 5;                                                     // 32-channel phases (even: checked by the launcher)
    const int NCT = a.CoutRows <= 64 ? 4 : (a.CoutRows + 127) / 128 * 8;           // 16-channel tiles per k-step of the packed weights (DsvtConv2dPlugin::packHalo)
    const int perImg = nitems / a.nb;
    auto decode = [&](int it, int& yy, int& xx, int& ch, int& bb) {
        bb = it / perImg; it -= bb * perImg;
        ch = it % nchunk; const int t = it / nchunk;
        yy = yBase + (t / tilesX) * RK_ROWS; xx = (t % tilesX) * RK_TW;      // (yBase: the launch covers the tile rows from image row yBase on)
    };
    const __amdgpu_buffer_rsrc_t inRsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.in), 0, (int)((size_t)a.nb * a.H * a.W * a.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wRsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(Wp), 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t bRsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);

    // byte offsets of this lane's 16 bytes of the wave's five halo pieces of the tile at (yy, xx) of image bb, phase 0
    uint32_t hv[RK_PPW];
    auto haloOffsets = [&](int yy, int xx, int bb) {
#pragma unroll
        for (int i = 0; i < RK_PPW; ++i) {
            const int lp = (wave + RK_NW * i) * 16 + (lane >> 2), hy = lp / RK_HS, hx = lp - hy * RK_HS;
            const int chunk = (lane & 3) ^ ((hx >> 1) & 2);
            const int gy = yy - 1 + hy, gx = xx - 1 + hx;
            const bool ok = hy < RK_HH && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            hv[i] = ok ? (uint32_t)((((bb * a.H + gy) * a.W + gx) * a.Cin + chunk * 8) * 2) : RK_OOB;
        }
    };
    // byte offset (scalar) of the 32 channels of phase ph within a pixel: the phases of the third plane of [hi | lo | hi] read plane 0 (alias3, see ConvArgs)
    auto phaseOff = [&](int ph) { const int c = ph * 32; return (uint32_t)(((a.alias3 && c >= a.alias3) ? c - a.alias3 : c) * 2); };
    auto haloRequest = [&](int i, int buf, uint32_t soff) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(inRsrc, (glds_dst_t)(smem + buf * RK_HBYTES + (wave + RK_NW * i) * 1024), 16, (int)hv[i], (int)soff, 0, 0);
    };
    // weight rows of slab (ph, ky) of chunk ch: tap 3 ky + j, channel tile `wave` -> row (2 ((ph >> 1) 9 + tap) + (ph & 1)) NCT + 8 ch + wave of the packed image
    // slot j of wave w is row u = w + 8 j of the slab's 3 CT rows: tap u / CT, channel tile u % CT (scalars)
    const uint32_t wv = (uint32_t)lane * 16u;
    const uint32_t wstep = (uint32_t)(2 * NCT * 1024);
    auto weightBase = [&](int ph, int ky, int ch) { return (uint32_t)(((2 * ((ph >> 1) * 9 + 3 * ky) + (ph & 1)) * NCT + ch * CT) * 1024); };
    auto weightRequest = [&](uint32_t base, int buf, int j) {
        const int u = wave + RK_NW * j;
        if (!C::WUNI && u >= 3 * CT) return;                                       // (wave-uniform)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wRsrc, (glds_dst_t)(smem + RK_WOFF + buf * RK_WBYTES + u * 1024), 16, (int)wv, (int)(base + (u / CT) * wstep + (u % CT) * 1024), 0, 0);
    };
    auto weightRequests = [&](uint32_t base, int buf) {
#pragma unroll
        for (int j = 0; j < RPW; ++j) weightRequest(base, buf, j);
    };
    const int wreq = C::WUNI ? RPW : ((wave + RK_NW * (RPW - 1) < 3 * CT) ? RPW : RPW - 1);      // requests this wave issues per slab = what its slab-end wait leaves in flight
    // the chunk's bias (128 floats, zeros beyond Cout / without a bias: out-of-range lanes) as one LDS-DMA piece: the accumulators start from it
    auto biasRequest = [&](int ch) {
        const int co = ch * CT * 16 + lane * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(bRsrc, (glds_dst_t)(smem + RK_BIAS), 16, (int)((lane < CT * 4 && co < a.Cout) ? (uint32_t)co * 4u : RK_OOB), 0, 0, 0);
    };

    int item = blockIdx.x;
    if (item >= nitems) return;
#if RK_SKEW
    // Start skew (round 6).  Every item takes the same time, so the 256 persistent workgroups of a launch reach their item ends -- the store burst of 512 pixels x 128 channels
    // x two planes, the residual read -- together, round after round: HBM sees bursts and idles in between.  A launch whose item count is not a multiple of the grid has
    // workgroups that do one item FEWER than the others (b >= nitems % grid) and wait for the launch's last round anyway: those start late by ((b - rem) % 8) x
    // NP x RK_SKEW x 64 cycles (up to ~0.6 item times), which spreads the item ends of the chip over the item period at no cost to the launch.  468 x 468 128 -> 128, four
    // images (1800 items = 7 rounds + 8): -2.4 %, with a residual -4 %, one image -2.5 %, the 64-channel layers -0.4 .. -1.2 % (RK_SKEW = 10; 20: the same within noise; 35: loses on the
    // two-round layers; tools/ab_variants.sh, profiles/r06_conv_rows_ablations.txt).  A launch with whole rounds is not skewed.
    {
        const int rem = nitems % (int)gridDim.x;
        if (rem != 0 && (int)blockIdx.x >= rem && nitems > (int)gridDim.x)
            for (int k = (((int)blockIdx.x - rem) & 7) * NP * RK_SKEW; k > 0; k -= 100) __builtin_amdgcn_s_sleep(100);
    }
#endif
#if RK_SETPRIO
    if (upper) __builtin_amdgcn_s_setprio(1);                     // (MI355X_MICROARCH "Two waves per SIMD" item 4: static priority for the younger half)
#endif
    int nmark = 0;
    const bool tracing = TR && a.trace != nullptr && (wave == 0 || wave == RK_NW / 2);
    auto mark = [&]() {
        if constexpr (TR) {
            if (tracing) { if (lane == 0 && nmark < CONV_TRACE_N) a.trace[(size_t)(blockIdx.x * 2 + (wave != 0)) * CONV_TRACE_N + nmark] = clock64(); ++nmark; }
        }
    };
    mark();
    int y0, x0, chunk, bimg;
    decode(item, y0, x0, chunk, bimg);
    haloOffsets(y0, x0, bimg);
    if (wave == 0) biasRequest(chunk);
#pragma unroll
    for (int i = 0; i < RK_PPW; ++i) haloRequest(i, 0, phaseOff(0));
    weightRequests(weightBase(0, 0, chunk), 0);
    weightRequests(weightBase(0, 1, chunk), 1);
    slabBarrier(0);

    // fragment addresses: per-lane bases + immediates
    const int pb = ((RW * wave) * RK_HS + r) * 64;
    constexpr int NVB = C::BIGH ? 2 : 1;                                           // (BIGH: one set of bases per halo buffer)
    int vB[NVB][3];
#pragma unroll
    for (int pp = 0; pp < NVB; ++pp)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) vB[pp][kx] = pp * RK_HBYTES + pb + kx * 64 + ((g ^ (((r + kx) >> 1) & 2)) << 4);
    int vA[3];                                                                     // (one base per weight buffer: the immediates stay below 64 KB)
#pragma unroll
    for (int i = 0; i < 3; ++i) vA[i] = RK_WOFF + i * RK_WBYTES + (lane << 4);
    // (opaque to the optimizer: left visible, hipcc re-associates base + immediate into one constant per access, which no longer fits the 16-bit offset field,
    // and keeps 72 address registers alive around the loop -- 200 spilled registers, each reload an s_waitcnt vmcnt(0) in the middle of the LDS-DMA stream)
#pragma unroll
    for (int i = 0; i < 3; ++i) { asm volatile("" : "+v"(vA[i])); asm volatile("" : "+v"(vB[0][i])); if (C::BIGH) asm volatile("" : "+v"(vB[NVB - 1][i])); }

    floatx4 acc[CT][NM];
    half8 Bf[2][NM], Af[2][4];
    auto loadB1 = [&](int par, int ky, int kx, int m, half8 (&B)[NM]) {            // (par, ky, kx, m: constants after inlining)
        if (ABL & 8) return;
        B[m] = *reinterpret_cast<const half8*>(smem + vB[C::BIGH ? par : 0][kx] + ((C::BIGH ? 0 : par * RK_HBYTES) + ky * RK_HS * 64) + ((m >> 1) * RK_HS + (m & 1) * 16) * 64);
    };
    auto loadB = [&](int par, int ky, int kx, half8 (&B)[NM]) {
#pragma unroll
        for (int m = 0; m < NM; ++m) loadB1(par, ky, kx, m, B);
    };
    auto loadA1 = [&](int buf, int u, int c0, int ct, half8 (&A)[4]) {
        if (ABL & 8) return;
        A[ct] = *reinterpret_cast<const half8*>(smem + vA[buf] + (u * CT + c0 + ct) * 1024);
    };
    auto loadA = [&](int buf, int u, int c0, half8 (&A)[4]) {
        if (ABL & 8) return;
        const unsigned char* p = smem + vA[buf] + (u * CT + c0) * 1024;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) A[ct] = *reinterpret_cast<const half8*>(p + ct * 1024);
    };

    loadB(0, 0, 0, Bf[0]);
    for (;;) {
        int nitem = item + gridDim.x, ny0, nx0, nch, nbimg;
        const bool have_next = nitem < nitems;
        if (!have_next) nitem = item;                              // (no next item: the end-of-item requests fetch this item's first phase again, into buffers nobody reads)
        decode(nitem, ny0, nx0, nch, nbimg);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const floatx4 b4 = *reinterpret_cast<const floatx4*>(smem + RK_BIAS + (ct * 16 + 4 * g) * 4);
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[ct][m] = b4;
        }

        // one slab: kernel row KY of phase P (parity PAR)
        // TAIL: P is the item's last phase (the halo / weights requested now belong to the next item); compile-time: the last two phases are peeled
        auto slab = [&](const int P, auto kyTag, auto parTag, auto tailTag, auto upTag) {
            constexpr int KY = decltype(kyTag)::value, PAR = decltype(parTag)::value;
            constexpr bool tail = decltype(tailTag)::value;
            constexpr bool UP = decltype(upTag)::value;            // (RK_STAGGER) this wave is the second of its SIMD: the whole K loop exists twice
            // --- requests: halo pieces of phase P + 1 (ky = 0: three, ky = 1: two), then the weights of slab s + 2 (three): one request per batch of MFMAs
            const uint32_t hso = tail ? phaseOff(0) : phaseOff(P + 1);
            if (KY == 2 && tail && wave == 0) biasRequest(nch);   // (before the weights: retired with the older requests)
            constexpr bool wnext = tail && KY != 0;               // the slab two ahead is slab 0 / 1 of the next item
            const uint32_t wbase = weightBase(wnext ? 0 : (KY == 0 ? P : P + 1), (KY + 2) % 3, wnext ? nch : chunk);
            constexpr int NH0 = (RK_PPW + 1) / 2;                  // (five pieces: three in the first slab, two in the second; seven: four and three)
            constexpr int NH = RK_HALO_IN_KY0 ? (KY == 0 ? RK_PPW : 0) : (KY == 0 ? NH0 : KY == 1 ? RK_PPW - NH0 : 0);      // halo requests of this slab
            constexpr int NREQ = NH + RPW;
            constexpr int NPOS = NB == 6 ? 6 : 4 * NB;             // where a request may go: before a batch (six batches per slab) or before any group of four / six MFMAs (three)
            auto request = [&](int j) {                            // j = 0 .. NREQ - 1 (a constant after unrolling)
                if (j < NH) { if (!(ABL & 1)) haloRequest((KY == 0 ? 0 : NH0) + j, PAR ^ 1, hso); }
                else if (j < NREQ) { if (!(ABL & 2)) weightRequest(wbase, (KY + 2) % 3, j - NH); }
            };
            mark();
            // --- 3 steps x 2 batches of 16 MFMAs; the fragments of batch b + 1 are read before the MFMAs of batch b
            // Fragment reads and requests are spread over the batch instead of issued as a burst before it (RK_SPREAD): eight waves leave the barrier together, and a
            // burst of eight ds_read_b128 per wave at every batch boundary is 256 LDS cycles during which no wave issues an MFMA (without any fragment read the layer
            // takes 525 instead of 685 us, tools/abl_conv_rows.sh).  Group g of a batch = the four MFMAs of channel tile g; before it: fragment g of the next batch's A
            // (and B) tiles.  The requests of a batch go out before group 0 (waves 0-3) or group 2 (their SIMD partners 4-7: RK_STAGGER).
            loadA(KY, 0, 0, Af[0]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int u = b / BPS, c0 = (b % BPS) * 4, t = PAR + KY + u;      // t & 1: the B buffer of step (P, KY, u)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    if (b + 1 < NB) {
                        if ((b + 1) % BPS == 0) {
#pragma unroll
                            for (int m = gq; m < NM; m += 4) loadB1(PAR, KY, u + 1, m, Bf[(t + 1) & 1]);
                        }
                        loadA1(KY, (b + 1) / BPS, ((b + 1) % BPS) * 4, gq, Af[(b + 1) & 1]);
                    } else {                                       // the next slab's first B fragments (its halo phase was published a slab ago at the latest); the item's
#pragma unroll
                        for (int m = gq; m < NM; m += 4) {         // last slab reads the NEXT item's: they wait in Bf[0] through the epilogue
                            if (KY == 2) loadB1(PAR ^ 1, 0, 0, m, Bf[(t + 1) & 1]); else loadB1(PAR, KY + 1, 0, m, Bf[(t + 1) & 1]);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < NREQ; ++j)                 // request j goes out at position j * NPOS / NREQ of the slab (the weights last)
                        if ((NB == 6 ? (gq == ((RK_STAGGER && UP) ? 2 : 0) && (j * NPOS) / NREQ == b) : (j * NPOS) / NREQ == b * 4 + gq)) request(j);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < NM; ++m)
                        acc[c0 + gq][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Af[b & 1][gq], Bf[t & 1][m], acc[c0 + gq][m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // everything but this slab's three weight requests (RK_HALO_IN_KY0: and the five halo requests of a first slab) has landed
            const int KEEP = ((RK_HALO_IN_KY0 && KY == 0) ? RK_PPW : 0) + wreq;
            if constexpr (TR) {
                mark();
                slabWait(KEEP);
                mark();
                __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
                mark();
            } else slabBarrier(KEEP);
        };
        using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
        using NT = std::false_type; using TL = std::true_type;
        auto kloop = [&](auto up) {
#pragma unroll 1
            for (int P = 0; P < NP - 2; P += 2) {
                slab(P, K0{}, K0{}, NT{}, up); slab(P, K1{}, K0{}, NT{}, up); slab(P, K2{}, K0{}, NT{}, up);
                slab(P + 1, K0{}, K1{}, NT{}, up); slab(P + 1, K1{}, K1{}, NT{}, up); slab(P + 1, K2{}, K1{}, NT{}, up);
            }
            slab(NP - 2, K0{}, K0{}, NT{}, up); slab(NP - 2, K1{}, K0{}, NT{}, up); slab(NP - 2, K2{}, K0{}, NT{}, up);
            haloOffsets(ny0, nx0, nbimg);                          // (this item's halo phases have all been requested: the last phase requests the next item's first)
            slab(NP - 1, K0{}, K1{}, TL{}, up); slab(NP - 1, K1{}, K1{}, TL{}, up); slab(NP - 1, K2{}, K1{}, TL{}, up);
        };
        if (RK_STAGGER && upper) kloop(std::true_type{}); else kloop(std::false_type{});
        mark();
        // residual / ReLU / store (the bias is in the accumulators): conv_wide_kernel's epilogue for a.wide layers
        if (!(ABL & 4)) {
            const int n0 = chunk * CT * 16;
            int ctn = (a.CoutRows - n0 + 15) / 16; ctn = ctn > CT ? CT : ctn;
            constexpr int TP = CT / 2, NBLK = NM * TP;            // blocks b = (pixel tile m, channel-tile pair tp)
            const int cg8 = (g & 1) * 16 + (g >> 1) * 8;
            bool valid[NM]; size_t opix[NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int oy = y0 + RW * wave + (m >> 1), ox = x0 + (m & 1) * 16 + r;
                valid[m] = oy < a.Ho && ox < a.Wo;
                opix[m] = valid[m] ? (size_t)(bimg * a.Ho + oy) * a.Wo + ox : 0;
            }
            const bool hasRes = EPI == 0 ? a.res != nullptr : EPI == 2;
            if constexpr (SPL) {
                constexpr int RS = 2;                              // residual blocks in flight.  4 / 8 / 16 (27 / 52 / 130 spilled registers): 676 / 700 / 753 us against 669 on the
                                                                   // 468 x 468 128 -> 128 residual layer -- the chip reads its residuals in one burst either way (profiles/r06_conv_rows_ablations.txt 9)
                const bool splitRes = EPI == 0 ? (hasRes && a.res_split != 0) : EPI == 2, resX8 = EPI == 0 && splitRes && a.res_x8 != 0;
#pragma unroll
                for (int b0 = 0; b0 < NBLK; b0 += RS) {
                    half8 rh[RS], rl[RS];
                    if (hasRes) {
#pragma unroll
                        for (int j = 0; j < RS; ++j) {
                            const int b = b0 + j, m = b / TP, tp = b % TP, co = n0 + tp * 32 + cg8;
                            const bool ok = valid[m] && co < a.Cout && 2 * tp < ctn;
                            rh[j] = *reinterpret_cast<const half8*>(a.res + (ok ? opix[m] * a.res_ld + co : 0));
                            if (resX8) {
                                const uint2 q = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(a.res + (ok ? opix[m] * a.res_ld + 2 * a.res_split : 0)) + (ok ? x8Offset(co) : 0));
                                rl[j] = __builtin_bit_cast(half8, make_uint4(q.x, q.y, 0u, 0u));
                            } else rl[j] = *reinterpret_cast<const half8*>(a.res + ((ok && splitRes) ? opix[m] * a.res_ld + a.res_split + co : 0));
                        }
                    }
#pragma unroll
                    for (int j = 0; j < RS; ++j) {
                        const int b = b0 + j, m = b / TP, tp = b % TP, co = n0 + tp * 32 + cg8;
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
                        if constexpr (EPI == 0) storeHalf8<true>(a, v, opix[m], co);
                        else {                                     // [hi | lo | -]: two 16-byte stores
                            _Float16* o = static_cast<_Float16*>(a.out) + opix[m] * a.out_ld + a.out_coff + co;
                            half8 hi, lo;
                            splitPlanes<8>(v, hi, lo);
                            *reinterpret_cast<half8*>(o) = hi;
                            *reinterpret_cast<half8*>(o + a.split_out) = lo;
                        }
                    }
                }
            } else {
                constexpr int RB = NBLK % 8 == 0 ? 8 : 4;
#pragma unroll
                for (int b0 = 0; b0 < NBLK; b0 += RB) {
                    half8 rv[RB];
                    if (hasRes) {
#pragma unroll
                        for (int j = 0; j < RB; ++j) {
                            const int b = b0 + j, m = b / TP, tp = b % TP, co = n0 + tp * 32 + cg8;
                            const bool ok = valid[m] && co < a.Cout && 2 * tp < ctn;
                            rv[j] = *reinterpret_cast<const half8*>(a.res + (ok ? opix[m] * a.res_ld + co : 0));
                        }
                    }
#pragma unroll
                    for (int j = 0; j < RB; ++j) {
                        const int b = b0 + j, m = b / TP, tp = b % TP, co = n0 + tp * 32 + cg8;
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
            }
        }
        mark();
        if (!have_next) break;
        item = nitem; y0 = ny0; x0 = nx0; chunk = nch; bimg = nbimg;
    }
}

// the layers these kernels take: 3 x 3, stride 1, pad 1, no pixel shuffle, 16-byte epilogue accesses legal, an even number of 32-channel phases, whole-chip grids.
// ct4 = false: 128-channel chunks on 16-row items; ct4 = true: 64-channel chunks on 24-row items (layers with <= 64 output rows, or whose last 128-channel chunk
// would be half empty)
static bool rowsCommon(const ConvArgs& a) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.up != 1 || !a.wide || a.out_f32 || a.xscale) return false;
    if (a.CoutRows != a.Cout || a.Cin % 64 != 0 || a.Cin < 64 || a.Ho != a.H || a.Wo != a.W) return false;
    return (size_t)a.nb * a.H * a.W * a.Cin * 2 < 0xF0000000ull;                       // (byte offsets in 32 bits, RK_OOB beyond num_records)
}
bool convRowsEligible(const ConvArgs& a, int ncu) {
    if (!rowsCommon(a) || a.CoutRows <= 64) return false;
    return cdiv(a.Ho, 16) * cdiv(a.Wo, RK_TW) * cdiv(a.CoutRows, 128) * a.nb >= ncu;
}
bool convRows64Eligible(const ConvArgs& a, int ncu) {
    if (!rowsCommon(a) || a.CoutRows <= 32 || !(a.CoutRows <= 64 || a.CoutRows % 128 == 64)) return false;
    if (!(a.split_out != 0 || a.res_split != 0)) return false;                         // (the fp16 frame keeps its own kernels)
    return cdiv(a.Ho, 24) * cdiv(a.Wo, RK_TW) * cdiv(a.CoutRows, 64) * a.nb >= ncu;
}

template <int CT, int RW>
static int launchRows(const ConvArgs& a, const _Float16* Wp, int ncu, hipStream_t stream, int yBase = 0, int tileRows = -1) {
    const int tilesX = cdiv(a.Wo, RK_TW), nchunk = cdiv(a.CoutRows, CT * 16);
    if (tileRows < 0) tileRows = cdiv(a.Ho - yBase, 8 * RW);                    // (default: every tile row from yBase to the image's last row)
    const int nitems = tileRows * tilesX * nchunk * a.nb;
    const bool spl = a.split_out != 0 || a.res_split != 0;
    const dim3 grid(nitems < ncu ? nitems : ncu), block(64 * RK_NW);
#define RK_LAUNCH(...) hipLaunchKernelGGL((conv_rows_kernel<CT, RW, __VA_ARGS__>), grid, block, 0, stream, a, Wp, tilesX, nitems, nchunk, yBase)
    if constexpr (kAblate) {
        static int dbg = -1; if (dbg < 0) dbg = ablateEnv("DSVT_CONV_DBG", 0);
        if (a.trace && spl) { RK_LAUNCH(true, true); return lastError(); }
#define RK_ABL(N_) if (spl && dbg == N_) { RK_LAUNCH(true, false, N_); return lastError(); }
        RK_ABL(1) RK_ABL(2) RK_ABL(3) RK_ABL(8)
#undef RK_ABL
    }
    if (spl && a.split_out && a.x8_out == 3 && !a.res) { RK_LAUNCH(true, false, 0, 1); return lastError(); }
    if (spl && a.split_out && a.x8_out == 3 && a.res && a.res_split && !a.res_x8) { RK_LAUNCH(true, false, 0, 2); return lastError(); }
    if (spl) RK_LAUNCH(true);
    else if constexpr (CT == 8) RK_LAUNCH(false);
    else return -3;
#undef RK_LAUNCH
    return lastError();
}
// The 128-channel items, with the image's partial last tile row as its own launch when that saves a round of the chip.  468 rows are 29 tile rows of 16 and FOUR more:
// at four images per launch 1800 items = 7 rounds of 256 workgroups + 8 items that run alone for an item time (75 us of ~600), and 60 of the 1800 compute twelve rows of
// padding each.  Without the partial tile row the launch is 1740 items = 7 rounds; the four rows go to conv_rows_kernel<4, 1> (8-row x 64-channel items: 120 items, one
// partial round of ~25 us) behind it.  Same (phase, tap) order in every item shape: the bits do not depend on the split.
static int launchRows128(const ConvArgs& a, const _Float16* Wp, int ncu, hipStream_t stream) {
    const int tilesX = cdiv(a.Wo, RK_TW), nchunk = cdiv(a.CoutRows, 128), per = tilesX * nchunk * a.nb;
    const int rest = a.Ho % 16, fullRows = a.Ho / 16;
    static int on = -1; if (on < 0) on = ablateEnv("DSVT_CONV_ROWS_LASTROW", 1);
    const bool spl = a.split_out != 0 || a.res_split != 0;
    if (on && spl && !a.trace && rest != 0 && rest <= 8 && fullRows > 0 && a.CoutRows % 64 == 0 && cdiv(fullRows * per, ncu) < cdiv((fullRows + 1) * per, ncu)) {
        const int rc = launchRows<8, 2>(a, Wp, ncu, stream, 0, fullRows);
        if (rc != 0) return rc;
        return launchRows<4, 1>(a, Wp, ncu, stream, fullRows * 16, 1);
    }
    return launchRows<8, 2>(a, Wp, ncu, stream);
}
int launchConvRows(const ConvArgs& a, const _Float16* Wp, int ncu, hipStream_t stream) { return launchRows128(a, Wp, ncu, stream); }
int launchConvRows64(const ConvArgs& a, const _Float16* Wp, int ncu, hipStream_t stream) { return launchRows<4, 3>(a, Wp, ncu, stream); }

// Launches that do not fill the chip with the items above (one frame per forward: 234 x 234 x 128 = 120 items of 16 rows x 128 channels, 117 x 117 x 256 = 64; the shared
// 384 -> 64 convolution = 300 items of 24 rows, a second round for 44 of them): 64-channel chunks on 16-row items (conv_rows_kernel<4, 2>: 240 / 450 items) or 8-row items
// (<4, 1>: 240 items at 117 x 117 x 256) -- the items round 5's launcher already chose for conv_wide_kernel<4, 8, 40, 4, 2, 2> / <4, 8, 36, 4, 2, 1>, on the ky-row slab loop.
// Same (phase, tap) order into the same accumulators: bit-identical to those and to every other tile shape (tests/test_conv_gpu.py::test_rows_kernel_equals_wide_kernel).
bool convRowsSmallEligible(const ConvArgs& a) {
    if (!rowsCommon(a) || a.CoutRows % 64 != 0) return false;
    return a.split_out != 0 || a.res_split != 0;                                       // (the fp16 frame keeps its own kernels)
}
int launchConvRowsSmall(const ConvArgs& a, const _Float16* Wp, int rowsPerWave, int ncu, hipStream_t stream) {
    return rowsPerWave == 1 ? launchRows<4, 1>(a, Wp, ncu, stream) : launchRows<4, 2>(a, Wp, ncu, stream);
}

}  // namespace dsvt
