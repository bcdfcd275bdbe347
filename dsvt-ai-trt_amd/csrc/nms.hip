// nms.hip -- RotatedNmsPlugin: the reference's host-side rotated BEV NMS on the device.
//
// SURVEY.md section 8(f)-2, second half.  The reference copies FilterBoxByScorePlugin's rows to the
// host and runs nms_cpu (include/helper.h:257-283) on them: sort by score, then greedily keep a box
// and drop every later box whose rotated-rectangle IoU with it is >= NMS_THRESH (0.01, class
// agnostic; params.h:334).  box_overlap (helper.h:166-255) clips the two rectangles by collecting
// edge intersections + contained corners, ordering them by atan2 around their centroid and summing
// the fan triangles.  The arithmetic below is that code line by line in fp32 (the file is built with
// -ffp-contract=off); the only shortcut is that a pair whose centres are farther apart than the two half
// diagonals (+ margin) skips it, which is exactly the case where the reference finds no intersection point
// and no contained corner and returns 0.
// Trigonometry: helper.h calls cos / sin / atan2 / fabs on floats under libstdc++, i.e. the FLOAT overloads
// (cosf, sinf, atan2f of the host's libm: helper.h:117-118, 194-195, 236-237).  glibc's float functions are
// not correctly rounded (and not one function: ifunc picks an FMA build where the CPU has it), and the device
// library's are a different algorithm again, so "the same bits as the host" does not exist for these values.
// The kernel uses the CORRECTLY ROUNDED float of each -- (float)cos((double)x) etc. --, which is what glibc's
// cosf / sinf return for ~98.7 % of inputs and what the device's own cosf / sinf return less often
// (tools/nms_trig_rates.py prints both rates on the box).  oracle/dsvt_oracle.c restates both arithmetics:
// orc_nms_cpu (the reference's overloads) and orc_nms_cpu_cr (this file's); the GPU tests pin the kernel to
// the second bit for bit and tests/test_host_post_cpu.py + the tool bound how often the two keep lists differ.
//
//   nms_sort   one wave per 64 rows: stable order of the n <= 512 rows by descending score (a row's place = the number of larger keys)
//   nms_mask   one wave per (row j, 64-row word): bit i = IoU(i, j) >= thresh for i < j (ballot) -- the transposed mask
//   nms_scan   mask in LDS; one wave sweeps 64 rows at a time (suppression by earlier blocks is a parallel test
//              against their kept sets, inside a block the greedy walk is the fixed point of a triangular system,
//              reached in a few whole-wave rounds), then the kept rows are written in score order
// Outputs: rows [1, max_boxes, 9] (the input row of every kept box, i.e. x, y, z, l, w, h, rt, id,
// score as helper.h:452-460 prints them), keep_idx [1, max_boxes] (input row numbers), count [1].
#include "plugin_base.h"
#include "device_utils.h"

namespace dsvt {

static bool f32Lin(const DsvtPluginTensorDesc& t) { return t.type == DSVT_FLOAT && t.format == DSVT_FORMAT_LINEAR; }
static bool i32Lin(const DsvtPluginTensorDesc& t) { return t.type == DSVT_INT32 && t.format == DSVT_FORMAT_LINEAR; }

constexpr int NMS_MAX = 512, NMS_WORDS = NMS_MAX / 64;
constexpr float kThresHold = 1e-8f;                                  // helper.h:26

struct Bnd { float x, y, z, w, l, h, rt; int id; float score; };     // helper.h:93-106
struct F2 { float x, y; };

__device__ __forceinline__ Bnd rowToBox(const float* o) {            // src/dsvt-ai-trt.cpp:1940-1950: dim0 -> l, dim1 -> w
    return Bnd{o[0], o[1], o[2], o[4], o[3], o[5], o[6], (int)o[7], o[8]};
}
__device__ __forceinline__ float crossf(F2 p1, F2 p2, F2 p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }   // :109-111

// The four trigonometric values box_overlap / check_box2d derive from a box's yaw (helper.h:115-116, 190-193: cos / sin of rt and of
// -rt, each correctly rounded to float -- see the header).  They depend on the box alone, so they are computed once per box (nms_sort) instead
// of once per PAIR (the host does the latter); the values, and with them every IoU, are the same bits.
struct Trig { float c, s, cn, sn; };                                  // cos(rt), sin(rt), cos(-rt), sin(-rt)
__device__ __forceinline__ Trig boxTrig(float rt) {
    return Trig{(float)cos((double)rt), (float)sin((double)rt), (float)cos((double)-rt), (float)sin((double)-rt)};
}

__device__ __forceinline__ bool checkBox2d(const Bnd& box, const Trig& tg, F2 p) {   // :113-123
    const float MARGIN = 1e-2f;
    const float angle_cos = tg.cn, angle_sin = tg.sn;
    const float rot_x = (p.x - box.x) * angle_cos + (p.y - box.y) * (-angle_sin);
    const float rot_y = (p.x - box.x) * angle_sin + (p.y - box.y) * angle_cos;
    return fabsf(rot_x) < box.w / 2 + MARGIN && fabsf(rot_y) < box.l / 2 + MARGIN;
}

__device__ __forceinline__ bool intersection(F2 p1, F2 p0, F2 q1, F2 q0, F2& ans) {   // :125-156
    if (!(fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
          fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y)))
        return false;
    const float s1 = crossf(q0, p1, p0), s2 = crossf(p1, q1, p0), s3 = crossf(p0, q1, q0), s4 = crossf(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = crossf(q1, p1, p0);
    if (fabsf(s5 - s1) > kThresHold) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

__device__ __forceinline__ void rotateAround(F2 c, float ac, float as, F2& p) {       // :158-163
    const float nx = (p.x - c.x) * ac + (p.y - c.y) * (-as) + c.x;
    const float ny = (p.x - c.x) * as + (p.y - c.y) * ac + c.y;
    p.x = nx; p.y = ny;
}

// cp / ang: the clip polygon's points and their angles, indexed by a run-time count.  They live in LDS, one column per lane (element k of
// lane l at [k * 64 + l]), not in a private array: a private array indexed at run time is scratch memory (400 bytes per lane here), and
// this was the only kernel of the frame with any (under two processes time-sharing the device its result was not
// reproducible run to run).
__device__ float boxOverlap(const Bnd& a, const Bnd& b, const Trig& ta, const Trig& tb, F2* __restrict__ cp, float* __restrict__ ang) {   // :166-255
    constexpr int L = 64;                                            // column stride
    const float a_dx = a.w / 2, b_dx = b.w / 2, a_dy = a.l / 2, b_dy = b.l / 2;
    F2 ac[5], bc[5], pc = {0.f, 0.f};                  // the host's cross_points[16] cannot hold the 16 + 8 worst case either: 24 here
    const F2 ca = {a.x, a.y}, cb = {b.x, b.y};
    int cnt = 0;
    ac[0] = F2{a.x - a_dx, a.y - a_dy}; ac[1] = F2{a.x + a_dx, a.y - a_dy}; ac[2] = F2{a.x + a_dx, a.y + a_dy}; ac[3] = F2{a.x - a_dx, a.y + a_dy};
    bc[0] = F2{b.x - b_dx, b.y - b_dy}; bc[1] = F2{b.x + b_dx, b.y - b_dy}; bc[2] = F2{b.x + b_dx, b.y + b_dy}; bc[3] = F2{b.x - b_dx, b.y + b_dy};
    const float a_cos = ta.c, a_sin = ta.s, b_cos = tb.c, b_sin = tb.s;
#pragma unroll
    for (int k = 0; k < 4; ++k) { rotateAround(ca, a_cos, a_sin, ac[k]); rotateAround(cb, b_cos, b_sin, bc[k]); }
    ac[4] = ac[0]; bc[4] = bc[0];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            F2 x;
            if (intersection(ac[i + 1], ac[i], bc[j + 1], bc[j], x)) { pc.x += x.x; pc.y += x.y; cp[cnt * L] = x; ++cnt; }
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (checkBox2d(a, ta, bc[k])) { pc.x += bc[k].x; pc.y += bc[k].y; cp[cnt * L] = bc[k]; ++cnt; }
        if (checkBox2d(b, tb, ac[k])) { pc.x += ac[k].x; pc.y += ac[k].y; cp[cnt * L] = ac[k]; ++cnt; }
    }
    if (cnt == 0) return 0.f;                                        // reference: 0/0 centroid, empty fan, area 0
    pc.x /= cnt; pc.y /= cnt;
    // the host recomputes atan2 inside every comparison (float overload: helper.h:236-237); same values, correctly rounded to float here
    for (int i = 0; i < cnt; ++i) { const F2 q = cp[i * L]; ang[i * L] = (float)atan2((double)(q.y - pc.y), (double)(q.x - pc.x)); }
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i) {
            const float a0 = ang[i * L], a1 = ang[(i + 1) * L];
            if (a0 > a1) {
                const F2 t = cp[i * L]; cp[i * L] = cp[(i + 1) * L]; cp[(i + 1) * L] = t;
                ang[i * L] = a1; ang[(i + 1) * L] = a0;
            }
        }
    float area = 0.f;
    const F2 c0 = cp[0];
    for (int k = 0; k < cnt - 1; ++k) {
        const F2 ck = cp[k * L], cn = cp[(k + 1) * L];
        const F2 u = {ck.x - c0.x, ck.y - c0.y}, v = {cn.x - c0.x, cn.y - c0.y};
        area += (u.x * v.y - u.y * v.x);
    }
    return (float)(fabsf(area) / 2.0);                               // :254: fabs(float), then the double division
}

// ---- kernels -------------------------------------------------------------------------------
// blockIdx.x = 64-row slice of the frame, blockIdx.y = frame of a stack; one wavefront
__global__ void __launch_bounds__(64)
nms_sort(const float* __restrict__ rows, const uint32_t* __restrict__ count, int max_boxes, uint32_t* __restrict__ order, float4* __restrict__ trig)
{
    __shared__ unsigned long long sk[NMS_MAX];       // (score key << 32) | ~row : descending = score desc, row asc (stable)
    const int lane = threadIdx.x, t = blockIdx.x * 64 + lane;
    rows += (size_t)blockIdx.y * max_boxes * 9; count += blockIdx.y; order += (size_t)blockIdx.y * NMS_MAX; trig += (size_t)blockIdx.y * NMS_MAX;
    int n = (int)*count; if (n > max_boxes) n = max_boxes;
    if (blockIdx.x * 64 >= n) return;
    auto keyOf = [&](int r) {
        const uint32_t u = __float_as_uint(rows[(size_t)r * 9 + 8]);
        const uint32_t key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        return ((unsigned long long)key << 32) | (uint32_t)~(uint32_t)r;
    };
    bool desc = true;                                // rows already in the output order?  (FilterBoxByScore keeps the decode's descending order)
    for (int j = lane; j < n; j += 64) {
        const unsigned long long kj = keyOf(j);
        sk[j] = kj;
        if (j + 1 < n) desc = desc && kj > keyOf(j + 1);
    }
    __syncthreads();
    if (__ballot(!desc) == 0ull) {                   // the frame pipeline's case: the order is the identity, no comparisons
        if (t < n) {
            order[t] = (uint32_t)t;
            const Trig tg = boxTrig(rows[(size_t)t * 9 + 6]);
            trig[t] = make_float4(tg.c, tg.s, tg.cn, tg.sn);
        }
        return;
    }
    // The keys are distinct (they carry the row number), so a row's place is the number of larger keys: n broadcast reads of LDS, no
    // exchange network (the 512-thread bitonic sort this replaces was 45 barriers on ONE CU, 9.7 us; a slice per wavefront spreads the
    // n^2 comparisons over n / 64 CUs).
    if (t < n) {
        const unsigned long long e = sk[t];
        int rank = 0;
        const int n8 = n & ~7;
        for (int j = 0; j < n8; j += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += sk[j + u] > e ? 1 : 0;
        }
        for (int j = n8; j < n; ++j) rank += sk[j] > e ? 1 : 0;
        order[rank] = (uint32_t)t;
        const Trig tg = boxTrig(rows[(size_t)t * 9 + 6]);             // trig[] is indexed by SORTED row
        trig[rank] = make_float4(tg.c, tg.s, tg.cn, tg.sn);
    }
}

// word (j, wi) of the TRANSPOSED suppression matrix: bit b = "sorted row i = 64 wi + b (i < j) suppresses sorted row j"
__global__ void __launch_bounds__(64)
nms_mask(const float* __restrict__ rows, const uint32_t* __restrict__ count, const uint32_t* __restrict__ order,
         const float4* __restrict__ trig, int max_boxes, float thresh, unsigned long long* __restrict__ maskT)
{
    rows += (size_t)blockIdx.z * max_boxes * 9; count += blockIdx.z; order += (size_t)blockIdx.z * NMS_MAX; trig += (size_t)blockIdx.z * NMS_MAX;
    maskT += (size_t)blockIdx.z * NMS_MAX * NMS_WORDS;                                                  // blockIdx.z = frame of a stack
    int n = (int)*count; if (n > max_boxes) n = max_boxes;
    __shared__ F2 s_cp[24 * 64];
    __shared__ float s_ang[24 * 64];
    const int j = blockIdx.x, wi = blockIdx.y, lane = threadIdx.x, i = wi * 64 + lane;
    if (j >= n) return;
    bool sup = false;
    if (i < j) {
        const Bnd bi = rowToBox(rows + (size_t)order[i] * 9), bj = rowToBox(rows + (size_t)order[j] * 9);
        // centres farther apart than both half diagonals + the containment margin: no intersection point, no contained corner
        const float dx = bi.x - bj.x, dy = bi.y - bj.y;
        const float ri = 0.5f * sqrtf(bi.w * bi.w + bi.l * bi.l), rj = 0.5f * sqrtf(bj.w * bj.w + bj.l * bj.l);
        const float reach = ri + rj + 0.1f;
        if (dx * dx + dy * dy <= reach * reach) {
            const float sa = bi.w * bi.l, sb = bj.w * bj.l;                        // helper.h:272-275 (i is the kept box, j the later one)
            const float4 ti = trig[i], tj = trig[j];
            const float so = boxOverlap(bi, bj, Trig{ti.x, ti.y, ti.z, ti.w}, Trig{tj.x, tj.y, tj.z, tj.w}, s_cp + lane, s_ang + lane);
            const float iou = so / fmaxf(sa + sb - so, kThresHold);
            sup = iou >= thresh;
        }
    }
    const unsigned long long word = __ballot(sup);
    if (lane == 0) maskT[(size_t)j * NMS_WORDS + wi] = word;
}

__global__ void __launch_bounds__(512)
nms_scan(const float* __restrict__ rows, const uint32_t* __restrict__ count, const uint32_t* __restrict__ order,
         const unsigned long long* __restrict__ maskT, int max_boxes, float* __restrict__ out_rows, int32_t* __restrict__ keep_idx,
         uint32_t* __restrict__ out_count, bool zero_fill)
{
    __shared__ unsigned long long sm[NMS_MAX * NMS_WORDS];
    __shared__ unsigned long long keptw[NMS_WORDS];
    __shared__ uint32_t kept[NMS_MAX];
    const int t = threadIdx.x, lane = t & 63;
    {   // blockIdx.x = frame of a stack
        const size_t b = blockIdx.x;
        rows += b * max_boxes * 9; count += b; order += b * NMS_MAX; maskT += b * NMS_MAX * NMS_WORDS;
        out_rows += b * max_boxes * 9; keep_idx += b * max_boxes; out_count += b;
    }
    int n = (int)*count; if (n > max_boxes) n = max_boxes;
    {                                                // all loads of the mask in flight at once (8 per thread)
        unsigned long long v[NMS_WORDS];
#pragma unroll
        for (int u = 0; u < NMS_WORDS; ++u) { const int e = t + u * 512; v[u] = e < n * NMS_WORDS ? maskT[e] : 0ull; }
#pragma unroll
        for (int u = 0; u < NMS_WORDS; ++u) sm[t + u * 512] = v[u];
    }
    __syncthreads();
    if (t < 64) {
        // Greedy sweep (helper.h:260-281), 64 sorted rows at a time, one wave.  Lane = row j of the block.  Whether an
        // EARLIER block suppresses j is a parallel test of j's transposed mask words against the kept sets of those blocks;
        // inside the block row j is kept unless suppressed from outside or by a row kept earlier in this block (its own word of the block).
        unsigned long long ks[NMS_WORDS];            // kept set, wave-uniform
#pragma unroll
        for (int w = 0; w < NMS_WORDS; ++w) ks[w] = 0ull;
#pragma unroll
        for (int w = 0; w < NMS_WORDS; ++w) {
            const int base = w * 64;
            if (base < n) {
                const int nb = n - base < 64 ? n - base : 64;
                const int j = base + lane;
                unsigned long long pre = 0ull;
#pragma unroll
                for (int wi = 0; wi < NMS_WORDS; ++wi)
                    if (wi < w) pre |= sm[j * NMS_WORDS + wi] & ks[wi];
                const unsigned long long own = lane < nb ? sm[j * NMS_WORDS + w] : 0ull;
                const unsigned long long outside = __ballot(pre != 0ull || lane >= nb);
                // Inside the block row b is kept iff nothing outside suppresses it and no KEPT earlier row of the block does: a triangular system
                // (own has bits below the lane only), so its fixed point is unique and is the serial walk's answer.  Iterate all 64 rows at once
                // from "everything not suppressed from outside is kept": after t rounds rows 0 .. t - 1 are final, and real frames settle in two
                // or three rounds (the 64-step readlane walk this replaces was 12 of the kernel's 18 us).
                unsigned long long kw = ~outside;
                for (int it = 0; it < 64; ++it) {
                    const unsigned long long nk2 = __ballot((own & kw) == 0ull) & ~outside;
                    if (nk2 == kw) break;
                    kw = nk2;
                }
                ks[w] = kw;
            }
        }
#pragma unroll
        for (int w = 0; w < NMS_WORDS; ++w) if (lane == 0) keptw[w] = ks[w];
    }
    __syncthreads();
    // kept sorted rows -> compact list, in order (rank = number of kept rows before it)
    int nk = 0;
#pragma unroll
    for (int w = 0; w < NMS_WORDS; ++w) nk += __popcll(keptw[w]);
    if (t < n && ((keptw[t >> 6] >> (t & 63)) & 1ull)) {
        int rank = __popcll(keptw[t >> 6] & ((1ull << (t & 63)) - 1ull));
        for (int w = 0; w < (t >> 6); ++w) rank += __popcll(keptw[w]);
        kept[rank] = order[t];                       // input row number
    }
    __syncthreads();
    constexpr int PER = (NMS_MAX * 9 + 511) / 512;
    float v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {                  // independent loads first, stores after
        const int e = t + u * 512, k = e / 9, c = e - k * 9;
        v[u] = (e < max_boxes * 9 && k < nk) ? rows[(size_t)kept[k] * 9 + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = t + u * 512;
        if (e < max_boxes * 9 && (e / 9 < nk || zero_fill)) out_rows[e] = v[u];
    }
    if (t < max_boxes && (t < nk || zero_fill)) keep_idx[t] = t < nk ? (int32_t)kept[t] : 0;
    if (t == 0) *out_count = (uint32_t)nk;
}

// ---- plugin ----------------------------------------------------------------------------------
class RotatedNmsPlugin : public Plugin {
public:
    int max_boxes_; float thresh_;
    RotatedNmsPlugin(int m, float t) : max_boxes_(m), thresh_(t) {}
    const char* type() const override { return "RotatedNmsPlugin"; }
    bool handlesBatch() const override { return true; }              // blockIdx = frame of a stack of row sets
    int nbOutputs() const override { return 3; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i == 0) { *out = dims3(in[0].d[0], max_boxes_, 9); return 0; }
        if (i == 1) { *out = dims2(in[0].d[0], max_boxes_); return 0; }
        if (i == 2) { *out = dims1(in[0].d[0]); return 0; }
        return -1;
    }
    int outputType(int i, const int32_t*, int) const override { return i == 0 ? DSVT_FLOAT : DSVT_INT32; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {
        if (pos == 0 || pos == 2) return f32Lin(io[pos]);
        return pos >= 0 && pos <= 4 && i32Lin(io[pos]);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc* in, int nbIn, const DsvtPluginTensorDesc*, int) const override {
        const size_t nb = (in && nbIn > 0 && in[0].dims.nbDims >= 1 && in[0].dims.d[0] > 1) ? (size_t)in[0].dims.d[0] : 1;
        return alignUp(sizeof(uint32_t) * NMS_MAX * nb) + alignUp(sizeof(unsigned long long) * NMS_MAX * NMS_WORDS * nb) + alignUp(sizeof(float4) * NMS_MAX * nb);
    }
    int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void* ws,
                hipStream_t stream) override {
        const int nb = (inDesc && inDesc[0].dims.nbDims >= 1 && inDesc[0].dims.d[0] > 1) ? inDesc[0].dims.d[0] : 1;
        WsCarver c(ws);
        uint32_t* order = c.take<uint32_t>((size_t)NMS_MAX * nb);
        unsigned long long* mask = c.take<unsigned long long>((size_t)NMS_MAX * NMS_WORDS * nb);
        float4* trig = c.take<float4>((size_t)NMS_MAX * nb);
        const float* rows = static_cast<const float*>(in[0]);
        const uint32_t* count = static_cast<const uint32_t*>(in[1]);
        hipLaunchKernelGGL(nms_sort, dim3(NMS_MAX / 64, nb), dim3(64), 0, stream, rows, count, max_boxes_, order, trig);
        hipLaunchKernelGGL(nms_mask, dim3(max_boxes_, cdiv(max_boxes_, 64), nb), dim3(64), 0, stream, rows, count, order, trig, max_boxes_, thresh_, mask);
        hipLaunchKernelGGL(nms_scan, dim3(nb), dim3(512), 0, stream, rows, count, order, mask, max_boxes_, static_cast<float*>(out[0]),
                           static_cast<int32_t*>(out[1]), static_cast<uint32_t*>(out[2]), zeroFill);
        return lastError();
    }
    size_t serializationSize() const override { return sizeof(int) + sizeof(float); }
    void serialize(void* b) const override { char* d = static_cast<char*>(b); wr<int>(d, max_boxes_); wr<float>(d, thresh_); }
    Plugin* clone() const override { return new RotatedNmsPlugin(max_boxes_, thresh_); }
};
static Plugin* nmsCreate(const DsvtPluginFieldCollection* fc) {
    const int m = fieldInt(fc, "max_boxes"); const float t = fieldFloat(fc, "nms_thresh", 0.01f);
    return (m > 0 && m <= NMS_MAX) ? new RotatedNmsPlugin(m, t) : nullptr;
}
static Plugin* nmsDeser(const void* data, size_t len) {
    if (len < sizeof(int) + sizeof(float)) return nullptr;
    const char* d = static_cast<const char*>(data);
    const int m = rd<int>(d); const float t = rd<float>(d);
    return (m > 0 && m <= NMS_MAX) ? new RotatedNmsPlugin(m, t) : nullptr;
}
static Creator g_nmsCreator{"RotatedNmsPlugin", {{"max_boxes", DSVT_FIELD_INT32}, {"nms_thresh", DSVT_FIELD_FLOAT32}}, nmsCreate, nmsDeser, {}, {}};
static Registrar g_nmsReg(&g_nmsCreator);

}  // namespace dsvt
