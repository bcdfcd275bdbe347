// partition_ops.hip -- rotated-set partitioning for gfx950:
//   WindowPartitionPlugin  plugins/src/windowPartition.cu:278-470
//   GetSetPlugin           plugins/src/getSet.cu:267-704
//
// The reference assigns window ids with an atomic counter plus a spin-wait hack
// (windowPartition.cu:308-331, a real data race) and sorts each window with a per-THREAD
// iterative quicksort on 4.6 kB of local memory (getSet.cu:293-324, <=406 threads active on
// the whole GPU).  Here nothing depends on arrival order:
//   * windows are numbered by an exclusive scan over the dense window grid (ascending
//     window linear id); a window's voxels are ordered by voxel id (= the serial arrival order
//     of the reference): for pillars that arrive in cell order that is the row-major order of
//     the in-window cells, i.e. rank = popcount of an occupancy bitmap; otherwise a workgroup
//     bitonic sort in LDS;
//   * the two per-window sorts use the fact that the in-window keys are unique and smaller
//     than the window volume: each voxel is dropped into an LDS table at its key and the
//     table is compacted with a workgroup scan -- no comparison sort at all;
//   * set bases come from a scan over windows (ascending (window, j)).
// All integer work; results are bit-exact against the oracle.
#include "plugin_base.h"
#include "device_utils.h"
#include <algorithm>

namespace dsvt {

constexpr uint32_t kNoneU = 0xffffffffu;
static bool f32L(const DsvtPluginTensorDesc& t) { return t.type == DSVT_FLOAT && t.format == DSVT_FORMAT_LINEAR; }
static bool i32L(const DsvtPluginTensorDesc& t) { return t.type == DSVT_INT32 && t.format == DSVT_FORMAT_LINEAR; }

struct WPParams {
    int max_win_num, max_voxel_num_per_win;
    int sx, sy, sz;       // sparse_shape
    int wx, wy, wz;       // win_shape
    int hx, hy, hz;       // shift_list
    int nwx, nwy, nwz;    // dense window grid (windowPartition.cu:425-427)
};

__device__ __forceinline__ void winOf(const uint4 co, const WPParams& p, uint32_t& win, uint32_t& ix, uint32_t& iy, uint32_t& iz) {
    uint32_t x = co.w + (uint32_t)p.hx, y = co.z + (uint32_t)p.hy, z = co.y + (uint32_t)p.hz;       // :292-294
    uint32_t cx = x / (uint32_t)p.wx, cy = y / (uint32_t)p.wy, cz = z / (uint32_t)p.wz;            // :296-298
    win = cz * (uint32_t)(p.nwy * p.nwx) + cy * (uint32_t)p.nwx + cx;                               // :301
    if (cx >= (uint32_t)p.nwx || cy >= (uint32_t)p.nwy || cz >= (uint32_t)p.nwz) win = kNoneU;      // outside the dense grid
    ix = x % (uint32_t)p.wx; iy = y % (uint32_t)p.wy; iz = z % (uint32_t)p.wz;                      // :352-354
}

__global__ void __launch_bounds__(256)
wp_count(const uint4* __restrict__ coords, const uint32_t* __restrict__ voxel_num, int max_pillars, WPParams p,
         uint32_t* __restrict__ win_cnt, uint32_t* __restrict__ vox_win, uint32_t* __restrict__ vox_slot)
{
    uint32_t n = *voxel_num; if (n > (uint32_t)max_pillars) n = max_pillars;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t win = kNoneU, ix, iy, iz;
    if (v < n) { winOf(coords[v], p, win, ix, iy, iz); vox_win[v] = win; }
    // Pillars are sorted by y * GX + x, so consecutive lanes mostly fall into the same window (12 / 24 cells wide): one atomic
    // per RUN of equal windows instead of one per voxel (same-address atomics serialise in L2).  Slots only have to be unique
    // inside a window; their order is fixed in wp_fill.
    const uint32_t prev = __shfl_up(win, 1, kWave);
    const bool head = lane == 0 || prev != win;
    const unsigned long long heads = __ballot(head);
    const unsigned long long below = heads & ((2ull << lane) - 1ull);            // run heads at or before this lane
    const int leader = 63 - __builtin_clzll(below);
    uint32_t base = 0;
    if (head && win != kNoneU) {
        // run length: up to the next head
        const unsigned long long nxt = heads & ~((2ull << lane) - 1ull);
        const int len = (nxt ? __builtin_ctzll(nxt) : 64) - lane;
        base = atomicAdd(&win_cnt[win], (uint32_t)len);
    }
    base = __shfl(base, leader, kWave);
    if (v < n) vox_slot[v] = win == kNoneU ? 0u : base + (uint32_t)(lane - leader);
}

// single workgroup: scan the dense window grid
__global__ void __launch_bounds__(1024)
wp_scan(const uint32_t* __restrict__ win_cnt, int dense, WPParams p, uint32_t* __restrict__ win_seg,
        uint32_t* __restrict__ rank2win, uint32_t* __restrict__ vcnt, uint32_t* __restrict__ win_num, bool zero_fill)
{
    __shared__ uint32_t smem[1024 / kWave + 1];
    uint32_t carry_o = 0, carry_f = 0;
    for (int b = 0; b < dense; b += 1024) {
        int w = b + threadIdx.x;
        uint32_t c = w < dense ? win_cnt[w] : 0, tot;
        uint32_t eo = blockExclusiveScan<1024>(c > 0 ? 1u : 0u, smem, &tot) + carry_o; carry_o += tot;
        uint32_t ef = blockExclusiveScan<1024>(c, smem, &tot) + carry_f; carry_f += tot;
        if (w < dense) {
            win_seg[w] = ef;
            if (c > 0 && eo < (uint32_t)p.max_win_num) {          // capacity guard the reference lacks (:311)
                rank2win[eo] = (uint32_t)w;
                vcnt[eo] = c > (uint32_t)p.max_voxel_num_per_win ? (uint32_t)p.max_voxel_num_per_win : c;    // :336-340
            }
        }
    }
    uint32_t W = carry_o < (uint32_t)p.max_win_num ? carry_o : (uint32_t)p.max_win_num;
    if (zero_fill) for (uint32_t r = W + threadIdx.x; r < (uint32_t)p.max_win_num; r += 1024) vcnt[r] = 0;
    if (threadIdx.x == 0) *win_num = W;
}

__global__ void __launch_bounds__(256)
wp_scatter(const uint32_t* __restrict__ voxel_num, int max_pillars, const uint32_t* __restrict__ vox_win,
           const uint32_t* __restrict__ vox_slot, const uint32_t* __restrict__ win_seg, uint32_t* __restrict__ sorted_vox)
{
    uint32_t n = *voxel_num; if (n > (uint32_t)max_pillars) n = max_pillars;
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    uint32_t w = vox_win[v];
    if (w != kNoneU) sorted_vox[win_seg[w] + vox_slot[v]] = v;
}

// workgroup bitonic sort of m (power of two) uint32 in LDS, ascending
__device__ __forceinline__ void bitonicSortLds(uint32_t* a, int m) {
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < m; i += blockDim.x) {
                int l = i ^ j;
                if (l > i) {
                    uint32_t x = a[i], y = a[l];
                    bool up = (i & k) == 0;
                    if ((x > y) == up) { a[i] = y; a[l] = x; }
                }
            }
            __syncthreads();
        }
}

// one workgroup per non-empty window (rank order)
__global__ void __launch_bounds__(256)
wp_fill(const uint4* __restrict__ coords, WPParams p, const uint32_t* __restrict__ win_num, const uint32_t* __restrict__ rank2win,
        const uint32_t* __restrict__ win_cnt, const uint32_t* __restrict__ win_seg, const uint32_t* __restrict__ sorted_vox,
        uint32_t* __restrict__ gidx, uint32_t* __restrict__ cinw, uint32_t* __restrict__ c2d, float* __restrict__ xy)
{
    extern __shared__ uint32_t lds[];
    const uint32_t r = blockIdx.x;
    if (r >= *win_num) return;
    // A window can only hold more voxels than its dynamic LDS (2 x pow2(volume) words) when the caller hands in DUPLICATE pillar
    // coordinates, which Points2Features cannot produce; the surplus segment entries are then dropped (which ones is as unspecified as
    // the reference's racy order beyond its cap, windowPartition.cu:305) instead of being sorted out of bounds.
    uint32_t cap2 = 2; { const uint32_t v3 = (uint32_t)(p.wx * p.wy * p.wz); uint32_t c1 = 1; while (c1 < v3) c1 <<= 1; cap2 = 2 * c1; }
    const uint32_t w = rank2win[r], n = win_cnt[w] < cap2 ? win_cnt[w] : cap2, seg = win_seg[w];
    const uint32_t Vw = p.max_voxel_num_per_win;
    // Voxel ids ascend with the cell key y * GX + x (Points2Features' canonical order), so inside a window of a one-level grid
    // "ascending voxel id" IS the row-major order of the in-window cells: the rank of a voxel is the number of occupied cells
    // before its own -- an occupancy bitmap + popcount instead of a bitonic sort (55 barrier passes for a 24 x 24 window)
    const uint32_t vol = (uint32_t)(p.wx * p.wy);
    int m = 1; while ((uint32_t)m < n) m <<= 1;
    bool sorted = false;
    if (p.wz == 1 && p.sz == 1 && vol <= 1024u && n <= vol) {
        // (that holds for Points2Features' output; a caller may hand over pillars in any order, so the result is checked and the
        // sort below remains the fallback)
        __shared__ unsigned long long bits[17];
        uint32_t* ord = lds + m;                                  // second half of the dynamic LDS (2 x capacity)
        if (threadIdx.x < 17) bits[threadIdx.x] = 0ull;
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < n; s += blockDim.x) {
            const uint32_t v = sorted_vox[seg + s];
            uint32_t win, ix, iy, iz;
            winOf(coords[v], p, win, ix, iy, iz);
            const uint32_t cell = iy * (uint32_t)p.wx + ix;
            atomicOr(&bits[cell >> 6], 1ull << (cell & 63));
            lds[s] = v;
        }
        __syncthreads();
        for (uint32_t s0 = threadIdx.x; s0 < n; s0 += blockDim.x) {
            const uint32_t v = lds[s0];
            uint32_t win, ix, iy, iz;
            winOf(coords[v], p, win, ix, iy, iz);
            const uint32_t cell = iy * (uint32_t)p.wx + ix;
            uint32_t s = (uint32_t)__popcll(bits[cell >> 6] & ((1ull << (cell & 63)) - 1ull));
            for (uint32_t q = 0; q < (cell >> 6); ++q) s += (uint32_t)__popcll(bits[q]);
            ord[s] = v;
        }
        __syncthreads();
        int bad = 0;
        for (uint32_t i = threadIdx.x; i + 1 < n; i += blockDim.x) bad |= ord[i] >= ord[i + 1];
        sorted = !__syncthreads_or(bad);
        if (sorted)
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) lds[i] = ord[i];
        __syncthreads();
    }
    if (!sorted) {
        for (int i = threadIdx.x; i < m; i += blockDim.x) lds[i] = (uint32_t)i < n ? sorted_vox[seg + i] : kNoneU;
        __syncthreads();
        bitonicSortLds(lds, m);
    }
    for (uint32_t s = threadIdx.x; s < n; s += blockDim.x) {
        uint32_t v = lds[s], win, ix, iy, iz;
        winOf(coords[v], p, win, ix, iy, iz);
        if (s < Vw) {                                                                  // :305
            gidx[(size_t)r * Vw + s] = v;                                              // :343-344
            uint32_t* c = cinw + ((size_t)r * Vw + s) * 3;
            c[0] = iz; c[1] = iy; c[2] = ix;                                           // :357-359
            c2d[(size_t)v * 3 + 0] = iz; c2d[(size_t)v * 3 + 1] = iy; c2d[(size_t)v * 3 + 2] = ix;   // :362-364
            xy[(size_t)v * 2 + 0] = (float)ix - (float)p.wx / 2;                       // :367-368
            xy[(size_t)v * 2 + 1] = (float)iy - (float)p.wy / 2;
        } else {                                                                       // dropped voxel: the reference returns early
            c2d[(size_t)v * 3 + 0] = 0; c2d[(size_t)v * 3 + 1] = 0; c2d[(size_t)v * 3 + 2] = 0;
            xy[(size_t)v * 2 + 0] = 0.f; xy[(size_t)v * 2 + 1] = 0.f;
        }
    }
}

class WindowPartitionPlugin : public Plugin {
public:
    WPParams p_;
    explicit WindowPartitionPlugin(const WPParams& p) : p_(p) {}
    const char* type() const override { return "WindowPartitionPlugin"; }
    int nbOutputs() const override { return 6; }
    int dense() const { return p_.nwx * p_.nwy * p_.nwz; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {    // windowPartition.cu: getOutputDimensions
        int b = in[0].d[0], mp = in[0].d[1];          // reference: MAX_PILLARS_NUM macro; here the coords tensor's row count
        switch (i) {
            case 0: *out = dims3(b, p_.max_win_num, p_.max_voxel_num_per_win); return 0;
            case 1: *out = dims4(b, p_.max_win_num, p_.max_voxel_num_per_win, 3); return 0;
            case 2: *out = dims2(b, p_.max_win_num); return 0;
            case 3: *out = dims1(b); return 0;
            case 4: *out = dims3(b, mp, 3); return 0;
            case 5: *out = dims3(b, mp, 2); return 0;
        }
        return -1;
    }
    int outputType(int i, const int32_t*, int) const override { return i == 5 ? DSVT_FLOAT : DSVT_INT32; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {
        return pos == 7 ? f32L(io[pos]) : pos >= 0 && pos <= 6 && i32L(io[pos]);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc* in, int, const DsvtPluginTensorDesc*, int) const override {
        int mp = in[0].dims.d[1];
        return 2 * alignUp(sizeof(uint32_t) * dense()) + alignUp(sizeof(uint32_t) * p_.max_win_num) + 3 * alignUp(sizeof(uint32_t) * mp);
    }
    int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc*, const void* const* in, void* const* out,
                void* workspace, hipStream_t stream) override {
        const int mp = inDesc[0].dims.d[1];
        const uint4* coords = static_cast<const uint4*>(in[0]);
        const uint32_t* voxel_num = static_cast<const uint32_t*>(in[1]);
        uint32_t* gidx = static_cast<uint32_t*>(out[0]);
        uint32_t* cinw = static_cast<uint32_t*>(out[1]);
        uint32_t* vcnt = static_cast<uint32_t*>(out[2]);
        uint32_t* win_num = static_cast<uint32_t*>(out[3]);
        uint32_t* c2d = static_cast<uint32_t*>(out[4]);
        float* xy = static_cast<float*>(out[5]);
        WsCarver ws(workspace);
        uint32_t* win_cnt = ws.take<uint32_t>(dense());
        uint32_t* win_seg = ws.take<uint32_t>(dense());
        uint32_t* rank2win = ws.take<uint32_t>(p_.max_win_num);
        uint32_t* vox_win = ws.take<uint32_t>(mp);
        uint32_t* vox_slot = ws.take<uint32_t>(mp);
        uint32_t* sorted_vox = ws.take<uint32_t>(mp);
        DSVT_CHECK(hipMemsetAsync(win_cnt, 0, sizeof(uint32_t) * dense(), stream));
        if (zeroFill) {                                                                // :445-450
            size_t wv = (size_t)p_.max_win_num * p_.max_voxel_num_per_win;
            DSVT_CHECK(hipMemsetAsync(gidx, 0, sizeof(uint32_t) * wv, stream));
            DSVT_CHECK(hipMemsetAsync(cinw, 0, sizeof(uint32_t) * wv * 3, stream));
            DSVT_CHECK(hipMemsetAsync(c2d, 0, sizeof(uint32_t) * (size_t)mp * 3, stream));
            DSVT_CHECK(hipMemsetAsync(xy, 0, sizeof(float) * (size_t)mp * 2, stream));
        }
        hipLaunchKernelGGL(wp_count, dim3(cdiv(mp, 256)), dim3(256), 0, stream, coords, voxel_num, mp, p_, win_cnt, vox_win, vox_slot);
        hipLaunchKernelGGL(wp_scan, dim3(1), dim3(1024), 0, stream, win_cnt, dense(), p_, win_seg, rank2win, vcnt, win_num, zeroFill);
        hipLaunchKernelGGL(wp_scatter, dim3(cdiv(mp, 256)), dim3(256), 0, stream, voxel_num, mp, vox_win, vox_slot, win_seg, sorted_vox);
        int vol = p_.wx * p_.wy * p_.wz, cap = 1; while (cap < vol) cap <<= 1;
        hipLaunchKernelGGL(wp_fill, dim3(p_.max_win_num), dim3(256), sizeof(uint32_t) * cap * 2, stream, coords, p_, win_num, rank2win,
                           win_cnt, win_seg, sorted_vox, gidx, cinw, c2d, xy);
        return lastError();
    }
    size_t serializationSize() const override { return 11 * sizeof(int); }
    void serialize(void* b) const override {                                           // windowPartition.cu:511-525
        char* d = static_cast<char*>(b);
        wr<int>(d, p_.sx); wr<int>(d, p_.sy); wr<int>(d, p_.sz); wr<int>(d, p_.wx); wr<int>(d, p_.wy); wr<int>(d, p_.wz);
        wr<int>(d, p_.hx); wr<int>(d, p_.hy); wr<int>(d, p_.hz); wr<int>(d, p_.max_win_num); wr<int>(d, p_.max_voxel_num_per_win);
    }
    Plugin* clone() const override { return new WindowPartitionPlugin(p_); }
};
static Plugin* wpNew(WPParams p) {
    if (p.max_win_num <= 0 || p.max_voxel_num_per_win <= 0 || p.wx <= 0 || p.wy <= 0 || p.wz <= 0 ||
        p.sx <= 0 || p.sy <= 0 || p.sz <= 0 || p.hx < 0 || p.hy < 0 || p.hz < 0) return nullptr;
    if ((long)p.wx * p.wy * p.wz > 8192) return nullptr;           // LDS sort capacity
    // windowPartition.cu:425-427: int(ceilf(shape / win) + 1) with INTEGER division inside
    p.nwx = (int)(ceilf((float)(p.sx / p.wx)) + 1);
    p.nwy = (int)(ceilf((float)(p.sy / p.wy)) + 1);
    p.nwz = (int)(ceilf((float)(p.sz / p.wz)) + 1);
    return new WindowPartitionPlugin(p);
}
static Plugin* wpCreate(const DsvtPluginFieldCollection* fc) {
    WPParams p{}; int s[3], w[3], h[3];
    p.max_win_num = fieldInt(fc, "max_win_num"); p.max_voxel_num_per_win = fieldInt(fc, "max_voxel_num_per_win");
    fieldInts(fc, "sparse_shape", s, 3); fieldInts(fc, "win_shape", w, 3); fieldInts(fc, "shift_list", h, 3);
    p.sx = s[0]; p.sy = s[1]; p.sz = s[2]; p.wx = w[0]; p.wy = w[1]; p.wz = w[2]; p.hx = h[0]; p.hy = h[1]; p.hz = h[2];
    return wpNew(p);
}
static Plugin* wpDeser(const void* data, size_t len) {
    if (len < 11 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data); WPParams p{};
    p.sx = rd<int>(d); p.sy = rd<int>(d); p.sz = rd<int>(d); p.wx = rd<int>(d); p.wy = rd<int>(d); p.wz = rd<int>(d);
    p.hx = rd<int>(d); p.hy = rd<int>(d); p.hz = rd<int>(d); p.max_win_num = rd<int>(d); p.max_voxel_num_per_win = rd<int>(d);
    return wpNew(p);
}
static Creator g_wpCreator{"WindowPartitionPlugin",
    {{"max_win_num", DSVT_FIELD_INT32}, {"max_voxel_num_per_win", DSVT_FIELD_INT32}, {"sparse_shape", DSVT_FIELD_INT32},
     {"win_shape", DSVT_FIELD_INT32}, {"shift_list", DSVT_FIELD_INT32}},               // :549-553
    wpCreate, wpDeser, {}, {}};
static Registrar g_wpReg(&g_wpCreator);

// =====================================================================================
// GetSet
// =====================================================================================
// max_set_num: capacity of the SET dimension of the outputs.  The reference sizes it with MAX_WIN_NUM too (getSet.cu:147,242); a frame has up
// to ceil(P / 36) + W sets, so the pipeline sizes it separately (non-reference field "max_set_num", default = max_win_num).
struct GSParams { int max_win_num, max_voxel_num_per_win, voxel_num_set, wx, wy, wz, num_heads, max_set_num; };

__device__ __forceinline__ uint32_t setsOf(uint32_t n, uint32_t L) { return (uint32_t)(int)ceilf((float)n / (float)(int)L); }   // getSet.cu:335

// single workgroup: set base of every window
__global__ void __launch_bounds__(1024)
gs_scan(const uint32_t* __restrict__ vcnt, const uint32_t* __restrict__ win_num, GSParams p, uint32_t* __restrict__ set_base,
        uint32_t* __restrict__ set_num)
{
    __shared__ uint32_t smem[1024 / kWave + 1];
    __shared__ uint32_t smax;
    if (threadIdx.x == 0) smax = 0;
    __syncthreads();
    uint32_t W = *win_num; if (W > (uint32_t)p.max_win_num) W = p.max_win_num;
    uint32_t carry = 0;
    for (uint32_t b = 0; b < W; b += 1024) {
        uint32_t w = b + threadIdx.x, tot;
        uint32_t ns = w < W ? setsOf(vcnt[w], p.voxel_num_set) : 0;
        uint32_t base = blockExclusiveScan<1024>(ns, smem, &tot) + carry; carry += tot;
        if (w < W) {
            // capacity guard the reference lacks (:337): the set list stops at the first window that does not fit
            bool ok = base + ns <= (uint32_t)p.max_set_num;
            set_base[w] = ok ? base : kNoneU;
            if (ok && ns) atomicMax(&smax, base + ns);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *set_num = smax;
}

// one workgroup per window
__global__ void __launch_bounds__(256)
gs_sets(const uint32_t* __restrict__ gidx, const uint32_t* __restrict__ cinw, const uint32_t* __restrict__ vcnt,
        const uint32_t* __restrict__ win_num, const uint32_t* __restrict__ set_base, GSParams p,
        uint32_t* __restrict__ inds, float* __restrict__ mask, float* __restrict__ mask0_h, float* __restrict__ mask1_h)
{
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t smem[256 / kWave + 1];
    const uint32_t w = blockIdx.x;
    uint32_t W = *win_num; if (W > (uint32_t)p.max_win_num) W = p.max_win_num;
    if (w >= W) return;
    const uint32_t base = set_base[w];
    if (base == kNoneU) return;
    const uint32_t Vw = p.max_voxel_num_per_win, L = p.voxel_num_set, MW = p.max_set_num;
    const uint32_t n = vcnt[w] < Vw ? vcnt[w] : Vw;
    const uint32_t vol = (uint32_t)(p.wx * p.wy * p.wz);
    uint32_t* ty = lds;             // [vol]  key_y -> voxel id
    uint32_t* tx = lds + vol;       // [vol]  key_x -> voxel id
    uint32_t* sy = lds + 2 * vol;   // [Vw]   voxel ids ascending by key_y
    uint32_t* sx = sy + Vw;         // [Vw]   voxel ids ascending by key_x
    for (uint32_t i = threadIdx.x; i < vol; i += blockDim.x) { ty[i] = kNoneU; tx[i] = kNoneU; }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { sy[i] = 0; sx[i] = 0; }
    __syncthreads();
    for (uint32_t m = threadIdx.x; m < n; m += blockDim.x) {
        uint32_t v = gidx[(size_t)w * Vw + m];
        const uint32_t* c = cinw + ((size_t)w * Vw + m) * 3;
        uint32_t z = c[0], y = c[1], x = c[2];
        uint32_t ky = y * (uint32_t)(p.wx * p.wz) + x * (uint32_t)p.wz + z;          // getSet.cu:386-387
        uint32_t kx = x * (uint32_t)(p.wy * p.wz) + y * (uint32_t)p.wz + z;          // :461-462
        if (ky < vol) ty[ky] = v;
        if (kx < vol) tx[kx] = v;
    }
    __syncthreads();
    // compact both tables in key order (ascending sort of unique keys)
    uint32_t cy = 0, cx = 0;
    for (uint32_t b = 0; b < vol; b += blockDim.x) {
        uint32_t i = b + threadIdx.x, tot;
        uint32_t vy = i < vol ? ty[i] : kNoneU, vx = i < vol ? tx[i] : kNoneU;
        uint32_t ey = blockExclusiveScan<256>(vy != kNoneU ? 1u : 0u, smem, &tot) + cy; cy += tot;
        uint32_t ex = blockExclusiveScan<256>(vx != kNoneU ? 1u : 0u, smem, &tot) + cx; cx += tot;
        if (vy != kNoneU && ey < Vw) sy[ey] = vy;
        if (vx != kNoneU && ex < Vw) sx[ex] = vx;
    }
    __syncthreads();
    const uint32_t ns = setsOf(n, L);
    const int ni = (int)n, Li = (int)L, nsi = (int)ns, H = p.num_heads;
    for (uint32_t t = threadIdx.x; t < ns * L; t += blockDim.x) {
        int j = (int)(t / L), k = (int)(t % L);
        int local = (j * Li + k) * ni / Li / nsi;                                     // :346 paper eq.(3), int arithmetic
        int prev = k > 0 ? (j * Li + k - 1) * ni / Li / nsi : -1;
        size_t s = base + (uint32_t)j;
        uint32_t iy = sy[local], ix = sx[local];
        inds[(size_t)0 * MW * L + s * L + k] = iy;                                    // :535-538
        inds[(size_t)1 * MW * L + s * L + k] = ix;
        // :544-565 slot k>0 holding the same voxel as slot k-1 is a padding key
        float my = (k > 0 && iy == sy[prev]) ? -3.4028235e+38f : 0.0f;
        float mx = (k > 0 && ix == sx[prev]) ? -3.4028235e+38f : 0.0f;
        mask[(size_t)0 * MW * L + s * L + k] = my;
        mask[(size_t)1 * MW * L + s * L + k] = mx;
        for (int h = 0; h < H; ++h) {                                                 // splitAndExpandMask :589-606
            mask0_h[(s * H + h) * L + k] = my;
            mask1_h[(s * H + h) * L + k] = mx;
        }
    }
}

class GetSetPlugin : public Plugin {
public:
    GSParams p_;
    explicit GetSetPlugin(const GSParams& p) : p_(p) {}
    const char* type() const override { return "GetSetPlugin"; }
    int nbOutputs() const override { return 5; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        int b = in[0].d[0];
        switch (i) {
            case 0: case 1: *out = dims4(b, 2, p_.max_set_num, p_.voxel_num_set); return 0;
            case 2: *out = dims1(b); return 0;
            case 3: case 4: *out = dims4(b, p_.max_set_num, p_.num_heads, p_.voxel_num_set); return 0;
        }
        return -1;
    }
    int outputType(int i, const int32_t*, int) const override { return (i == 1 || i == 3 || i == 4) ? DSVT_FLOAT : DSVT_INT32; }   // getSet.cu:706-713
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {
        if (pos == 5 || pos == 7 || pos == 8) return f32L(io[pos]);
        return pos >= 0 && pos <= 6 && i32L(io[pos]);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override {
        return alignUp(sizeof(uint32_t) * p_.max_win_num);
    }
    int enqueue(const DsvtPluginTensorDesc*, const DsvtPluginTensorDesc*, const void* const* in, void* const* out,
                void* workspace, hipStream_t stream) override {
        const uint32_t* gidx = static_cast<const uint32_t*>(in[0]);
        const uint32_t* cinw = static_cast<const uint32_t*>(in[1]);
        const uint32_t* vcnt = static_cast<const uint32_t*>(in[2]);
        const uint32_t* win_num = static_cast<const uint32_t*>(in[3]);
        uint32_t* inds = static_cast<uint32_t*>(out[0]);
        float* mask = static_cast<float*>(out[1]);
        uint32_t* set_num = static_cast<uint32_t*>(out[2]);
        float* m0 = static_cast<float*>(out[3]);
        float* m1 = static_cast<float*>(out[4]);
        uint32_t* set_base = static_cast<uint32_t*>(workspace);
        if (zeroFill) {                                                                // getSet.cu:681-686
            size_t e = (size_t)2 * p_.max_set_num * p_.voxel_num_set, eh = (size_t)p_.max_set_num * p_.num_heads * p_.voxel_num_set;
            DSVT_CHECK(hipMemsetAsync(inds, 0, sizeof(uint32_t) * e, stream));
            DSVT_CHECK(hipMemsetAsync(mask, 0, sizeof(float) * e, stream));
            DSVT_CHECK(hipMemsetAsync(m0, 0, sizeof(float) * eh, stream));
            DSVT_CHECK(hipMemsetAsync(m1, 0, sizeof(float) * eh, stream));
        }
        hipLaunchKernelGGL(gs_scan, dim3(1), dim3(1024), 0, stream, vcnt, win_num, p_, set_base, set_num);
        size_t lds = sizeof(uint32_t) * (2 * (size_t)p_.wx * p_.wy * p_.wz + 2 * (size_t)p_.max_voxel_num_per_win);
        hipLaunchKernelGGL(gs_sets, dim3(p_.max_win_num), dim3(256), lds, stream, gidx, cinw, vcnt, win_num, set_base, p_, inds, mask, m0, m1);
        return lastError();
    }
    // the reference's six ints (getSet.cu:749-758); a seventh only when the set capacity differs from the window capacity
    bool ownSetCap() const { return p_.max_set_num != p_.max_win_num; }
    size_t serializationSize() const override { return (ownSetCap() ? 7 : 6) * sizeof(int); }
    void serialize(void* b) const override {
        char* d = static_cast<char*>(b);
        wr<int>(d, p_.voxel_num_set); wr<int>(d, p_.max_win_num); wr<int>(d, p_.max_voxel_num_per_win);
        wr<int>(d, p_.wx); wr<int>(d, p_.wy); wr<int>(d, p_.wz);
        if (ownSetCap()) wr<int>(d, p_.max_set_num);
    }
    Plugin* clone() const override { return new GetSetPlugin(p_); }
};
static Plugin* gsNew(GSParams p) {
    p.num_heads = 8;                                                                   // NUM_HEADS, include/params.h:73
    if (p.max_set_num <= 0) p.max_set_num = p.max_win_num;
    if (p.max_win_num <= 0 || p.max_voxel_num_per_win <= 0 || p.voxel_num_set <= 0 || p.wx <= 0 || p.wy <= 0 || p.wz <= 0) return nullptr;
    size_t lds = sizeof(uint32_t) * (2 * (size_t)p.wx * p.wy * p.wz + 2 * (size_t)p.max_voxel_num_per_win);
    if (lds > 150 * 1024) return nullptr;
    return new GetSetPlugin(p);
}
static Plugin* gsCreate(const DsvtPluginFieldCollection* fc) {
    GSParams p{}; int w[3];
    p.max_win_num = fieldInt(fc, "max_win_num"); p.max_voxel_num_per_win = fieldInt(fc, "max_voxel_num_per_win");
    p.voxel_num_set = fieldInt(fc, "voxel_num_set"); fieldInts(fc, "win_shape", w, 3);
    p.wx = w[0]; p.wy = w[1]; p.wz = w[2];
    p.max_set_num = fieldInt(fc, "max_set_num", 0);
    return gsNew(p);
}
static Plugin* gsDeser(const void* data, size_t len) {
    const int extra = trailingInts(len, 6 * sizeof(int), 1);
    if (extra < 0) return nullptr;
    const char* d = static_cast<const char*>(data); GSParams p{};
    p.voxel_num_set = rd<int>(d); p.max_win_num = rd<int>(d); p.max_voxel_num_per_win = rd<int>(d);
    p.wx = rd<int>(d); p.wy = rd<int>(d); p.wz = rd<int>(d);
    if (extra >= 1) p.max_set_num = rd<int>(d);
    return gsNew(p);
}
static Creator g_gsCreator{"GetSetPlugin",
    {{"max_win_num", DSVT_FIELD_INT32}, {"max_voxel_num_per_win", DSVT_FIELD_INT32}, {"voxel_num_set", DSVT_FIELD_INT32},
     {"win_shape", DSVT_FIELD_INT32}},                                                 // :782-785
    gsCreate, gsDeser, {}, {}};
static Registrar g_gsReg(&g_gsCreator);


// =====================================================================================
// DsvtSetPartitionPlugin -- WindowPartition + GetSet of ALL window configurations of a frame in four launches
// (the per-configuration plugins above take 2 x (4 + 2) launches and two workspace memsets: ~75 us of launch latency per frame for
// ~3 MB of integer work).  blockIdx.y = configuration:
//   sp_count    window id of every voxel, run-aggregated window counters                 (wp_count)
//   sp_scan     per configuration ONE workgroup: window ranks / segments (wp_scan) AND the set bases of the ranked windows (gs_scan)
//   sp_scatter  voxels into their window's segment                                        (wp_scatter)
//   sp_window   per non-empty window: order its voxels (wp_fill), write the in-window coordinates, sort by the two keys through the
//               LDS tables and emit the sets (gs_sets) -- gidx / cinw never exist in memory
// Outputs per configuration k: c2d_k [1,P,3], inds_k [1,2,S,36], mask_k [1,2,S,36], set_num_k [1]: exactly the tensors of
// WindowPartitionPlugin output 4 and GetSetPlugin outputs 0..2 (bit-identical; tests/test_plugins_gpu.py), i.e. what the fused
// attention / QKV ops consume.  Same reference lines as above: windowPartition.cu:278-470, getSet.cu:267-704.
// =====================================================================================
constexpr int kMaxCfg = 4;
struct SPParams { int K, max_set_num, voxel_num_set, max_pillars; WPParams wp[kMaxCfg]; int dense_off[kMaxCfg + 1]; int frames; };       // frames: coords.x = frame index < frames (Points2Features "frames")
// the output tensors of one enqueue, passed by value as a kernel argument (no device-side pointer table that could go stale)
struct SPOuts { uint32_t* c2d[kMaxCfg]; uint32_t* inds[kMaxCfg]; float* mask[kMaxCfg]; uint32_t* snum[kMaxCfg]; };

__global__ void __launch_bounds__(256)
sp_count(const uint4* __restrict__ coords, const uint32_t* __restrict__ voxel_num, SPParams sp,
         uint32_t* __restrict__ win_cnt, uint32_t* __restrict__ vox_win, uint32_t* __restrict__ vox_slot)
{
    const int k = blockIdx.y;
    const WPParams& p = sp.wp[k];
    win_cnt += sp.dense_off[k]; vox_win += (size_t)k * sp.max_pillars; vox_slot += (size_t)k * sp.max_pillars;
    uint32_t n = *voxel_num; if (n > (uint32_t)sp.max_pillars) n = sp.max_pillars;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t win = kNoneU, ix, iy, iz;
    if (v < n) {
        const uint4 co = coords[v];
        winOf(co, p, win, ix, iy, iz);
        // several frames per call: the frame index (coords.x) is the slowest window coordinate
        if (win != kNoneU) win = co.x < (uint32_t)sp.frames ? win + co.x * (uint32_t)(p.nwx * p.nwy * p.nwz) : kNoneU;
        vox_win[v] = win;
    }
    const uint32_t prev = __shfl_up(win, 1, kWave);                   // one atomic per run of equal windows (see wp_count)
    const bool head = lane == 0 || prev != win;
    const unsigned long long heads = __ballot(head);
    const unsigned long long below = heads & ((2ull << lane) - 1ull);
    const int leader = 63 - __builtin_clzll(below);
    uint32_t base = 0;
    if (head && win != kNoneU) {
        const unsigned long long nxt = heads & ~((2ull << lane) - 1ull);
        const int len = (nxt ? __builtin_ctzll(nxt) : 64) - lane;
        base = atomicAdd(&win_cnt[win], (uint32_t)len);
    }
    base = __shfl(base, leader, kWave);
    if (v < n) vox_slot[v] = win == kNoneU ? 0u : base + (uint32_t)(lane - leader);
}

// one workgroup per configuration
template <int WPT>                                // slabs of 1024 windows per round (the host picks it from the dense grid's size)
__global__ void __launch_bounds__(1024)
sp_scan(const uint32_t* __restrict__ win_cnt, SPParams sp, uint32_t* __restrict__ win_seg, uint32_t* __restrict__ rank2win,
        uint32_t* __restrict__ set_base, uint32_t* __restrict__ win_num, SPOuts outs)
{
    __shared__ uint32_t smem[2 * WPT * (1024 / kWave + 1)];
    const int k = blockIdx.x;
    const WPParams& p = sp.wp[k];
    const int dense = sp.dense_off[k + 1] - sp.dense_off[k];
    win_cnt += sp.dense_off[k]; win_seg += sp.dense_off[k];
    rank2win += (size_t)k * p.max_win_num; set_base += (size_t)k * p.max_win_num;
    const uint32_t L = (uint32_t)sp.voxel_num_set, Vw = (uint32_t)p.max_voxel_num_per_win;
    uint32_t carry_o = 0, carry_f = 0, carry_s = 0;
    // A round covers WPT slabs of 1024 windows (a frame's 3200 dense windows: one round of four slabs, four frames' 12,800: two rounds of
    // eight): thread t holds window t of every slab, so every load and store of the round is
    // lane-contiguous (the ranks of a slab's non-empty windows ascend with the lane too).  All slabs are scanned across the workgroup at
    // once (one set of barriers for the 2 WPT rank / voxel-offset scans, one for the WPT set-base scans); the slab totals chain in registers.
    // (Eight CONSECUTIVE windows per thread, the layout before, made every access 32 bytes apart from its neighbour's: 64 cache lines per
    // wavefront instruction through ONE CU's memory pipeline -- 21 us per four-frame launch, 8 of them for a single frame's 3200 windows.)
    for (int b = 0; b < dense; b += 1024 * WPT) {
        uint32_t c[WPT], v[2 * WPT], tot2[2 * WPT];
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int w = b + i * 1024 + (int)threadIdx.x;
            c[i] = w < dense ? win_cnt[w] : 0u;
            v[i] = c[i] > 0 ? 1u : 0u; v[WPT + i] = c[i];
        }
        blockExclusiveScanK<1024, 2 * WPT>(v, smem, tot2);
        // sets of a ranked window: ceil(min(c, Vw) / L) (getSet.cu:335 on the clamped count of windowPartition.cu:336-340); windows beyond
        // the window capacity get none
        uint32_t eo[WPT], ns[WPT], ts[WPT];
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            eo[i] = v[i] + carry_o; v[WPT + i] += carry_f;                  // exclusive rank / voxel offset of window (i, t)
            carry_o += tot2[i]; carry_f += tot2[WPT + i];
            const bool ranked = c[i] > 0 && eo[i] < (uint32_t)p.max_win_num;
            ns[i] = ranked ? setsOf(c[i] > Vw ? Vw : c[i], L) : 0u;
        }
        uint32_t es[WPT];
#pragma unroll
        for (int i = 0; i < WPT; ++i) es[i] = ns[i];
        blockExclusiveScanK<1024, WPT>(es, smem, ts);
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int w = b + i * 1024 + (int)threadIdx.x;
            const uint32_t e = es[i] + carry_s; carry_s += ts[i];
            if (w < dense) {
                win_seg[w] = v[WPT + i];
                if (ns[i] > 0u) {                                                   // ranked (a non-empty window has at least one set)
                    rank2win[eo[i]] = (uint32_t)w;
                    set_base[eo[i]] = e + ns[i] <= (uint32_t)sp.max_set_num ? e : kNoneU;       // capacity guard the reference lacks (getSet.cu:337)
                }
            }
        }
    }
    if (threadIdx.x == 0) {
        win_num[k] = carry_o < (uint32_t)p.max_win_num ? carry_o : (uint32_t)p.max_win_num;
        // the sets that fit form a prefix (bases ascend): their number is the largest base + ns that still fits = min(total, ...) only when
        // nothing overflowed; with an overflow the last fitting window is found by sp_window's writers, so publish the clamped total here
        *outs.snum[k] = carry_s <= (uint32_t)sp.max_set_num ? carry_s : kNoneU;
    }
}

__global__ void __launch_bounds__(256)
sp_scatter(const uint32_t* __restrict__ voxel_num, SPParams sp, const uint32_t* __restrict__ vox_win, const uint32_t* __restrict__ vox_slot,
           const uint32_t* __restrict__ win_seg, uint32_t* __restrict__ sorted_vox)
{
    const int k = blockIdx.y;
    vox_win += (size_t)k * sp.max_pillars; vox_slot += (size_t)k * sp.max_pillars; sorted_vox += (size_t)k * sp.max_pillars;
    win_seg += sp.dense_off[k];
    uint32_t n = *voxel_num; if (n > (uint32_t)sp.max_pillars) n = sp.max_pillars;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint32_t w = vox_win[v];
    if (w != kNoneU) sorted_vox[win_seg[w] + vox_slot[v]] = v;
}

// the ordering step of wp_fill: afterwards lds[0 .. n) holds the window's voxel ids ascending.  lds: 2 x cap words (cap = pow2 >= volume)
__device__ __forceinline__ void orderWindowVoxels(const uint4* __restrict__ coords, const WPParams& p, uint32_t n, uint32_t seg,
                                                  const uint32_t* __restrict__ sorted_vox, uint32_t* lds, unsigned long long* bits)
{
    const uint32_t vol = (uint32_t)(p.wx * p.wy);
    int m = 1; while ((uint32_t)m < n) m <<= 1;
    bool sorted = false;
    if (p.wz == 1 && p.sz == 1 && vol <= 1024u && n <= vol) {           // occupancy bitmap + popcount, verified (see wp_fill)
        uint32_t* ord = lds + m;
        if (threadIdx.x < 17) bits[threadIdx.x] = 0ull;
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < n; s += blockDim.x) {
            const uint32_t v = sorted_vox[seg + s];
            uint32_t win, ix, iy, iz;
            winOf(coords[v], p, win, ix, iy, iz);
            const uint32_t cell = iy * (uint32_t)p.wx + ix;
            atomicOr(&bits[cell >> 6], 1ull << (cell & 63));
            lds[s] = v;
        }
        __syncthreads();
        for (uint32_t s0 = threadIdx.x; s0 < n; s0 += blockDim.x) {
            const uint32_t v = lds[s0];
            uint32_t win, ix, iy, iz;
            winOf(coords[v], p, win, ix, iy, iz);
            const uint32_t cell = iy * (uint32_t)p.wx + ix;
            uint32_t s = (uint32_t)__popcll(bits[cell >> 6] & ((1ull << (cell & 63)) - 1ull));
            for (uint32_t q = 0; q < (cell >> 6); ++q) s += (uint32_t)__popcll(bits[q]);
            ord[s] = v;
        }
        __syncthreads();
        int bad = 0;
        for (uint32_t i = threadIdx.x; i + 1 < n; i += blockDim.x) bad |= ord[i] >= ord[i + 1];
        sorted = !__syncthreads_or(bad);
        if (sorted)
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) lds[i] = ord[i];
        __syncthreads();
    }
    if (!sorted) {
        for (int i = threadIdx.x; i < m; i += blockDim.x) lds[i] = (uint32_t)i < n ? sorted_vox[seg + i] : kNoneU;
        __syncthreads();
        bitonicSortLds(lds, m);
    }
}

// one workgroup per (ranked window, configuration).  NT threads: ONE wavefront when the windows are small (a 12 x 12 window holds 39 voxels
// on average: the dozen barriers of this kernel are free inside one wavefront, and four times as many windows are resident), 256 otherwise
template <int NT>
__global__ void __launch_bounds__(NT)
sp_window(const uint4* __restrict__ coords, SPParams sp, const uint32_t* __restrict__ win_num, const uint32_t* __restrict__ rank2win,
          const uint32_t* __restrict__ win_cnt, const uint32_t* __restrict__ win_seg, const uint32_t* __restrict__ sorted_vox,
          const uint32_t* __restrict__ set_base, int cap_words, SPOuts outs)
{
    extern __shared__ uint32_t lds[];
    __shared__ unsigned long long bits[17];
    __shared__ uint32_t smem[2 * (NT / kWave + 1)];
    const int k = blockIdx.y;
    const WPParams& p = sp.wp[k];
    const uint32_t r = blockIdx.x;
    if (r >= win_num[k]) return;
    rank2win += (size_t)k * p.max_win_num; set_base += (size_t)k * p.max_win_num;
    win_cnt += sp.dense_off[k]; win_seg += sp.dense_off[k]; sorted_vox += (size_t)k * sp.max_pillars;
    uint32_t* c2d = outs.c2d[k]; uint32_t* inds = outs.inds[k]; float* mask = outs.mask[k];
    // (a window with more voxels than its LDS region -- duplicate pillar coordinates only -- is truncated: see wp_fill)
    const uint32_t w = rank2win[r], nall = win_cnt[w] < (uint32_t)cap_words ? win_cnt[w] : (uint32_t)cap_words, seg = win_seg[w];
    const uint32_t Vw = (uint32_t)p.max_voxel_num_per_win, L = (uint32_t)sp.voxel_num_set, MS = (uint32_t)sp.max_set_num;
    orderWindowVoxels(coords, p, nall, seg, sorted_vox, lds, bits);
    // ---- WindowPartition outputs that survive: in-window coordinates per voxel (:362-364); voxels beyond the cap are dropped (:305)
    const uint32_t n = nall < Vw ? nall : Vw;
    const uint32_t vol = (uint32_t)(p.wx * p.wy * p.wz);
    uint32_t* ty = lds + cap_words;     // [vol]  key_y -> voxel id
    uint32_t* tx = ty + vol;            // [vol]
    uint32_t* sy = tx + vol;            // [Vw]   voxel ids ascending by key_y
    uint32_t* sx = sy + Vw;             // [Vw]
    for (uint32_t i = threadIdx.x; i < vol; i += blockDim.x) { ty[i] = kNoneU; tx[i] = kNoneU; }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { sy[i] = 0; sx[i] = 0; }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < nall; s += blockDim.x) {
        const uint32_t v = lds[s];
        uint32_t win, ix, iy, iz;
        winOf(coords[v], p, win, ix, iy, iz);
        if (s < Vw) {
            c2d[(size_t)v * 3 + 0] = iz; c2d[(size_t)v * 3 + 1] = iy; c2d[(size_t)v * 3 + 2] = ix;
            const uint32_t ky = iy * (uint32_t)(p.wx * p.wz) + ix * (uint32_t)p.wz + iz;          // getSet.cu:386-387
            const uint32_t kx = ix * (uint32_t)(p.wy * p.wz) + iy * (uint32_t)p.wz + iz;          // :461-462
            if (ky < vol) ty[ky] = v;
            if (kx < vol) tx[kx] = v;
        } else {
            c2d[(size_t)v * 3 + 0] = 0; c2d[(size_t)v * 3 + 1] = 0; c2d[(size_t)v * 3 + 2] = 0;
        }
    }
    __syncthreads();
    const uint32_t base = set_base[r];
    if (base == kNoneU) {                     // this window's sets do not fit: the FIRST such window publishes the set count
        // (bases ascend with the rank: the previous window fits iff its base is valid)
        if (threadIdx.x == 0 && (r == 0 || set_base[r - 1] != kNoneU)) {
            uint32_t prev = 0;
            if (r > 0) { const uint32_t pw = rank2win[r - 1]; const uint32_t pc = win_cnt[pw] < Vw ? win_cnt[pw] : Vw; prev = set_base[r - 1] + setsOf(pc, L); }
            *outs.snum[k] = prev;
        }
        return;
    }
    // compact both tables in key order (ascending sort of unique keys)
    uint32_t cy = 0, cx = 0;
    for (uint32_t b = 0; b < vol; b += blockDim.x) {
        const uint32_t i = b + threadIdx.x;
        const uint32_t vy = i < vol ? ty[i] : kNoneU, vx = i < vol ? tx[i] : kNoneU;
        uint32_t e2[2] = {vy != kNoneU ? 1u : 0u, vx != kNoneU ? 1u : 0u}, tot2[2];
        blockExclusiveScanK<NT, 2>(e2, smem, tot2);                                      // both tables behind one set of barriers
        const uint32_t ey = e2[0] + cy, ex = e2[1] + cx; cy += tot2[0]; cx += tot2[1];
        if (vy != kNoneU && ey < Vw) sy[ey] = vy;
        if (vx != kNoneU && ex < Vw) sx[ex] = vx;
    }
    __syncthreads();
    const uint32_t ns = setsOf(n, L);
    const int ni = (int)n, Li = (int)L, nsi = (int)ns;
    for (uint32_t t = threadIdx.x; t < ns * L; t += blockDim.x) {
        const int j = (int)(t / L), kk = (int)(t % L);
        const int local = (j * Li + kk) * ni / Li / nsi;                                // getSet.cu:346 paper eq.(3), int arithmetic
        const int prev = kk > 0 ? (j * Li + kk - 1) * ni / Li / nsi : -1;
        const size_t s = base + (uint32_t)j;
        const uint32_t iy = sy[local], ix = sx[local];
        inds[(size_t)0 * MS * L + s * L + kk] = iy;                                     // :535-538
        inds[(size_t)1 * MS * L + s * L + kk] = ix;
        mask[(size_t)0 * MS * L + s * L + kk] = (kk > 0 && iy == sy[prev]) ? -3.4028235e+38f : 0.0f;      // :544-565
        mask[(size_t)1 * MS * L + s * L + kk] = (kk > 0 && ix == sx[prev]) ? -3.4028235e+38f : 0.0f;
    }
}

class DsvtSetPartitionPlugin : public Plugin {
public:
    SPParams sp_{};
    explicit DsvtSetPartitionPlugin(const SPParams& sp) : sp_(sp) {}
    const char* type() const override { return "DsvtSetPartitionPlugin"; }
    int nbOutputs() const override { return 4 * sp_.K; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i < 0 || i >= nbOutputs()) return -1;
        const int b = in[0].d[0];
        switch (i % 4) {
            case 0: *out = dims3(b, sp_.max_pillars, 3); return 0;                                  // c2d
            case 1: case 2: *out = dims4(b, 2, sp_.max_set_num, sp_.voxel_num_set); return 0;       // inds, mask
            default: *out = dims1(b); return 0;                                                     // set_num
        }
    }
    int outputType(int i, const int32_t*, int) const override { return i % 4 == 2 ? DSVT_FLOAT : DSVT_INT32; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {
        if (pos < 2) return i32L(io[pos]);
        return (pos - 2) % 4 == 2 ? f32L(io[pos]) : i32L(io[pos]);
    }
    int denseAll() const { return sp_.dense_off[sp_.K]; }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override {
        const size_t mw = sp_.wp[0].max_win_num;
        return 2 * alignUp(sizeof(uint32_t) * denseAll()) + 2 * alignUp(sizeof(uint32_t) * sp_.K * mw) + alignUp(sizeof(uint32_t) * kMaxCfg) +
               3 * alignUp(sizeof(uint32_t) * (size_t)sp_.K * sp_.max_pillars);
    }
    int enqueue(const DsvtPluginTensorDesc*, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void* workspace,
                hipStream_t stream) override {
        const uint4* coords = static_cast<const uint4*>(in[0]);
        const uint32_t* voxel_num = static_cast<const uint32_t*>(in[1]);
        const int K = sp_.K, mp = sp_.max_pillars, mw = sp_.wp[0].max_win_num;
        WsCarver ws(workspace);
        uint32_t* win_cnt = ws.take<uint32_t>(denseAll());
        uint32_t* win_seg = ws.take<uint32_t>(denseAll());
        uint32_t* rank2win = ws.take<uint32_t>((size_t)K * mw);
        uint32_t* set_base = ws.take<uint32_t>((size_t)K * mw);
        uint32_t* win_num = ws.take<uint32_t>(kMaxCfg);
        uint32_t* vox_win = ws.take<uint32_t>((size_t)K * mp);
        uint32_t* vox_slot = ws.take<uint32_t>((size_t)K * mp);
        uint32_t* sorted_vox = ws.take<uint32_t>((size_t)K * mp);
        SPOuts o{};
        for (int k = 0; k < K; ++k) {
            o.c2d[k] = static_cast<uint32_t*>(out[4 * k]); o.inds[k] = static_cast<uint32_t*>(out[4 * k + 1]);
            o.mask[k] = static_cast<float*>(out[4 * k + 2]); o.snum[k] = static_cast<uint32_t*>(out[4 * k + 3]);
        }
        DSVT_CHECK(hipMemsetAsync(win_cnt, 0, sizeof(uint32_t) * denseAll(), stream));
        if (zeroFill)
            for (int k = 0; k < K; ++k) {
                const size_t e = (size_t)2 * sp_.max_set_num * sp_.voxel_num_set;
                DSVT_CHECK(hipMemsetAsync(o.c2d[k], 0, sizeof(uint32_t) * (size_t)mp * 3, stream));
                DSVT_CHECK(hipMemsetAsync(o.inds[k], 0, sizeof(uint32_t) * e, stream));
                DSVT_CHECK(hipMemsetAsync(o.mask[k], 0, sizeof(float) * e, stream));
            }
        hipLaunchKernelGGL(sp_count, dim3(cdiv(mp, 256), K), dim3(256), 0, stream, coords, voxel_num, sp_, win_cnt, vox_win, vox_slot);
        int dmax = 0;
        for (int k = 0; k < K; ++k) dmax = std::max(dmax, sp_.dense_off[k + 1] - sp_.dense_off[k]);
        if (dmax <= 4 * 1024) hipLaunchKernelGGL(sp_scan<4>, dim3(K), dim3(1024), 0, stream, win_cnt, sp_, win_seg, rank2win, set_base, win_num, o);
        else hipLaunchKernelGGL(sp_scan<8>, dim3(K), dim3(1024), 0, stream, win_cnt, sp_, win_seg, rank2win, set_base, win_num, o);     // (sixteen slabs spill: 22 against 14 us for four frames)
        hipLaunchKernelGGL(sp_scatter, dim3(cdiv(mp, 256), K), dim3(256), 0, stream, voxel_num, sp_, vox_win, vox_slot, win_seg, sorted_vox);
        int capw = 2, ldsw = 0;
        for (int k = 0; k < K; ++k) {
            const int vol = sp_.wp[k].wx * sp_.wp[k].wy * sp_.wp[k].wz; int cap = 1; while (cap < vol) cap <<= 1;
            capw = std::max(capw, 2 * cap);
            ldsw = std::max(ldsw, 2 * vol + 2 * sp_.wp[k].max_voxel_num_per_win);
        }
        if (capw <= 2 * 256)                                          // windows of up to 256 cells: one wavefront each (576-cell windows: 45 us against 34 with 256 threads)
            hipLaunchKernelGGL(sp_window<64>, dim3(mw, K), dim3(64), sizeof(uint32_t) * (size_t)(capw + ldsw), stream, coords, sp_, win_num, rank2win,
                               win_cnt, win_seg, sorted_vox, set_base, capw, o);
        else
            hipLaunchKernelGGL(sp_window<256>, dim3(mw, K), dim3(256), sizeof(uint32_t) * (size_t)(capw + ldsw), stream, coords, sp_, win_num, rank2win,
                               win_cnt, win_seg, sorted_vox, set_base, capw, o);
        return lastError();
    }
    size_t serializationSize() const override { return sizeof(int) * (5 + 9 * (size_t)sp_.K + 2); }
    void serialize(void* b) const override {
        char* d = static_cast<char*>(b);
        wr<int>(d, sp_.K); wr<int>(d, sp_.max_set_num); wr<int>(d, sp_.voxel_num_set); wr<int>(d, sp_.max_pillars);
        wr<int>(d, sp_.wp[0].max_win_num); wr<int>(d, sp_.wp[0].max_voxel_num_per_win); wr<int>(d, sp_.frames);
        for (int k = 0; k < sp_.K; ++k) {
            const WPParams& p = sp_.wp[k];
            wr<int>(d, p.sx); wr<int>(d, p.sy); wr<int>(d, p.sz); wr<int>(d, p.wx); wr<int>(d, p.wy); wr<int>(d, p.wz); wr<int>(d, p.hx); wr<int>(d, p.hy); wr<int>(d, p.hz);
        }
    }
    Plugin* clone() const override { return new DsvtSetPartitionPlugin(sp_); }
};
static Plugin* spNew(int K, int mw, int vw, int L, int ms, int mp, const int* shape, const int* wins, const int* shifts, int frames = 1) {
    if (K < 1 || K > kMaxCfg || mw <= 0 || vw <= 0 || L <= 0 || mp <= 0 || frames < 1 || frames > 64) return nullptr;
    SPParams sp{}; sp.K = K; sp.max_set_num = ms > 0 ? ms : mw; sp.voxel_num_set = L; sp.max_pillars = mp; sp.frames = frames;
    for (int k = 0; k < K; ++k) {
        WPParams& p = sp.wp[k];
        p.max_win_num = mw; p.max_voxel_num_per_win = vw; p.sx = shape[0]; p.sy = shape[1]; p.sz = shape[2];
        p.wx = wins[3 * k]; p.wy = wins[3 * k + 1]; p.wz = wins[3 * k + 2]; p.hx = shifts[3 * k]; p.hy = shifts[3 * k + 1]; p.hz = shifts[3 * k + 2];
        if (p.wx <= 0 || p.wy <= 0 || p.wz <= 0 || p.sx <= 0 || p.sy <= 0 || p.sz <= 0 || p.hx < 0 || p.hy < 0 || p.hz < 0) return nullptr;
        if ((long)p.wx * p.wy * p.wz > 8192) return nullptr;
        p.nwx = (int)(ceilf((float)(p.sx / p.wx)) + 1); p.nwy = (int)(ceilf((float)(p.sy / p.wy)) + 1); p.nwz = (int)(ceilf((float)(p.sz / p.wz)) + 1);   // windowPartition.cu:425-427
        sp.dense_off[k + 1] = sp.dense_off[k] + p.nwx * p.nwy * p.nwz * frames;
    }
    return new DsvtSetPartitionPlugin(sp);
}
static Plugin* spCreate(const DsvtPluginFieldCollection* fc) {
    const int K = fieldInt(fc, "num_configs");
    if (K < 1 || K > kMaxCfg) return nullptr;
    int shape[3], wins[3 * kMaxCfg], shifts[3 * kMaxCfg];
    fieldInts(fc, "sparse_shape", shape, 3); fieldInts(fc, "win_shapes", wins, 3 * K); fieldInts(fc, "shift_lists", shifts, 3 * K);
    return spNew(K, fieldInt(fc, "max_win_num"), fieldInt(fc, "max_voxel_num_per_win"), fieldInt(fc, "voxel_num_set"), fieldInt(fc, "max_set_num", 0),
                 fieldInt(fc, "max_pillars_num"), shape, wins, shifts, fieldInt(fc, "frames", 1));
}
static Plugin* spDeser(const void* data, size_t len) {
    if (len < 7 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    const int K = rd<int>(d), ms = rd<int>(d), L = rd<int>(d), mp = rd<int>(d), mw = rd<int>(d), vw = rd<int>(d), frames = rd<int>(d);
    if (K < 1 || K > kMaxCfg || len < sizeof(int) * (7 + 9 * (size_t)K)) return nullptr;
    int shape[3] = {0, 0, 0}, wins[3 * kMaxCfg], shifts[3 * kMaxCfg];
    for (int k = 0; k < K; ++k) {
        shape[0] = rd<int>(d); shape[1] = rd<int>(d); shape[2] = rd<int>(d);
        for (int e = 0; e < 3; ++e) wins[3 * k + e] = rd<int>(d);
        for (int e = 0; e < 3; ++e) shifts[3 * k + e] = rd<int>(d);
    }
    return spNew(K, mw, vw, L, ms, mp, shape, wins, shifts, frames);
}
static Creator g_spCreator{"DsvtSetPartitionPlugin",
    {{"max_win_num", DSVT_FIELD_INT32}, {"max_voxel_num_per_win", DSVT_FIELD_INT32}, {"voxel_num_set", DSVT_FIELD_INT32},
     {"max_set_num", DSVT_FIELD_INT32}, {"max_pillars_num", DSVT_FIELD_INT32}, {"sparse_shape", DSVT_FIELD_INT32},
     {"num_configs", DSVT_FIELD_INT32}, {"win_shapes", DSVT_FIELD_INT32}, {"shift_lists", DSVT_FIELD_INT32}, {"frames", DSVT_FIELD_INT32}},
    spCreate, spDeser, {}, {}};
static Registrar g_spReg(&g_spCreator);

}  // namespace dsvt
