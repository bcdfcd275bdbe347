// points2features.hip -- Points2FeaturesPlugin for gfx950.
//
// Replaces plugins/src/points2Features.cu:669-990 (generateVoxels_random_kernel,
// generateBaseFeatures_kernel, generateFeatures_kernel).  Same tensors, different machine:
// the reference scatters every point into a dense 468x468x48x4 float scratch (168 MB, zeroed
// every frame) with racy atomic slot order.  Here
//   1. p2f_count     one coalesced float4 read per point, range filter, cell key, one atomic
//                    per in-range point on a 0.9 MB cell histogram (order-independent counts);
//   2. p2f_scan      ONE single-pass exclusive scan over the cells (decoupled look-back between 1024-cell tiles, tile
//                    order by ticket) gives, in ascending cell-key order, pillar ids, point-segment offsets and the
//                    compact point offsets (deterministic pillar order), the pillar records and the two counts;
//   3. p2f_scatter   point ids land in their cell's segment (arbitrary order inside it);
//   4. p2f_pillar    one wavefront per pillar ranks the segment by point id (= input order),
//                    keeps the first T, sums the cluster mean sequentially in slot order like
//                    the reference, and writes the 10-d features / point-id rows.
// Canonical order: pillars ascending by y*GX+x, points in input order, first T.
// grid_size z > 1 (BASELINE configs[4], a 3-D voxel grid; NOT in the reference, whose voxel z index is forced to 0,
// points2Features.cu:689-690,755): the cell key gains the z index computed like x and y (the reference computes it the same way for
// the centre offset, :846), key = (z*GY + y)*GX + x, voxels ascending by that key, coords = (0, z, y, x).
//
// Compiled with -ffp-contract=off: cell indices and features follow the reference's
// expression order exactly (fp32 subtract, IEEE divide, floorf; the centre offset in double).
#include "plugin_base.h"
#include "device_utils.h"

namespace dsvt {

struct P2FParams {
    int max_points_num, max_points_num_voxel_filter, max_pillars_num;
    int point_feature_num, feature_num, max_num_points_per_voxel;
    float min_x, max_x, min_y, max_y, min_z, max_z;
    float vx, vy, vz;
    int gx, gy, gz;
    int frames;       // >= 1: the points tensor holds `frames` slabs of max_points_num rows with one count each (see the plugin class)
};

constexpr uint32_t kNone = 0xffffffffu;
constexpr int kCPT = 16;         // consecutive cells per thread of the scan
constexpr int kTile = 256 * kCPT;   // cells per scan tile: the look-back chain is one link per tile (1024-cell tiles, 428 links for two frames: 43 us)

// cell key of one point, or kNone (out of range / out of the grid)
__device__ __forceinline__ uint32_t p2fCellOf(const float4 q, const P2FParams& p, uint32_t fr) {
    if (q.x < p.min_x || q.x >= p.max_x || q.y < p.min_y || q.y >= p.max_y || q.z < p.min_z || q.z >= p.max_z) return kNone;   // points2Features.cu:683-685
    int ix = (int)floorf((q.x - p.min_x) / p.vx);             // :687
    int iy = (int)floorf((q.y - p.min_y) / p.vy);             // :688
    // fp32 rounding can give ix == gx for a point just below max_x; like the reference
    // (:689-690) the linear index then aliases the first cell of the next row.  Only an
    // index past the last cell (undefined behaviour in the reference) is dropped.
    uint32_t c = (uint32_t)(iy * p.gx + ix);
    if (p.gz > 1) {                                           // 3-D grid (not in the reference): same floorf rule for z
        int iz = (int)floorf((q.z - p.min_z) / p.vz);
        c = (uint32_t)((iz * p.gy + iy) * p.gx + ix);
    }
    if (c >= (uint32_t)(p.gx * p.gy * p.gz)) return kNone;
    return c + fr * (uint32_t)(p.gx * p.gy * p.gz);           // frames are one more (slowest) grid dimension: pillars ascend by (frame, cell)
}

// FOUR points per thread (round 3): the kernel is bound by the round trip of its returning atomic -- 246 GB/s on a pure float4 stream with
// one point per thread (profiles/r02_h_*: 46.8 us for four frames) -- so a thread issues its four loads, then its four atomics, and only
// then waits: four round trips in flight per lane instead of one.  Rows i, i + 256, i + 512, i + 768 of a 1024-row block: every access
// of a wave stays a contiguous 1 KB.
constexpr int kPPT = 4;
__global__ void __launch_bounds__(256)
p2f_count(const float4* __restrict__ pts, const uint32_t* __restrict__ n_ptr, P2FParams p,
          uint32_t* __restrict__ cell_cnt, uint32_t* __restrict__ pt_cell, uint32_t* __restrict__ pt_slot)
{
    const uint32_t i0 = blockIdx.x * (256u * kPPT) + threadIdx.x;
    const uint32_t total = (uint32_t)p.max_points_num * (uint32_t)p.frames;
    float4 q[kPPT]; uint32_t fr[kPPT]; bool live[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
        const uint32_t i = i0 + 256u * k;
        fr[k] = i / (uint32_t)p.max_points_num;                   // frame of this row (0 when frames == 1)
        live[k] = i < total;
        if (live[k]) {
            uint32_t n = n_ptr[fr[k]];
            if (n > (uint32_t)p.max_points_num) n = p.max_points_num;
            live[k] = i - fr[k] * (uint32_t)p.max_points_num < n;
        }
        q[k] = live[k] ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    uint32_t cell[kPPT], slot[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
        cell[k] = live[k] ? p2fCellOf(q[k], p, fr[k]) : kNone;
        slot[k] = cell[k] != kNone ? atomicAdd(&cell_cnt[cell[k]], 1u) : 0u;      // count only; order fixed later
    }
#pragma unroll
    for (int k = 0; k < kPPT; ++k)
        if (live[k]) { pt_cell[i0 + 256u * k] = cell[k]; pt_slot[i0 + 256u * k] = slot[k]; }
}

__device__ __forceinline__ void cellTriple(uint32_t c, uint32_t T, uint32_t& occ, uint32_t& full, uint32_t& kept) {
    occ = c > 0 ? 1u : 0u; full = c; kept = c < T ? c : T;        // :746-748
}

// ---- single-pass scan over the cells ------------------------------------------------------------------------------
// Three running sums travel together: occupied cells (-> pillar id), full point counts (-> segment offset) and kept point
// counts (-> compact point offset).  Round 2 packed all three into ONE 64-bit word behind a 2-bit flag (20 + 21 + 21 bits), which capped
// max_points_num x frames at 2^20: five frames per launch, or a million-point cap, were rejected.  Round 3 (CUB's layout for wide
// types): a tile owns one FLAG word and two value slots of three 32-bit sums -- its aggregate and its inclusive prefix.  A slot is
// written once, BEFORE the release store that moves the flag to the state that names it (1 = aggregate valid, 2 = inclusive prefix valid),
// and a reader that acquires a flag value reads the slot that value names: the look-back polls one 4-byte word per predecessor, as before,
// and the sums are full 32-bit counters.  (A first version with two independently flagged 64-bit words cost 41 us per four-frame launch
// against 28 for the packed word: twice the polling traffic.)
// Tiles take their index from a ticket counter, so a tile only ever waits for tiles that are already running.
constexpr uint32_t kFlagAgg = 1u, kFlagInc = 2u;
constexpr int kStateWords = 8;                    // per tile: flag | pad | aggregate (occupied, full, kept) | inclusive (occupied, full, kept)

// scan_state: 32-bit words; [0] ticket counter, [8 + 8 t ..] the kStateWords of tile t -- zeroed by the same memset as cell_cnt
__global__ void __launch_bounds__(256)
p2f_scan(const uint32_t* __restrict__ cell_cnt, int ncell, P2FParams p, uint32_t* __restrict__ scan_state, int ntiles,
         uint32_t* __restrict__ cell_seg, uint32_t* __restrict__ pil_seg, uint32_t* __restrict__ pil_full,
         uint32_t* __restrict__ pil_ptoff, uint32_t* __restrict__ coords, uint32_t* __restrict__ pcnt,
         uint32_t* __restrict__ pillar_num, uint32_t* __restrict__ point_num)
{
    __shared__ uint32_t smem[256 / kWave + 1];
    __shared__ uint32_t s_tile, s_pref[3];
    const uint32_t T = p.max_num_points_per_voxel;
    if (threadIdx.x == 0) s_tile = atomicAdd(scan_state, 1u);
    __syncthreads();
    const int tile = (int)s_tile;
    uint32_t* state = scan_state + kStateWords;
    const int base = tile * kTile + threadIdx.x * kCPT;
    uint32_t c[kCPT], o[kCPT], f[kCPT], k[kCPT], so = 0, sf = 0, sk = 0;
    if (base + kCPT - 1 < ncell) {
#pragma unroll
        for (int q = 0; q < kCPT / 4; ++q) {
            const uint4 v = *reinterpret_cast<const uint4*>(cell_cnt + base + 4 * q);
            c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < kCPT; ++j) c[j] = (base + j) < ncell ? cell_cnt[base + j] : 0;
    }
#pragma unroll
    for (int j = 0; j < kCPT; ++j) { cellTriple(c[j], T, o[j], f[j], k[j]); so += o[j]; sf += f[j]; sk += k[j]; }
    uint32_t to, tf, tk;
    uint32_t eo = blockExclusiveScan<256>(so, smem, &to);
    uint32_t ef = blockExclusiveScan<256>(sf, smem, &tf);
    uint32_t ek = blockExclusiveScan<256>(sk, smem, &tk);
    // ---- decoupled look-back (wave 0): prefix of all earlier tiles --------------------------------------------------
    if (threadIdx.x < kWave) {
        const int lane = threadIdx.x;
        uint32_t* mine = state + (size_t)tile * kStateWords;
        if (lane == 0) {
            const int slot = tile == 0 ? 5 : 2;                               // (tile 0's aggregate IS its inclusive prefix)
            mine[slot] = to; mine[slot + 1] = tf; mine[slot + 2] = tk;
            __hip_atomic_store(mine, tile == 0 ? kFlagInc : kFlagAgg, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t po = 0, pf = 0, pk = 0;
        int back = tile - 1;
        while (back >= 0) {
            const int t = back - lane;
            const uint32_t* st = state + (size_t)(t >= 0 ? t : 0) * kStateWords;
            uint32_t fl;
            // every lane polls its predecessor's flag until it has published something
            do { fl = t >= 0 ? __hip_atomic_load(st, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : kFlagInc; } while (__any(fl == 0u));
            // nearest predecessor holding an inclusive prefix: lanes below it contribute aggregates, it contributes the prefix
            const int first = __ffsll((long long)__ballot(fl == kFlagInc)) - 1;       // >= 0 when any (lanes with t < 0 count as "inclusive 0")
            const bool take = t >= 0 && (first < 0 || lane <= first);
            uint32_t a = 0, b_ = 0, d = 0;
            if (take) {                                                       // the slot the acquired flag value names (written before that flag)
                const int slot = fl == kFlagInc ? 5 : 2;
                a = __hip_atomic_load(st + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b_ = __hip_atomic_load(st + slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                d = __hip_atomic_load(st + slot + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            po += waveSum(a); pf += waveSum(b_); pk += waveSum(d);
            if (first >= 0) break;
            back -= kWave;
        }
        if (lane == 0) {
            s_pref[0] = po; s_pref[1] = pf; s_pref[2] = pk;
            if (tile > 0) {
                mine[5] = po + to; mine[6] = pf + tf; mine[7] = pk + tk;
                __hip_atomic_store(mine, kFlagInc, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();
    eo += s_pref[0]; ef += s_pref[1]; ek += s_pref[2];
    // ---- per-cell segment offsets, per-pillar records ---------------------------------------------------------------
    const uint32_t maxP = (uint32_t)p.max_pillars_num, maxN = (uint32_t)p.max_points_num_voxel_filter;
    const uint32_t gxy = (uint32_t)(p.gx * p.gy);
#pragma unroll
    for (int j = 0; j < kCPT; ++j) {
        const int cell = base + j;
        if (cell < ncell) {
            cell_seg[cell] = ef;
            if (o[j]) {
                // capacity guard the reference lacks: the pillar list is truncated at the first pillar that overflows
                // max_pillars_num or max_points_num_voxel_filter (ek + k is monotonic, so the valid pillars form a prefix)
                const bool valid = eo < maxP && ek + k[j] <= maxN;
                if (valid) {
                    pil_seg[eo] = ef; pil_full[eo] = f[j]; pil_ptoff[eo] = ek;
                    pcnt[eo] = k[j];                                              // :753
                    const uint32_t gxyz = gxy * (uint32_t)p.gz, fr = (uint32_t)cell / gxyz, cf = (uint32_t)cell % gxyz;
                    const uint32_t cz = cf / gxy, cyx = cf % gxy;
                    reinterpret_cast<uint4*>(coords)[eo] = make_uint4(fr, cz, cyx / (uint32_t)p.gx, cyx % (uint32_t)p.gx);       // :755-756 (batch = z = 0 there)
                } else if (eo == 0 || (eo - 1 < maxP && ek <= maxN)) {
                    *pillar_num = eo; *point_num = ek;                              // the FIRST pillar that does not fit: P and Nk are its offsets
                }
            }
        }
        eo += o[j]; ef += f[j]; ek += k[j];
    }
    // nothing overflowed: the last tile knows the totals
    if (tile == ntiles - 1 && threadIdx.x == 255 && eo <= maxP && ek <= maxN) { *pillar_num = eo; *point_num = ek; }
}

__global__ void __launch_bounds__(256)
p2f_scatter(const uint32_t* __restrict__ n_ptr, int max_points, int frames, const uint32_t* __restrict__ pt_cell,
            const uint32_t* __restrict__ pt_slot, const uint32_t* __restrict__ cell_seg, uint32_t* __restrict__ sorted_idx)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t fr = i / (uint32_t)max_points;
    if (fr >= (uint32_t)frames) return;
    uint32_t n = n_ptr[fr];
    if (n > (uint32_t)max_points) n = max_points;
    if (i - fr * (uint32_t)max_points >= n) return;
    uint32_t cell = pt_cell[i];
    if (cell == kNone) return;
    sorted_idx[cell_seg[cell] + pt_slot[i]] = i;
}

// one wavefront, one pillar (any point count)
__device__ __forceinline__ void p2fPillarWave(uint32_t pid, const float4* __restrict__ pts, const P2FParams& p,
           const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ pil_seg,
           const uint32_t* __restrict__ pil_full, const uint32_t* __restrict__ pil_ptoff,
           float* __restrict__ feat, uint32_t* __restrict__ pidx, uint32_t* sel_lds)
{
    const int lane = laneId();
    const uint32_t T = p.max_num_points_per_voxel;
    const uint32_t seg = pil_seg[pid], nfull = pil_full[pid], ptoff = pil_ptoff[pid];
    const uint32_t kept = nfull < T ? nfull : T;

    // lane s (< kept) ends up holding the point id with the s-th smallest index in the cell
    uint32_t sel = kNone;
    if (nfull <= (uint32_t)kWave) {
        uint32_t mine = lane < (int)nfull ? sorted_idx[seg + lane] : kNone;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nfull; ++j) rank += (__shfl(mine, (int)j, kWave) < mine) ? 1u : 0u;
        if (lane >= (int)nfull) rank = lane;              // idle lanes push onto themselves
        sel = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank * 4), (int)mine);
    } else if (nfull <= 256u) {
        // 65 .. 256 points (a few hundred cells next to the sensor): rank every id against all the others with wave shuffles and
        // drop it at its rank through LDS.  (The selection loop below costs `kept` dependent global-load rounds, ~50 us per cell:
        // it set the duration of the whole kernel.)
        uint32_t vals[4], rk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t idx = (uint32_t)lane + 64u * e;
            vals[e] = idx < nfull ? sorted_idx[seg + idx] : kNone;
            rk[e] = 0;
        }
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            if (64u * e2 >= nfull) break;
            const int cnt = (int)(nfull - 64u * e2 < 64u ? nfull - 64u * e2 : 64u);
            for (int j = 0; j < cnt; ++j) {
                const uint32_t o = __shfl(vals[e2], j, kWave);
#pragma unroll
                for (int e = 0; e < 4; ++e) rk[e] += o < vals[e] ? 1u : 0u;
            }
        }
        uint32_t* sm = sel_lds + (threadIdx.x / kWave) * kWave;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (vals[e] != kNone && rk[e] < (uint32_t)kWave) sm[rk[e]] = vals[e];
        __builtin_amdgcn_wave_barrier();
        if (lane < (int)kept) sel = sm[lane];
    } else {
        // more than 256 points in one cell: select the `kept` smallest ids one by one
        uint32_t last = 0; bool first = true;
        for (uint32_t s = 0; s < kept; ++s) {
            uint32_t m = kNone;
            for (uint32_t j = lane; j < nfull; j += kWave) {
                uint32_t v = sorted_idx[seg + j];
                if ((first || v > last) && v < m) m = v;
            }
            m = waveMinU(m);
            if (lane == (int)s) sel = m;
            last = m; first = false;
        }
    }

    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < (int)kept) q = pts[sel];
    // cluster mean: sequential fp32 sum in slot order, then divide by the int count (:813-824)
    float cx = 0.f, cy = 0.f, cz = 0.f;
    for (uint32_t s = 0; s < kept; ++s) {
        cx += __shfl(q.x, (int)s, kWave);
        cy += __shfl(q.y, (int)s, kWave);
        cz += __shfl(q.z, (int)s, kWave);
    }
    const int ni = (int)kept;
    cx = cx / ni; cy = cy / ni; cz = cz / ni;

    if (lane < (int)T) pidx[(size_t)pid * T + lane] = lane < (int)kept ? ptoff + lane : 0u;   // :829-830
    if (lane < (int)kept) {
        float* f = feat + (size_t)(ptoff + lane) * p.feature_num;
        int index_x = (int)floorf((q.x - p.min_x) / p.vx);                                     // :844-846
        int index_y = (int)floorf((q.y - p.min_y) / p.vy);
        int index_z = (int)floorf((q.z - p.min_z) / p.vz);
        // :849-851 -- the 0.5 literal is a double, so the bracket is evaluated in double
        float fx = (float)((double)q.x - ((index_x + 0.5) * (double)p.vx + (double)p.min_x));
        float fy = (float)((double)q.y - ((index_y + 0.5) * (double)p.vy + (double)p.min_y));
        float fz = (float)((double)q.z - ((index_z + 0.5) * (double)p.vz + (double)p.min_z));
        f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;                                        // :838-841
        f[4] = q.x - cx; f[5] = q.y - cy; f[6] = q.z - cz;                                     // :859-861
        f[7] = fx; f[8] = fy; f[9] = fz;                                                       // :854-856
    }
}


// the arithmetic of one point row (reference lines as in p2fPillarWave)
__device__ __forceinline__ void p2fWriteFeat(float* f, const float4 q, float cx, float cy, float cz, const P2FParams& p) {
    int index_x = (int)floorf((q.x - p.min_x) / p.vx);                                     // :844-846
    int index_y = (int)floorf((q.y - p.min_y) / p.vy);
    int index_z = (int)floorf((q.z - p.min_z) / p.vz);
    float fx = (float)((double)q.x - ((index_x + 0.5) * (double)p.vx + (double)p.min_x));    // :849-851 (double bracket)
    float fy = (float)((double)q.y - ((index_y + 0.5) * (double)p.vy + (double)p.min_y));
    float fz = (float)((double)q.z - ((index_z + 0.5) * (double)p.vz + (double)p.min_z));
    f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;                                        // :838-841
    f[4] = q.x - cx; f[5] = q.y - cy; f[6] = q.z - cz;                                     // :859-861
    f[7] = fx; f[8] = fy; f[9] = fz;                                                       // :854-856
}

// A wavefront owns FOUR consecutive pillars.  Pillars hold 4.8 points on average, so when all four have <= 16 points (the
// common case) each takes a 16-lane group: same ranking, same sequential fp32 sums, a quarter of the wavefronts.  A pillar
// with more points gets the whole wavefront first.
__global__ void __launch_bounds__(256)
p2f_pillar(const float4* __restrict__ pts, P2FParams p, const uint32_t* __restrict__ pillar_num,
           const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ pil_seg,
           const uint32_t* __restrict__ pil_full, const uint32_t* __restrict__ pil_ptoff,
           float* __restrict__ feat, uint32_t* __restrict__ pidx)
{
    __shared__ uint32_t sel_lds[256];
    const int lane = laneId(), sub = lane >> 4, sl = lane & 15;
    const uint32_t P = *pillar_num;
    // (pillars gw, gw + Q, gw + 2Q, gw + 3Q: dense cells come in runs of neighbours, a wavefront should not get four of them)
    const uint32_t Q = (P + 3u) / 4u, gw = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (gw >= Q) return;
    const uint32_t pid = gw + (uint32_t)sub * Q;
    const bool have = pid < P;
    const uint32_t nfull = have ? pil_full[pid] : 0u;
    // pillars with more than 16 points: the whole wavefront, one after the other (every lane still here)
    for (int k = 0; k < 4; ++k) {
        const uint32_t nk = __shfl(nfull, 16 * k, kWave);
        if (nk > 16u) p2fPillarWave(gw + (uint32_t)k * Q, pts, p, sorted_idx, pil_seg, pil_full, pil_ptoff, feat, pidx, sel_lds);
    }
    // loop bound of the group loops: the largest small pillar of this wavefront (wave-uniform)
    int nmax = 0;
    for (int k = 0; k < 4; ++k) { const int nk = (int)__shfl(nfull, 16 * k, kWave); if (nk <= 16 && nk > nmax) nmax = nk; }
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    if (!have || nfull > 16u) return;                                           // (whole 16-lane groups: the group shuffles below stay inside live groups)
    const uint32_t T = p.max_num_points_per_voxel;
    const uint32_t seg = pil_seg[pid], ptoff = pil_ptoff[pid];
    const uint32_t kept = nfull < T ? nfull : T;
    // lane s (< kept) of the group ends up holding the point id with the s-th smallest index in the cell
    const uint32_t mine = sl < (int)nfull ? sorted_idx[seg + sl] : kNone;
    uint32_t rank = 0;
    for (int j = 0; j < nmax; ++j) rank += (__shfl(mine, j, 16) < mine) ? 1u : 0u;
    if (sl >= (int)nfull) rank = (uint32_t)sl;                                  // idle lanes push onto themselves
    const uint32_t sel = (uint32_t)__builtin_amdgcn_ds_permute((int)(((uint32_t)(sub << 4) + rank) * 4), (int)mine);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sl < (int)kept) q = pts[sel];
    float cx = 0.f, cy = 0.f, cz = 0.f;                                         // sequential fp32 sum in slot order (:813-824)
    for (int s_ = 0; s_ < nmax; ++s_) {
        const float vx = __shfl(q.x, s_, 16), vy = __shfl(q.y, s_, 16), vz = __shfl(q.z, s_, 16);
        if (s_ < (int)kept) { cx += vx; cy += vy; cz += vz; }
    }
    const int ni = (int)kept;
    cx = cx / ni; cy = cy / ni; cz = cz / ni;
    for (uint32_t e = (uint32_t)sl; e < T; e += 16u) pidx[(size_t)pid * T + e] = e < kept ? ptoff + e : 0u;   // :829-830
    if (sl < (int)kept) p2fWriteFeat(feat + (size_t)(ptoff + sl) * p.feature_num, q, cx, cy, cz, p);
}

// frames > 1 (optional field "frames", not in the reference): SEVERAL frames per enqueue with their rows CONCATENATED -- the layout the
// fused backbone ops want (rows = sum of the frames' pillars, one launch per layer for all of them), unlike the per-frame slabs a batched
// enqueue of the C ABI gives.  Input 0 is [1, frames * max_points_num, 4] (frame f owns rows f * max_points_num ...), input 1 holds
// `frames` counts; the frame index is one more, slowest, grid dimension: pillars ascend by (frame, cell), coords = (frame, z, y, x) -- the
// batch slot the reference's coordinate layout already has (:755) -- and max_pillars_num / max_points_num_voxel_filter bound the totals.
class Points2FeaturesPlugin : public Plugin {
public:
    P2FParams p_;
    explicit Points2FeaturesPlugin(const P2FParams& p) : p_(p) {}
    const char* type() const override { return "Points2FeaturesPlugin"; }
    int nbOutputs() const override { return 6; }                                               // :1018-1021
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {            // :133-189
        int b = in[0].d[0];
        switch (i) {
            case 0: *out = dims3(b, p_.max_points_num_voxel_filter, p_.feature_num); return 0;
            case 1: *out = dims3(b, p_.max_pillars_num, p_.max_num_points_per_voxel); return 0;
            case 2: *out = dims3(b, p_.max_pillars_num, 4); return 0;
            case 3: *out = dims3(b, p_.max_pillars_num, 1); return 0;
            case 4: case 5: *out = dims1(b); return 0;
        }
        return -1;
    }
    int outputType(int i, const int32_t* t, int) const override { return i == 0 ? t[0] : t[1]; }   // :1000-1006
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {   // :203-255
        if (io[pos].format != DSVT_FORMAT_LINEAR) return false;
        if (pos == 0 || pos == 2) return io[pos].type == DSVT_FLOAT;
        return pos >= 1 && pos <= 7 && io[pos].type == DSVT_INT32;
    }
    int ncell() const { return p_.gx * p_.gy * p_.gz * p_.frames; }
    int rowsAll() const { return p_.max_points_num * p_.frames; }
    // cell histogram followed by the scan's ticket + tile states (one memset covers both); 16-byte aligned rows
    size_t cntWords() const { return ((size_t)ncell() + 3) / 4 * 4; }
    size_t headBytes() const { return sizeof(uint32_t) * (cntWords() + (size_t)kStateWords * (1 + (size_t)ntiles())); }
    int ntiles() const { return cdiv(ncell(), kTile); }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override {
        size_t s = 0;
        s += alignUp(headBytes()) + alignUp(sizeof(uint32_t) * ncell());   // cell_cnt + scan state, cell_seg
        s += 3 * alignUp(sizeof(uint32_t) * rowsAll());               // pt_cell, pt_slot, sorted_idx
        s += 3 * alignUp(sizeof(uint32_t) * p_.max_pillars_num);      // pil_seg, pil_full, pil_ptoff
        return s;                                                     // ~4.6 MB at 180k caps (reference: 176.8 MB, :262-277)
    }
    int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc*, const void* const* inputs,
                void* const* outputs, void* workspace, hipStream_t stream) override {
        if (inDesc && inDesc[0].dims.d[0] != 1) return -2;            // batch 1 only, as the reference
        const float4* pts = static_cast<const float4*>(inputs[0]);
        const uint32_t* n_ptr = static_cast<const uint32_t*>(inputs[1]);
        float* feat = static_cast<float*>(outputs[0]);
        uint32_t* pidx = static_cast<uint32_t*>(outputs[1]);
        uint32_t* coords = static_cast<uint32_t*>(outputs[2]);
        uint32_t* pcnt = static_cast<uint32_t*>(outputs[3]);
        uint32_t* pillar_num = static_cast<uint32_t*>(outputs[4]);
        uint32_t* point_num = static_cast<uint32_t*>(outputs[5]);
        WsCarver ws(workspace);
        char* head = ws.take<char>(headBytes());
        uint32_t* cell_cnt = reinterpret_cast<uint32_t*>(head);
        uint32_t* scan_state = reinterpret_cast<uint32_t*>(head + sizeof(uint32_t) * cntWords());
        uint32_t* cell_seg = ws.take<uint32_t>(ncell());
        uint32_t* pt_cell = ws.take<uint32_t>(rowsAll());
        uint32_t* pt_slot = ws.take<uint32_t>(rowsAll());
        uint32_t* sorted_idx = ws.take<uint32_t>(rowsAll());
        uint32_t* pil_seg = ws.take<uint32_t>(p_.max_pillars_num);
        uint32_t* pil_full = ws.take<uint32_t>(p_.max_pillars_num);
        uint32_t* pil_ptoff = ws.take<uint32_t>(p_.max_pillars_num);

        DSVT_CHECK(hipMemsetAsync(head, 0, headBytes(), stream));
        if (zeroFill) {   // reference zero-fills every output each call (:928-937)
            DSVT_CHECK(hipMemsetAsync(feat, 0, sizeof(float) * (size_t)p_.max_points_num_voxel_filter * p_.feature_num, stream));
            DSVT_CHECK(hipMemsetAsync(pidx, 0, sizeof(uint32_t) * (size_t)p_.max_pillars_num * p_.max_num_points_per_voxel, stream));
            DSVT_CHECK(hipMemsetAsync(coords, 0, sizeof(uint32_t) * (size_t)p_.max_pillars_num * 4, stream));
            DSVT_CHECK(hipMemsetAsync(pcnt, 0, sizeof(uint32_t) * (size_t)p_.max_pillars_num, stream));
        }
        const int nt = ntiles();
        hipLaunchKernelGGL(p2f_count, dim3(cdiv(rowsAll(), 256 * kPPT)), dim3(256), 0, stream, pts, n_ptr, p_, cell_cnt, pt_cell, pt_slot);
        hipLaunchKernelGGL(p2f_scan, dim3(nt), dim3(256), 0, stream, cell_cnt, ncell(), p_, scan_state, nt, cell_seg,
                           pil_seg, pil_full, pil_ptoff, coords, pcnt, pillar_num, point_num);
        hipLaunchKernelGGL(p2f_scatter, dim3(cdiv(rowsAll(), 256)), dim3(256), 0, stream, n_ptr, p_.max_points_num, p_.frames,
                           pt_cell, pt_slot, cell_seg, sorted_idx);
        hipLaunchKernelGGL(p2f_pillar, dim3(cdiv(p_.max_pillars_num, 16)), dim3(256), 0, stream, pts, p_, pillar_num,
                           sorted_idx, pil_seg, pil_full, pil_ptoff, feat, pidx);
        return lastError();
    }
    // the reference's 18 words (:1033-1036); one more only for a multi-frame plugin
    size_t serializationSize() const override { return 9 * sizeof(float) + (p_.frames > 1 ? 10 : 9) * sizeof(int); }
    void serialize(void* buffer) const override {                                                  // :1038-1060
        char* d = static_cast<char*>(buffer);
        wr<int>(d, p_.max_points_num); wr<int>(d, p_.max_points_num_voxel_filter); wr<int>(d, p_.max_pillars_num);
        wr<int>(d, p_.point_feature_num); wr<int>(d, p_.feature_num); wr<int>(d, p_.max_num_points_per_voxel);
        wr<float>(d, p_.min_x); wr<float>(d, p_.max_x); wr<float>(d, p_.min_y); wr<float>(d, p_.max_y);
        wr<float>(d, p_.min_z); wr<float>(d, p_.max_z); wr<float>(d, p_.vx); wr<float>(d, p_.vy); wr<float>(d, p_.vz);
        wr<int>(d, p_.gx); wr<int>(d, p_.gy); wr<int>(d, p_.gz);
        if (p_.frames > 1) wr<int>(d, p_.frames);
    }
    Plugin* clone() const override { return new Points2FeaturesPlugin(p_); }
};

static bool validP2F(const P2FParams& p) {
    return p.max_points_num > 0 && p.max_points_num_voxel_filter > 0 && p.max_pillars_num > 0 &&
           p.point_feature_num == 4 && p.feature_num == 10 &&
           p.max_num_points_per_voxel > 0 && p.max_num_points_per_voxel <= kWave &&
           p.gx > 0 && p.gy > 0 && p.gz > 0 && p.frames >= 1 && (long)p.gx * p.gy * p.gz * p.frames < (1l << 30) && p.vx > 0 && p.vy > 0 && p.vz > 0 &&
           (long)p.max_points_num * p.frames < (1l << 31);    // the scan's running sums and the row indices are 32-bit
}

static Plugin* p2fCreate(const DsvtPluginFieldCollection* fc) {                                    // :1113-1195
    P2FParams p{};
    p.max_points_num = fieldInt(fc, "max_points_num");
    p.max_points_num_voxel_filter = fieldInt(fc, "max_points_num_voxel_filter");
    p.max_pillars_num = fieldInt(fc, "max_pillars_num");
    p.point_feature_num = fieldInt(fc, "point_feature_num");
    p.feature_num = fieldInt(fc, "feature_num");
    p.max_num_points_per_voxel = fieldInt(fc, "max_num_points_per_voxel");
    float r[6], v[3]; int g[3];
    fieldFloats(fc, "point_cloud_range", r, 6);   // (xmin,ymin,zmin,xmax,ymax,zmax) plugin_helper.h:33-38
    fieldFloats(fc, "voxel_size", v, 3);
    fieldInts(fc, "grid_size", g, 3);
    p.min_x = r[0]; p.max_x = r[3]; p.min_y = r[1]; p.max_y = r[4]; p.min_z = r[2]; p.max_z = r[5];
    p.vx = v[0]; p.vy = v[1]; p.vz = v[2]; p.gx = g[0]; p.gy = g[1]; p.gz = g[2];
    p.frames = fieldInt(fc, "frames", 1);       // not a reference field: see the class comment
    return validP2F(p) ? new Points2FeaturesPlugin(p) : nullptr;
}
static Plugin* p2fDeserialize(const void* data, size_t len) {                                      // ctor :60-84
    if (len < 9 * sizeof(float) + 9 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    P2FParams p{};
    p.max_points_num = rd<int>(d); p.max_points_num_voxel_filter = rd<int>(d); p.max_pillars_num = rd<int>(d);
    p.point_feature_num = rd<int>(d); p.feature_num = rd<int>(d); p.max_num_points_per_voxel = rd<int>(d);
    p.min_x = rd<float>(d); p.max_x = rd<float>(d); p.min_y = rd<float>(d); p.max_y = rd<float>(d);
    p.min_z = rd<float>(d); p.max_z = rd<float>(d); p.vx = rd<float>(d); p.vy = rd<float>(d); p.vz = rd<float>(d);
    p.gx = rd<int>(d); p.gy = rd<int>(d); p.gz = rd<int>(d);
    p.frames = len >= 9 * sizeof(float) + 10 * sizeof(int) ? rd<int>(d) : 1;
    return validP2F(p) ? new Points2FeaturesPlugin(p) : nullptr;
}

static Creator g_p2fCreator{"Points2FeaturesPlugin",
    {{"max_points_num", DSVT_FIELD_INT32}, {"max_points_num_voxel_filter", DSVT_FIELD_INT32},
     {"max_pillars_num", DSVT_FIELD_INT32}, {"point_feature_num", DSVT_FIELD_INT32},
     {"feature_num", DSVT_FIELD_INT32}, {"max_num_points_per_voxel", DSVT_FIELD_INT32},
     {"point_cloud_range", DSVT_FIELD_FLOAT32}, {"voxel_size", DSVT_FIELD_FLOAT32},
     {"grid_size", DSVT_FIELD_INT32}},                                                             // :1084-1092
    p2fCreate, p2fDeserialize, {}, {}};
static Registrar g_p2fReg(&g_p2fCreator);

}  // namespace dsvt
