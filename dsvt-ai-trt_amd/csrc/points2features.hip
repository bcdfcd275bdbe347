// points2features.hip -- Points2FeaturesPlugin for gfx950.
//
// Replaces plugins/src/points2Features.cu:669-990 (generateVoxels_random_kernel,
// generateBaseFeatures_kernel, generateFeatures_kernel).  Same tensors, different machine:
// the reference scatters every point into a dense 468x468x48x4 float scratch (168 MB, zeroed
// every frame) with racy atomic slot order.  Here (round 3; rounds 1-2 spent one returning GLOBAL atomic per point on a cell
// histogram -- 16 G atomics/s, 44 us for the 720k points of four frames whatever else the kernel did -- plus a chained scan over all
// 876k cells and a scatter) the points are bucketed by an MSD pass and everything per cell happens in LDS:
//   1. p2f_partition  a block of 2048 points: range filter, cell key, bin = key / 2048; LDS histogram over the bins (the LDS
//                     atomic's return value is the point's rank inside its (block, bin) piece), LDS scan, and the block writes
//                     its points grouped by bin into ITS OWN 2048 slots of the partitioned arrays (point, key, row index) plus
//                     one row of piece offsets.  No global atomic at all; runs of ~5 points per piece.
//   2. p2f_bins       one workgroup per bin (= 2048 consecutive cells): walks the bin's pieces (one per block, found in the
//                     offset table; the table also gives the bin's first slot in the sorted array, sum over blocks of
//                     offset[block][bin]), histograms the cells in LDS, scans them (occupied -> pillar ordinal, point counts ->
//                     segment offsets, kept counts -> compact offsets), gets the two prefixes that cross bins (pillars, kept
//                     points) by decoupled look-back between the bins, writes the pillar records (coordinates, counts, segment)
//                     and drops every point's slot into its cell's segment of the sorted array;
//   3. p2f_pillar     one 16-lane group per pillar (a whole wavefront for pillars with more than 16 points) ranks the segment
//                     by row index (= input order), keeps the first T, sums the cluster mean sequentially in slot order like
//                     the reference, and writes the 10-d features / point-id rows.
// Canonical order: pillars ascending by y*GX+x, points in input order, first T.
// grid_size z > 1 (BASELINE configs[4], a 3-D voxel grid; NOT in the reference, whose voxel z index is forced to 0,
// points2Features.cu:689-690,755): the cell key gains the z index computed like x and y (the reference computes it the same way for
// the centre offset, :846), key = (z*GY + y)*GX + x, voxels ascending by that key, coords = (0, z, y, x).
//
// Compiled with -ffp-contract=off: cell indices and features follow the reference's
// expression order exactly (fp32 subtract, IEEE divide, floorf; the centre offset in double).
#include "plugin_base.h"
#include "device_utils.h"
#include <cstdio>
#include <string>

namespace dsvt {

struct P2FParams {
    int max_points_num, max_points_num_voxel_filter, max_pillars_num;
    int point_feature_num, feature_num, max_num_points_per_voxel;
    float min_x, max_x, min_y, max_y, min_z, max_z;
    float vx, vy, vz;
    int gx, gy, gz;
    int frames;       // >= 1: the points tensor holds `frames` slabs of max_points_num rows with one count each (see the plugin class)
    int pid_slots;    // optional field "point_id_slots" (default max_num_points_per_voxel = the reference's table): how many slots of a pillar's row of the
                      // [P, T] point-id table (output 1) are written.  The rows of a pillar are consecutive, so a consumer that walks them needs slot 0 (the
                      // pillar's first row) and the count: the fused pillar feature net reads nothing else, and the other 47 slots were 26 MB of stores per
                      // four-frame launch that no kernel of the frame pipeline ever read (round 4 PMC: 155 MB of p2f_pillar's 198 MB)
};

constexpr uint32_t kNone = 0xffffffffu;
#ifdef DSVT_ABLATE
#define P2F_DBG(bits) ((dbg) & (bits))
#else
#define P2F_DBG(bits) 0
#endif
constexpr int kBlk = 2048;          // points of a partition block (eight per thread)
constexpr int kBinCells = 2048;     // cells whose tables one pass of p2f_bins keeps in LDS
constexpr int kMaxBins = 8192;      // bins p2f_partition histograms in LDS (32 KB); beyond, a bin holds several 2048-cell sub-ranges
constexpr int kBT = 1024;           // threads of a p2f_bins workgroup

// launch geometry, derived from the plugin's fields on the host
struct P2FPlan {
    int bpf;          // partition blocks per frame
    int nblk;         // partition blocks (frames * bpf)
    int bin_shift;    // log2(cells per bin)
    int nsub;         // 2048-cell sub-ranges per bin (1 unless the grid has more than kMaxBins * 2048 cells)
    int nbins;
    int ncell;        // cells of all frames
    int dbg;          // timing ablations of p2f_pillar (wrong results; compiled only into the ablate build, tools/trace_p2f.sh): 64 no feature stores, 128 no point-id stores, 256 no whole-wavefront passes
    unsigned long long* trace;      // debugging (tools/, the ablate build): wall-clock stamps of p2f_bins' phases, 8 per bin, or nullptr
};

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

// ---- 1. partition ------------------------------------------------------------------------------------------------------
// tab: [nblk][nbins + 1] uint16 -- exclusive offsets of the block's pieces (entry nbins = the block's in-range points)
__global__ void __launch_bounds__(256)
p2f_partition(const float4* __restrict__ pts, const uint32_t* __restrict__ n_ptr, P2FParams p, P2FPlan pl,
              uint16_t* __restrict__ tab, float4* __restrict__ part_pts, uint32_t* __restrict__ part_key, uint32_t* __restrict__ part_idx,
              uint32_t* __restrict__ scan_state, int state_words)
{
    extern __shared__ uint32_t hist[];               // nbins + 1
    __shared__ uint32_t smem[256 / kWave + 1];
    __shared__ float4 out_pts[kBlk]; __shared__ uint32_t out_key[kBlk], out_idx[kBlk];
    const int tid = threadIdx.x, g = blockIdx.x;
    const uint32_t fr = (uint32_t)(g / pl.bpf), row0 = (uint32_t)(g - (int)fr * pl.bpf) * kBlk;
    for (int w = g * 256 + tid; w < state_words; w += gridDim.x * 256) scan_state[w] = 0;     // p2f_bins' ticket + look-back state
    for (int i = tid; i <= pl.nbins; i += 256) hist[i] = 0;
    uint32_t n = n_ptr[fr];
    if (n > (uint32_t)p.max_points_num) n = p.max_points_num;
    constexpr int PPT = kBlk / 256;
    float4 q[PPT]; uint32_t cell[PPT], rk[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const uint32_t r = row0 + 256u * k + tid;
        cell[k] = r < n ? 0u : kNone;
        q[k] = r < n ? pts[(size_t)fr * p.max_points_num + r] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        if (cell[k] != kNone) cell[k] = p2fCellOf(q[k], p, fr);
        rk[k] = cell[k] != kNone ? atomicAdd(&hist[cell[k] >> pl.bin_shift], 1u) : 0u;       // LDS: rank inside the (block, bin) piece
    }
    __syncthreads();
    // exclusive scan of the bin counts, in place: thread t owns a contiguous run of bins
    const int per = (pl.nbins + 256) / 256, b0 = tid * per, b1 = b0 + per < pl.nbins + 1 ? b0 + per : pl.nbins + 1;
    uint32_t sum = 0;
    for (int b = b0; b < b1; ++b) sum += hist[b];
    uint32_t tot;
    uint32_t run = blockExclusiveScan<256>(sum, smem, &tot);
    uint16_t* row = tab + (size_t)g * (pl.nbins + 1);
    for (int b = b0; b < b1; ++b) { const uint32_t c = hist[b]; hist[b] = run; row[b] = (uint16_t)run; run += c; }
    __syncthreads();
    // the block's 2048 slots are one contiguous range of the partitioned arrays: the points take their places in LDS and leave with
    // lane-contiguous stores (three lane-scattered stores per point cost this kernel 15 us; see p2f_pillar)
#pragma unroll
    for (int k = 0; k < PPT; ++k)
        if (cell[k] != kNone) {
            const uint32_t pos = hist[cell[k] >> pl.bin_shift] + rk[k];
            out_pts[pos] = q[k]; out_key[pos] = cell[k];
            out_idx[pos] = fr * (uint32_t)p.max_points_num + row0 + 256u * k + tid;
        }
    __syncthreads();
    const uint32_t total = hist[pl.nbins];
    const size_t base = (size_t)g * kBlk;
    for (uint32_t i = tid; i < total; i += 256) { part_pts[base + i] = out_pts[i]; part_key[base + i] = out_key[i]; part_idx[base + i] = out_idx[i]; }
}

__device__ __forceinline__ void cellTriple(uint32_t c, uint32_t T, uint32_t& occ, uint32_t& full, uint32_t& kept) {
    occ = c > 0 ? 1u : 0u; full = c; kept = c < T ? c : T;        // :746-748
}
// the arithmetic of one point row
__device__ __forceinline__ void p2fWriteFeat(float* f, const float4 q, float cx, float cy, float cz, const P2FParams& p) {
    int index_x = (int)floorf((q.x - p.min_x) / p.vx);                                     // :844-846
    int index_y = (int)floorf((q.y - p.min_y) / p.vy);
    int index_z = (int)floorf((q.z - p.min_z) / p.vz);
    float fx = (float)((double)q.x - ((index_x + 0.5) * (double)p.vx + (double)p.min_x));    // :849-851 (the 0.5 literal is a double: the bracket is evaluated in double)
    float fy = (float)((double)q.y - ((index_y + 0.5) * (double)p.vy + (double)p.min_y));
    float fz = (float)((double)q.z - ((index_z + 0.5) * (double)p.vz + (double)p.min_z));
    f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;                                        // :838-841
    f[4] = q.x - cx; f[5] = q.y - cy; f[6] = q.z - cz;                                     // :859-861
    f[7] = fx; f[8] = fy; f[9] = fz;                                                       // :854-856
}

// one wavefront, one pillar (any point count).  srt[seg .. seg + nfull): slots (positions in the partitioned arrays) of the cell's points
__device__ __forceinline__ void p2fPillarWave(uint32_t pid, uint32_t seg, uint32_t nfull, uint32_t ptoff, const P2FParams& p,
           const uint32_t* __restrict__ srt, const float4* __restrict__ part_pts, const uint32_t* __restrict__ part_idx,
           float* __restrict__ feat, uint32_t* __restrict__ pidx, uint32_t* sel_lds /* 64 words of this wave */)
{
    const int lane = laneId();
    const uint32_t T = p.max_num_points_per_voxel;
    const uint32_t kept = nfull < T ? nfull : T;
    // lane s (< kept) ends up holding the slot of the point with the s-th smallest row index in the cell
    uint32_t sel = kNone;
    if (nfull <= (uint32_t)kWave) {
        const uint32_t slot = lane < (int)nfull ? srt[seg + lane] : kNone;
        const uint32_t mine = lane < (int)nfull ? part_idx[slot] : kNone;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nfull; ++j) rank += ((uint32_t)__builtin_amdgcn_readlane((int)mine, (int)j) < mine) ? 1u : 0u;   // (j is wave-uniform: v_readlane, no LDS)
        if (lane >= (int)nfull) rank = lane;              // idle lanes push onto themselves
        sel = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank * 4), (int)slot);
    } else if (nfull <= 256u) {
        // 65 .. 256 points (a few hundred cells next to the sensor): rank every row index against all the others with wave shuffles
        // and drop its slot at its rank through LDS
        uint32_t vals[4], sl[4], rk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t i = (uint32_t)lane + 64u * e;
            sl[e] = i < nfull ? srt[seg + i] : kNone;
            vals[e] = i < nfull ? part_idx[sl[e]] : kNone;
            rk[e] = 0;
        }
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            if (64u * e2 >= nfull) break;
            const int cnt = (int)(nfull - 64u * e2 < 64u ? nfull - 64u * e2 : 64u);
            for (int j = 0; j < cnt; ++j) {
                const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)vals[e2], j);
#pragma unroll
                for (int e = 0; e < 4; ++e) rk[e] += o < vals[e] ? 1u : 0u;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (vals[e] != kNone && rk[e] < (uint32_t)kWave) sel_lds[rk[e]] = sl[e];
        __builtin_amdgcn_wave_barrier();
        if (lane < (int)kept) sel = sel_lds[lane];
        __builtin_amdgcn_wave_barrier();
    } else {
        // more than 256 points in one cell: select the `kept` smallest row indices one by one ((index << 32) | slot: the index decides)
        unsigned long long last = 0ull; bool first = true;
        for (uint32_t s = 0; s < kept; ++s) {
            unsigned long long m = ~0ull;
            for (uint32_t j = lane; j < nfull; j += kWave) {
                const uint32_t slot = srt[seg + j];
                const unsigned long long v = ((unsigned long long)part_idx[slot] << 32) | slot;
                if ((first || v > last) && v < m) m = v;
            }
#pragma unroll
            for (int o = kWave / 2; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o, kWave); m = t < m ? t : m; }
            if (lane == (int)s) sel = (uint32_t)m;
            last = m; first = false;
        }
    }
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < (int)kept) q = part_pts[sel];
    // cluster mean: sequential fp32 sum in slot order, then divide by the int count (:813-824)
    float cx = 0.f, cy = 0.f, cz = 0.f;
    for (uint32_t s = 0; s < kept; ++s) {
        cx += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.x), (int)s));
        cy += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.y), (int)s));
        cz += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.z), (int)s));
    }
    const int ni = (int)kept;
    cx = cx / ni; cy = cy / ni; cz = cz / ni;
    if (lane < p.pid_slots) pidx[(size_t)pid * T + lane] = lane < (int)kept ? ptoff + lane : 0u;   // :829-830 (pid_slots = T unless the caller asked for fewer)
    if (lane < (int)kept) p2fWriteFeat(feat + (size_t)(ptoff + lane) * p.feature_num, q, cx, cy, cz, p);
}

// ---- 2. bins ---------------------------------------------------------------------------------------------------------------
// Two running sums cross the bins: occupied cells (-> pillar id, < 2^30) and kept point counts (-> compact point offset, < 2^31).  Both
// travel in ONE 64-bit word behind a 2-bit flag (1 = the bin's aggregate, 2 = its inclusive prefix): flag << 62 | occupied << 32 | kept,
// stored and polled with relaxed agent-scope atomics -- value and flag arrive together, so no release / acquire pair is needed (on this
// part an agent-scope release is an L2 write-back and an acquire an invalidate: ~4 us per publish, measured with the two-slot layout that
// round 3 started with; the third sum that layout made room for, the full point count, no longer crosses bins).  Bins take their index
// from a ticket counter, so a bin only ever waits for bins that are already running.  scan_state (64-bit words): [0] ticket (low half),
// [1 + b] the word of bin b -- zeroed by p2f_partition.
// A bin's time is its memory round trips and the work of its busiest thread (the bins that cross the sensor hold 7 x the average), so
// (1) the slots of the bin's entries are listed ONCE, in LDS, by the threads that own the pieces (a few LDS writes each), and every pass
// walks that flat list -- no search per entry; (2) a thread takes FOUR entries at a time and issues their loads together; (3) the bin
// publishes its aggregate right after the histogram and looks back only after the placement pass, which needs no prefix from other
// bins: the wait for the slowest predecessor is hidden behind the bin's own work.  Bins with more than kCap points, and launches with
// more than kBT partition blocks, take a slower path (binary search per entry) for what the list does not hold.
constexpr unsigned long long kFlagAgg = 1ull << 62, kFlagInc = 2ull << 62, kFlagMask = 3ull << 62;
constexpr int kCap = 16384;        // entries of a bin whose slots p2f_bins lists in LDS (64 KB; the 180k-point bench clouds put 12.1k - 12.5k into the bins that cross the sensor)

__global__ void __launch_bounds__(kBT)
p2f_bins(P2FParams p, P2FPlan pl, const uint16_t* __restrict__ tab, const uint32_t* __restrict__ part_key,
         uint32_t* __restrict__ srt, uint32_t* __restrict__ scan_state, uint4* __restrict__ pil_rec,
         uint32_t* __restrict__ coords, uint32_t* __restrict__ pcnt, uint32_t* __restrict__ pillar_num, uint32_t* __restrict__ point_num)
{
    __shared__ uint32_t cnt[kBinCells], segx[kBinCells], cur[kBinCells];
    __shared__ uint32_t pst[kBT]; __shared__ uint16_t pof[kBT];
    __shared__ uint32_t slots[kCap];                 // slots of the bin's first kCap entries (pieces of the first kBT partition blocks)
    __shared__ uint32_t smem[3 * (kBT / kWave + 1)];
    __shared__ uint32_t s_bin, s_pref[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t T = p.max_num_points_per_voxel;
    const unsigned long long t_start = pl.trace ? wall_clock64() : 0ull;
    if (tid == 0) s_bin = atomicAdd(scan_state, 1u);
    __syncthreads();
    const int b = (int)s_bin;
    int nmark = 0;
    auto mark = [&]() { if (pl.trace && tid == 0 && nmark < 16) pl.trace[(size_t)b * 16 + nmark] = nmark == 0 ? t_start : wall_clock64(); ++nmark; };
    mark(); mark();                                  // [0] start, [1] ticket
    unsigned long long* state = reinterpret_cast<unsigned long long*>(scan_state) + 1;
    const size_t trow = (size_t)pl.nbins + 1;
    const uint32_t cell_bin0 = (uint32_t)b << pl.bin_shift;

    // ---- the list: slots of the bin's entries, piece after piece (the pieces of the first kBT blocks, the first kCap entries) ---------
    uint32_t binbase;                                // first position of the bin in the sorted arrays = sum over blocks of offset[block][bin]
    uint32_t nfast;                                  // (uniform) entries in the list
    bool spill = pl.nblk > kBT;                      // (uniform) entries exist that the list does not hold
    {
        uint32_t o = 0, sz = 0;
        if (tid < pl.nblk) { o = tab[(size_t)tid * trow + b]; sz = (uint32_t)tab[(size_t)tid * trow + b + 1] - o; }
        uint32_t osum = o;
        for (int g = kBT + tid; g < pl.nblk; g += kBT) osum += tab[(size_t)g * trow + b];
        uint32_t v[2] = {sz, osum}, tot[2];
        blockExclusiveScanK<kBT, 2>(v, smem, tot);
        const uint32_t ex = v[0];
        binbase = tot[1];
        nfast = tot[0] < (uint32_t)kCap ? tot[0] : (uint32_t)kCap;
        if (tot[0] > (uint32_t)kCap) spill = true;
        const uint32_t base = (uint32_t)tid * kBlk + o;
        for (uint32_t k = 0; k < sz && ex + k < (uint32_t)kCap; ++k) slots[ex + k] = base + k;
        __syncthreads();
    }
    // every entry of the bin: fn(slot in the partitioned arrays, key) -- the list, then (slow: binary search per entry) what it does not hold
    auto forEntries = [&](auto&& fn) {
        for (uint32_t e0 = tid; e0 < nfast; e0 += 4 * kBT) {
            uint32_t sl[4], ky[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t e = e0 + (uint32_t)u * kBT; sl[u] = e < nfast ? slots[e] : kNone; }
#pragma unroll
            for (int u = 0; u < 4; ++u) ky[u] = sl[u] != kNone ? part_key[sl[u]] : kNone;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (ky[u] != kNone) fn(sl[u], ky[u]);
        }
        if (!spill) return;
        for (int g0 = 0; g0 < pl.nblk; g0 += kBT) {
            const int g = g0 + tid;
            uint32_t o = 0, sz = 0;
            if (g < pl.nblk) { o = tab[(size_t)g * trow + b]; sz = (uint32_t)tab[(size_t)g * trow + b + 1] - o; }
            uint32_t tot;
            const uint32_t ex = blockExclusiveScan<kBT>(sz, smem, &tot);
            pst[tid] = ex; pof[tid] = (uint16_t)o;
            __syncthreads();
            for (uint32_t e = tid + (g0 == 0 ? (uint32_t)kCap : 0u); e < tot; e += kBT) {
                int lo = 0, hi = kBT - 1;            // the LAST piece whose start is <= e (empty pieces share their successor's start)
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pst[mid] <= e) lo = mid; else hi = mid - 1; }
                const uint32_t slot = (uint32_t)(g0 + lo) * kBlk + pof[lo] + (e - pst[lo]);
                fn(slot, part_key[slot]);
            }
            __syncthreads();
        }
    };
    // The common case -- one sub-range, every entry in the list: a thread keeps its entries' slots in REGISTERS from the histogram to
    // the permutation and loads their keys four at a time (a bin's time is the round trips of its busiest thread, and the bins that
    // cross the sensor hold seven times the average).
    const bool fast = pl.nsub == 1 && !spill;        // (uniform)
    constexpr int EPT = kCap / kBT;
    uint32_t sl[EPT];
    auto histSub = [&](int s) {                      // cnt[] = points per cell of sub-range s (general path)
        for (int i = tid; i < kBinCells; i += kBT) cnt[i] = 0;
        __syncthreads();
        const uint32_t c0 = cell_bin0 + (uint32_t)s * kBinCells;
        forEntries([&](uint32_t, uint32_t key) { const uint32_t cl = key - c0; if (cl < (uint32_t)kBinCells) atomicAdd(&cnt[cl], 1u); });
        __syncthreads();
    };
    constexpr int CPT = kBinCells / kBT;             // consecutive cells per thread of the scans
    auto tripleSums = [&](uint32_t& so, uint32_t& sf, uint32_t& sk) {
        so = sf = sk = 0;
#pragma unroll
        for (int j = 0; j < CPT; ++j) { uint32_t o, f, k; cellTriple(cnt[tid * CPT + j], T, o, f, k); so += o; sf += f; sk += k; }
    };

    // ---- phase I: the bin's aggregate (occupied cells, kept points), published at once ----------------------------------------
    uint32_t agg_o = 0, agg_k = 0;
    if (fast) {
        for (int i = tid; i < kBinCells; i += kBT) cnt[i] = 0;
#pragma unroll
        for (int i = 0; i < EPT; ++i) { const uint32_t e = tid + (uint32_t)i * kBT; sl[i] = e < nfast ? slots[e] : kNone; }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < EPT; h += 4) {           // four key loads in flight
            uint32_t ky[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ky[u] = sl[h + u] != kNone ? part_key[sl[h + u]] : kNone;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (ky[u] != kNone) atomicAdd(&cnt[ky[u] - cell_bin0], 1u);
            __builtin_amdgcn_sched_barrier(0);       // (the next batch's loads stay behind this one's atomics: registers)
        }
        __syncthreads();
    }
    uint32_t pre[3] = {0, 0, 0}, tot3[3] = {0, 0, 0};   // (one sub-range: phase II reuses these exclusive prefixes and totals)
    for (int s = 0; s < pl.nsub; ++s) {
        if (!fast) histSub(s);
        tripleSums(pre[0], pre[1], pre[2]);
        blockExclusiveScanK<kBT, 3>(pre, smem, tot3);
        agg_o += tot3[0]; agg_k += tot3[2];
    }
    mark();                                          // [2] histogram + aggregate
    auto pack = [](unsigned long long flag, uint32_t o, uint32_t k) { return flag | ((unsigned long long)o << 32) | k; };
    if (tid == 0)                                                             // (bin 0's aggregate IS its inclusive prefix)
        __hip_atomic_store(state + b, pack(b == 0 ? kFlagInc : kFlagAgg, agg_o, agg_k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // decoupled look-back (wave 0): prefix of all earlier bins
    auto lookBack = [&]() {
        if (wave == 0) {
            uint32_t po = 0, pk = 0;
            int back = b - 1;
            while (back >= 0) {
                const int t = back - lane;
                unsigned long long w;
                // every lane polls its predecessor's word until it has published something
                do { w = t >= 0 ? __hip_atomic_load(state + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kFlagInc; } while (__any((w & kFlagMask) == 0ull));
                // nearest predecessor holding an inclusive prefix: lanes below it contribute aggregates, it contributes the prefix
                const int first = __ffsll((long long)__ballot((w & kFlagMask) == kFlagInc)) - 1;      // >= 0 when any (lanes with t < 0 count as "inclusive 0")
                const bool take = t >= 0 && (first < 0 || lane <= first);
                po += waveSum(take ? (uint32_t)((w & ~kFlagMask) >> 32) : 0u);
                pk += waveSum(take ? (uint32_t)w : 0u);
                if (first >= 0) break;
                back -= kWave;
            }
            if (lane == 0) {
                s_pref[0] = po; s_pref[1] = pk;
                if (b > 0) __hip_atomic_store(state + b, pack(kFlagInc, po + agg_o, pk + agg_k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    };
    const uint32_t maxP = (uint32_t)p.max_pillars_num, maxN = (uint32_t)p.max_points_num_voxel_filter;
    const uint32_t gxy = (uint32_t)(p.gx * p.gy), gxyz = gxy * (uint32_t)p.gz;
    if (agg_o == 0) lookBack();                      // (uniform) an empty bin only passes the prefix on
    uint32_t Po = 0, Pk = 0;                         // pillars / kept points before the current sub-range
    uint32_t subbase = 0;                            // points of this bin's earlier sub-ranges

    // ---- phase II, per sub-range: cell scan -> slots into their segments of the sorted array -> [look-back] -> pillar records ----
    if (agg_o != 0)
    for (int s = 0; s < pl.nsub; ++s) {
        if (pl.nsub > 1) histSub(s);                 // (one sub-range: cnt[] is still phase I's)
        if (pl.nsub > 1) { tripleSums(pre[0], pre[1], pre[2]); blockExclusiveScanK<kBT, 3>(pre, smem, tot3); }
        const uint32_t eo0 = pre[0], ef0 = pre[1], ek0 = pre[2], to = tot3[0], tf = tot3[1], tk = tot3[2];
        {
            uint32_t ef = ef0;
#pragma unroll
            for (int j = 0; j < CPT; ++j) { const int cl = tid * CPT + j; segx[cl] = ef; cur[cl] = 0; ef += cnt[cl]; }
        }
        __syncthreads();
        mark();                                      // [3] scans + segment offsets
        const uint32_t c0 = cell_bin0 + (uint32_t)s * kBinCells;
        uint32_t* sseg = srt + binbase + subbase;     // the sorted array: slots (positions in the partitioned arrays) in segment order
        if (fast) {
            // The list becomes the bin's permutation in LDS -- slots[position] = slot, position = segment offset + LDS cursor -- and leaves
            // with lane-contiguous stores.  (Moving the POINTS here -- lane-scattered stores, or lane-contiguous stores fed by
            // lane-scattered loads -- costs the CU that owns the bin 4-7 cycles per lane: 35-40 us for the 12k points of a bin that
            // crosses the sensor, which then is the duration of the launch; p2f_pillar's gathers are spread over all CUs.)
#pragma unroll
            for (int h = 0; h < EPT; h += 4) {       // (every thread took its slots into registers before phase I's barrier; the keys come from L2 again)
                uint32_t ky[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) ky[u] = sl[h + u] != kNone ? part_key[sl[h + u]] : kNone;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ky[u] != kNone) { const uint32_t cl = ky[u] - c0; slots[segx[cl] + atomicAdd(&cur[cl], 1u)] = sl[h + u]; }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            mark();                                  // [4] permutation
            for (uint32_t e = tid; e < nfast; e += kBT) sseg[e] = slots[e];
        } else {
            // (bins with more than kCap points, launches with more than kBT partition blocks, grids whose bins span several sub-ranges:
            // every slot stored at its position as it is found)
            forEntries([&](uint32_t slot, uint32_t key) {
                const uint32_t cl = key - c0;
                if (cl < (uint32_t)kBinCells) sseg[segx[cl] + atomicAdd(&cur[cl], 1u)] = slot;
            });
        }
        __syncthreads();
        mark();                                      // [5] slots in their segments
        if (s == 0) { lookBack(); Po = s_pref[0]; Pk = s_pref[1]; }
        mark();                                      // [6] look-back
        {   // pillar records
            uint32_t eo = eo0, ef = ef0, ek = ek0;
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int cl = tid * CPT + j;
                uint32_t o, f, k;
                cellTriple(cnt[cl], T, o, f, k);
                if (o) {
                    // capacity guard the reference lacks: the pillar list is truncated at the first pillar that overflows
                    // max_pillars_num or max_points_num_voxel_filter (the running kept count is monotonic, so the valid pillars form a prefix)
                    const uint32_t go = Po + eo, gk = Pk + ek;
                    const bool valid = go < maxP && gk + k <= maxN;
                    if (valid) {
                        pil_rec[go] = make_uint4(binbase + subbase + ef, f, gk, 0u);   // segment, points, compact offset
                        pcnt[go] = k;                                             // :753
                        const uint32_t cell = c0 + (uint32_t)cl;
                        const uint32_t fr = cell / gxyz, cf = cell % gxyz, cz = cf / gxy, cyx = cf % gxy;
                        reinterpret_cast<uint4*>(coords)[go] = make_uint4(fr, cz, cyx / (uint32_t)p.gx, cyx % (uint32_t)p.gx);       // :755-756 (batch = z = 0 there)
                    } else if (go == 0 || (go - 1 < maxP && gk <= maxN)) {
                        *pillar_num = go; *point_num = gk;                          // the FIRST pillar that does not fit: P and Nk are its offsets
                    }
                }
                eo += o; ef += f; ek += k;
            }
        }
        Po += to; Pk += tk; subbase += tf;
        __syncthreads();
    }
    mark();                                          // [7] records
    // nothing overflowed: the last bin knows the totals
    if (b == pl.nbins - 1 && tid == 0) {
        const uint32_t eo = s_pref[0] + agg_o, ek = s_pref[1] + agg_k;
        if (eo <= maxP && ek <= maxN) { *pillar_num = eo; *point_num = ek; }
    }
}

constexpr uint32_t kPillarChunk = 64;            // consecutive (virtual) workgroups of p2f_pillar that share an XCD
// ---- 3. pillar rows ----------------------------------------------------------------------------------------------------
// A wavefront owns FOUR pillars.  Pillars hold 4.8 points on average, so when all four have <= 16 points (the common case) each takes
// a 16-lane group: same ranking, same sequential fp32 sums, a quarter of the wavefronts.  A pillar with more points gets the whole
// wavefront first.  (A first version of round 3 did this inside p2f_bins, one workgroup per bin: the bins that cross the sensor hold
// hundreds of pillars with more than 16 points, one wavefront pass each -- 496 us for the launch instead of ~40.)
__global__ void __launch_bounds__(256)
p2f_pillar(P2FParams p, int dbg, const uint32_t* __restrict__ pillar_num, const uint32_t* __restrict__ srt, const float4* __restrict__ part_pts,
           const uint32_t* __restrict__ part_idx, const uint4* __restrict__ pil_rec, float* __restrict__ feat, uint32_t* __restrict__ pidx)
{
    __shared__ uint32_t sel_lds[256];
    const int lane = laneId(), sub = lane >> 4, sl = lane & 15;
    // The four pillars of a wavefront are 1024 apart (dense cells come in runs of neighbours: a wavefront that gets four of them walks four
    // whole-wavefront passes -- 60 us for the launch with pillars 4 gw .. 4 gw + 3) and their numbers do not depend on the pillar COUNT,
    // so the records are loaded beside the count, not behind it: wavefront j of a group of S takes pillars base + j + {0, S, 2S, 3S} of
    // the group's 4S, S = a sixteenth of the capacity (the dense region of a frame is ~9000 pillars wide).
    const uint32_t S = ((uint32_t)p.max_pillars_num + 15u) / 16u;
    // XCD-aware (round 5): workgroups go to the eight XCDs round-robin (workgroup b -> XCD b % 8), and the 64-byte lines of the partitioned points hold the
    // points of FOUR slots of one (block, bin) piece, i.e. of neighbouring pillars.  With workgroup b on pillars 4 b .. each line was fetched into several
    // L2s; chunks of kPillarChunk consecutive workgroups (256 consecutive pillars of each of the four groups) now share an XCD: virtual workgroup
    // ((s / CH) * 8 + x) * CH + s % CH for x = b % 8, s = b / 8 (the grid is padded to a multiple of 8 CH).
    uint32_t vb = blockIdx.x;
    // PMC, four frames per launch (tools/pmc_fetch_kernel.sh, FETCH_SIZE as counted / x 2 per the guide): 48.8 / 97.7 MB with the plain mapping (DSVT_P2F_DBG=512),
    // chunks of 16 workgroups 39.3 / 78.5, of 64 28.5 / 57.0 (kept: the launch 0.7 us shorter), of 256 15.2 / 30.4 (the launch 1.5 us LONGER: the last chunks of a
    // frame's pillars leave XCDs idle); writes 31 MB either way.  Algorithmic bytes of the launch: 43.6 MB.
    if (!P2F_DBG(512)) { const uint32_t CH = P2F_DBG(1024) ? 16u : P2F_DBG(2048) ? 256u : kPillarChunk, x = vb & 7u, s_ = vb >> 3; vb = ((s_ / CH) * 8u + x) * CH + s_ % CH; }
    const uint32_t gw = vb * (blockDim.x / kWave) + threadIdx.x / kWave;
    const uint32_t pid0 = (gw / S) * (4u * S) + (gw % S), pid = pid0 + (uint32_t)sub * S;
    const uint4 rc0 = pid < (uint32_t)p.max_pillars_num ? pil_rec[pid] : make_uint4(0u, 0u, 0u, 0u);     // (stale beyond the count: masked below)
    const uint32_t P = *pillar_num;
    if (pid0 >= P) return;
    const bool have = pid < P;
    const uint4 rec = have ? rc0 : make_uint4(0u, 0u, 0u, 0u);                   // segment, points, compact offset
    const uint32_t nfull = rec.y;
    // pillars with more than 16 points: the whole wavefront, one after the other (every lane still here).  (Issuing the loads of all
    // four pillars before processing any -- two round trips per wavefront instead of two or three per pillar -- was slower: 89 registers,
    // five waves per SIMD instead of seven.)
    for (int k = 0; k < 4; ++k) {
        const uint32_t nk = __shfl(nfull, 16 * k, kWave);
        if (nk > 16u && !P2F_DBG(256))
            p2fPillarWave(pid0 + (uint32_t)k * S, __shfl(rec.x, 16 * k, kWave), nk, __shfl(rec.z, 16 * k, kWave), p, srt, part_pts, part_idx, feat, pidx,
                          sel_lds + (threadIdx.x / kWave) * kWave);
    }
    // loop bound of the group loops: the largest small pillar of this wavefront (wave-uniform)
    int nmax = 0;
    for (int k = 0; k < 4; ++k) { const int nk = (int)__shfl(nfull, 16 * k, kWave); if (nk <= 16 && nk > nmax) nmax = nk; }
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    if (!have || nfull > 16u) return;                                           // (whole 16-lane groups: the group shuffles below stay inside live groups)
    const uint32_t T = p.max_num_points_per_voxel;
    const uint32_t seg = rec.x, ptoff = rec.z;
    const uint32_t kept = nfull < T ? nfull : T;
    // lane s (< kept) of the group ends up holding the point with the s-th smallest row index in the cell (index and point are loaded
    // together and the point moves by ds_permute: one round trip less than fetching it after the ranking)
    const uint32_t slot = sl < (int)nfull ? srt[seg + sl] : kNone;
    const uint32_t mine = slot != kNone ? part_idx[slot] : kNone;
    const float4 mq = slot != kNone ? part_pts[slot] : make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t rank = 0;
    for (int j = 0; j < nmax; ++j) rank += (__shfl(mine, j, 16) < mine) ? 1u : 0u;
    if (slot == kNone) rank = (uint32_t)sl;                                     // idle lanes push onto themselves
    const int dst = (int)(((uint32_t)(sub << 4) + rank) * 4);
    float4 q;
    q.x = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(mq.x)));
    q.y = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(mq.y)));
    q.z = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(mq.z)));
    q.w = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(mq.w)));
    float cx = 0.f, cy = 0.f, cz = 0.f;                                         // sequential fp32 sum in slot order (:813-824)
    for (int s_ = 0; s_ < nmax; ++s_) {
        const float vx = __shfl(q.x, s_, 16), vy = __shfl(q.y, s_, 16), vz = __shfl(q.z, s_, 16);
        if (s_ < (int)kept) { cx += vx; cy += vy; cz += vz; }
    }
    const int ni = (int)kept;
    cx = cx / ni; cy = cy / ni; cz = cz / ni;
    if (!P2F_DBG(128))
    for (uint32_t e = (uint32_t)sl; e < (uint32_t)p.pid_slots; e += 16u) pidx[(size_t)pid * T + e] = e < kept ? ptoff + e : 0u;   // :829-830
    if (sl < (int)kept && !P2F_DBG(64)) p2fWriteFeat(feat + (size_t)(ptoff + sl) * p.feature_num, q, cx, cy, cz, p);
    if (P2F_DBG(64) && cx + q.x == 123.f) feat[0] = cx;       // (keeps the arithmetic alive)
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
    P2FPlan plan() const {
        P2FPlan pl{};
        pl.bpf = cdiv(p_.max_points_num, kBlk); pl.nblk = pl.bpf * p_.frames; pl.ncell = ncell();
        pl.nsub = 1; pl.bin_shift = 11;                               // log2(kBinCells)
        while (cdiv(pl.ncell, kBinCells * pl.nsub) > kMaxBins) { pl.nsub *= 2; ++pl.bin_shift; }
        pl.nbins = cdiv(pl.ncell, kBinCells * pl.nsub);
        return pl;
    }
    size_t slots() const { return (size_t)plan().nblk * kBlk; }                                  // slots of the partitioned / sorted arrays
    size_t stateWords() const { return 2 * (1 + (size_t)plan().nbins); }            // 32-bit words: ticket + one 64-bit word per bin
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override {
        const P2FPlan pl = plan();
        size_t s = alignUp(sizeof(uint32_t) * stateWords());                                     // ticket + look-back state
        s += alignUp(sizeof(uint16_t) * (size_t)pl.nblk * (pl.nbins + 1));                       // piece offsets
        s += alignUp(sizeof(float4) * slots()) + 3 * alignUp(sizeof(uint32_t) * slots());        // partitioned points, keys, row indices; sorted slots
        s += alignUp(sizeof(uint4) * p_.max_pillars_num);                                        // pillar records (segment, points, compact offset)
        return s;                                                     // ~23 MB at four 196k-point frames (reference: 176.8 MB per frame, :262-277)
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
        P2FPlan pl = plan();
        static unsigned long long* tr = nullptr; static int tron = -1;     // tools/: DSVT_P2F_TRACE=1 with the ablate build
        if (tron < 0) { tron = ablateEnv("DSVT_P2F_TRACE", 0) ? 1 : 0; if (tron) (void)hipMallocManaged(&tr, 8 * 16 * 65536); }
        pl.trace = tr;
        static int dbg = -1; if (dbg < 0) dbg = ablateEnv("DSVT_P2F_DBG", 0);
        pl.dbg = dbg;
        WsCarver ws(workspace);
        uint32_t* scan_state = ws.take<uint32_t>(stateWords());
        uint16_t* tab = ws.take<uint16_t>((size_t)pl.nblk * (pl.nbins + 1));
        float4* part_pts = ws.take<float4>(slots());
        uint32_t* part_key = ws.take<uint32_t>(slots());
        uint32_t* part_idx = ws.take<uint32_t>(slots());
        uint32_t* srt = ws.take<uint32_t>(slots());
        uint4* pil_rec = ws.take<uint4>(p_.max_pillars_num);

        if (zeroFill) {   // reference zero-fills every output each call (:928-937)
            DSVT_CHECK(hipMemsetAsync(feat, 0, sizeof(float) * (size_t)p_.max_points_num_voxel_filter * p_.feature_num, stream));
            DSVT_CHECK(hipMemsetAsync(pidx, 0, sizeof(uint32_t) * (size_t)p_.max_pillars_num * p_.max_num_points_per_voxel, stream));
            DSVT_CHECK(hipMemsetAsync(coords, 0, sizeof(uint32_t) * (size_t)p_.max_pillars_num * 4, stream));
            DSVT_CHECK(hipMemsetAsync(pcnt, 0, sizeof(uint32_t) * (size_t)p_.max_pillars_num, stream));
        }
        // 48 KB of static LDS + up to 32 KB of histogram: the opt-in is issued once per DEVICE (a host with several detectors, one per GPU)
        static bool lds_set[64] = {};
        int devId = 0; (void)hipGetDevice(&devId);
        if (devId < 0 || devId >= 64 || !lds_set[devId]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(p2f_partition), hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(uint32_t) * (kMaxBins + 1));
            if (devId >= 0 && devId < 64) lds_set[devId] = true;
        }
        hipLaunchKernelGGL(p2f_partition, dim3(pl.nblk), dim3(256), sizeof(uint32_t) * (pl.nbins + 1), stream, pts, n_ptr, p_, pl,
                           tab, part_pts, part_key, part_idx, scan_state, (int)stateWords());
        hipLaunchKernelGGL(p2f_bins, dim3(pl.nbins), dim3(kBT), 0, stream, p_, pl, tab, part_key, srt, scan_state,
                           pil_rec, coords, pcnt, pillar_num, point_num);
        #ifdef DSVT_ABLATE
        constexpr int kPillarPad = 8 * 256;                     // (the ablation build's largest chunk, DSVT_P2F_DBG=2048)
#else
        constexpr int kPillarPad = 8 * (int)kPillarChunk;
#endif
        hipLaunchKernelGGL(p2f_pillar, dim3(cdiv(cdiv(cdiv(p_.max_pillars_num, 16), 4) * 4, kPillarPad) * kPillarPad), dim3(256), 0, stream,     // four groups of S = cap / 16 wavefronts (padded to whole XCD chunks)
                           p_, pl.dbg, pillar_num, srt, part_pts, part_idx,
                           pil_rec, feat, pidx);
        if (tron) {
            (void)hipStreamSynchronize(stream);
            unsigned long long t0 = ~0ull;
            for (int b = 0; b < pl.nbins; ++b) if (tr[b * 16] < t0) t0 = tr[b * 16];
            for (int b = 0; b < pl.nbins; b += (pl.nbins > 64 ? pl.nbins / 48 : 1)) {
                fprintf(stderr, "[p2f bin %4d]", b);
                for (int i = 0; i < 16; ++i) fprintf(stderr, " %6.1f", tr[b * 16 + i] ? (double)(tr[b * 16 + i] - t0) * 0.01 : 0.0);
                fprintf(stderr, " us\n");
            }
        }
        return lastError();
    }
    // the reference's 18 words (:1033-1036); trailing ints [frames [point_id_slots]], each present when it or a later one is not the default
    int nTrail() const { return p_.pid_slots != p_.max_num_points_per_voxel ? 2 : p_.frames > 1 ? 1 : 0; }
    size_t serializationSize() const override { return 9 * sizeof(float) + (9 + nTrail()) * sizeof(int); }
    void serialize(void* buffer) const override {                                                  // :1038-1060
        char* d = static_cast<char*>(buffer);
        wr<int>(d, p_.max_points_num); wr<int>(d, p_.max_points_num_voxel_filter); wr<int>(d, p_.max_pillars_num);
        wr<int>(d, p_.point_feature_num); wr<int>(d, p_.feature_num); wr<int>(d, p_.max_num_points_per_voxel);
        wr<float>(d, p_.min_x); wr<float>(d, p_.max_x); wr<float>(d, p_.min_y); wr<float>(d, p_.max_y);
        wr<float>(d, p_.min_z); wr<float>(d, p_.max_z); wr<float>(d, p_.vx); wr<float>(d, p_.vy); wr<float>(d, p_.vz);
        wr<int>(d, p_.gx); wr<int>(d, p_.gy); wr<int>(d, p_.gz);
        if (nTrail() >= 1) wr<int>(d, p_.frames);
        if (nTrail() >= 2) wr<int>(d, p_.pid_slots);
    }
    Plugin* clone() const override { return new Points2FeaturesPlugin(p_); }
};

static bool validP2F(const P2FParams& p) {
    auto no = [](const char* why) { createError() = std::string("Points2FeaturesPlugin: ") + why; return false; };
    if (p.max_points_num <= 0 || p.max_points_num_voxel_filter <= 0 || p.max_pillars_num <= 0) return no("max_points_num, max_points_num_voxel_filter and max_pillars_num must be positive");
    if (p.point_feature_num != 4 || p.feature_num != 10) return no("point_feature_num must be 4 and feature_num 10 (x, y, z, intensity -> the reference's ten features)");
    if (p.max_num_points_per_voxel <= 0 || p.max_num_points_per_voxel > kWave) return no("max_num_points_per_voxel must be in 1 .. 64 (one wavefront lane per kept point)");
    if (p.gx <= 0 || p.gy <= 0 || p.gz <= 0 || p.frames < 1) return no("grid_size and frames must be positive");
    if (!(p.vx > 0 && p.vy > 0 && p.vz > 0)) return no("voxel_size must be positive");
    if (p.pid_slots < 1 || p.pid_slots > p.max_num_points_per_voxel) return no("point_id_slots must be in 1 .. max_num_points_per_voxel");
    if ((long)p.gx * p.gy * p.gz * p.frames >= (1l << 30)) return no("grid cells x frames must stay below 2^30 (the look-back word holds a 30-bit pillar count)");
    if (((long)p.max_points_num + kBlk) * p.frames >= (1l << 31)) return no("(max_points_num + 2048) x frames must stay below 2^31 (32-bit row indices and slot numbers)");
    return true;
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
    p.pid_slots = fieldInt(fc, "point_id_slots", p.max_num_points_per_voxel);      // not a reference field: see P2FParams
    return validP2F(p) ? new Points2FeaturesPlugin(p) : nullptr;
}
static Plugin* p2fDeserialize(const void* data, size_t len) {                                      // ctor :60-84
    const int extra = trailingInts(len, 9 * sizeof(float) + 9 * sizeof(int), 2);
    if (extra < 0) return nullptr;
    const char* d = static_cast<const char*>(data);
    P2FParams p{};
    p.max_points_num = rd<int>(d); p.max_points_num_voxel_filter = rd<int>(d); p.max_pillars_num = rd<int>(d);
    p.point_feature_num = rd<int>(d); p.feature_num = rd<int>(d); p.max_num_points_per_voxel = rd<int>(d);
    p.min_x = rd<float>(d); p.max_x = rd<float>(d); p.min_y = rd<float>(d); p.max_y = rd<float>(d);
    p.min_z = rd<float>(d); p.max_z = rd<float>(d); p.vx = rd<float>(d); p.vy = rd<float>(d); p.vz = rd<float>(d);
    p.gx = rd<int>(d); p.gy = rd<int>(d); p.gz = rd<int>(d);
    p.frames = extra >= 1 ? rd<int>(d) : 1;
    p.pid_slots = extra >= 2 ? rd<int>(d) : p.max_num_points_per_voxel;
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
