// proj_fill.hpp -- pass 3 of (Depth)FlowProjection, round 4: hole filling from BIT MASKS.  Textually included by
// flow_projection.hip (namespace memc).
//
// Replaces my_package/src/my_lib_kernel.cu:1742-1836 (and the identical :2169-2264) of the reference.  The reference
// walks from every hole (count <= 0) to the nearest cell with a non-zero count to its left, to its right and above (the
// downward search is dead code, :1799) and writes the mean of the output at those of them whose count is positive.
//
// Rounds 1-3 detected holes, wrote per-tile summaries with LDS atomics and row reductions, stored the planes through the
// cache and ran a second kernel that re-read the count cells of every tile holding a hole (8.4x its algorithmic bytes)
// and walked them cell by cell: +46 us on a 160 us projection (720p, batch 32).  Now:
//   * the owner kernel, which has every count of its 64 x TH tile in registers at its store epilogue, builds the tile's
//     NON-ZERO MASKS -- one 64-bit word per row (an OR over the 16 lanes of a row: DPP), one TH-bit word per column
//     (ds_or_b32) -- only when the tile holds a hole (one vote; 82 % of the tiles of the benchmark's flow do not);
//   * a walk inside the tile is then a mask, a count-leading-zeros and nothing else: the owner fills, from the staged
//     planes in LDS and BEFORE its one store of the cell, every hole whose three walks end inside the tile or at the
//     image border (round 3 tried this with cell-by-cell walks and lost 59 us);
//   * the holes that need a neighbouring tile are left as bits of a PENDING mask; tiles with a hole write their masks
//     (TH + TH 64-bit words + 64 32-bit words) and all tiles their three summaries (last non-zero row per column,
//     first / last non-zero column per row: a ctz / clz of the masks) to the call's scratch;
//   * proj_fill_pending visits the flagged tiles, reads the masks (768 B instead of 8 KiB of counts) and finishes the
//     pending holes: in-tile part from the masks, beyond the tile through the neighbours' summaries, nearest first.
// Same cells, same flags, same arithmetic as the reference's walks.
#pragma once

struct FillWs {
    int *up, *left, *right;       // per-tile summaries: last non-zero row per column, last / first non-zero column per row
    int *hole;                    // hole[tile] != 0: tile (id (b * tiles_y + ty) * tiles_x + tx) holds a hole and wrote masks
    unsigned long long *masks;    // per tile TileMasks<TH>, valid where hole[tile] != 0 (round 4; unused by the round 1-3 arms)
};

template <int TH>
struct TileMasks {                // global scratch, per tile
    unsigned long long row[TH];   // bit c: cell (row, c) of the tile has a non-zero count (stops a walk, :1778-1797)
    unsigned long long pend[TH];  // bit c: the cell is a hole (count <= 0, :1757) that is still to be filled
    unsigned col[64];             // bit r of word c: the same as row[r] bit c
};
template <int TH>
constexpr size_t tile_mask_words() { return sizeof(TileMasks<TH>) / 8; }

template <int TH>
struct FillLds {                  // LDS of the owner kernels
    unsigned long long row[TH];   // non-zero mask of every row
    unsigned long long pend[TH];  // pending holes of every row
    unsigned col[64];             // non-zero mask of every column
    int n_holes;                  // (the holes' list itself lives behind the staged planes, in the point planes' bytes)
};

template <int TH>
__device__ __forceinline__ void fill_lds_init(FillLds<TH> &f, int tid)
{
    static_assert(TH <= 32, "a column mask is one 32-bit word");
    if (tid < 64) f.col[tid] = 0u;
    if (tid < TH) f.pend[tid] = 0ull;
    if (tid == 0) f.n_holes = 0;
}

// OR over each group of 16 consecutive lanes (one DPP row), result in ALL 16 lanes (row rotates)
__device__ __forceinline__ unsigned row16_or_u32(unsigned v)
{
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false);      // row_ror:1
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false);      // row_ror:2
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);      // row_ror:4
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);      // row_ror:8
    return v;
}

__device__ __forceinline__ int last_bit64(unsigned long long m) { return 63 - __builtin_clzll(m); }   // m != 0

// The three walks of the hole at (r, c) of a tile, as far as the tile's masks can tell.
// Returns per direction: >= 0 the LOCAL position found; -1 "ran into the image border, nothing found" (count 0 there,
// my_lib_kernel.cu:1778-1797 with the loop's bound); -2 unresolved: the walk leaves the tile.
struct TileWalk {
    int l, r, u;
};
__device__ __forceinline__ TileWalk tile_walk_masks(unsigned long long rowm, unsigned colm, int r, int c, bool first_tx,
                                                    bool last_tx, bool first_ty)
{
    const unsigned long long below = (1ull << c) - 1ull;            // columns < c
    const unsigned long long lm = rowm & below, rm = rowm & ~(below | (1ull << c));
    const unsigned um = colm & ((1u << r) - 1u);
    TileWalk w;
    w.l = lm ? last_bit64(lm) : (first_tx ? -1 : -2);
    w.r = rm ? (int)__builtin_ctzll(rm) : (last_tx ? -1 : -2);     // (cells past the image edge carry no bit)
    w.u = um ? 31 - (int)__builtin_clz(um) : (first_ty ? -1 : -2);
    return w;
}

// my_lib_kernel.cu:1801-1832: the fill value of one component from the three stops (counts lt / rt / ut: 0 where the walk
// found nothing; values vl / vr / vu there; `self` the cell's own value: the dead downward search contributes 0 * self)
__device__ __forceinline__ float fill_value(float lt, float rt, float ut, float vl, float vr, float vu, float self)
{
    const float fl = lt > 0.0f ? 1.0f : 0.0f, fr = rt > 0.0f ? 1.0f : 0.0f;
    const float fu = ut > 0.0f ? 1.0f : 0.0f, fd = 0.0f;
    return (fl * vl + fr * vr + fu * vu + fd * self) / (fl + fr + fu + fd);
}

// Store epilogue of an owner tile when pass 3 follows.  Lane `tid` owns the cells (4 q .. 4 q + 3, r) of the tile, r = tid / 16,
// q = tid % 16, with count / output in oc / ox / oy (updated in place for the holes filled here).  `stage`: LDS the point
// planes occupied (>= 3 TH 64 floats + TH 64 shorts), free once every wave is past its read-out -- which the vote below
// establishes.
// Holes are a per cent of the cells: handled by the lanes that own them, a wave would run the walk four times (once per
// cell of a lane) with a lane or two active.  They are LISTED in LDS and dealt out one per lane instead -- with the
// benchmark's flows the whole list is one pass of the first wave -- and the owners read their cells back from the stage.
// Converged code only (barriers).  ws.hole[tile]: 0 no hole, 1 holes, all filled here, 2 holes pending.
// PENDT (measurement build; -1 = the product): a tile in which more than PENDT lanes hold a hole skips the in-tile fill -- no
// hole list, no staged planes, no walks, no read-back: masks and summaries only, every hole of it pending -- and leaves its
// holes to proj_fill_pending (0: every tile with a hole).  Same results either way.  Round-5 review, item 5: does the epilogue's
// ~2000 clocks per phase, once nearly every tile holds holes (the flow twice as large, a camera pan), pay better there?
template <int TH, int NT, bool TRACE = false, int PENDT = -1>
__device__ __forceinline__ void owner_fill_epilogue(FillLds<TH> &fl, float *stage, const FillWs &ws, int tid, int b, int tx,
                                                    int ty, int W, int H, int tiles_x, int tiles_y, bool inb, f32x4 &ox,
                                                    f32x4 &oy, const f32x4 &oc)
{
    static_assert(NT == 16 * TH, "one lane per four cells");
    const int r = tid / 16, q = tid % 16;
    const int tx0 = tx * 64, ty0 = ty * TH;
    const int64_t tile_id = ((int64_t)b * tiles_y + ty) * tiles_x + tx;
    unsigned nz = 0, hole = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool in_row = inb && tx0 + 4 * q + j < W;        // (a ragged row's last quad: the cells past the row are nothing)
        if (in_row && oc[j] != 0.0f) nz |= 1u << j;            // what stops a walk (my_lib_kernel.cu:1778-1797)
        if (in_row && oc[j] <= 0.0f) hole |= 1u << j;          // what pass 3 fills (:1757)
    }
    const int hole_lanes = PENDT >= 0 ? __syncthreads_count(hole != 0) : __syncthreads_or(hole != 0);
    const bool any_hole = hole_lanes != 0;
    trace_mark_proj<TRACE>(6);                 // (timestamp instance: tools/probes/proj_pan_phases.py)
    if (!any_hole) {
        // No hole: every cell of the tile inside the image has a positive count -- every walk that enters the tile stops at
        // its first cell.  The trivial summaries, no masks.
        if (tid < 64 && tx0 + tid < W) ws.up[((int64_t)b * tiles_y + ty) * W + tx0 + tid] = min(ty0 + TH - 1, H - 1);
        if (tid < TH && ty0 + tid < H) {
            const int64_t i = ((int64_t)b * tiles_x + tx) * H + ty0 + tid;      // [b][tx][y]: a tile's rows are one run
            ws.right[i] = tx0;
            ws.left[i] = min(tx0 + 63, W - 1);
        }
        if (tid == 0) ws.hole[tile_id] = 0;
        trace_mark_proj<TRACE>(7);
        trace_mark_proj<TRACE>(8);
        trace_mark_proj<TRACE>(9);
        return;
    }
    // masks: the row's 64 bits (an OR over the 16 lanes of a row), the columns' TH bits (LDS); the holes' list
    {
        const unsigned lo = row16_or_u32(q < 8 ? nz << (4 * q) : 0u), hi = row16_or_u32(q >= 8 ? nz << (4 * (q - 8)) : 0u);
        if (q == 15) fl.row[r] = ((unsigned long long)hi << 32) | lo;
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        if ((nz >> j) & 1u) atomicOr(&fl.col[4 * q + j], 1u << r);
    // (round 5: the column masks from four ballots, one conflict-free LDS atomic per wave instead of these four with four
    // lanes on every word -- no difference in the phase's clocks, tools/probes/proj_pan_phases.py; left as it was)
    if (PENDT >= 0 && hole_lanes > PENDT) {                    // (workgroup-uniform) every hole of this tile pending
        const unsigned plo = row16_or_u32(q < 8 ? hole << (4 * q) : 0u), phi = row16_or_u32(q >= 8 ? hole << (4 * (q - 8)) : 0u);
        if (q == 15) fl.pend[r] = ((unsigned long long)phi << 32) | plo;
        __syncthreads();
        if (tid < TH && ty0 + tid < H) {
            const unsigned long long rowm = fl.row[tid];
            const int64_t i = ((int64_t)b * tiles_x + tx) * H + ty0 + tid;
            ws.right[i] = rowm ? tx0 + (int)__builtin_ctzll(rowm) : -1;
            ws.left[i] = rowm ? tx0 + last_bit64(rowm) : -1;
        }
        if (tid < 64 && tx0 + tid < W) {
            const unsigned cm = fl.col[tid];
            ws.up[((int64_t)b * tiles_y + ty) * W + tx0 + tid] = cm ? ty0 + 31 - (int)__builtin_clz(cm) : -1;
        }
        TileMasks<TH> *tm = reinterpret_cast<TileMasks<TH> *>(ws.masks) + tile_id;
        if (tid < TH) {
            tm->row[tid] = fl.row[tid];
            tm->pend[tid] = fl.pend[tid];
        }
        if (tid < 64) tm->col[tid] = fl.col[tid];
        if (tid == 0) ws.hole[tile_id] = 2;
        return;
    }
    unsigned short *const hole_list = reinterpret_cast<unsigned short *>(stage + 3 * TH * 64);
    {   // one atomic per wave (a tile in an uncovered band is ALL holes: 64 lanes adding to one LDS word four times over)
        unsigned long long m[4];
        int tot = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            m[j] = __builtin_amdgcn_ballot_w64(((hole >> j) & 1u) != 0);
            tot += __builtin_popcountll(m[j]);
        }
        if (tot) {                                             // (wave-uniform)
            const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            int base = 0;
            if (lane == 0) base = atomicAdd(&fl.n_holes, tot);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if ((hole >> j) & 1u)
                    hole_list[base + __builtin_popcountll(m[j] & ((1ull << lane) - 1))] = (unsigned short)((r << 6) | (4 * q + j));
                base += __builtin_popcountll(m[j]);
            }
        }
    }
    // planes staged for the gathers below: [count, x, y][r][64]
    *reinterpret_cast<f32x4 *>(stage + r * 64 + 4 * q) = oc;
    *reinterpret_cast<f32x4 *>(stage + TH * 64 + r * 64 + 4 * q) = ox;
    *reinterpret_cast<f32x4 *>(stage + 2 * TH * 64 + r * 64 + 4 * q) = oy;
    __syncthreads();
    trace_mark_proj<TRACE>(7);
    const int n = fl.n_holes;
    float *const sx = stage + TH * 64, *const sy = stage + 2 * TH * 64;
    for (int i = tid; i < n; i += NT) {
        const int cell = hole_list[i], hr = cell >> 6, hc = cell & 63;
        const TileWalk w = tile_walk_masks(fl.row[hr], fl.col[hc], hr, hc, tx == 0, tx == tiles_x - 1, ty == 0);
        if (w.l == -2 || w.r == -2 || w.u == -2) {             // a walk leaves the tile: proj_fill_pending finishes it
            atomicOr(&fl.pend[hr], 1ull << hc);
            continue;
        }
        // a walk that found nothing stops at the border cell, whose count is 0: its flag is 0 and the reference
        // multiplies that cell's value by it -- 0 here
        const int il = hr * 64 + max(w.l, 0), ir = hr * 64 + max(w.r, 0), iu = max(w.u, 0) * 64 + hc;
        const float lt = w.l >= 0 ? stage[il] : 0.0f, rt = w.r >= 0 ? stage[ir] : 0.0f, ut = w.u >= 0 ? stage[iu] : 0.0f;
        if (lt + rt + ut + 0.0f <= 0.0f) continue;             // my_lib_kernel.cu:1801: the cell keeps its value
        // (reads touch cells with a non-zero count -- or times 0 --, writes cells with count <= 0: no ordering needed)
        const float vx = fill_value(lt, rt, ut, w.l >= 0 ? sx[il] : 0.0f, w.r >= 0 ? sx[ir] : 0.0f, w.u >= 0 ? sx[iu] : 0.0f, sx[cell]);
        const float vy = fill_value(lt, rt, ut, w.l >= 0 ? sy[il] : 0.0f, w.r >= 0 ? sy[ir] : 0.0f, w.u >= 0 ? sy[iu] : 0.0f, sy[cell]);
        sx[cell] = vx;
        sy[cell] = vy;
    }
    __syncthreads();
    trace_mark_proj<TRACE>(8);
    ox = *reinterpret_cast<const f32x4 *>(sx + r * 64 + 4 * q);     // the owners take their cells back
    oy = *reinterpret_cast<const f32x4 *>(sy + r * 64 + 4 * q);
    // summaries (what a walk from ANOTHER tile needs) and, for proj_fill_pending, the masks of a tile with pending holes
    int pending = 0;
#pragma unroll
    for (int i = 0; i < TH; i += 16) pending |= (fl.pend[i + q] != 0) ? 1 : 0;      // (every lane: 2 broadcast-free reads)
    pending = __builtin_amdgcn_ballot_w64(pending != 0) != 0;
    if (tid < TH && ty0 + tid < H) {
        const unsigned long long rowm = fl.row[tid];
        const int64_t i = ((int64_t)b * tiles_x + tx) * H + ty0 + tid;
        ws.right[i] = rowm ? tx0 + (int)__builtin_ctzll(rowm) : -1;
        ws.left[i] = rowm ? tx0 + last_bit64(rowm) : -1;
    }
    if (tid < 64 && tx0 + tid < W) {
        const unsigned cm = fl.col[tid];
        ws.up[((int64_t)b * tiles_y + ty) * W + tx0 + tid] = cm ? ty0 + 31 - (int)__builtin_clz(cm) : -1;
    }
    if (pending) {                                             // (wave-uniform; the same in every wave)
        TileMasks<TH> *tm = reinterpret_cast<TileMasks<TH> *>(ws.masks) + tile_id;
        if (tid < TH) {
            tm->row[tid] = fl.row[tid];
            tm->pend[tid] = fl.pend[tid];
        }
        if (tid < 64) tm->col[tid] = fl.col[tid];
    }
    if (tid == 0) ws.hole[tile_id] = pending ? 2 : 1;
    trace_mark_proj<TRACE>(9);
}

// Summaries and masks from the count plane, for the paths on which no owner kernel wrote them: the general path on its own
// (far_flag == nullptr: every tile) -- every hole is pending there.  Grid-stride over tiles, 16 TH lanes.
template <int TH>
__global__ __launch_bounds__(16 * TH) void proj_fill_masks(
    int W, int H, int tiles_x, int tiles_y, int batch, int64_t scb, int sch, const float *__restrict__ count, FillWs ws)
{
    __shared__ FillLds<TH> fl;
    const unsigned ntiles = (unsigned)tiles_x * tiles_y * batch;
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / ((unsigned)tiles_x * tiles_y);
        const int tid = tid_now(), r = tid / 16, q = tid % 16;
        const int tx0 = tx * 64, ty0 = ty * TH, x = tx0 + 4 * q, y = ty0 + r;
        const bool inb = x < W && y < H;
        fill_lds_init(fl, tid);
        const f32x4 oc = ld_cached4(count + b * scb + (int64_t)min(y, H - 1) * sch + min(x, W - 4));
        unsigned nz = 0, hole = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (inb && oc[j] != 0.0f) nz |= 1u << j;
            if (inb && oc[j] <= 0.0f) hole |= 1u << j;
        }
        const int any_hole = __syncthreads_or(hole != 0);      // (also: fl.col is zero)
        const unsigned lo = row16_or_u32(q < 8 ? nz << (4 * q) : 0u), hi = row16_or_u32(q >= 8 ? nz << (4 * (q - 8)) : 0u);
        const unsigned long long rowm = ((unsigned long long)hi << 32) | lo;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if ((nz >> j) & 1u) atomicOr(&fl.col[4 * q + j], 1u << r);
        __syncthreads();
        const unsigned plo = row16_or_u32(q < 8 ? hole << (4 * q) : 0u), phi = row16_or_u32(q >= 8 ? hole << (4 * (q - 8)) : 0u);
        TileMasks<TH> *tm = reinterpret_cast<TileMasks<TH> *>(ws.masks) + tile;
        if (q == 15 && y < H) {
            const int64_t i = ((int64_t)b * tiles_x + tx) * H + y;
            ws.right[i] = rowm ? tx0 + (int)__builtin_ctzll(rowm) : -1;
            ws.left[i] = rowm ? tx0 + last_bit64(rowm) : -1;
        }
        if (q == 15 && any_hole) {
            tm->row[r] = rowm;
            tm->pend[r] = ((unsigned long long)phi << 32) | plo;
        }
        if (tid < 64) {
            const unsigned cm = fl.col[tid];
            if (any_hole) tm->col[tid] = cm;
            if (tx0 + tid < W) ws.up[((int64_t)b * tiles_y + ty) * W + tx0 + tid] = cm ? ty0 + 31 - (int)__builtin_clz(cm) : -1;
        }
        if (tid == 0) ws.hole[tile] = any_hole ? 2 : 0;        // (2: pending holes -- here every hole is)
        __syncthreads();                                       // fl is re-initialised by the next tile
    }
}

// One direction of a walk beyond the tile: table[t * stride] for t = t0, t0 + dir, ... (t != tend) holds, per neighbouring
// tile, the position of its nearest non-zero cell in this row / column (-1: none).
struct TableWalk {
    const int *table;
    int64_t stride;
    int t, dir, tend, res;
    bool open;                                 // still looking
};
__device__ __forceinline__ TableWalk table_walk(bool wanted, const int *table, int64_t stride, int t0, int dir, int tend)
{
    TableWalk w = {table, stride, t0, dir, tend, -1, wanted};
    w.open = wanted && (dir > 0 ? t0 < tend : t0 > tend);
    return w;
}
// The three directions advance TOGETHER, eight neighbours each per round trip: an uncovered strip along an image border
// makes every hole of the strip look at all tiles of its row or column, and walked one dependent load at a time that
// chain was the whole kernel.
__device__ __forceinline__ void table_walks(TableWalk (&w)[3])
{
    constexpr int CH = 8;
    while (w[0].open || w[1].open || w[2].open) {
        int v[3][CH];
#pragma unroll
        for (int d = 0; d < 3; d++)
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int tt = w[d].t + k * w[d].dir;
                v[d][k] = (w[d].open && (w[d].dir > 0 ? tt < w[d].tend : tt > w[d].tend)) ? w[d].table[(int64_t)tt * w[d].stride] : -1;
            }
#pragma unroll
        for (int d = 0; d < 3; d++) {
            if (!w[d].open) continue;
#pragma unroll
            for (int k = CH - 1; k >= 0; k--) w[d].res = v[d][k] >= 0 ? v[d][k] : w[d].res;     // (the nearest valid one wins)
            w[d].t += CH * w[d].dir;
            w[d].open = w[d].res < 0 && (w[d].dir > 0 ? w[d].t < w[d].tend : w[d].t > w[d].tend);
        }
    }
}

// The pending holes of the flagged tiles (flag 2).  One WAVE per tile -- the pending lists are mostly short (the owner kernel
// filled what it could), a tile's masks are 768 bytes, and a tile's chain of round trips (flag -> masks -> neighbours'
// summaries, per row and column -> counts and values, four holes per lane at a time -> store) is what matters: how many
// tiles are in flight.  Wave-synchronous from the list on: the LDS of a wave is touched by that wave only.
// Which wave takes which tile (round 5): workgroup i reads the flags of the tiles i, i + grid, i + 2 grid, ... (one per
// thread), lists the flagged ones in LDS, and its SIXTEEN waves draw from that list until it is empty; one workgroup per CU.
// Before, wave g owned the tiles g and g + 8192 whether flagged or not: at this kernel's 120 VGPRs the chip holds 4096
// waves, so the launch ran as two rounds, and a camera pan -- whose uncovered bands are runs of tiles with 1000+ pending
// holes each, 15-20 us of dependent round trips per tile -- put such tiles in both rounds (55 us for the pan of 40 px, against
// 14 us for the benchmark's flow).  Lists per workgroup of four waves made it one round but left the luck of the draw:
// 1.35 heavy tiles per workgroup on average, six in the unluckiest, four waves to take them (41 us).  With sixteen waves
// sharing one list the heavy tiles of a pan are 5-6 per workgroup, nearly evenly (the grid is coprime to the tiles per
// row), and no wave takes two.
constexpr int kFillWaves = 16;
template <int TH>
__global__ __launch_bounds__(kFillWaves * kWave) void proj_fill_pending(
    int W, int H, int tiles_x, int tiles_y, int batch, int64_t s1b, int64_t s1c, int s1h, int64_t scb, int sch,
    const float *__restrict__ count, float *out, FillWs ws)
{
    constexpr unsigned kNT = kFillWaves * kWave;
    __shared__ TileMasks<TH> tms[kFillWaves];
    __shared__ unsigned short lists[kFillWaves][TH * 64];
    __shared__ struct { int lr[kWave], up[kWave]; } edges[kFillWaves];   // per wave: the walks' answers beyond the tile, per row side / column
    __shared__ unsigned flagged[kNT];
    __shared__ int nflagged, next_flagged;
    const int wv = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    auto &edge = edges[wv];
    TileMasks<TH> &tm = tms[wv];
    unsigned short *hole_list = lists[wv];
    const unsigned ntiles = (unsigned)tiles_x * tiles_y * batch;
    for (unsigned span = 0; span < ntiles; span += kNT * gridDim.x) {         // (one span: up to 262144 tiles)
    if (threadIdx.x == 0) nflagged = next_flagged = 0;
    __syncthreads();
    {
        const unsigned mine = span + blockIdx.x + threadIdx.x * gridDim.x;
        const bool is = mine < ntiles && ws.hole[mine < ntiles ? mine : 0] == 2;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(is);
        int at = 0;
        if (lane == 0 && m) at = atomicAdd(&nflagged, __builtin_popcountll(m));
        at = __builtin_amdgcn_readfirstlane(at);
        if (is) flagged[at + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = mine;
    }
    __syncthreads();
    const int nfl = nflagged;
    for (;;) {
        int pick = 0;
        if (lane == 0) pick = atomicAdd(&next_flagged, 1);
        pick = __builtin_amdgcn_readfirstlane(pick);
        if (pick >= nfl) break;
        const unsigned tile = flagged[pick];
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / ((unsigned)tiles_x * tiles_y);
        const int tx0 = tx * 64, ty0 = ty * TH;
        const TileMasks<TH> *g = reinterpret_cast<const TileMasks<TH> *>(ws.masks) + tile;
        unsigned long long pd = 0;
        if (lane < TH) {
            pd = g->pend[lane];
            tm.row[lane] = g->row[lane];
        }
        tm.col[lane] = g->col[lane];
        // the list: lane r holds row r's pending bits; every lane appends its row's cells behind the rows before it
        const unsigned long long pd0 = pd;
        const int mycount = __builtin_popcountll(pd);
        int before = 0;                                        // exclusive prefix sum over the lanes (6 DPP-free steps: shuffles)
        {
            int acc = mycount;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const int up = __shfl_up(acc, off, kWave);
                acc += lane >= off ? up : 0;
            }
            before = acc - mycount;
        }
        const int n = __shfl(before + mycount, kWave - 1, kWave);
        for (int k = before; pd; pd &= pd - 1, k++) hole_list[k] = (unsigned short)((lane << 6) | __builtin_ctzll(pd));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // Beyond the tile a walk's answer depends on the hole's ROW (left, right) or COLUMN (up) only: the nearest non-zero
        // cell of that row / column outside the tile.  One walk per row side and per column that holds a pending hole
        // (lanes: TH rows left, TH rows right; 64 columns up; both directions of a lane advance together), kept in LDS --
        // not one per hole: the uncovered band a camera pan of 18 px leaves along two image edges is 1000+ pending holes
        // per border tile, and walked per hole that band cost 110 us on a 115 us call (profiles/r04_proj_small_pans.txt).
        {
            // Which walks?  Row r (lanes 0 .. TH-1 hold its pending bits pd0 and its non-zero mask): left iff its first
            // pending hole has no non-zero cell before it in the tile, right iff its last one has none behind it (<=: a hole
            // with a NEGATIVE depth sum is a non-zero cell itself).  Column c:
            // up iff it holds a pending hole and no non-zero cell above the tile's FIRST pending row (conservative: the
            // column's own first pending row may lie lower).
            const unsigned long long rowm = lane < TH ? tm.row[lane] : 0ull;
            const bool has = pd0 != 0;
            const unsigned long long need_l = __builtin_amdgcn_ballot_w64(has && (rowm == 0 || __builtin_ctzll(pd0) <= __builtin_ctzll(rowm)));
            const unsigned long long need_r = __builtin_amdgcn_ballot_w64(has && (rowm == 0 || __builtin_clzll(pd0) <= __builtin_clzll(rowm)));
            const unsigned long long rows_pending = __builtin_amdgcn_ballot_w64(has);
            unsigned lo32 = (unsigned)pd0, hi32 = (unsigned)(pd0 >> 32);           // columns with a pending hole: OR over the rows
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                lo32 |= (unsigned)__shfl_xor((int)lo32, off, kWave);
                hi32 |= (unsigned)__shfl_xor((int)hi32, off, kWave);
            }
            const unsigned long long cols_pending = ((unsigned long long)hi32 << 32) | lo32;
            const int first_row = rows_pending ? __builtin_ctzll(rows_pending) : 0;
            const bool col_wanted = ((cols_pending >> lane) & 1) != 0 && (tm.col[lane] & ((1u << first_row) - 1u)) == 0;
            const int r = lane % TH, side = lane / TH;     // side 0: left of row r, 1: right (TH = 16: lanes 32 .. 63 idle)
            const bool row_wanted = side < 2 && (((side == 0 ? need_l : need_r) >> r) & 1) != 0;
            const int64_t rowbase = (int64_t)b * tiles_x * H + ty0 + r;
            TableWalk tw[3] = {side == 0 ? table_walk(row_wanted, ws.left + rowbase, H, tx - 1, -1, -1)
                                         : table_walk(row_wanted, ws.right + rowbase, H, tx + 1, +1, tiles_x),
                               table_walk(false, ws.up, 0, 0, -1, -1),
                               table_walk(col_wanted, ws.up + (int64_t)b * tiles_y * W + tx0 + lane, W, ty - 1, -1, -1)};
            table_walks(tw);
            if (side < 2) edge.lr[lane] = tw[0].res;       // [r]: left of row r, [TH + r]: right
            edge.up[lane] = tw[2].res;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const float *cn = count + b * scb;
        float *o = out + b * s1b;
        constexpr int kPer = 4;                                // holes per lane in flight: one round trip for their counts and values
        for (int i0 = 0; i0 < n; i0 += kPer * kWave) {
            int gxs[kPer], gys[kPer], found[kPer];
            bool live[kPer];
            float lt[kPer], rt[kPer], ut[kPer], vl[kPer][2], vr[kPer][2], vu[kPer][2], self[kPer][2];
#pragma unroll
            for (int k = 0; k < kPer; k++) {
                const int i = i0 + k * kWave + lane;
                live[k] = i < n;
                if (i0 + k * kWave >= n) continue;             // (wave-uniform) nothing in this slot: no loads either
                const int cell = live[k] ? hole_list[i] : 0, hx = cell & 63, hy = cell >> 6;
                const int gx = tx0 + hx, gy = ty0 + hy;
                const TileWalk w = tile_walk_masks(tm.row[hy], tm.col[hx], hy, hx, tx == 0, tx == tiles_x - 1, ty == 0);
                // beyond the tile (-2): the row's / column's answer from above; -1: the walk ran into the image border
                const int lo = w.l >= 0 ? tx0 + w.l : (w.l == -2 ? edge.lr[hy] : -1);
                const int ro = w.r >= 0 ? tx0 + w.r : (w.r == -2 ? edge.lr[TH + hy] : -1);
                const int uo = w.u >= 0 ? ty0 + w.u : (w.u == -2 ? edge.up[hx] : -1);
                // a walk that found nothing ends at the border cell (column 0 / W-1, row 0): its flag is 0, but the reference
                // still multiplies that cell's value by it -- keep the operand identical.  Counts and values are requested
                // together (the values do not depend on the counts).
                const int lc = lo >= 0 ? lo : 0, rc = ro >= 0 ? ro : W - 1, ur = uo >= 0 ? uo : 0;
                // (NOTHING below may look at a loaded value before the last slot's requests are out: `lo >= 0 ? cl : 0` here
                // made every slot wait for its counts -- one round trip per 64 holes, 35 of the kernel's 40 us under a pan)
                lt[k] = cn[(int64_t)gy * sch + lc];
                rt[k] = cn[(int64_t)gy * sch + rc];
                ut[k] = cn[(int64_t)ur * sch + gx];
                found[k] = (lo >= 0 ? 1 : 0) | (ro >= 0 ? 2 : 0) | (uo >= 0 ? 4 : 0);
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const float *pl = o + c * s1c;
                    vl[k][c] = pl[(int64_t)gy * s1h + lc];
                    vr[k][c] = pl[(int64_t)gy * s1h + rc];
                    vu[k][c] = pl[(int64_t)ur * s1h + gx];
                    self[k][c] = pl[(int64_t)gy * s1h + gx];
                }
                gxs[k] = gx;
                gys[k] = gy;
            }
            // (reads touch cells with a non-zero count -- or times 0 --, writes cells with count <= 0: no ordering needed)
#pragma unroll
            for (int k = 0; k < kPer; k++) {
                if (!live[k]) continue;
                // the counts the walks stopped at (0 when they ran into the image border)
                const float l = (found[k] & 1) ? lt[k] : 0.0f, r = (found[k] & 2) ? rt[k] : 0.0f, u = (found[k] & 4) ? ut[k] : 0.0f;
                if (l + r + u + 0.0f <= 0.0f) continue;
#pragma unroll
                for (int c = 0; c < 2; c++)
                    o[c * s1c + (int64_t)gys[k] * s1h + gxs[k]] = fill_value(l, r, u, vl[k][c], vr[k][c], vu[k][c], self[k][c]);
            }
        }
        __builtin_amdgcn_wave_barrier();                       // the masks and the list are reused by the wave's next tile
    }
    __syncthreads();                                           // (the list is rebuilt by the next span)
    }
}
