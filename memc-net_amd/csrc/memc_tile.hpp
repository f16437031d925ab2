// memc_tile.hpp -- LDS tiling machinery shared by the vectorised (16 B per lane) kernels.
//
// What MI355X wants from these operators (measured, DESIGN.md "Kernel ladder"): a CU retires roughly one
// wave-level vector-memory instruction per ~30 clocks whatever its width, so 4-byte-per-lane accesses top out
// near 2 TB/s while 16-byte ones reach the copy rate.  Hence:
//   * every streamed tensor (flow, filter taps, outputs, gradients) moves as dwordx4 -- one lane owns FOUR
//     consecutive sites of a row;
//   * the data-dependent gathers (and scatters) never touch global memory: the source window of a workgroup's
//     output tile -- its bounding box under the tile's own flow vectors -- is staged into LDS with coalesced
//     dwordx4 loads and gathered from there with ds_read_b128 (DS is a separate, much faster pipe);
//   * sites whose window falls outside the staged region (box larger than the LDS budget: violent or
//     discontinuous motion) fall back to global gathers lane by lane: slower, never wrong.
//
// LDS image layout: one float4 per pixel = up to four channels of that pixel ("pixel quad"), row pitch
// kPitch pixels, so a single ds_read_b128 returns all (<= 4) channels of one tap.  Columns are XOR-swizzled in
// their low four bits with bits 2..5 (col ^= (col >> 2) & 15): lanes own 4-pixel-strided sites, which
// unswizzled would put the 16 lanes of a ds_read_b128 group on 4 bank slots (4-way conflict); the swizzle
// is a bijection of every aligned 64-column span that spreads them over all 16 slots.
#pragma once

#include "memc_common.hpp"

namespace memc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// A quad in GLOBAL memory: four consecutive floats at ANY 4-byte boundary.  global_load / global_store_dwordx4 need dword
// alignment only (the HSA queues run in unaligned-access mode and the compiler selects the 16-byte instruction for this
// type), so the tiled kernels serve views whose base or strides are not multiples of four elements as well (round 5;
// until then such a view fell back to the scalar kernels: 13-41x slower for the scattering passes).  An aligned quad
// costs nothing extra.  LDS quads stay f32x4 (ds_read_b128 wants 16 bytes).
typedef f32x4 f32x4u __attribute__((aligned(4)));

// threadIdx.x through an opaque asm: values derived from it are recomputed where they are used instead of
// being hoisted out of a persistent kernel's tile loop and kept (or spilled) for the whole kernel -- a spill
// reload is a scratch LOAD, and waiting for it means waiting for every store and atomic issued before it.
__device__ __forceinline__ unsigned tid_now()
{
    unsigned t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

__device__ __forceinline__ f32x4 ld_stream4(const float *p)
{
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4u *>(p));
}
__device__ __forceinline__ void st_stream4(float *p, f32x4 v)
{
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4u *>(p));
}
__device__ __forceinline__ f32x4 ld_cached4(const float *p) { return *reinterpret_cast<const f32x4u *>(p); }
__device__ __forceinline__ void st_cached4(float *p, f32x4 v) { *reinterpret_cast<f32x4u *>(p) = v; }

// Wave-uniform base + 32-bit per-lane BYTE offset: compiles to the saddr form (global_load ... v_off, s[base]),
// so a kernel that touches many planes at the same site keeps ONE offset register instead of a 64-bit pointer
// per plane (sixteen tap planes = 32 VGPRs of addresses, which the big kernels then spill).
// (the empty asm pins the base in an SGPR pair at the point of use: otherwise LLVM re-associates
// (base + k * plane) + offset into per-lane 64-bit pointers again and hoists them out of loops.  The pointer
// is rebuilt in the GLOBAL address space: an inline asm hides where it came from, and a generic pointer would
// compile to flat_load/flat_store, which also count on lgkmcnt -- every LDS barrier would then wait for them.)
#define MEMC_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ uintptr_t pin_sgpr(const void *ubase)
{
    uintptr_t u = reinterpret_cast<uintptr_t>(ubase);
    asm volatile("" : "+s"(u));
    return u;
}
// The offset goes through an asm as well so that its zero-extension sits in the SAME basic block as the access
// (instruction selection is per block: a zext CSE'd into an earlier block is just a 64-bit VGPR pair to it, and
// the access falls back to v_lshl_add_u64 + a 64-bit vaddr).
__device__ __forceinline__ uintptr_t addr_u(const void *ubase, unsigned byte_off)
{
    asm volatile("" : "+v"(byte_off));
    return pin_sgpr(ubase) + byte_off;
}
__device__ __forceinline__ f32x4 ld_stream4_u(const float *ubase, unsigned byte_off)
{
    return __builtin_nontemporal_load(reinterpret_cast<const MEMC_GLOBAL f32x4u *>(addr_u(ubase, byte_off)));
}
__device__ __forceinline__ void st_stream4_u(float *ubase, unsigned byte_off, f32x4 v)
{
    __builtin_nontemporal_store(v, reinterpret_cast<MEMC_GLOBAL f32x4u *>(addr_u(ubase, byte_off)));
}
__device__ __forceinline__ f32x4 ld_cached4_u(const float *ubase, unsigned byte_off)
{
    return *reinterpret_cast<const MEMC_GLOBAL f32x4u *>(addr_u(ubase, byte_off));
}
// plain (cached) accesses through the same addressing
__device__ __forceinline__ MEMC_GLOBAL float *at_u(float *ubase, unsigned byte_off)
{
    return reinterpret_cast<MEMC_GLOBAL float *>(addr_u(ubase, byte_off));
}

// RAGGED rows (round 5): a width that is not a multiple of four.  The last quad of a row then holds W % 4 sites, and a 16-byte
// access there would run past the row -- past the tensor, in its last row.  The load is moved left so that it ENDS at the
// row's end and the registers are rotated back (the sites past the row read as `pad`); the store writes the sites inside
// the row one by one.  Only the kernels' RAG instantiations pay for this (a wave-uniform branch: only the waves that hold a
// row's last quad take it); widths that are multiples of four run the code they always ran.
__device__ __forceinline__ int tail_shift(int x, int W) { return (x < W && x + 4 > W) ? x + 4 - W : 0; }   // sites past the row
__device__ __forceinline__ f32x4 tail_fix(const f32x4 &v, int r, float pad)
{
    if (__builtin_amdgcn_ballot_w64(r != 0) == 0) return v;
    f32x4 o;
    o[0] = r == 0 ? v[0] : r == 1 ? v[1] : r == 2 ? v[2] : v[3];
    o[1] = r == 0 ? v[1] : r == 1 ? v[2] : r == 2 ? v[3] : pad;
    o[2] = r == 0 ? v[2] : r == 1 ? v[3] : pad;
    o[3] = r == 0 ? v[3] : pad;
    return o;
}
template <bool STREAM>
__device__ __forceinline__ void st_tail4(float *p, const f32x4 &v, int r)
{
    if (__builtin_amdgcn_ballot_w64(r != 0) == 0) {
        if (STREAM) st_stream4(p, v); else st_cached4(p, v);
        return;
    }
    if (r == 0) {
        if (STREAM) st_stream4(p, v); else st_cached4(p, v);
    } else {
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (j < 4 - r) p[j] = v[j];
    }
}

__device__ __forceinline__ int swz_col(int c) { return c ^ ((c >> 2) & 15); }

// Kernels that address a plane as wave-uniform base + 32-bit byte offset (ld_stream4_u & co.) need every in-plane
// byte offset (h - 1) * row_stride + w to fit 32 bits; a view with a gigantic row stride takes the 64-bit kernels.
inline bool plane_fits_u32(int w, int h, std::initializer_list<long> row_strides)
{
    for (long s : row_strides)
        if (((long long)(h > 0 ? h - 1 : 0) * s + w) * 4LL >= (1LL << 32)) return false;
    return true;
}

// Quad path preconditions: the width a multiple of 4 sites (a lane owns four consecutive sites of a row).  Strides and
// base pointers may be anything (f32x4u above); rounds 1-4 also asked for 16-byte aligned bases and strides.
inline bool vec4_ok(int w, std::initializer_list<long> strides, std::initializer_list<const void *> ptrs)
{
    (void)strides;
    (void)ptrs;
    return w % 4 == 0;
}

constexpr int kStageItsMax = 3;                 // float4 staging slots per lane (see stage_slots)

// Tile geometry: 256 threads, LX lanes per tile row, 4 sites per lane.
// CAP: the LDS budget in pixel quads.  3072 px * 16 B = 48 KiB -> 3 workgroups per CU (the 4x4-window kernels, whose
// boxes need it); the 2x2-footprint kernels stage smaller boxes and trade budget for workgroups per CU.
template <int LX, int CAP = 3072, int NT = 256>
struct TileGeom {
    static constexpr int kThreads = NT;                // 256: 64 x 16 tiles; 512: 64 x 32 (less box halo per site)
    static constexpr int kTW = 4 * LX;                 // tile width  (sites)
    static constexpr int kTH = kThreads / LX;          // tile height (sites)
    static constexpr int kPitch = kTW + 32;            // LDS row pitch in pixels; multiple of 16 (swizzle span)
    static constexpr int kCapPx = CAP;
    static constexpr int kRows = kCapPx / kPitch;      // staged rows that fit
    static_assert(kPitch % 16 == 0, "swizzle needs a pitch that is a multiple of 16 pixels");
};

// LDS carve: [0, kCapPx*16) pixel quads, then 16 ints of per-wave bounding boxes.
template <int LX, int CAP = 3072>
constexpr int tile_lds_bytes() { return TileGeom<LX, CAP>::kCapPx * 16 + 64; }

// min over the 64 lanes of a wave, returned in every lane (wave-uniform)
__device__ __forceinline__ int wave_min_i32(int v)
{
    // row_shr:1,2,4,8 inside each row of 16 lanes; lanes without a source keep their own value
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    // lane 15 of every row now holds the row minimum
    const int a = __builtin_amdgcn_readlane(v, 15), b = __builtin_amdgcn_readlane(v, 31);
    const int c = __builtin_amdgcn_readlane(v, 47), d = __builtin_amdgcn_readlane(v, 63);
    return min(min(a, b), min(c, d));
}

struct Region {
    int x0, y0;      // image coordinates of LDS pixel (0,0); x0 % 4 == 0
    int w, h;        // staged extent (w % 4 == 0); 0 when nothing is staged
    int pitch;       // LDS row pitch of the pixel-quad image, a multiple of 16 pixels (swizzle span)
    int wimg;        // 0, or the image's width when that is NOT a multiple of four: the staged box's last quad may then
                     // reach past the row's end and the staging loads take care (tile_stage_load_planes); set by the kernel
    __device__ __forceinline__ bool covers(int cmin, int cmax, int rmin, int rmax) const
    {
        return cmin >= x0 && cmax < x0 + w && rmin >= y0 && rmax < y0 + h;
    }
};

// Workgroup-wide bounding box of the clamped source windows [cmin,cmax] x [rmin,rmax] of all valid sites,
// fitted to the LDS budget.  Each lane passes the box of its own (up to four) valid sites, or an empty box
// (cmin > cmax).  One __syncthreads.
// DYN: the pixel-quad image uses the narrowest pitch (multiple of 16) that holds the box, which buys rows:
// 96 -> 32 rows, 80 -> 38, 64 -> 48.  Kernels that also keep fixed-shape accumulator planes pass DYN = false.
template <int LX, bool DYN = false, int CAP = 3072, int NT = 256>
__device__ __forceinline__ Region tile_region(int cmin, int cmax, int rmin, int rmax, int tile_x0, int tile_y0,
                                              int *bb /* 4 ints per wave in LDS */)
{
    using G = TileGeom<LX, CAP, NT>;
    static_assert(CAP <= kStageItsMax * NT * 4, "the staging slots cover the budget");
    // wave-level reduction on the VALU/SALU only (DPP row shifts, then the four row leaders through readlane);
    // ds_bpermute-based shuffles would put ~700 clocks of LDS round trips on the tile's critical chain
    cmin = wave_min_i32(cmin);
    cmax = -wave_min_i32(-cmax);
    rmin = wave_min_i32(rmin);
    rmax = -wave_min_i32(-rmax);
    const unsigned tid = tid_now();
    const int wave = tid / kWave;
    if ((tid & (kWave - 1)) == 0) {
        bb[wave * 4 + 0] = cmin;
        bb[wave * 4 + 1] = cmax;
        bb[wave * 4 + 2] = rmin;
        bb[wave * 4 + 3] = rmax;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < G::kThreads / kWave; w++) {
        cmin = min(cmin, bb[w * 4 + 0]);
        cmax = max(cmax, bb[w * 4 + 1]);
        rmin = min(rmin, bb[w * 4 + 2]);
        rmax = max(rmax, bb[w * 4 + 3]);
    }
    cmin = __builtin_amdgcn_readfirstlane(cmin);
    cmax = __builtin_amdgcn_readfirstlane(cmax);
    rmin = __builtin_amdgcn_readfirstlane(rmin);
    rmax = __builtin_amdgcn_readfirstlane(rmax);
    Region r;
    r.wimg = 0;
    if (cmin > cmax) {          // no valid site in this tile
        r.x0 = r.y0 = r.w = r.h = 0;
        r.pitch = G::kPitch;
        return r;
    }
    int x0 = cmin & ~3, w = (cmax | 3) + 1 - x0;
    int y0 = rmin, h = rmax + 1 - y0;
    if (w > G::kPitch) {        // clip around the tile centre, keep 4-alignment
        const int lo = x0, hi = x0 + w - G::kPitch;
        x0 = min(max((tile_x0 + G::kTW / 2 - G::kPitch / 2) & ~3, lo), hi);
        w = G::kPitch;
    }
    const int pitch = DYN ? ((w + 15) & ~15) : G::kPitch;
    const int rows = DYN ? G::kCapPx / pitch : G::kRows;
    if (h > rows) {
        const int lo = y0, hi = y0 + h - rows;
        y0 = min(max(tile_y0 + G::kTH / 2 - rows / 2, lo), hi);
        h = rows;
    }
    r.x0 = x0; r.y0 = y0; r.w = w; r.h = h; r.pitch = pitch;
    return r;
}

// Stage NCH (1..4) channel planes of the region into LDS pixel quads (missing channels -> 0).
// Thread layout: 8 region rows x 32 lanes per pass; lane q loads the float4 at columns 4q..4q+3 of every
// channel (coalesced: a region row is one contiguous run) and writes four swizzled pixel quads.
// Rows are handled in batches of 32 (4 passes): all loads of a batch are issued before the first LDS write,
// and they are unconditional -- lanes/rows outside the region read the plane's first element instead -- so
// that no load result becomes a phi (see the note in fi_fwd_tiled_fs4) and one latency covers the batch.
// ---------------------------------------------------------------------------------------------------------
// Bands: when a tile's source box does not fit the LDS budget (large or discontinuous motion) the pure-gather
// kernels do not drop to scalar code; they sweep the box in overlapping sub-boxes ("bands"), one stage ->
// gather round per band.  Bands overlap by >= 3 rows / columns, so every 4x4 (or 2x2) window lies wholly inside
// at least one band.  At most kMaxBands rounds; what is still uncovered after that (pathological motion) goes
// to the scalar path.  Ordinary tiles have exactly one band and pay nothing for this.
// ---------------------------------------------------------------------------------------------------------
struct BBox {
    int x0, w, y0, h;          // unclipped box, x0 % 4 == 0, w % 4 == 0; w == 0: no valid site
};

constexpr int kMaxBands = 6;

struct Bands {
    int nbx, nby, n;           // band grid, n = min(nbx * nby, kMaxBands)
    int bw, bh, pitch;         // band extent and LDS pitch
    int sx, sy;                // band steps
};

// Workgroup-wide box of the lanes' boxes (DPP / readlane reduction + one barrier), not clipped.
template <int LX, int NT = 256>
__device__ __forceinline__ BBox tile_bbox(int cmin, int cmax, int rmin, int rmax, int *bb /* 4 ints per wave in LDS */)
{
    using G = TileGeom<LX, 3072, NT>;
    cmin = wave_min_i32(cmin);
    cmax = -wave_min_i32(-cmax);
    rmin = wave_min_i32(rmin);
    rmax = -wave_min_i32(-rmax);
    const unsigned tid = tid_now();
    const int wave = tid / kWave;
    if ((tid & (kWave - 1)) == 0) {
        bb[wave * 4 + 0] = cmin;
        bb[wave * 4 + 1] = cmax;
        bb[wave * 4 + 2] = rmin;
        bb[wave * 4 + 3] = rmax;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < G::kThreads / kWave; w++) {
        cmin = min(cmin, bb[w * 4 + 0]);
        cmax = max(cmax, bb[w * 4 + 1]);
        rmin = min(rmin, bb[w * 4 + 2]);
        rmax = max(rmax, bb[w * 4 + 3]);
    }
    // the box is workgroup-uniform: keep it (and everything derived from it) in scalar registers
    cmin = __builtin_amdgcn_readfirstlane(cmin);
    cmax = __builtin_amdgcn_readfirstlane(cmax);
    rmin = __builtin_amdgcn_readfirstlane(rmin);
    rmax = __builtin_amdgcn_readfirstlane(rmax);
    BBox b;
    if (cmin > cmax) {
        b.x0 = b.y0 = b.w = b.h = 0;
    } else {
        b.x0 = cmin & ~3;
        b.w = (cmax | 3) + 1 - b.x0;
        b.y0 = rmin;
        b.h = rmax + 1 - rmin;
    }
    return b;
}

// DYN = false: fixed 96 x 32 geometry (kernels whose accumulator planes have a compile-time pitch)
template <int LX, bool DYN = true, int CAP = 3072>
__device__ __forceinline__ Bands make_bands(const BBox &b)
{
    using G = TileGeom<LX, CAP>;
    Bands d;
    d.bw = min(b.w, G::kPitch);
    d.pitch = DYN ? max((d.bw + 15) & ~15, 16) : G::kPitch;
    const int rows = G::kCapPx / d.pitch;
    d.bh = min(b.h, rows);
    d.sx = (G::kPitch - 4) & ~3;                               // 92: overlap 4 columns
    d.sy = rows - 3;                                           // overlap 3 rows
    d.nbx = b.w > d.bw ? (b.w - d.bw + d.sx - 1) / d.sx + 1 : 1;
    d.nby = b.h > d.bh ? (b.h - d.bh + d.sy - 1) / d.sy + 1 : 1;
    d.n = b.w == 0 ? 1 : min(d.nbx * d.nby, kMaxBands);
    return d;
}

__device__ __forceinline__ Region band_region(const BBox &b, const Bands &d, int i, int wimg = 0)
{
    Region r;
    const int bi = i % d.nbx, bj = i / d.nbx;
    r.x0 = min(b.x0 + bi * d.sx, b.x0 + b.w - d.bw);
    r.y0 = min(b.y0 + bj * d.sy, b.y0 + b.h - d.bh);
    r.w = d.bw;
    r.h = d.bh;
    r.pitch = d.pitch;
    r.wimg = wimg;             // (the image's width when it is ragged and the kernel stages ragged-safely, else 0: see Region)
    return r;
}

// Work split of the staging: the box is r.h rows of r.w / 4 float4 columns; the 256 lanes take float4 slots
// round-robin in row-major order, kStageIts = 3 slots each (3 * 256 * 4 px = the whole 3072-pixel budget).
// All loads are issued before the first LDS write and are unconditional -- slots past the end of the box read
// the plane's first element -- so that no load result becomes a phi (see fi_fwd_tiled_fs4).
constexpr int kStageIts = kStageItsMax;

// ITS: staging slots per lane -- kStageIts (3: the 3072-pixel budget on 256 lanes) everywhere but in kernels instantiated
// with a larger budget (CAP <= ITS * NT * 4)
template <int ITS = kStageIts>
struct StageSlotT {
    int row[ITS], q[ITS];                  // q = float4 column; row >= r.h marks an empty slot
};
using StageSlot = StageSlotT<>;

template <int NT = 256, int ITS = kStageIts>
__device__ __forceinline__ StageSlotT<ITS> stage_slots(const Region &r)
{
    StageSlotT<ITS> s;
    const int wq = max(r.w >> 2, 1);
    const unsigned tid = tid_now();
    int row = tid / wq, q = tid % wq;          // one run-time division per kernel
    const int drow = NT / wq, dq = NT % wq;
#pragma unroll
    for (int it = 0; it < ITS; it++) {
        s.row[it] = r.w > 0 ? row : r.h;
        s.q[it] = q;
        q += dq;
        row += drow + (q >= wq ? 1 : 0);
        q -= q >= wq ? wq : 0;
    }
    return s;
}

template <int NCH, int ITS = kStageIts>
struct StageRegs {
    f32x4 v[ITS][NCH];
};

// RAG: a ragged image (r.wimg = its width, not a multiple of four): the box's last quad may reach past the row's end.  It
// is loaded to END at the row's end (rs sites further left) and rotated back where it is consumed (tile_stage_store);
// what lies past the row is never gathered: coordinates are clamped.  Kernels instantiate RAG only for such widths.
template <int NCH, bool RAG = false, int ITS = kStageIts>
__device__ __forceinline__ void tile_stage_load_planes(const Region &r, const StageSlotT<ITS> &sl,
                                                       const float *const (&plane)[NCH], const int (&hstride)[NCH],
                                                       StageRegs<NCH, ITS> &sr)
{
#pragma unroll
    for (int it = 0; it < ITS; it++) {
        const bool on = sl.row[it] < r.h;
        const int rs = RAG ? tail_shift(r.x0 + 4 * sl.q[it], r.wimg) : 0;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const float *p = on ? plane[c] + (int64_t)(r.y0 + sl.row[it]) * hstride[c] + r.x0 + 4 * sl.q[it] - rs : plane[c];
            sr.v[it][c] = ld_cached4(p);
        }
    }
}

template <int NCH, bool RAG = false, int ITS = kStageIts>
__device__ __forceinline__ void tile_stage_store(const Region &r, const StageSlotT<ITS> &sl, const StageRegs<NCH, ITS> &sr,
                                                 f32x4 *tile)
{
#pragma unroll
    for (int it = 0; it < ITS; it++) {
        if (sl.row[it] < r.h) {
            f32x4 *dst = tile + sl.row[it] * r.pitch;
            f32x4 v[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) v[c] = sr.v[it][c];
            if (RAG) {                         // the row's last quad, rotated back (see the loads)
                const int rs = tail_shift(r.x0 + 4 * sl.q[it], r.wimg);
#pragma unroll
                for (int c = 0; c < NCH; c++) v[c] = tail_fix(v[c], rs, 0.0f);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f32x4 px = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NCH; c++) px[c] = v[c][i];
                dst[swz_col(4 * sl.q[it] + i)] = px;
            }
        }
    }
}

template <int NCH, bool RAG = false, int ITS = kStageIts>
__device__ __forceinline__ void tile_stage_load(const Region &r, const StageSlotT<ITS> &sl, const float *plane0,
                                                int64_t cstride, int hstride, StageRegs<NCH, ITS> &sr)
{
    const float *plane[NCH];
    int hs[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        plane[c] = plane0 + c * cstride;
        hs[c] = hstride;
    }
    tile_stage_load_planes<NCH, RAG, ITS>(r, sl, plane, hs, sr);
}

template <int LX, int NCH, int NT = 256, bool RAG = false, int ITS = kStageIts>
__device__ __forceinline__ void tile_stage_planes(const Region &r, const float *const (&plane)[NCH],
                                                  const int (&hstride)[NCH], f32x4 *tile)
{
    static_assert(TileGeom<LX>::kCapPx <= kStageIts * 256 * 4, "three float4 slots per lane cover the budget");
    const StageSlotT<ITS> sl = stage_slots<NT, ITS>(r);
    StageRegs<NCH, ITS> sr;
    tile_stage_load_planes<NCH, RAG, ITS>(r, sl, plane, hstride, sr);
    tile_stage_store<NCH, RAG, ITS>(r, sl, sr, tile);
}

// channel planes of ONE tensor: plane c = plane0 + c * cstride, common row stride
template <int LX, int NCH, int NT = 256, bool RAG = false, int ITS = kStageIts>
__device__ __forceinline__ void tile_stage(const Region &r, const float *plane0, int64_t cstride, int hstride,
                                           f32x4 *tile)
{
    const float *plane[NCH];
    int hs[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        plane[c] = plane0 + c * cstride;
        hs[c] = hstride;
    }
    tile_stage_planes<LX, NCH, NT, RAG, ITS>(r, plane, hs, tile);
}

// ---------------------------------------------------------------------------------------------------------
// LDS-privatised scatter.  The forward splat and every backward pass scatter fp32 adds to data-dependent
// cells.  Issued straight to global memory those atomics arrive in ragged, misaligned runs and the chip
// retires only ~50 G of them per second (measured on the reference-shaped kernels); fully coalesced ones run
// at ~300 G/s (tools/probes).  So a workgroup accumulates its tile's contributions in LDS planes covering the
// bounding box of its targets (ds_add_f32), then flushes each non-zero cell once with row-coalesced global
// atomics -- still atomics, because neighbouring tiles' boxes overlap.  Targets outside the box (clipped to
// the LDS budget) go to global memory directly.
//
// Accumulator layout: NP planes of kAccRows x kAccPitch floats; the odd pitch staggers rows over the banks.
// ---------------------------------------------------------------------------------------------------------
template <int LX>
struct AccGeom {
    using G = TileGeom<LX>;
    static constexpr int kPitch = G::kPitch + 1;
    static constexpr int kRows = G::kRows;
    static constexpr int kPlane = kPitch * kRows;      // floats per plane
};

__device__ __forceinline__ void lds_add_f32(float *p, float v) { (void)unsafeAtomicAdd(p, v); }   // ds_add_f32

// ---------------------------------------------------------------------------------------------------------
// fp64 accumulation in LDS.  Measured on MI355X (tools/probes, `run_probe.py lds`): ds_add_f32 retires 0.33
// lane-ops per clock per CU whatever the access pattern (an emulated path), ds_add_f64 6.7 conflict-free --
// twenty times faster.  The backward passes therefore accumulate their scattered image gradient in DOUBLE in
// LDS (also more accurate than the reference's fp32 atomics) and round to fp32 once, when a cell is flushed.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_add_f64(double *p, double v) { (void)unsafeAtomicAdd(p, v); }   // ds_add_f64

// ---------------------------------------------------------------------------------------------------------
// Transposed fp64 accumulator plane.  Measured (tools/probes/run_probe.py ldspat): a ds_add_f64 costs ~8.8 clk
// per wave-instruction whatever the number of active lanes, and the LDS serialises it over groups of 16 lanes
// that must hit 16 different 8-byte slots (mod 128 B) to stay at that cost -- with one lane per pixel quad a
// group is 16 quads of one tile row, i.e. cells 4 apart (+ flow), which in a row-major plane is 2..4 passes
// (16 / 28 clk measured).  Here column c of a row is stored at (c & 3) * 32 + (c >> 2): cells 4 apart are
// adjacent, rows are 128 cells (a multiple of 16 slots) apart, so lane i of a group lands in slot
// (i + (offset >> 2)) mod 16 -- distinct unless the flow folds over itself.
// ---------------------------------------------------------------------------------------------------------
struct AccT {
    static constexpr int kPitch = 128, kRows = 32, kPlane = kPitch * kRows;      // 32 KiB of doubles
    static constexpr int kMaxW = 128;
};
__device__ __forceinline__ int acct_col(int c) { return ((c & 3) << 5) | (c >> 2); }

template <int NPLANES = 1>
__device__ __forceinline__ void acct_zero(double *acc)
{
    f32x4 *p = reinterpret_cast<f32x4 *>(acc);
    const unsigned tid = tid_now();
#pragma unroll
    for (int i = 0; i < NPLANES * AccT::kPlane / 2 / 256; i++) p[tid + i * 256] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Adds every non-zero cell of the region to dst (row-coalesced global atomics) and leaves the plane zeroed.
// A wave takes rows wave, wave + 4, ...; four rows (eight cells per lane) are read before any is consumed so
// that the LDS latency is paid once per batch, not once per cell.
// STORAGE_ORDER: lanes walk the plane in the order it is stored (lane l reads slots l and l + 64 of a row, i.e.
// columns 4 * (slot & 31) + (slot >> 5)): conflict-free LDS reads and zero-writes, global atomics 16 bytes apart
// (four cache lines per wave-instruction instead of two).  Otherwise lanes walk columns: row-coalesced atomics,
// 4-way bank conflicts on the 8-byte LDS accesses.  Measured equal (FI backward 1822 vs 1820 us, Interpolation
// backward 541 vs 544): the flush is bound by neither; the coalesced walk is the default.
template <bool PLAIN_STORE = false, bool STORAGE_ORDER = false>
__device__ __forceinline__ void acct_flush_zero(const Region &r, double *acc, float *dst, int hstride)
{
    const unsigned tid = tid_now();
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    const int s0 = STORAGE_ORDER ? lane : acct_col(lane), s1 = STORAGE_ORDER ? lane + kWave : acct_col(lane + kWave);
    const int col0 = STORAGE_ORDER ? 4 * (lane & 31) + (lane >> 5) : lane;
    const int col1 = STORAGE_ORDER ? col0 + 2 : lane + kWave;
    const bool ok0 = col0 < r.w, ok1 = col1 < r.w;
    // wave-uniform base + 32-bit byte offsets
    const uintptr_t base = pin_sgpr(dst);
    unsigned off = 4u * (unsigned)((r.y0 + wave) * hstride + r.x0 + col0);
    const unsigned step = 16u * (unsigned)hstride;          // four rows down
    const unsigned dcol = 4u * (unsigned)(col1 - col0);
#pragma unroll 1
    for (int row0 = wave; row0 < r.h; row0 += 16, off += 4u * step) {
        double val[4][2];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int row = min(row0 + 4 * u, AccT::kRows - 1);
            val[u][0] = acc[row * AccT::kPitch + s0];
            val[u][1] = acc[row * AccT::kPitch + s1];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int row = row0 + 4 * u;
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const float f = (float)val[u][hh];
                if (row < r.h && (hh ? ok1 : ok0) && f != 0.0f) {
                    acc[row * AccT::kPitch + (hh ? s1 : s0)] = 0.0;
                    MEMC_GLOBAL float *q = reinterpret_cast<MEMC_GLOBAL float *>(base + (off + u * step + hh * dcol));
                    if (PLAIN_STORE) *q = f; else (void)__builtin_amdgcn_global_atomic_fadd_f32(q, f);
                }
            }
        }
    }
}

}  // namespace memc
