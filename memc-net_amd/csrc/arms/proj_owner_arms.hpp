// arms/proj_owner_arms.hpp -- MEASUREMENT BUILD ONLY: the owner kernels that proj_owner4 superseded, kept as A/B arms of
// flow_projection.hip (tools/bench_ops.py --proj-variants, tools/trace_kernel.py proj*).  Textually included by
// flow_projection.hip under MEMC_MEASURE, behind the helpers they share with the product kernels; never part of
// libmemc_hip.so.
//   proj_owner   (round 1): 64 x 16 tile, per-source locate + window test + three ds_add_f64 under exec masks
//   proj_owner2  (round 2): the scan as a test, hits compacted per wave into LDS rings, three planes
//   proj_owner3  (round 2): persistent, the next tile's fy prefetched into a second register set -- LOST

#ifndef MEMC_MEASURE
#error "measurement arms: build with -DMEMC_MEASURE (make measure)"
#endif

// ---- part A: proj_owner2, proj_owner3 (need TileSummary / FillWs, defined above the include) ----
#ifdef MEMC_PROJ_ARMS_PART_A
// ---- rounds 1-3: per-tile summaries by LDS atomics / row reductions (superseded by the masks of proj_fill.hpp) ----
template <int TH>
struct TileSummary {              // LDS
    int col_last[64], row_first[TH], row_last[TH];
};

// (the summary helpers take the thread index as an argument: a kernel that rebuilds it late -- proj_owner4 -- must not
// keep threadIdx.x alive in a register just for them)
template <int TH>
__device__ __forceinline__ void summary_init(TileSummary<TH> &t, int tid = threadIdx.x)
{
    if (tid < 64) t.col_last[tid] = -1;
    if (tid < TH) {
        t.row_first[tid] = INT_MAX;
        t.row_last[tid] = -1;
    }
}

// the lane's four counts at (x .. x+3, y), local coordinates (lx .. lx+3, ly); returns "one of them is a hole".
// A barrier must separate summary_init from this, and this from summary_store.
// min over each group of 16 consecutive lanes (one DPP row), valid in the group's LAST lane
__device__ __forceinline__ int row16_min_i32(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));      // row_shr:1,2,4,8
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    return v;
}

// The lane's four counts at (x .. x+3, y), local coordinates (lx .. lx+3, ly); returns "one of them is a hole".
// Lane layout: the 16 lanes of a tile row are 16 CONSECUTIVE lanes (lx = 4 * (lane % 16)) -- a row's first / last
// non-zero column is then a DPP reduction and one plain LDS store by the row's last lane (sixteen lanes bumping
// one LDS word with atomics serialise); the columns' last non-zero rows span waves and stay LDS atomics.
// Converged code only.  A barrier must separate summary_init from this, and this from summary_store.
template <int TH>
__device__ __forceinline__ bool summary_add(TileSummary<TH> &t, bool inb, const f32x4 &c4, int lx, int ly, int x, int y,
                                            int tid = threadIdx.x)
{
    bool hole = false;
    int first = INT_MAX, last = -1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!inb) continue;
        hole = hole || c4[j] <= 0.0f;                      // what pass 3 fills (my_lib_kernel.cu:1757)
        if (c4[j] != 0.0f) {                               // what stops a walk (:1778-1797)
            atomicMax(&t.col_last[lx + j], y);
            first = min(first, x + j);
            last = max(last, x + j);
        }
    }
    first = row16_min_i32(first);
    last = -row16_min_i32(-last);
    if ((tid & 15) == 15) {                                // one writer per row
        t.row_first[ly] = first;
        t.row_last[ly] = last;
    }
    return hole;
}

template <int TH>
__device__ __forceinline__ void summary_store(const TileSummary<TH> &t, int any_hole, const FillWs &ws, int b, int tx,
                                              int ty, int W, int H, int ntx, int nty, int tid = threadIdx.x)
{
    const int tx0 = tx * 64, ty0 = ty * TH;
    if (tid < 64 && tx0 + (int)tid < W)
        ws.up[((int64_t)b * nty + ty) * W + tx0 + tid] = t.col_last[tid];
    if (tid < TH && ty0 + (int)tid < H) {
        // [b][tx][y]: a tile's rows are one contiguous run (row-major [b][y][tx] made these TH scattered 4-byte stores)
        const int64_t i = ((int64_t)b * ntx + tx) * H + ty0 + tid;
        ws.right[i] = t.row_first[tid] == INT_MAX ? -1 : t.row_first[tid];
        ws.left[i] = t.row_last[tid];
    }
    if (tid == 0) ws.hole[((int64_t)b * nty + ty) * ntx + tx] = any_hole;
}

// ---- measurement build only: proj_owner2 (LDS compaction rings, three planes) and proj_owner3 (persistent) ----
// ABL / TRACE: measurement build only (timing arms, results WRONG for ABL != 0): 1 no scan, 2 no fp64 adds,
// 3 no read-out, 4 no halo loads (own tile only), 5 no loads and no scan; TRACE: per-workgroup phase timestamps.
template <bool DEPTH, int TH, int kReach, int ABL = 0, bool TRACE = false>
__global__ __launch_bounds__(16 * TH) void proj_owner2(
    int W, int H, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t sdb, int sdh, int64_t scb, int sch,
    const float *__restrict__ flow, const float *__restrict__ depth,
    float *__restrict__ count, float *__restrict__ out, int *__restrict__ far_flag, FillWs ws, int sw)
{
    constexpr int NT = 16 * TH, NW = NT / kWave;      // one lane per four owned cells
    constexpr int kPtH = TH + 1, kPlane = kPtH * kPtW;
    constexpr int kScanPadX = kReach + 4;         // dilated tile: columns, kept 4-aligned
    constexpr int kScanW = 64 + 2 * kScanPadX;    // source columns
    constexpr int kScanH = TH + 2 * kReach + 1;   // source rows: [ty0 - kReach - 1, ty0 + TH + kReach)
    constexpr int kCols4 = kScanW / 4, kSlots = kCols4 * kScanH, kIts = (kSlots + NT - 1) / NT;
    constexpr int kRing = 128;                    // entries per wave; at most 63 + 64 wait at any time
    static_assert(kPlane % 2 == 0, "P is zeroed 16 bytes at a time");
    constexpr int kPlanes = ABL == 6 ? 2 : 3;     // (timing arm 6: two planes -> 52 KiB at TH = 32, three workgroups per CU)
    __shared__ __attribute__((aligned(16))) double P[kPlanes * kPlane];
    __shared__ __attribute__((aligned(16))) f32x4 ring[NW * kRing];
    __shared__ TileSummary<TH> sm;                // for the hole filler, when one follows (ws.up != nullptr)

    const TileCoord tc = tile_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, sw);
    if (tc.tx >= tiles_x) return;                 // virtual column of the last stripe
    const int b = tc.b, tx0 = tc.tx * 64, ty0 = tc.ty * TH;
    const int tid = threadIdx.x;
    trace_mark_proj<TRACE>(0);
    summary_init(sm);
    {
        f32x4 *pz = reinterpret_cast<f32x4 *>(P);
        for (int i = tid; i < kPlanes * kPlane / 2; i += NT) pz[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // scan: kCols4 float4 columns x kScanH rows of slots, kIts per lane; all loads first.  Slot -> (row, column) by
    // one division and increments, addresses as wave-uniform base + 32-bit lane offset.
    // Slot `it` of the NT lanes covers scan rows [NT it / kCols4, (NT it + NT - 1) / kCols4]; the tile is rows
    // [kReach + 1, kReach + 1 + TH): the slot is "far" when all of its rows are at least kNearRows away from the
    // tile -- such rows almost never pass the row test, so only their fy is requested up front (fx / depth follow
    // inside the branch if they do): the scan moves several times the tile's own bytes through the CU's 64 B/clk
    // L1 path, which is a bound of its own.
    constexpr int kNearRows = 8;
    auto far_it = [](int it) {
        const int first = NT * it / kCols4, last = (NT * it + NT - 1) / kCols4;
        return last <= kReach + 1 - kNearRows || first >= kReach + 1 + TH + kNearRows;
    };
    const float *flow_b = flow + b * s1b;
    const float *depth_b = DEPTH ? depth + b * sdb : nullptr;
    f32x4 fx[kIts], fy[kIts], dd[kIts];
    int sx[kIts], sy[kIts];
    bool live[kIts];
    int row = tid / kCols4, c4 = tid % kCols4;
#pragma unroll
    for (int it = 0; it < kIts; it++) {
        sx[it] = tx0 - kScanPadX + 4 * c4;
        sy[it] = ty0 - kReach - 1 + row;
        live[it] = row < kScanH && sx[it] >= 0 && sx[it] < W && sy[it] >= 0 && sy[it] < H;   // W % 4 == 0
        if (ABL == 4)                          // (timing arm: the tile's own sources only)
            live[it] = live[it] && (unsigned)(sy[it] - ty0) < (unsigned)TH && (unsigned)(sx[it] - tx0) < 64u;
        // dead slots read the plane's first pixels (unconditional loads)
        const unsigned off = live[it] ? 4u * (unsigned)(sy[it] * s1h + sx[it]) : 0u;
        if (ABL == 5) {
            fx[it] = fy[it] = dd[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            fy[it] = ld_cached4_u(flow_b + s1c, off);
            if (!far_it(it)) {
                fx[it] = ld_cached4_u(flow_b, off);
                if (DEPTH) dd[it] = ld_cached4_u(depth_b, live[it] ? 4u * (unsigned)(sy[it] * sdh + sx[it]) : 0u);
            }
        }
        row += NT / kCols4;                    // the next slot of this lane is NT further on
        c4 += NT % kCols4;
        if (c4 >= kCols4) {
            c4 -= kCols4;
            row++;
        }
    }
    __syncthreads();                           // P is zero
    trace_mark_proj<TRACE>(1);                 // loads issued, P zeroed
    if (TRACE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    trace_mark_proj<TRACE>(2);                 // loads arrived

    // wave-uniform window bounds.  A source is a hit when its point (T, L) = ((int)y2, (int)x2) lies in the window
    // and the site is valid (x2, y2 inside the image, my_lib_kernel.cu:1670): x2 >= max(tx0 - 1, 0) and
    // x2 < tx0 + 64 and x2 <= W - 1.  For x2 >= 0 the float order is the order of the bit patterns, so the last two
    // are ONE integer compare against min(bits(tx0 + 64), bits(W - 1) + 1).
    const float xlo = (float)max(tx0 - 1, 0), ylo = (float)max(ty0 - 1, 0);
    const int xhi_bits = min(__float_as_int((float)(tx0 + 64)), __float_as_int((float)(W - 1)) + 1);
    const int yhi_bits = min(__float_as_int((float)(ty0 + TH)), __float_as_int((float)(H - 1)) + 1);
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    f32x4 *const my_ring = ring + wave * kRing;
    unsigned head = 0, tail = 0;               // wave-uniform ring positions
    bool far = false;

    // all 64 lanes splat one waiting entry each
    auto flush64 = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the entries are other lanes' LDS writes
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const f32x4 e = my_ring[(head + lane) & (kRing - 1)];
        double *q = P + __float_as_int(e[0]);
        if (ABL == 2) {
            asm volatile("" ::"v"(q), "v"(e[1]), "v"(e[2]), "v"(e[3]));
        } else {
            lds_add_f64(q, (double)e[1]);
            lds_add_f64(q + kPlane, (double)e[2]);
            if (kPlanes == 3) lds_add_f64(q + 2 * kPlane, (double)e[3]);
        }
        head += kWave;
    };

#pragma unroll
    for (int it = 0; it < kIts; it++) {
        if (ABL == 1 || ABL == 5) {            // (timing arm: no scan; the loads stay)
            asm volatile("" ::"v"(fx[it]), "v"(fy[it]), "v"(dd[it]));
            continue;
        }
        const bool lv = live[it];
        const float syf = (float)sy[it], sxf = (float)sx[it];
        // the quad lies inside the tile itself (tx0, the pad and sx are multiples of 4: all four sites or none)
        const bool homeq = lv && (unsigned)(sy[it] - ty0) < (unsigned)TH && (unsigned)(sx[it] - tx0) < 64u;
        float y2[4];
        bool wy[4], rowany = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            y2[j] = syf + fy[it][j];
            wy[j] = lv && y2[j] >= ylo && __float_as_int(y2[j]) < yhi_bits;
            rowany = rowany || wy[j];
        }
        // rows farther from the tile than the local motion: the whole wave leaves after the four y tests
        if (__builtin_amdgcn_ballot_w64(rowany || homeq) == 0) continue;
        f32x4 fxq = fx[it], ddq = dd[it];
        if (far_it(it)) {                      // rare: requested only now (and consumed inside this branch)
            const unsigned off = lv ? 4u * (unsigned)(sy[it] * s1h + sx[it]) : 0u;
            fxq = ld_cached4_u(flow_b, off);
            if (DEPTH) ddq = ld_cached4_u(depth_b, lv ? 4u * (unsigned)(sy[it] * sdh + sx[it]) : 0u);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float fxv = fxq[j], fyv = fy[it][j];
            const float x2 = (sxf + (float)j) + fxv;           // (float)x + fx, as the reference rounds it
            const bool nearj = fabsf(fxv) < (float)kReach && fabsf(fyv) < (float)kReach;
            if (homeq && !nearj) {             // a far source whose home is this tile: the image takes the general path
                const bool valid = x2 >= 0.0f && y2[j] >= 0.0f && x2 <= (float)(W - 1) && y2[j] <= (float)(H - 1);
                far = far || valid;
            }
            const bool hit = wy[j] && nearj && x2 >= xlo && __float_as_int(x2) < xhi_bits;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            if (m == 0) continue;              // wave-uniform
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (hit) {
                const int py = (int)y2[j] - (ty0 - 1), px = (int)x2 - (tx0 - 1);
                float vx = -fxv, vy = -fyv, vc = 1.0f;
                if (DEPTH) {                   // my_lib_kernel.cu:2102-2114
                    vx = -ddq[j] * fxv;
                    vy = -ddq[j] * fyv;
                    vc = ddq[j] * 1.0f;
                }
                my_ring[(tail + rank) & (kRing - 1)] = f32x4{__int_as_float(py * kPtW + px), vx, vy, vc};
            }
            tail += (unsigned)__builtin_popcountll(m);
            if (tail - head >= (unsigned)kWave) flush64();
        }
    }
    {                                          // what is still waiting (< 64 entries)
        const unsigned n = tail - head;
        if (n != 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if ((unsigned)lane < n) {
                const f32x4 e = my_ring[(head + lane) & (kRing - 1)];
                double *q = P + __float_as_int(e[0]);
                lds_add_f64(q, (double)e[1]);
                lds_add_f64(q + kPlane, (double)e[2]);
                if (kPlanes == 3) lds_add_f64(q + 2 * kPlane, (double)e[3]);
            }
        }
    }
    if (far) {                                 // this image needs the general path
        far_flag[b % kFlagWords] = 1;
        far_flag[kFlagWords] = 1;
    }
    trace_mark_proj<TRACE>(3);                 // scan + splat done (wave 0)
    __syncthreads();                           // every wave's points are in P
    trace_mark_proj<TRACE>(4);                 // all waves done

    // every lane owns four cells of a row
    const int cx = tx0 + 4 * (tid % 16), cy = ty0 + tid / 16;
    const bool inb = cx < W && cy < H;            // (no early exit: the summary below has a barrier)
    const float wy0 = (cy == H - 1) ? 2.0f : 1.0f;
    f32x4 ox, oy, oc;
    // The lane's four cells need the point sums of columns c-1 .. c+3 of two rows, per plane: read them once as
    // 2 x (two 16-byte pairs + one double) instead of 16 single doubles -- lanes are four cells apart, which for
    // 8-byte reads is a 4-way bank conflict.
    double top[3][5], bot[3][5];               // [plane][column c-1 .. c+3], rows cy-1 and cy
    if (ABL == 3) {
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
#pragma unroll
            for (int i = 0; i < 5; i++) top[pl][i] = bot[pl][i] = 0.0;
    } else {
        const int col0 = cx - tx0;             // P column of cell cx-1 (a multiple of 4: 16-byte aligned pairs)
        const double *r0 = P + (cy - ty0) * kPtW + col0, *r1 = r0 + kPtW;
        typedef double f64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            const double *a = r0 + (pl % kPlanes) * kPlane, *c = r1 + (pl % kPlanes) * kPlane;
            const f64x2 a01 = *reinterpret_cast<const f64x2 *>(a), a23 = *reinterpret_cast<const f64x2 *>(a + 2);
            const f64x2 c01 = *reinterpret_cast<const f64x2 *>(c), c23 = *reinterpret_cast<const f64x2 *>(c + 2);
            top[pl][0] = a01[0]; top[pl][1] = a01[1]; top[pl][2] = a23[0]; top[pl][3] = a23[1]; top[pl][4] = a[4];
            bot[pl][0] = c01[0]; bot[pl][1] = c01[1]; bot[pl][2] = c23[0]; bot[pl][3] = c23[1]; bot[pl][4] = c[4];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float wx0 = (cx + j == W - 1) ? 2.0f : 1.0f;
        float v[3];
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            // the four contributions are rounded to fp32 one by one and added in a fixed order (the
            // reference's order is arbitrary: fp32 atomics)
            float t = 0.0f;
            t += wy0 * wx0 * (float)bot[pl][j + 1];
            t += wy0 * (float)bot[pl][j];
            t += wx0 * (float)top[pl][j + 1];
            t += (float)top[pl][j];
            v[pl] = t;
        }
        if (v[2] > 0.0f) {                     // my_lib_kernel.cu:1730-1735; one reciprocal for both components
            const float inv = 1.0f / v[2];     // (<= 1 ulp from the two divisions)
            v[0] = v[0] * inv;
            v[1] = v[1] * inv;
        }
        ox[j] = v[0];  oy[j] = v[1];  oc[j] = v[2];
    }
    if (inb) {
        float *o = out + b * s1b + (int64_t)cy * s1h + cx;
        *reinterpret_cast<f32x4 *>(o) = ox;    // plain stores: pass 3 (hole fill) re-reads them
        *reinterpret_cast<f32x4 *>(o + s1c) = oy;
        *reinterpret_cast<f32x4 *>(count + b * scb + (int64_t)cy * sch + cx) = oc;
    }
    trace_mark_proj<TRACE>(5);                 // outputs stored (issued)
    if (ws.up) {                               // the counts are in registers: the filler's per-tile summaries are free
        const bool hole = summary_add(sm, inb, oc, 4 * (tid % 16), tid / 16, cx, cy);
        const int any_hole = __syncthreads_or(hole);
        summary_store(sm, any_hole, ws, b, tc.tx, tc.ty, W, H, tiles_x, tiles_y);
    }
}

// --------------------------------------------------------------------------------------------------
// proj_owner3: proj_owner2 as a PERSISTENT, software-pipelined kernel.
// Measured on proj_owner2 (tools/bench_ops.py arms 200..251, tools/trace_kernel.py proj2_32): without any scan work
// the kernel still takes 178 us where its stores alone take 82 -- the scan's loads are issued at the start of a
// workgroup's life and nothing is in flight while it tests, splats and reads out (39 % of a workgroup's life is
// "issue the loads, wait for them"): with 2 - 4 workgroups per CU the bytes in flight average ~30 KB per CU, a
// quarter of what the latency needs.  Here WGCU workgroups per CU walk the tiles of their XCD's chunk of the stripe
// order (tile positions p, p + grid, ...; grid % 8 == 0) and the NEXT tile's flow is requested before the current
// tile is scanned, into a second register set (the loop is unrolled by two so that the sets swap by name).
// --------------------------------------------------------------------------------------------------
template <bool DEPTH, int TH, int kReach, int WGCU>
__global__ __launch_bounds__(16 * TH, (16 * TH / 256) * WGCU) void proj_owner3(
    int W, int H, int tiles_x, int tiles_y, unsigned npos,
    int64_t s1b, int64_t s1c, int s1h, int64_t sdb, int sdh, int64_t scb, int sch,
    const float *__restrict__ flow, const float *__restrict__ depth,
    float *__restrict__ count, float *__restrict__ out, int *__restrict__ far_flag, FillWs ws, int sw)
{
    constexpr int NT = 16 * TH, NW = NT / kWave;
    constexpr int kPtH = TH + 1, kPlane = kPtH * kPtW;
    constexpr int kScanPadX = kReach + 4, kScanW = 64 + 2 * kScanPadX, kScanH = TH + 2 * kReach + 1;
    constexpr int kCols4 = kScanW / 4, kSlots = kCols4 * kScanH, kIts = (kSlots + NT - 1) / NT;
    constexpr int kRing = 128;
    constexpr int kNearRows = 8;
    static_assert(kPlane % 2 == 0, "P is zeroed 16 bytes at a time");
    __shared__ __attribute__((aligned(16))) double P[3 * kPlane];
    __shared__ __attribute__((aligned(16))) f32x4 ring[NW * kRing];
    __shared__ TileSummary<TH> sm;

    // What is prefetched one tile ahead is the fy plane of the scan region only: the row test needs nothing else,
    // and two full register sets (fy + fx [+ depth], twice) do not fit the 128 VGPRs that 16 waves per CU leave a
    // lane -- the allocator then spills freshly loaded values, which waits for them on the spot.  fx (and depth) of
    // the rows near the tile are requested at the start of the tile's own turn, ahead of the P zeroing, the next
    // tile's fy requests and the barrier.
    struct Regs {
        f32x4 fy[kIts];
    };
    auto far_it = [](int it) {                 // see proj_owner2
        const int first = NT * it / kCols4, last = (NT * it + NT - 1) / kCols4;
        return last <= kReach + 1 - kNearRows || first >= kReach + 1 + TH + kNearRows;
    };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    f32x4 *const my_ring = ring + wave * kRing;
    // (threadIdx through an opaque asm, once per tile and phase: everything derived from it -- slot rows, columns,
    // byte offsets -- would otherwise be hoisted out of the tile loop and kept, i.e. spilled, kernel-wide)

    // position of the next real tile of this workgroup's walk (stripes have virtual columns past the image)
    auto real_from = [&](unsigned p) {
        while (p < npos && tile_walk(p, npos, tiles_x, tiles_y, sw).tx >= tiles_x) p += gridDim.x;
        return p;
    };
    // slot `it` of this lane in the scan region of the tile at (tx0, ty0)
    auto slot = [&](int tid, int it, int tx0, int ty0, int &sx, int &sy, bool &live) {
        const int s = it * NT + tid, row = s / kCols4, c4 = s - row * kCols4;
        sx = tx0 - kScanPadX + 4 * c4;
        sy = ty0 - kReach - 1 + row;
        live = row < kScanH && sx >= 0 && sx < W && sy >= 0 && sy < H;            // W % 4 == 0
    };
    auto request = [&](unsigned p, Regs &r) {
        const TileCoord tc = tile_walk(p, npos, tiles_x, tiles_y, sw);
        const float *flow_b = flow + tc.b * s1b;
        const int tid = tid_now();
#pragma unroll
        for (int it = 0; it < kIts; it++) {
            int sx, sy;
            bool live;
            slot(tid, it, tc.tx * 64, tc.ty * TH, sx, sy, live);
            const unsigned off = live ? 4u * (unsigned)(sy * s1h + sx) : 0u;      // dead slots read pixel 0
            r.fy[it] = ld_cached4_u(flow_b + s1c, off);
        }
    };

    // one tile: `cur` holds its flow (requested one tile earlier); the flow of the tile at `pn` goes into `nxt`
    auto process = [&](unsigned p, Regs &cur, unsigned pn, Regs &nxt) {
        const TileCoord tc = tile_walk(p, npos, tiles_x, tiles_y, sw);
        const int b = tc.b, tx0 = tc.tx * 64, ty0 = tc.ty * TH;
        const float *flow_b = flow + b * s1b;
        const float *depth_b = DEPTH ? depth + b * sdb : nullptr;
        const int tid = tid_now();
        const int lane = tid & (kWave - 1);
        f32x4 cfx[kIts], cdd[kIts];            // this tile's fx / depth, rows near the tile
#pragma unroll
        for (int it = 0; it < kIts; it++) {
            if (far_it(it)) continue;
            int sx, sy;
            bool lv;
            slot(tid, it, tx0, ty0, sx, sy, lv);
            cfx[it] = ld_cached4_u(flow_b, lv ? 4u * (unsigned)(sy * s1h + sx) : 0u);
            if (DEPTH) cdd[it] = ld_cached4_u(depth_b, lv ? 4u * (unsigned)(sy * sdh + sx) : 0u);
        }
        summary_init(sm);
        {
            f32x4 *pz = reinterpret_cast<f32x4 *>(P);
            for (int i = tid; i < 3 * kPlane / 2; i += NT) pz[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        request(pn, nxt);                      // in flight while this tile is scanned, splatted and read out
        __syncthreads();                       // P is zero

        const float xlo = (float)max(tx0 - 1, 0), ylo = (float)max(ty0 - 1, 0);
        const int xhi_bits = min(__float_as_int((float)(tx0 + 64)), __float_as_int((float)(W - 1)) + 1);
        const int yhi_bits = min(__float_as_int((float)(ty0 + TH)), __float_as_int((float)(H - 1)) + 1);
        unsigned head = 0, tail = 0;
        bool far = false;
        auto splat = [&](unsigned n) {         // the first n (<= 64) waiting entries, one per lane
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if ((unsigned)lane < n) {
                const f32x4 e = my_ring[(head + lane) & (kRing - 1)];
                double *q = P + __float_as_int(e[0]);
                lds_add_f64(q, (double)e[1]);
                lds_add_f64(q + kPlane, (double)e[2]);
                lds_add_f64(q + 2 * kPlane, (double)e[3]);
            }
            head += n;
        };
#pragma unroll
        for (int it = 0; it < kIts; it++) {
            int sx, sy;
            bool lv;
            slot(tid, it, tx0, ty0, sx, sy, lv);
            const float syf = (float)sy, sxf = (float)sx;
            const bool homeq = lv && (unsigned)(sy - ty0) < (unsigned)TH && (unsigned)(sx - tx0) < 64u;
            float y2[4];
            bool wy[4], rowany = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                y2[j] = syf + cur.fy[it][j];
                wy[j] = lv && y2[j] >= ylo && __float_as_int(y2[j]) < yhi_bits;
                rowany = rowany || wy[j];
            }
            if (__builtin_amdgcn_ballot_w64(rowany || homeq) == 0) continue;
            f32x4 fxq = cfx[it], ddq = cdd[it];
            if (far_it(it)) {
                const unsigned off = lv ? 4u * (unsigned)(sy * s1h + sx) : 0u;
                fxq = ld_cached4_u(flow_b, off);
                if (DEPTH) ddq = ld_cached4_u(depth_b, lv ? 4u * (unsigned)(sy * sdh + sx) : 0u);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float fxv = fxq[j], fyv = cur.fy[it][j];
                const float x2 = (sxf + (float)j) + fxv;
                const bool nearj = fabsf(fxv) < (float)kReach && fabsf(fyv) < (float)kReach;
                if (homeq && !nearj) {
                    const bool valid = x2 >= 0.0f && y2[j] >= 0.0f && x2 <= (float)(W - 1) && y2[j] <= (float)(H - 1);
                    far = far || valid;
                }
                const bool hit = wy[j] && nearj && x2 >= xlo && __float_as_int(x2) < xhi_bits;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                if (m == 0) continue;
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (hit) {
                    const int py = (int)y2[j] - (ty0 - 1), px = (int)x2 - (tx0 - 1);
                    float vx = -fxv, vy = -fyv, vc = 1.0f;
                    if (DEPTH) {
                        vx = -ddq[j] * fxv;
                        vy = -ddq[j] * fyv;
                        vc = ddq[j] * 1.0f;
                    }
                    my_ring[(tail + rank) & (kRing - 1)] = f32x4{__int_as_float(py * kPtW + px), vx, vy, vc};
                }
                tail += (unsigned)__builtin_popcountll(m);
                if (tail - head >= (unsigned)kWave) splat(kWave);
            }
        }
        if (tail != head) splat(tail - head);
        if (far) {
            far_flag[b % kFlagWords] = 1;
            far_flag[kFlagWords] = 1;
        }
        __syncthreads();                       // every wave's points are in P

        const int cx = tx0 + 4 * (tid % 16), cy = ty0 + tid / 16;
        const bool inb = cx < W && cy < H;
        const float wy0 = (cy == H - 1) ? 2.0f : 1.0f;
        // one plane at a time (the next tile's flow occupies a register set of its own: reading all three planes'
        // thirty doubles at once would not fit beside it)
        f32x4 val[3];
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            typedef double f64x2 __attribute__((ext_vector_type(2)));
            const double *a = P + pl * kPlane + (cy - ty0) * kPtW + (cx - tx0), *c = a + kPtW;
            const f64x2 a01 = *reinterpret_cast<const f64x2 *>(a), a23 = *reinterpret_cast<const f64x2 *>(a + 2);
            const f64x2 c01 = *reinterpret_cast<const f64x2 *>(c), c23 = *reinterpret_cast<const f64x2 *>(c + 2);
            const double top[5] = {a01[0], a01[1], a23[0], a23[1], a[4]};
            const double bot[5] = {c01[0], c01[1], c23[0], c23[1], c[4]};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float wx0 = (cx + j == W - 1) ? 2.0f : 1.0f;
                float t = 0.0f;                // same order as proj_owner2
                t += wy0 * wx0 * (float)bot[j + 1];
                t += wy0 * (float)bot[j];
                t += wx0 * (float)top[j + 1];
                t += (float)top[j];
                val[pl][j] = t;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 ox, oy, oc;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float v0 = val[0][j], v1 = val[1][j];
            const float v2 = val[2][j];
            if (v2 > 0.0f) {
                const float inv = 1.0f / v2;
                v0 = v0 * inv;
                v1 = v1 * inv;
            }
            ox[j] = v0;  oy[j] = v1;  oc[j] = v2;
        }
        if (inb) {
            float *o = out + b * s1b + (int64_t)cy * s1h + cx;
            *reinterpret_cast<f32x4 *>(o) = ox;
            *reinterpret_cast<f32x4 *>(o + s1c) = oy;
            *reinterpret_cast<f32x4 *>(count + b * scb + (int64_t)cy * sch + cx) = oc;
        }
        if (ws.up) {
            const bool hole = summary_add(sm, inb, oc, 4 * (tid % 16), tid / 16, cx, cy);
            const int any_hole = __syncthreads_or(hole);
            summary_store(sm, any_hole, ws, b, tc.tx, tc.ty, W, H, tiles_x, tiles_y);
        }
        __syncthreads();                       // P, the summary and the rings are rebuilt by the next tile
    };

    unsigned p = real_from(blockIdx.x);
    if (p >= npos) return;
    Regs ra, rb;
    request(p, ra);
#pragma unroll 1
    for (;;) {
        unsigned pn = real_from(p + gridDim.x);
        process(p, ra, pn < npos ? pn : p, rb);             // (past the end: re-request this tile -- unconditional loads)
        if (pn >= npos) break;
        p = pn;
        pn = real_from(p + gridDim.x);
        process(p, rb, pn < npos ? pn : p, ra);
        if (pn >= npos) break;
        p = pn;
    }
}

#endif  // MEMC_PROJ_ARMS_PART_A

// ---- part B: proj_owner (round 1) ----
#ifdef MEMC_PROJ_ARMS_PART_B
// ---- round-1 owner kernel, kept as the A/B arm of proj_owner2 (variants -10 / -7 / -6) ----

template <bool DEPTH, int kReach, bool TRACE = false>
__global__ __launch_bounds__(256) void proj_owner(
    int W, int H, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t sdb, int sdh, int64_t scb, int sch,
    const float *__restrict__ flow, const float *__restrict__ depth,
    float *__restrict__ count, float *__restrict__ out, int *__restrict__ far_flag, FillWs ws)
{
    constexpr int kScanPadX = kReach + 4;         // dilated tile: columns, kept 4-aligned
    constexpr int kScanW = 64 + 2 * kScanPadX;    // source columns
    constexpr int kScanH = 16 + 2 * kReach + 1;   // source rows: [ty0 - kReach - 1, ty0 + 16 + kReach)
    constexpr int kPtH = 17;                      // point window 65 x 17
    __shared__ __attribute__((aligned(16))) double P[3 * kPtH * kPtW];
    const TileCoord tc = strip_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, gridDim.x / (tiles_x * tiles_y));
    const int b = tc.b, tx0 = tc.tx * 64, ty0 = tc.ty * 16;
    trace_mark_proj<TRACE>(0);
    __shared__ TileSummary<16> sm;                   // for the hole filler, when one follows (ws.up != nullptr)
    summary_init(sm);
    {
        static_assert((3 * kPtH * kPtW) % 2 == 0, "P is zeroed 16 bytes at a time");
        f32x4 *pz = reinterpret_cast<f32x4 *>(P);
        for (int i = threadIdx.x; i < 3 * kPtH * kPtW / 2; i += 256) pz[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // scan: kScanW / 4 float4 columns x kScanH rows of slots, kIts per lane; all loads first.  Slot -> (row,
    // column) by one division and increments, addresses as wave-uniform base + 32-bit lane offset (the address
    // arithmetic of this prologue was a quarter of the kernel's VALU instructions, and VALU is its bound).
    constexpr int kCols4 = kScanW / 4, kSlots = kCols4 * kScanH, kIts = (kSlots + 255) / 256;
    // slot `it` of the 256 lanes covers scan rows [256 it / kCols4, (256 it + 255) / kCols4]; the tile is rows
    // [kReach + 1, kReach + 17): the slot is "far" when all of its rows are at least kNearRows away from the tile
    constexpr int kNearRows = 8;
    auto far_it = [](int it) {
        const int first = 256 * it / kCols4, last = (256 * it + 255) / kCols4;
        return last <= kReach + 1 - kNearRows || first >= kReach + 17 + kNearRows;
    };
    const float *flow_b = flow + b * s1b;
    const float *depth_b = DEPTH ? depth + b * sdb : nullptr;
    f32x4 fx[kIts], fy[kIts], dd[kIts];
    int sx[kIts], sy[kIts];
    bool live[kIts];
    int row = (int)threadIdx.x / kCols4, c4 = (int)threadIdx.x % kCols4;
#pragma unroll
    for (int it = 0; it < kIts; it++) {
        sx[it] = tx0 - kScanPadX + 4 * c4;
        sy[it] = ty0 - kReach - 1 + row;
        live[it] = row < kScanH && sx[it] >= 0 && sx[it] < W && sy[it] >= 0 && sy[it] < H;   // W % 4 == 0
        // dead slots read the plane's first pixels (unconditional loads)
        const unsigned off = live[it] ? 4u * (unsigned)(sy[it] * s1h + sx[it]) : 0u;
        fy[it] = ld_cached4_u(flow_b + s1c, off);
        // Rows more than ~8 px from the tile (the first and last two slots of a lane) almost never pass the row
        // test below: only their fy is requested here, fx / depth follow inside the branch if they do.  The scan
        // moves 7.6x the tile's own bytes through the CU's 64 B/clk L1 path, which is a bound of its own.
        if (!far_it(it)) {
            fx[it] = ld_cached4_u(flow_b, off);
            if (DEPTH) dd[it] = ld_cached4_u(depth_b, live[it] ? 4u * (unsigned)(sy[it] * sdh + sx[it]) : 0u);
        }
        row += 256 / kCols4;                   // the next slot of this lane is 256 further on
        c4 += 256 % kCols4;
        if (c4 >= kCols4) {
            c4 -= kCols4;
            row++;
        }
    }
    __syncthreads();                           // P is zero
    trace_mark_proj<TRACE>(1);                 // loads issued, P zeroed
    if (TRACE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    trace_mark_proj<TRACE>(2);                 // loads arrived
    bool far = false;
#pragma unroll
    for (int it = 0; it < kIts; it++) {
        if (!live[it]) continue;
        const bool home_row = sy[it] >= ty0 && sy[it] < ty0 + 16;
        // conservative row test first (one pixel of slack covers the rounding of y + fy): a wave scans ~2.5 rows
        // of the dilated tile, and in the rows farther from the tile than the local motion no lane can land --
        // the whole wave then skips the per-source work (VALU is what bounds this kernel)
        if (!home_row) {
            const float lo = (float)(ty0 - 2 - sy[it]), hi = (float)(ty0 + 17 - sy[it]);
            const f32x4 f = fy[it];
            if (!((f[0] >= lo && f[0] < hi) || (f[1] >= lo && f[1] < hi) || (f[2] >= lo && f[2] < hi) ||
                  (f[3] >= lo && f[3] < hi)))
                continue;
        }
        f32x4 fxq = fx[it], ddq = dd[it];
        if (far_it(it)) {                      // rare: requested only now (and consumed inside this branch)
            const unsigned off = 4u * (unsigned)(sy[it] * s1h + sx[it]);
            fxq = ld_cached4_u(flow_b, off);
            if (DEPTH) ddq = ld_cached4_u(depth_b, 4u * (unsigned)(sy[it] * sdh + sx[it]));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int x = sx[it] + j, y = sy[it];
            const float fxv = fxq[j], fyv = fy[it][j];
            const BlSite s = bl_locate<false>(x, y, W, H, fxv, fyv);
            if (!s.valid) continue;
            const bool near = fabsf(fxv) < (float)kReach && fabsf(fyv) < (float)kReach;
            if (!near) {
                far = far || (home_row && x >= tx0 && x < tx0 + 64);
                continue;
            }
            const int py = s.T - (ty0 - 1), px = s.L - (tx0 - 1);
            if ((unsigned)py < (unsigned)kPtH && (unsigned)px < 65u) {
                float vx = -fxv, vy = -fyv, vc = 1.0f;
                if (DEPTH) {
                    vx = -ddq[j] * fxv;
                    vy = -ddq[j] * fyv;
                    vc = ddq[j] * 1.0f;
                }
                double *q = P + py * kPtW + px;
                lds_add_f64(q, (double)vx);
                lds_add_f64(q + kPtH * kPtW, (double)vy);
                lds_add_f64(q + 2 * kPtH * kPtW, (double)vc);
            }
        }
    }
    if (far) {                                 // this image needs the general path
        far_flag[b % kFlagWords] = 1;
        far_flag[kFlagWords] = 1;
    }
    trace_mark_proj<TRACE>(3);                 // scan + splat done (wave 0)
    __syncthreads();
    trace_mark_proj<TRACE>(4);                 // all waves done

    // every lane owns four cells of a row
    const int cx = tx0 + 4 * (threadIdx.x % 16), cy = ty0 + threadIdx.x / 16;
    const bool inb = cx < W && cy < H;            // (no early exit: the summary below has a barrier)
    const float wy0 = (cy == H - 1) ? 2.0f : 1.0f;
    f32x4 ox, oy, oc;
    // The lane's four cells need the point sums of columns c-1 .. c+3 of two rows, per plane: read them once as
    // 2 x (two 16-byte pairs + one double) instead of 16 single doubles -- lanes are four cells apart, which for
    // 8-byte reads is a 4-way bank conflict, and this read-out was most of the kernel's LDS time.
    double top[3][5], bot[3][5];               // [plane][column c-1 .. c+3], rows cy-1 and cy
    {
        const int col0 = cx - tx0;             // P column of cell cx-1 (a multiple of 4: 16-byte aligned pairs)
        const double *r0 = P + (cy - ty0) * kPtW + col0, *r1 = r0 + kPtW;
        typedef double f64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            const double *a = r0 + pl * kPtH * kPtW, *c = r1 + pl * kPtH * kPtW;
            const f64x2 a01 = *reinterpret_cast<const f64x2 *>(a), a23 = *reinterpret_cast<const f64x2 *>(a + 2);
            const f64x2 c01 = *reinterpret_cast<const f64x2 *>(c), c23 = *reinterpret_cast<const f64x2 *>(c + 2);
            top[pl][0] = a01[0]; top[pl][1] = a01[1]; top[pl][2] = a23[0]; top[pl][3] = a23[1]; top[pl][4] = a[4];
            bot[pl][0] = c01[0]; bot[pl][1] = c01[1]; bot[pl][2] = c23[0]; bot[pl][3] = c23[1]; bot[pl][4] = c[4];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float wx0 = (cx + j == W - 1) ? 2.0f : 1.0f;
        float v[3];
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            // the four contributions are rounded to fp32 one by one and added in a fixed order (the
            // reference's order is arbitrary: fp32 atomics)
            float t = 0.0f;
            t += wy0 * wx0 * (float)bot[pl][j + 1];
            t += wy0 * (float)bot[pl][j];
            t += wx0 * (float)top[pl][j + 1];
            t += (float)top[pl][j];
            v[pl] = t;
        }
        if (v[2] > 0.0f) {                     // my_lib_kernel.cu:1730-1735; one reciprocal for both components
            const float inv = 1.0f / v[2];     // (<= 1 ulp from the two divisions; VALU is this kernel's bound)
            v[0] = v[0] * inv;
            v[1] = v[1] * inv;
        }
        ox[j] = v[0];  oy[j] = v[1];  oc[j] = v[2];
    }
    if (inb) {
        float *o = out + b * s1b + (int64_t)cy * s1h + cx;
        *reinterpret_cast<f32x4 *>(o) = ox;    // plain stores: pass 3 (hole fill) re-reads them
        *reinterpret_cast<f32x4 *>(o + s1c) = oy;
        *reinterpret_cast<f32x4 *>(count + b * scb + (int64_t)cy * sch + cx) = oc;
    }
    trace_mark_proj<TRACE>(5);                 // outputs stored (issued)
    if (ws.up) {                               // the counts are in registers: the filler's per-tile summaries are free
        const bool hole = summary_add(sm, inb, oc, 4 * (threadIdx.x % 16), threadIdx.x / 16, cx, cy);
        const int any_hole = __syncthreads_or(hole);
        summary_store(sm, any_hole, ws, b, tc.tx, tc.ty, W, H, tiles_x, tiles_y);
    }
}
#endif  // MEMC_PROJ_ARMS_PART_B

// ---- part C: round 3's production set -- proj_owner4 (register compaction of the hits), its far-source kernel, the
// summaries from the count plane and the carry filler that re-read the counts of every tile holding a hole ----
#ifdef MEMC_PROJ_ARMS_PART_C
// rounds 2-3: the owner tile of proj_owner_far_r3 (register batch of waiting hits)
// One owned 64 x TH tile: its point planes, window bounds and the wave's register batch of waiting hits.
// (Used by proj_owner_far.  Always THREE planes there: without a limit on |flow| the sum of vx at a point is not
// bounded by 2^19, which the packed count * 2^20 + sum(vx) plane of proj_owner5 relies on.)
template <bool DEPTH, int TH>
struct OwnerTile {
    static constexpr int NP = 3;                  // planes: count, vx, vy
    static constexpr int kPlane = (TH + 1) * kPtW4;
    static_assert(kPlane % 2 == 0, "P is zeroed 16 bytes at a time");
    double *P;
    int tx0, ty0;
    float xlo, ylo;
    int xhi_bits, yhi_bits;
    unsigned lane, fill;                          // fill: valid entries of the batch (wave-uniform); entry i in lane i
    int p_cell;
    float p_vx, p_vy, p_vc;

    // Window bounds.  A source is a hit when its point (T, L) = ((int)y2, (int)x2) lies in the window
    // [ty0 - 1, ty0 + TH - 1] x [tx0 - 1, tx0 + 63] and the site is valid (x2, y2 inside the image,
    // my_lib_kernel.cu:1670): x2 >= max(tx0 - 1, 0) and x2 < tx0 + 64 and x2 <= W - 1.  For x2 >= 0 the float order
    // is the order of the bit patterns, so the last two are ONE integer compare against
    // min(bits(tx0 + 64), bits(W - 1) + 1).
    __device__ __forceinline__ void begin(double *P_, int tx0_, int ty0_, int W, int H, unsigned lane_)
    {
        P = P_;  tx0 = tx0_;  ty0 = ty0_;  lane = lane_;  fill = 0;
        p_cell = 0;  p_vx = p_vy = 0.0f;  p_vc = 1.0f;         // (FlowProjection: every source counts 1)
        xlo = (float)max(tx0 - 1, 0);
        ylo = (float)max(ty0 - 1, 0);
        xhi_bits = min(__float_as_int((float)(tx0 + 64)), __float_as_int((float)(W - 1)) + 1);
        yhi_bits = min(__float_as_int((float)(ty0 + TH)), __float_as_int((float)(H - 1)) + 1);
    }
    template <int NT>
    __device__ __forceinline__ void zero(int tid) const
    {
        f32x4 *pz = reinterpret_cast<f32x4 *>(P);
        for (int i = tid; i < NP * kPlane / 2; i += NT) pz[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void splat(int cell, float vx, float vy, float vc) const
    {
        double *q = P + cell;
        lds_add_f64(q, (double)vc);
        lds_add_f64(q + kPlane, (double)vx);
        lds_add_f64(q + 2 * kPlane, (double)vy);
    }
    // the four y tests of a quad of sources in row sy
    __device__ __forceinline__ bool rows(bool lv, float syf, const f32x4 &fy4, float (&y2)[4], bool (&wy)[4]) const
    {
        bool any = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            y2[j] = syf + fy4[j];
            wy[j] = lv && y2[j] >= ylo && __float_as_int(y2[j]) < yhi_bits;
            any = any || wy[j];
        }
        return any;
    }
    // One source per lane (`pre`: passed the y test and whatever else the caller demands): x test, then the hits of
    // the wave are pushed to the consecutive lanes fill, fill + 1, ... (cyclically) of the batch -- lanes without
    // a hit aim at the LAST slot of the cycle, which a hit only takes when all 64 lanes hit (no such lane then).
    // Wave-uniform control flow: call from converged code only.
    __device__ __forceinline__ void source(bool pre, float x2, float y2, float fxv, float fyv, float d)
    {
        const bool hit = pre && x2 >= xlo && __float_as_int(x2) < xhi_bits;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
        if (m == 0) return;                    // wave-uniform
        const unsigned n = (unsigned)__builtin_popcountll(m);
        const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        const int dst = (int)((fill + (hit ? rank : 63u)) & 63u) << 2;
        const int py = (int)y2 - (ty0 - 1), px = (int)x2 - (tx0 - 1);                       // (garbage without a hit)
        float vx = -fxv, vy = -fyv, vc = 1.0f;
        if (DEPTH) {                           // my_lib_kernel.cu:2102-2114
            vx = -d * fxv;
            vy = -d * fyv;
            vc = d * 1.0f;
        }
        const int r_cell = __builtin_amdgcn_ds_permute(dst, py * kPtW4 + px);
        const float r_vx = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(vx)));
        const float r_vy = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(vy)));
        float r_vc = 1.0f;
        if (DEPTH) r_vc = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(vc)));
        if (fill + n < (unsigned)kWave) {      // (wave-uniform) not full yet: lanes [fill, fill + n) take theirs
            const bool recv = ((lane - fill) & 63u) < n;
            p_cell = recv ? r_cell : p_cell;
            p_vx = recv ? r_vx : p_vx;
            p_vy = recv ? r_vy : p_vy;
            if (DEPTH) p_vc = recv ? r_vc : p_vc;
            fill += n;
        } else {                               // full: lanes [fill, 64) hold new entries, lanes [0, fill) waiting ones
            const bool fresh = lane >= fill;
            splat(fresh ? r_cell : p_cell, fresh ? r_vx : p_vx, fresh ? r_vy : p_vy, fresh ? r_vc : p_vc);
            fill = fill + n - (unsigned)kWave; // the entries that wrapped around: lanes [0, fill)
            p_cell = r_cell;  p_vx = r_vx;  p_vy = r_vy;  p_vc = r_vc;
        }
    }
    __device__ __forceinline__ void finish() const
    {
        if (lane < fill) splat(p_cell, p_vx, p_vy, p_vc);      // what is still waiting
    }
    // After a barrier: the lane's four cells (cx .. cx + 3, cy) -- 2x2 box sums of the points of columns c-1 .. c+3,
    // rows cy-1 and cy (border duplicates as weights 2, see proj_scatter_tiled), normalised by the count.
    __device__ __forceinline__ void readout(int cx, int cy, int W, int H, f32x4 &ox, f32x4 &oy, f32x4 &oc) const
    {
        const float wy0 = (cy == H - 1) ? 2.0f : 1.0f;
        float top[3][5], bot[3][5];            // [count, vx, vy][column], each point sum rounded to fp32 once
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const double *r0 = P + (cy - ty0) * kPtW4 + (cx - tx0);   // column offset a multiple of 4: 16-byte pairs
#pragma unroll
        for (int pl = 0; pl < NP; pl++) {
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const double *a = r0 + pl * kPlane + rr * kPtW4;
                const f64x2 a01 = *reinterpret_cast<const f64x2 *>(a), a23 = *reinterpret_cast<const f64x2 *>(a + 2);
                const double v[5] = {a01[0], a01[1], a23[0], a23[1], a[4]};
                float (&dst_c)[5] = rr ? bot[0] : top[0];
                float (&dst_x)[5] = rr ? bot[1] : top[1];
                float (&dst_y)[5] = rr ? bot[2] : top[2];
#pragma unroll
                for (int i = 0; i < 5; i++) (pl == 0 ? dst_c : (pl == 1 ? dst_x : dst_y))[i] = (float)v[i];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float wx0 = (cx + j == W - 1) ? 2.0f : 1.0f;
            float v[3];
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                // the four contributions are added in a fixed order (the reference's order is arbitrary: fp32 atomics)
                float t = 0.0f;
                t += wy0 * wx0 * bot[pl][j + 1];
                t += wy0 * bot[pl][j];
                t += wx0 * top[pl][j + 1];
                t += top[pl][j];
                v[pl] = t;
            }
            if (v[0] > 0.0f) {                 // my_lib_kernel.cu:1730-1735; one reciprocal for both components
                const float inv = 1.0f / v[0]; // (<= 1 ulp from the two divisions)
                v[1] = v[1] * inv;
                v[2] = v[2] * inv;
            }
            oc[j] = v[0];  ox[j] = v[1];  oy[j] = v[2];
        }
    }
};

// stores, and the hole filler's per-tile summaries (the counts are in registers: they are free)
template <int TH>
__device__ __forceinline__ void owner_store(TileSummary<TH> &sm, const FillWs &ws, int tid, int b, int tx, int ty, int W,
                                            int H, int tiles_x, int tiles_y, int64_t s1b, int64_t s1c, int s1h,
                                            int64_t scb, int sch, float *count, float *out, const f32x4 &ox,
                                            const f32x4 &oy, const f32x4 &oc)
{
    const int cx = tx * 64 + 4 * (tid % 16), cy = ty * TH + tid / 16;
    const bool inb = cx < W && cy < H;            // (no early exit: the summary below has a barrier)
    if (inb) {
        float *o = out + b * s1b + (int64_t)cy * s1h + cx;
        *reinterpret_cast<f32x4 *>(o) = ox;    // plain stores: pass 3 (hole fill) re-reads them
        *reinterpret_cast<f32x4 *>(o + s1c) = oy;
        *reinterpret_cast<f32x4 *>(count + b * scb + (int64_t)cy * sch + cx) = oc;
    }
    if (ws.up) {
        const bool hole = summary_add(sm, inb, oc, 4 * (tid % 16), tid / 16, cx, cy);
        const int any_hole = __syncthreads_or(hole);
        summary_store(sm, any_hole, ws, b, tx, ty, W, H, tiles_x, tiles_y);
    }
}

// (Written out rather than built from OwnerTile's methods: at 64 VGPRs -- eight waves per SIMD, which is what lets
// four workgroups share a CU -- the allocator is at its limit, and the method form of the very same code spilled five
// registers and ran 25 % slower.)
// (Round 2 measured three knobs of this kernel that are no longer built: rows whose fx / depth loads are deferred --
// 0 / 4 / 12 / 16 instead of 8: equal / equal / 245 us / 245 us; no motion bounds, flagged images through the general path:
// four more normally idle launches; timing arms of the summaries: ~20 us of the call with hole filling.)
template <bool DEPTH, int TH, int kReach, int MINW>
__global__ __launch_bounds__(16 * TH, MINW) void proj_owner4(
    int W, int H, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t sdb, int sdh, int64_t scb, int sch,
    const float *__restrict__ flow, const float *__restrict__ depth,
    float *__restrict__ count, float *__restrict__ out, int *__restrict__ far_flag, int *__restrict__ bounds,
    FillWs ws, int sw, int nonce)
{
    constexpr int NT = 16 * TH;                   // one lane per four owned cells
    constexpr int NP = DEPTH ? 3 : 2;             // planes: (count, vx, vy) or (count * 2^20 + vx, vy)
    constexpr int kPtH = TH + 1, kPlane = kPtH * kPtW4;
    constexpr int kScanPadX = kReach + 4;         // dilated tile: columns, kept 4-aligned
    constexpr int kScanW = 64 + 2 * kScanPadX;    // source columns
    constexpr int kScanH = TH + 2 * kReach + 1;   // source rows: [ty0 - kReach - 1, ty0 + TH + kReach)
    constexpr int kCols4 = kScanW / 4, kSlots = kCols4 * kScanH, kIts = (kSlots + NT - 1) / NT;
    static_assert(kPlane % 2 == 0, "P is zeroed 16 bytes at a time");
    static_assert((2 * kReach + 1) * (2 * kReach + 1) < 4096 && kReach <= 128, "count * 2^20 + sum(vx) must split exactly");
    __shared__ __attribute__((aligned(16))) double P[NP * kPlane];
    __shared__ TileSummary<TH> sm;                // for the hole filler, when one follows (ws.up != nullptr)
    __shared__ int tile_max[2];                   // bit patterns of max |fx|, max |fy| over the tile's own FAR sources (0: none)

    const TileCoord tc = tile_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, sw);
    if (tc.tx >= tiles_x) return;                 // virtual column of the last stripe
    const int b = tc.b, tx0 = tc.tx * 64, ty0 = tc.ty * TH;
    const int tid0 = threadIdx.x;                 // (thread index of the first half of the kernel, see below)
    const int wave_index = __builtin_amdgcn_readfirstlane(tid0 / kWave);
    summary_init(sm, tid0);
    if (tid0 < 2) tile_max[tid0] = 0;
    {
        f32x4 *pz = reinterpret_cast<f32x4 *>(P);
        for (int i = tid0; i < NP * kPlane / 2; i += NT) pz[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // scan loads: see proj_owner2 (slots, far rows, unconditional addresses)
    constexpr int kNearRows = 8;
    auto far_it = [](int it) {
        const int first = NT * it / kCols4, last = (NT * it + NT - 1) / kCols4;
        return last <= kReach + 1 - kNearRows || first >= kReach + 1 + TH + kNearRows;
    };
    const float *flow_b = flow + b * s1b;
    const float *depth_b = DEPTH ? depth + b * sdb : nullptr;
    f32x4 fx[kIts], fy[kIts], dd[kIts];
    int sx[kIts], sy[kIts];
    bool live[kIts];
    int row = tid0 / kCols4, c4 = tid0 % kCols4;
#pragma unroll
    for (int it = 0; it < kIts; it++) {
        sx[it] = tx0 - kScanPadX + 4 * c4;
        sy[it] = ty0 - kReach - 1 + row;
        live[it] = row < kScanH && sx[it] >= 0 && sx[it] < W && sy[it] >= 0 && sy[it] < H;   // W % 4 == 0
        const unsigned off = live[it] ? 4u * (unsigned)(sy[it] * s1h + sx[it]) : 0u;         // dead slots read pixel 0
        fy[it] = ld_cached4_u(flow_b + s1c, off);
        if (!far_it(it)) {
            fx[it] = ld_cached4_u(flow_b, off);
            if (DEPTH) dd[it] = ld_cached4_u(depth_b, live[it] ? 4u * (unsigned)(sy[it] * sdh + sx[it]) : 0u);
        }
        row += NT / kCols4;
        c4 += NT % kCols4;
        if (c4 >= kCols4) {
            c4 -= kCols4;
            row++;
        }
    }
    __syncthreads();                           // P is zero

    // wave-uniform window bounds (see proj_owner2: the upper bounds are compared as bit patterns)
    const float xlo = (float)max(tx0 - 1, 0), ylo = (float)max(ty0 - 1, 0);
    const int xhi_bits = min(__float_as_int((float)(tx0 + 64)), __float_as_int((float)(W - 1)) + 1);
    const int yhi_bits = min(__float_as_int((float)(ty0 + TH)), __float_as_int((float)(H - 1)) + 1);
    const unsigned lane = tid0 & (kWave - 1);
    // the wave's batch of waiting hits: entry i sits in lane i; `fill` of them are valid (wave-uniform)
    int p_cell = 0;
    float p_vx = 0.0f, p_vy = 0.0f, p_vc = 0.0f;
    unsigned fill = 0;
    bool far = false;

    // (cell, vx, vy as compacted above: cell without its wave-uniform offset, +f instead of v = -f)
    const int cell0 = (ty0 - 1) * kPtW4 + (tx0 - 1);
    auto splat = [&](int cell, float vx, float vy, float vc) {
        double *q = P + (cell - cell0);
        if (DEPTH) {
            lds_add_f64(q, (double)vc);
            lds_add_f64(q + kPlane, -(double)vx);
            lds_add_f64(q + 2 * kPlane, -(double)vy);
        } else {
            lds_add_f64(q, kCountUnit - (double)vx);           // one source: count += 1, sum(vx) += -fx
            lds_add_f64(q + kPlane, -(double)vy);
        }
    };

#pragma unroll
    for (int it = 0; it < kIts; it++) {
        const bool lv = live[it];
        const float syf = (float)sy[it], sxf = (float)sx[it];
        // The quad lies inside the tile itself (tx0, the pad and sx are multiples of 4: all four sites or none) -- only
        // slots that can hold rows of the tile evaluate this (and the far-source test below) at all.
        const bool kHomeIt = NT * it / kCols4 < kReach + 1 + TH && (NT * it + NT - 1) / kCols4 >= kReach + 1;   // folds: `it` is unrolled
        const bool homeq = kHomeIt && lv && (unsigned)(sy[it] - ty0) < (unsigned)TH && (unsigned)(sx[it] - tx0) < 64u;
        float y2[4];
        bool wy[4], rowany = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            y2[j] = syf + fy[it][j];
            wy[j] = lv && y2[j] >= ylo && __float_as_int(y2[j]) < yhi_bits;
            rowany = rowany || wy[j];
        }
        // rows farther from the tile than the local motion: the whole wave leaves after the four y tests
        if (__builtin_amdgcn_ballot_w64(rowany || homeq) == 0) continue;
        f32x4 fxq = fx[it], ddq = dd[it];
        if (far_it(it)) {                      // rare: requested only now (and consumed inside this branch)
            const unsigned off = lv ? 4u * (unsigned)(sy[it] * s1h + sx[it]) : 0u;
            fxq = ld_cached4_u(flow_b, off);
            if (DEPTH) ddq = ld_cached4_u(depth_b, lv ? 4u * (unsigned)(sy[it] * sdh + sx[it]) : 0u);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float fxv = fxq[j], fyv = fy[it][j];
            const float x2 = (sxf + (float)j) + fxv;           // (float)x + fx, as the reference rounds it
            // A far source (|f| >= kReach) whose home is this tile: the image is redone by proj_owner_far.  The hit
            // test below does NOT ask for |f| < kReach: an image without a valid far source has only near hits, which
            // every owner of their point sees (they lie inside its scan region); in an image WITH one the owners may
            // disagree -- and every tile of that image is recomputed anyway.  Two compares less per scanned source.
            if (kHomeIt && homeq && !(fabsf(fxv) < (float)kReach && fabsf(fyv) < (float)kReach)) {
                const bool valid = x2 >= 0.0f && y2[j] >= 0.0f && x2 <= (float)(W - 1) && y2[j] <= (float)(H - 1);
                far = far || valid;
                if (valid) {                   // (cold) the tile's bound on its far sources' motion, for proj_owner_far:
                    atomicMax(&tile_max[0], __float_as_int(fabsf(fxv)));      // non-negative floats order like their bits
                    atomicMax(&tile_max[1], __float_as_int(fabsf(fyv)));
                }
            }
            const bool hit = wy[j] && x2 >= xlo && __float_as_int(x2) < xhi_bits;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            if (m == 0) continue;              // wave-uniform
            const unsigned n = (unsigned)__builtin_popcountll(m);
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            // Push the hits to the consecutive lanes fill, fill + 1, ... (cyclically) of the batch.  Lanes without a
            // hit aim at the LAST slot of the cycle, which a hit only takes when all 64 lanes hit (no such lane then).
            const int dst = (int)((fill + (hit ? rank : 63u)) & 63u) << 2;
            // cell = ((int)y2 - (ty0 - 1)) * pitch + (int)x2 - (tx0 - 1); the wave-uniform part is added at the splat
            const int cell = (int)y2[j] * kPtW4 + (int)x2;                                   // (garbage without a hit)
            // what travels is +f (or d * f): the sign of v = -f is applied where it is converted to double
            float vx = fxv, vy = fyv, vc = 1.0f;
            if (DEPTH) {                       // my_lib_kernel.cu:2102-2114: v = -d * f, count += d
                vx = ddq[j] * fxv;
                vy = ddq[j] * fyv;
                vc = ddq[j] * 1.0f;
            }
            const int r_cell = __builtin_amdgcn_ds_permute(dst, cell);
            const float r_vx = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(vx)));
            const float r_vy = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(vy)));
            float r_vc = 1.0f;
            if (DEPTH) r_vc = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(vc)));
            if (fill + n < (unsigned)kWave) {  // (wave-uniform) the batch is not full yet: lanes [fill, fill + n) take theirs
                const bool recv = ((lane - fill) & 63u) < n;
                p_cell = recv ? r_cell : p_cell;
                p_vx = recv ? r_vx : p_vx;
                p_vy = recv ? r_vy : p_vy;
                if (DEPTH) p_vc = recv ? r_vc : p_vc;
                fill += n;
            } else {                           // full: lanes [fill, 64) hold new entries, lanes [0, fill) waiting ones
                const bool fresh = lane >= fill;
                splat(fresh ? r_cell : p_cell, fresh ? r_vx : p_vx, fresh ? r_vy : p_vy, fresh ? r_vc : p_vc);
                fill = fill + n - (unsigned)kWave;            // the entries that wrapped around: lanes [0, fill)
                p_cell = r_cell;  p_vx = r_vx;  p_vy = r_vy;  p_vc = r_vc;
            }
        }
    }
    if (lane < fill) splat(p_cell, p_vx, p_vy, p_vc);          // what is still waiting
    if (far) {                                 // this image is redone by proj_owner_far.  The flag words are NOT cleared
        far_flag[b % kFlagWords] = nonce;      // before the call: "raised" = "holds this call's nonce" (launcher), so stale
        far_flag[kFlagWords] = nonce;          // or uninitialised words can at worst cause a needless redo, never a missed one
    }
    __syncthreads();                           // every wave's points are in P (and the tile's motion bound in tile_max)
    // (from here on the thread index is REBUILT from the wave's index, a scalar, and the lane's rank in the wave:
    // kept in a VGPR across the scan it was the one value the allocator spilled at 64 registers)
    const int tid = wave_index * kWave + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (tid < 2) bounds[2 * (((int64_t)b * tiles_y + tc.ty) * tiles_x + tc.tx) + tid] = tile_max[tid];

    // Every lane owns four cells of a row: 2x2 box sums of the points of columns c-1 .. c+3, rows cy-1 and cy (border
    // duplicates as weights 2, see proj_scatter_tiled), summed in DOUBLE -- exact, also for the packed plane:
    // sum_i w_i (count_i 2^20 + S_i) = (sum w count) 2^20 + sum w S with |sum w S| < 2^19 (at most 2500 sources reach a
    // 2x2 block, weights <= 4, |v| < kReach) -- then split and rounded to fp32 ONCE per cell (four splits per lane
    // instead of ten; the reference's fp32 atomics add in arbitrary order anyway).
    const int cx = tx0 + 4 * (tid % 16), cy = ty0 + tid / 16;
    const bool inb = cx < W && cy < H;            // (no early exit: the summary below has a barrier)
    const double wy0 = (cy == H - 1) ? 2.0 : 1.0;
    f32x4 ox, oy, oc;
    {
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const double *r0 = P + (cy - ty0) * kPtW4 + (cx - tx0);   // column offset a multiple of 4: 16-byte pairs
        double box[NP][4];
#pragma unroll
        for (int pl = 0; pl < NP; pl++) {
            const double *a = r0 + pl * kPlane, *c = a + kPtW4;
            const f64x2 a01 = *reinterpret_cast<const f64x2 *>(a), a23 = *reinterpret_cast<const f64x2 *>(a + 2);
            const f64x2 c01 = *reinterpret_cast<const f64x2 *>(c), c23 = *reinterpret_cast<const f64x2 *>(c + 2);
            const double top[5] = {a01[0], a01[1], a23[0], a23[1], a[4]};
            const double bot[5] = {c01[0], c01[1], c23[0], c23[1], c[4]};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const double wx0 = (cx + j == W - 1) ? 2.0 : 1.0;
                box[pl][j] = __builtin_fma(wy0, __builtin_fma(wx0, bot[j + 1], bot[j]), __builtin_fma(wx0, top[j + 1], top[j]));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float v0, v1, v2;
            if (DEPTH) {
                v0 = (float)box[0][j];  v1 = (float)box[1][j];  v2 = (float)box[NP - 1][j];
            } else {                           // A = count * 2^20 + sum(vx): split exactly
                const double cnt = __builtin_rint(box[0][j] * (1.0 / kCountUnit));
                v0 = (float)cnt;
                v1 = (float)__builtin_fma(cnt, -kCountUnit, box[0][j]);
                v2 = (float)box[1][j];
            }
            if (v0 > 0.0f) {                   // my_lib_kernel.cu:1730-1735; one reciprocal for both components
                const float inv = 1.0f / v0;   // (<= 1 ulp from the two divisions)
                v1 = v1 * inv;
                v2 = v2 * inv;
            }
            oc[j] = v0;  ox[j] = v1;  oy[j] = v2;
        }
    }
    if (inb) {
        float *o = out + b * s1b + (int64_t)cy * s1h + cx, *cn = count + b * scb + (int64_t)cy * sch + cx;
        if (ws.up) {                           // plain stores: pass 3 (hole fill) re-reads them
            *reinterpret_cast<f32x4 *>(o) = ox;
            *reinterpret_cast<f32x4 *>(o + s1c) = oy;
            *reinterpret_cast<f32x4 *>(cn) = oc;
        } else {                               // single-use streams otherwise
            st_stream4(o, ox);
            st_stream4(o + s1c, oy);
            st_stream4(cn, oc);
        }
    }
    if (ws.up) {                               // the filler's per-tile summaries, from the counts in registers
        // A tile whose every cell has a positive count (82 % of the tiles under the benchmark's smooth flow) has the
        // trivial summary -- every walk that enters it stops at its first cell -- and no hole: one vote instead of the
        // LDS atomics, the row reductions and their barrier.
        const bool full = !inb || (oc[0] > 0.0f && oc[1] > 0.0f && oc[2] > 0.0f && oc[3] > 0.0f);
        if (__syncthreads_and(full)) {
            const int tx0 = tc.tx * 64, ty0 = tc.ty * TH;
            if (tid < 64 && tx0 + (int)tid < W)
                ws.up[((int64_t)b * tiles_y + tc.ty) * W + tx0 + tid] = min(ty0 + TH - 1, H - 1);
            if (tid < TH && ty0 + (int)tid < H) {
                const int64_t i = ((int64_t)b * tiles_x + tc.tx) * H + ty0 + tid;
                ws.right[i] = tx0;
                ws.left[i] = min(tx0 + 63, W - 1);
            }
            if (tid == 0) ws.hole[((int64_t)b * tiles_y + tc.ty) * tiles_x + tc.tx] = 0;
            return;
        }
        const bool hole = summary_add(sm, inb, oc, 4 * (tid % 16), tid / 16, cx, cy, tid);
        const int any_hole = __syncthreads_or(hole);
        summary_store(sm, any_hole, ws, b, tc.tx, tc.ty, W, H, tiles_x, tiles_y, tid);
    }
}

// The images flagged by proj_owner4, redone exactly: every tile of such an image scans whole source tiles -- those
// whose motion bound (kReach, or what proj_owner4 recorded in bounds[] for the tile's far sources: max |fx|, max |fy|)
// lets one of their sources land in the window -- with no limit on |flow|.  Queued behind proj_owner4 as a short grid that strides over the tiles; returns
// at once when no flag was raised.
template <bool DEPTH, int TH, int kReach>
__global__ __launch_bounds__(16 * TH) void proj_owner_far_r3(
    int W, int H, int tiles_x, int tiles_y, int batch,
    int64_t s1b, int64_t s1c, int s1h, int64_t sdb, int sdh, int64_t scb, int sch,
    const float *__restrict__ flow, const float *__restrict__ depth,
    float *__restrict__ count, float *__restrict__ out, const int *__restrict__ far_flag,
    const int *__restrict__ bounds, FillWs ws, int nonce)
{
    using OT = OwnerTile<DEPTH, TH>;
    constexpr int NT = 16 * TH;
    __shared__ __attribute__((aligned(16))) double P[OT::NP * OT::kPlane];
    __shared__ TileSummary<TH> sm;
    if (far_flag[kFlagWords] != nonce) return;
    const unsigned per_image = (unsigned)tiles_x * tiles_y, ntiles = per_image * batch;
#pragma unroll 1
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / per_image, tx = (tile % per_image) % tiles_x, ty = (tile % per_image) / tiles_x;
        if (far_flag[b % kFlagWords] != nonce) continue;         // wave-uniform: this image was complete
        const int tid = tid_now();
        const int tx0 = tx * 64, ty0 = ty * TH;
        OT t;
        t.begin(P, tx0, ty0, W, H, tid & (kWave - 1));
        summary_init(sm);
        t.template zero<NT>(tid);
        __syncthreads();
        const float *flow_b = flow + b * s1b;
        const float *depth_b = DEPTH ? depth + b * sdb : nullptr;
        const int *bnd = bounds + 2 * (int64_t)b * per_image;
#pragma unroll 1
        for (unsigned s = 0; s < per_image; s++) {
            const int stx = s % tiles_x, sty = s / tiles_x;
            // (recorded: the tile's far sources; its other sources move by less than kReach)
            const float mx = fmaxf(__int_as_float(bnd[2 * s]), (float)kReach), my = fmaxf(__int_as_float(bnd[2 * s + 1]), (float)kReach);
            // can a source of tile s, moved by at most (mx, my) (+1: the rounding of x + fx), land in the window?
            const float sx0 = (float)(stx * 64), sy0 = (float)(sty * TH);
            const bool cand = sx0 + 64.0f + mx >= (float)(tx0 - 1) && sx0 - mx - 1.0f < (float)(tx0 + 64) &&
                              sy0 + (float)TH + my >= (float)(ty0 - 1) && sy0 - my - 1.0f < (float)(ty0 + TH);
            if (!cand) continue;               // wave-uniform (scalar data)
            const int sx = stx * 64 + 4 * (tid % 16), sy = sty * TH + tid / 16;   // one quad of sources per lane
            const bool lv = sx < W && sy < H;
            const unsigned off = lv ? 4u * (unsigned)(sy * s1h + sx) : 0u;
            const f32x4 fxq = ld_cached4_u(flow_b, off), fyq = ld_cached4_u(flow_b + s1c, off);
            f32x4 ddq = {1.f, 1.f, 1.f, 1.f};
            if (DEPTH) ddq = ld_cached4_u(depth_b, lv ? 4u * (unsigned)(sy * sdh + sx) : 0u);
            float y2[4];
            bool wy[4];
            const bool rowany = t.rows(lv, (float)sy, fyq, y2, wy);
            if (__builtin_amdgcn_ballot_w64(rowany) == 0) continue;
#pragma unroll
            for (int j = 0; j < 4; j++)
                t.source(wy[j], ((float)sx + (float)j) + fxq[j], y2[j], fxq[j], fyq[j], ddq[j]);
        }
        t.finish();
        __syncthreads();                       // every wave's points are in P
        f32x4 ox, oy, oc;
        t.readout(tx0 + 4 * (tid % 16), ty0 + tid / 16, W, H, ox, oy, oc);
        owner_store(sm, ws, tid, b, tx, ty, W, H, tiles_x, tiles_y, s1b, s1c, s1h, scb, sch, count, out, ox, oy, oc);
        __syncthreads();                       // P and the summary are rebuilt by the next tile
    }
}

// --------------------------------------------------------------------------------------------------
// Pass 3 with carries: the hole filler whose walks never leave a tile.
// The reference walks from every hole to the nearest cell with a non-zero count to its left, to its right and above
// (my_lib_kernel.cu:1776-1800).  Walked literally, a camera pan -- an uncovered strip along one image border --
// makes every hole of a vertical strip climb the whole strip (measured: projection + fill 765 .. 1320 us against
// 244 .. 266 us without, 720p batch 32).  Here a walk covers its own 64 x TH tile only; what lies beyond comes from
// three small carry tables built by two tiny scans over per-tile summaries:
//   up   [b][ty][x]   last row of band ty whose cell in column x has a non-zero count            (-1: none)
//   left [b][tx][y]   last column of tile column tx with a non-zero count in row y               (-1: none)
//   right[b][tx][y]   first such column
// (round 1 turned these into "nearest beyond the tile" tables with a scan kernel; now the filler walks the
// neighbouring tiles' entries itself, nearest first -- one launch and 13 us less, usually one step)
// plus the list of the tiles that contain a hole (the filler is launched over that list only).
// Same cells, same flags, same arithmetic as the walks -- identical results.  The tables (0.4 B per pixel) live in
// a stream-ordered allocation made and released by the launcher.
// --------------------------------------------------------------------------------------------------
// Summaries from the count plane, for the paths on which the owner kernel did not write them: the general path on
// its own (far_flag == nullptr: every tile) or behind the far flag (only the images it redid; returns at once when
// no image was flagged).  Grid-stride over tiles.
template <int TH>
__global__ __launch_bounds__(256) void proj_fill_summary(
    int W, int H, int tiles_x, int tiles_y, int batch, int64_t scb, int sch, const float *__restrict__ count,
    FillWs ws, const int *__restrict__ far_flag)
{
    __shared__ TileSummary<TH> sm;
    if (far_flag && far_flag[kFlagWords] == 0) return;
    const unsigned ntiles = (unsigned)tiles_x * tiles_y * batch;
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / ((unsigned)tiles_x * tiles_y);
        if (far_flag && far_flag[b % kFlagWords] == 0) continue;          // wave-uniform
        summary_init(sm);
        __syncthreads();
        bool hole = false;
#pragma unroll
        for (int r = 0; r < TH / 16; r++) {
            const int lx = 4 * (threadIdx.x % 16), ly = threadIdx.x / 16 + 16 * r;
            const int x = tx * 64 + lx, y = ty * TH + ly;
            const bool inb = x < W && y < H;
            const f32x4 own = ld_cached4(count + b * scb + (int64_t)min(y, H - 1) * sch + min(x, W - 4));
            hole = summary_add(sm, inb, own, lx, ly, x, y) || hole;
        }
        const int any_hole = __syncthreads_or(hole);       // (also orders the LDS atomics before the reads below)
        summary_store(sm, any_hole, ws, b, tx, ty, W, H, tiles_x, tiles_y);
        __syncthreads();                                   // before the next tile re-initialises the summary
    }
}

// (One WAVE per flagged tile instead of a 256-thread workgroup -- no workgroup barriers, four times the tiles in
// flight -- was measured and LOST: 61 us against 29; a tile's holes come in clusters of more than 64.)
template <int TH>
__global__ __launch_bounds__(256) void proj_fillhole_carry(
    int W, int H, int tiles_x, int tiles_y, int batch, int64_t s1b, int64_t s1c, int s1h, int64_t scb, int sch,
    const float *__restrict__ count, float *out, FillWs ws)
{
    __shared__ __attribute__((aligned(16))) float cnt[TH * 64];
    __shared__ int n_holes;
    __shared__ unsigned short hole_list[TH * 64];
    // workgroup i looks after the tiles i, i + grid, ...: their flags are fetched by one load (lane k: tile i + k grid),
    // then the flagged ones are taken in turn
    const unsigned ntiles = (unsigned)tiles_x * tiles_y * batch;
    const unsigned mine = blockIdx.x + (threadIdx.x & 63u) * gridDim.x;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(mine < ntiles && ws.hole[mine < ntiles ? mine : 0] != 0);
    for (; todo; todo &= todo - 1) {
    const unsigned tile = blockIdx.x + (unsigned)__builtin_ctzll(todo) * gridDim.x;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / ((unsigned)tiles_x * tiles_y);
    const int tx0 = tx * 64, ty0 = ty * TH;
    const float *cn = count + b * scb;
    if (threadIdx.x == 0) n_holes = 0;
    f32x4 own[TH / 16];
#pragma unroll
    for (int r = 0; r < TH / 16; r++) {
        const int lx = 4 * (threadIdx.x % 16), ly = threadIdx.x / 16 + 16 * r;
        const int x = tx0 + lx, y = ty0 + ly;
        const bool inb = x < W && y < H;
        own[r] = ld_cached4(cn + (int64_t)min(y, H - 1) * sch + min(x, W - 4));
        // cells past the image edge are staged as "non-zero": the walks below test the edge themselves
        *reinterpret_cast<f32x4 *>(cnt + ly * 64 + lx) = inb ? own[r] : f32x4{1.f, 1.f, 1.f, 1.f};
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TH / 16; r++) {
        const int lx = 4 * (threadIdx.x % 16), ly = threadIdx.x / 16 + 16 * r;
        if (tx0 + lx < W && ty0 + ly < H) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (own[r][j] <= 0.0f) hole_list[atomicAdd(&n_holes, 1)] = (unsigned short)((ly << 6) | (lx + j));
        }
    }
    __syncthreads();
    const int n = n_holes;
    float *o = out + b * s1b;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int cell = hole_list[i], hx = cell & 63, hy = cell >> 6;
        const int gx = tx0 + hx, gy = ty0 + hy;
        // inside the tile: LDS; beyond it: the carry tables (position only -- the count there is read back)
        int lo = -1, ro = -1, uo = -1;
        for (int c = hx - 1; c >= 0 && lo < 0; c--)
            if (cnt[hy * 64 + c] != 0.0f) lo = tx0 + c;
        // (tile summaries of this row, nearest first: the last non-zero column of each tile to the left)
        for (int t = tx - 1; t >= 0 && lo < 0; t--) lo = ws.left[((int64_t)b * tiles_x + t) * H + gy];
        for (int c = hx + 1; c < 64 && tx0 + c < W && ro < 0; c++)
            if (cnt[hy * 64 + c] != 0.0f) ro = tx0 + c;
        for (int t = tx + 1; t < tiles_x && ro < 0; t++) ro = ws.right[((int64_t)b * tiles_x + t) * H + gy];
        for (int r = hy - 1; r >= 0 && uo < 0; r--)
            if (cnt[r * 64 + hx] != 0.0f) uo = ty0 + r;
        for (int t = ty - 1; t >= 0 && uo < 0; t--) uo = ws.up[((int64_t)b * tiles_y + t) * W + gx];
        // the counts the walks stopped at (0 when they ran into the image border)
        const float lt = lo >= 0 ? cn[(int64_t)gy * sch + lo] : 0.0f;
        const float rt = ro >= 0 ? cn[(int64_t)gy * sch + ro] : 0.0f;
        const float ut = uo >= 0 ? cn[(int64_t)uo * sch + gx] : 0.0f;
        const float dt = 0.0f;                              // dead downward search (my_lib_kernel.cu:1799)
        if (lt + rt + ut + dt <= 0.0f) continue;
        const float fl = lt > 0.0f ? 1.0f : 0.0f, fr = rt > 0.0f ? 1.0f : 0.0f;
        const float fu = ut > 0.0f ? 1.0f : 0.0f, fd = 0.0f;
        // a walk that found nothing ends at the border cell (column 0 / W-1, row 0): its flag is 0, but the
        // reference still multiplies that cell's value by it -- keep the operand finite and identical
        const int lc = lo >= 0 ? lo : 0, rc = ro >= 0 ? ro : W - 1, ur = uo >= 0 ? uo : 0;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            float *pl = o + k * s1c;
            float *self = pl + (int64_t)gy * s1h + gx;
            *self = (fl * pl[(int64_t)gy * s1h + lc] + fr * pl[(int64_t)gy * s1h + rc] +
                     fu * pl[(int64_t)ur * s1h + gx] + fd * *self) / (fl + fr + fu + fd);
        }
    }
    __syncthreads();                                        // the LDS tile and list are reused by the next tile
    }   // listed tiles
}
#endif  // MEMC_PROJ_ARMS_PART_C
