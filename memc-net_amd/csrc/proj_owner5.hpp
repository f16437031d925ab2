// proj_owner5.hpp -- the owner-computes (Depth)FlowProjection forward, round 4.  Textually included by
// flow_projection.hip (namespace memc, behind FillWs / kPtW4 / kCountUnit / trace_mark_proj).
//
// Replaces my_package/src/my_lib_kernel.cu:1630-1735 (scatter + averaging) and :2053-2167 (depth) of the reference.
//
// What round 3's proj_owner4 was bound by (ISA of the product build): 1450 vector and 1900
// SCALAR instructions per wave, both pipes near their issue rate.  Two thirds of them were the per-wave register
// compaction of the hits (ballot, mbcnt, three ds_permute, cyclic batch bookkeeping -- executed by all 64 lanes for
// every one of the 20 sources a lane scans) and the mask algebra of four float compares per source.  Here:
//   * a window test is ONE subtract and ONE unsigned compare per axis: for x2 >= 0 the float order is the order of the
//     bit patterns, so  bits(x2) - bits(xlo) < bits(xhi) - bits(xlo)  (unsigned) is  xlo <= x2 < xhi; negative values,
//     NaNs and values below xlo wrap to huge unsigned numbers and fail;
//   * a hit goes STRAIGHT to the fp64 point planes under its exec mask -- ds_add_f64 costs the same 8.8 clocks per
//     wave-instruction whatever the number of active lanes (tools/probes, round 1), so the ~90 masked splat
//     instructions per plane and tile cost ~1.6 k LDS clocks, against the ~6 k issue clocks the compaction cost the
//     vector and scalar pipes;
//   * far-source detection of the home quads is a max3 chain and one compare per quad.
// Everything else is proj_owner4's: 64 x TH cell tiles, 16 TH lanes, the scan region dilated by kReach, deferred fx loads
// for rows far from the tile, two planes (count * 2^20 + sum vx, sum vy) for FlowProjection and three for the depth
// operator, box sums in double, one reciprocal per cell, far flags by nonce, per-tile landing boxes of the far sources.
//
// Round 5: the scan region is SHIFTED by the image's dominant motion m (a camera pan).  A source s lands at s + f; the
// sources that land in the tile T are those around T - m, so T scans [T - m] dilated by kReach, and "far" means
// |f - m| >= kReach on an axis.  m is the mean flow of 64 fixed sites of the image, rounded to multiples of 4 px (0 below 6 px) -- every
// workgroup (and proj_owner_far) computes the same value from the same loads in the same order.  Exactness: (i) every valid
// source with |f - m| < kReach on both axes lies inside the scan region of the tile that owns its point (s = p - f with p in
// the window: the argument of the unshifted scan with f - m in place of f), and the hit test asks for the window only;
// (ii) every other valid source is found by the tile it LIES in (its home), which raises the image's flag, records where
// its far sources land and stamps the tiles that box meets -- itself included when m != 0, because its own scan no longer
// covers its own sources then; proj_owner_far recomputes the stamped tiles from all sources.  m == 0 -- any flow without a
// dominant motion of 2 px or more, the benchmark's among them -- is the round-4 kernel: the scan's loads are issued for
// m = 0 before m is known (wave 0 requests the 64 samples FIRST, reduces them while the planes are zeroed and posts m
// before the barrier that was there anyway); only m != 0 pays a second round of loads.
//
// Round 6 took the estimate apart on the benchmark's flow, where it buys nothing (m = 0) and round 5's kernel ran 2-3 % behind
// round 4's (measurement build, template parameter MOT; profiles/r06_proj_motion_estimate_arms.txt, _scalar_arm.txt,
// r06_proj_speculative_scan_ab_*.txt):
//   * MOT 2, no estimate at all: -0 ... -3.4 % -- that is all there is to win, and it moves by more than a per cent from one build
//     of the library to the next (the arms' code shifts the kernel in instruction memory);
//   * MOT 1, the round-5 review's proposal -- scan speculatively for m = 0 without waiting for the samples, let wave 0 post m
//     after its first iteration (samples by LDS DMA: no register is held while they fly), read m behind the scan's closing
//     barrier and start over, shifted, when the image moves: +0 ... 2 % on the benchmark's flow instead of a gain (wave 0's
//     reduction now sits INSIDE the scan, where it delays the barrier; in front of the first barrier it ran in the dead time
//     of the scan's own loads), the depth operator lost a workgroup per CU to the samples' 512 bytes of LDS (+19 %), and a 40 px
//     pan paid the wasted pass: 190 instead of 135 us.  Lost on every count;
//   * MOT 5, the estimate cached per image in the call's scratch (the image's first workgroups publish, the others read one word):
//     +2.5 ... 4 % -- one hot word per image costs more than the 64 warm sample lines; read with device scope it doubled the call;
//   * MOT 3 / MOT 4, sixteen samples instead of sixty-four -- one lane each, or through the scalar unit (s_load_dword via the
//     constant address space): -1 ... 1.5 % on the depth operator, nothing on FlowProjection.
// The product stays round 5's order (MOT 0): wave 0 requests the samples first, reduces them while the planes are zeroed and
// posts m in front of the first barrier.
#pragma once

// (motion_sample_issue / motion_reduce: flow_projection.hip, shared with proj_owner_far)
// FIX64 (measurement build, projection variant -46; depth operator only): the three sums as 64-bit FIXED POINT on ds_add_u64
// (9.0 against 6.7 lane-operations per clock for ds_add_f64), at a fixed scale of 2^30 -- right for the benchmark's depths
// in [0.1, 1.1), a timing arm for anything else: what would the fixed-point planes of the round-4 review's item 5 buy
// before their per-tile scale, its barrier and the outlier route are built?
// RAG: rows whose width is not a multiple of four (tail_shift / tail_fix / st_tail4, flow_projection.hip).
// MOT: how the motion estimate reaches the scan (0 = the product; 1-4: measurement build, projection variants -47 ... -50, see
// "Round 6" above).
template <bool DEPTH, int TH, int kReach, int MINW, bool TRACE = false, bool FIX64 = false, bool RAG = false, int MOT = 0, int PENDT = -1>
__global__ __launch_bounds__(16 * TH, MINW) void proj_owner5(
    int W, int H, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t sdb, int sdh, int64_t scb, int sch,
    const float *__restrict__ flow, const float *__restrict__ depth,
    float *__restrict__ count, float *__restrict__ out, int *__restrict__ far_flag, int *__restrict__ bounds,
    int *__restrict__ stamps, FillWs ws, WalkPlan plan, int nonce_arg)
{
    // MOT: how the image's motion gets to the scan.  0: round 5's order (the product).  Measurement build: 1 = the speculative m = 0
    // pass ("Round 6" above: measured, lost); 2 = no estimate at all, m = 0 (timing arm: what does the estimate cost?); 3 = round 5's
    // order on 16 samples instead of 64 (timing arm; proj_owner_far still takes 64: only for flows where both say 0).
    constexpr bool SPEC = MOT == 1;
    // this call's tag: a host counter's value, or -- 0: the call was recorded into a HIP graph, every replay needs its own --
    // the device counter proj_bump_nonce advanced in front of this kernel
    const int nonce = nonce_arg ? nonce_arg : __builtin_amdgcn_readfirstlane(far_flag[kFlagWords + 1]);
    constexpr int NT = 16 * TH;                   // one lane per four owned cells
    constexpr int NP = DEPTH ? 3 : 2;             // planes: (count, vx, vy) or (count * 2^20 + vx, vy)
    constexpr int kPtH = TH + 1, kPlane = kPtH * kPtW4;
    // Scan region: the sources that can reach the tile's window, rows [ty0 - kReach - 1, ty0 + TH + kReach), columns
    // [tx0 - kReach - 4, tx0 + 64 + kReach + 4) -- laid out 32 quads (128 columns, [tx0 - 32, tx0 + 96)) wide: a lane's
    // column is tid % 32 for the whole scan and its row advances by NT / 32 per iteration, so a slot's coordinates,
    // its "inside the image" test and its load offset are one add each (round 4's first version laid the 30 needed quads
    // out densely: a division, a wrap test and a multiply per slot -- 180 of the kernel's 530 vector instructions per wave).
    // The two outermost quads on either side are never needed; what they hit can only be a far source's point, and an image
    // with a valid far source is redone anyway.
    static_assert(kReach + 4 <= 32, "32 quads per scan row");
    constexpr int kColsQ = 32, kRowsIt = NT / kColsQ;
    constexpr int kScanH = TH + 2 * kReach + 1;   // source rows
    constexpr int kIts = (kScanH + kRowsIt - 1) / kRowsIt;
    static_assert(kPlane % 2 == 0, "P is zeroed 16 bytes at a time");
    static_assert(kIts >= 3, "the motion is posted in front of iteration 1 and looked at in front of iteration 2");
    static_assert(NP * kPlane * 8 >= 3 * TH * 64 * 4 + TH * 64 * 2, "the fill epilogue stages three planes and the holes' list in P");
    static_assert((2 * kReach + 1) * (2 * kReach + 1) < 4096 && kReach <= 128, "count * 2^20 + sum(vx) must split exactly");
    __shared__ __attribute__((aligned(16))) double P[NP * kPlane];
    __shared__ FillLds<TH> fl;                    // the hole filler's masks (fill epilogue only)
    __shared__ int tile_box[8];                   // where the tile's own FAR sources land: bit patterns of min x2, max x2, min y2,
                                                  // max y2 (all >= 0: ordered like ints); max < 0: the tile has none.  [4]: the
                                                  // tile has sources that are NOT far.  [6], [7]: the image's motion (mx, my)

    const TileCoord tc = plan.fast ? tile_walk_plan(blockIdx.x, plan)
                                   : tile_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, plan.sw > 1 || plan.stripes_x != (unsigned)tiles_x ? (int)plan.sw : 0);
    if (tc.tx >= tiles_x) return;                 // virtual column of the last stripe
    const int b = tc.b, tx0 = tc.tx * 64, ty0 = tc.ty * TH;
    const int tid0 = threadIdx.x;                 // (thread index of the first half of the kernel, see below)
    const int wave_index = __builtin_amdgcn_readfirstlane(tid0 / kWave);
    trace_mark_proj<TRACE>(0);
    const float *flow_b = flow + b * s1b;
    const float *depth_b = DEPTH ? depth + b * sdb : nullptr;
    // wave 0: the image's 64 motion samples, requested before anything else -- round 5 into two registers, consumed in front of
    // the first barrier; SPEC into LDS (they are consumed after the first scan iteration, and the 64 registers of the eight
    // waves per SIMD have none free until then)
    __shared__ float msamp[SPEC ? 128 : 2];
    float msx = 0.0f, msy = 0.0f;
    unsigned long long mcached = 0;            // MOT 5: the image's cached estimate (requested before anything else)
    if (MOT == 5 && wave_index == 0)
        // (a PLAIN load: it is served by this XCD's L2, where the image's first workgroup ON THIS XCD published -- eight publishers per
        // image, each XCD its own; a device-scope load goes past the L2 to one hot line for all 14 400 workgroups: 2x the whole call)
        mcached = *(reinterpret_cast<const unsigned long long *>(far_flag + kMotionCacheAt) + (b % kMotionCacheWords));
    if (wave_index == 0 && MOT != 2 && MOT != 4 && MOT != 5) {
        if (SPEC) motion_sample_issue_lds(flow_b, s1c, s1h, W, H, tid0, msamp);
        else if (MOT == 3) motion_sample_issue16(flow_b, s1c, s1h, W, H, tid0, msx, msy);
        else motion_sample_issue(flow_b, s1c, s1h, W, H, tid0, msx, msy);
    }
    fill_lds_init(fl, tid0);
    constexpr int kUnposted = 0x7fffffff;       // tile_box[6] before wave 0 has posted the motion (never a motion: |m| <= 4096)
    auto tile_box_init = [&]() __attribute__((always_inline)) {                // (wave 1: wave 0 has its samples to look after)
        if (tid0 >= kWave && tid0 < kWave + 5) tile_box[tid0 - kWave] = tid0 == kWave + 4 ? 0 : (tid0 & 1) ? -1 : 0x7fffffff;
    };
    tile_box_init();
    if (SPEC && tid0 == kWave + 6) tile_box[6] = kUnposted;
    auto zero_planes = [&]() __attribute__((always_inline)) {
        f32x4 *pz = reinterpret_cast<f32x4 *>(P);
        for (int i = tid0; i < NP * kPlane / 2; i += NT) pz[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_planes();

    // scan loads: fy of every slot now, fx / depth now only for the slots whose rows lie within kNearRows of the tile (the
    // others almost never pass the row test and fetch theirs inside the branch); unconditional addresses (dead slots read
    // pixel 0)
    // near slots: rows less than 10 above / 7 below the tile (TH = 32: iterations 1 - 3 of six)
    auto far_it = [](int it) __attribute__((always_inline)) {
        const int first = kRowsIt * it, last = kRowsIt * it + kRowsIt - 1;
        return last <= kReach + 1 - 10 || first >= kReach + 1 + TH + 7;
    };
    // the last iteration may hold a row or two only (TH = 32: row 80 of 81, half of the first wave): nothing of it is
    // requested up front
    auto partial_it = [](int it) __attribute__((always_inline)) { return kRowsIt * (it + 1) > kScanH; };
    f32x4 fx[kIts], fy[kIts], dd[kIts];
    bool live[kIts];
    // loads scan_issue requests up front (per lane and wave: the addresses are unconditional)
    constexpr int kScanLoads = [&]() {
        int n = 0;
        for (int it = 0; it < kIts; it++) {
            const int first = kRowsIt * it, last = kRowsIt * it + kRowsIt - 1;
            const bool far_i = last <= kReach + 1 - 10 || first >= kReach + 1 + TH + 7, partial_i = kRowsIt * (it + 1) > kScanH;
            if (!partial_i) n += 1 + (far_i ? 0 : (DEPTH ? 2 : 1));
        }
        return n;
    }();
    int sx, sy0, rt = 0;                       // rt (RAG): sites of the lane's quads that lie past the row's end
    unsigned off0, offd0;
    const float kNaN = __int_as_float(0x7fc00000);
    // the scan's slots for the motion (mx, my) and their loads (first for (0, 0), before the motion is known; see the header)
    auto scan_issue = [&](int mx, int my) __attribute__((always_inline)) {
        sx = tx0 - mx - 32 + 4 * (tid0 % kColsQ);
        sy0 = ty0 - my - kReach - 1 + tid0 / kColsQ;
        const bool colok = sx >= 0 && sx < W;     // (W % 4 == 0, or RAG: the row's last quad is partly inside)
        if (RAG) rt = tail_shift(sx, W);
        off0 = 4u * (unsigned)(sy0 * s1h + sx - rt);
        offd0 = DEPTH ? 4u * (unsigned)(sy0 * sdh + sx - rt) : 0u;
#pragma unroll
        for (int it = 0; it < kIts; it++) {
            const int sy = sy0 + kRowsIt * it;
            live[it] = colok && sy >= 0 && sy < H && (kRowsIt * (it + 1) <= kScanH || tid0 / kColsQ + kRowsIt * it < kScanH);
            const unsigned off = live[it] ? off0 + (unsigned)(4 * kRowsIt * it) * (unsigned)s1h : 0u;
            if (!partial_it(it)) fy[it] = ld_cached4_u(flow_b + s1c, off);
            if (!far_it(it) && !partial_it(it)) {
                fx[it] = ld_cached4_u(flow_b, off);
                if (DEPTH) dd[it] = ld_cached4_u(depth_b, live[it] ? offd0 + (unsigned)(4 * kRowsIt * it) * (unsigned)sdh : 0u);
            }
        }
    };
    scan_issue(0, 0);
    auto motion_post = [&]() __attribute__((always_inline)) {                 // wave 0: the image's motion from its 64 samples (requested before the scan's loads:
        int pmx, pmy;                          // waiting for them does not wait for the scan), one 8-byte store
        if (SPEC) {                            // (the scan's up-front loads are all that was issued behind the samples by then)
            motion_samples_wait<kScanLoads>();
            msx = msamp[tid0 & 63];
            msy = msamp[64 + (tid0 & 63)];
        }
        if (MOT == 3) motion_reduce16(msx, msy, pmx, pmy);
        else motion_reduce(msx, msy, pmx, pmy);
        typedef int i32x2 __attribute__((ext_vector_type(2)));
        if (tid0 == 0) *reinterpret_cast<i32x2 *>(&tile_box[6]) = i32x2{pmx, pmy};
    };
    if ((MOT == 0 || MOT == 3) && wave_index == 0) motion_post();   // (round 5: in front of the first barrier)
    if (MOT == 5 && wave_index == 0) {             // the cached estimate, or -- the image's first workgroups -- sample and publish
        const unsigned tag = motion_tag(nonce, b);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(mcached >> 32));
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)mcached);
        int pmx, pmy;
        if (hi == tag) {                           // (wave-uniform)
            motion_unpack(lo, pmx, pmy);
        } else {
            motion_sample_issue(flow_b, s1c, s1h, W, H, tid0, msx, msy);
            motion_reduce(msx, msy, pmx, pmy);
            if (tid0 == 0)
                *(reinterpret_cast<unsigned long long *>(far_flag + kMotionCacheAt) + (b % kMotionCacheWords)) =
                    ((unsigned long long)tag << 32) | motion_pack(pmx, pmy);
        }
        typedef int i32x2 __attribute__((ext_vector_type(2)));
        if (tid0 == 0) *reinterpret_cast<i32x2 *>(&tile_box[6]) = i32x2{pmx, pmy};
    }
    if (MOT == 4 && wave_index == 0) {             // the estimate through the scalar unit, behind the scan's requests
        int pmx, pmy;
        motion_estimate_scalar(flow_b, s1c, s1h, W, H, pmx, pmy);
        typedef int i32x2 __attribute__((ext_vector_type(2)));
        if (tid0 == 0) *reinterpret_cast<i32x2 *>(&tile_box[6]) = i32x2{pmx, pmy};
    }
    // (posting m BEHIND the barrier and letting the other waves poll the LDS word, so that nobody sits at the barrier for the
    // samples' round trip, was measured: +5.1 % against round 4's kernel instead of +2.2 % -- profiles/r05_proj_motion_polled_ab.txt)
    trace_mark_proj<TRACE>(1);                 // loads issued, P zeroed
    // (nothing of the scan may move in front of this barrier: with m = 0 known at compile time the first iteration's row tests
    // depend on nothing behind it, and hoisted they made every wave wait for its first load BEFORE the barrier -- the
    // workgroup then started its scan when its slowest wave's data had arrived)
    if (MOT == 1 || MOT == 2) __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                           // P is zero (round 5: and the image's motion is posted)
    if (MOT == 1 || MOT == 2) __builtin_amdgcn_sched_barrier(0);
    if (TRACE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    trace_mark_proj<TRACE>(2);                 // loads arrived
    // the motion this pass assumes: SPEC starts with (0, 0) and learns the posted one behind the scan's closing barrier
    int mx = 0, my = 0;
    if (MOT == 0 || MOT == 3 || MOT == 4 || MOT == 5) {
        mx = __builtin_amdgcn_readfirstlane(tile_box[6]);
        my = __builtin_amdgcn_readfirstlane(tile_box[7]);
    }
    bool shifted = false;                      // (scalar)
    float mxf = 0.0f, myf = 0.0f;

    // Window [ty0 - 1, ty0 + TH - 1] x [tx0 - 1, tx0 + 63] of the points (T, L) = ((int)y2, (int)x2), and validity
    // (0 <= x2 <= W - 1, my_lib_kernel.cu:1670), as ONE range test per axis on the bit patterns.
    const int xlo_b = __float_as_int((float)max(tx0 - 1, 0)), ylo_b = __float_as_int((float)max(ty0 - 1, 0));
    const unsigned xrange = (unsigned)(min(__float_as_int((float)(tx0 + 64)), __float_as_int((float)(W - 1)) + 1) - xlo_b);
    const unsigned yrange = (unsigned)(min(__float_as_int((float)(ty0 + TH)), __float_as_int((float)(H - 1)) + 1) - ylo_b);
    // point (py, px) -> byte offset 8 (py * pitch + px - cell0) into a plane (the constants pinned in SGPRs: rematerialised at every use
    // they cost the scalar pipe, which is as busy as the vector one here, two or three moves per source)
    unsigned pitch8 = 8 * kPtW4, ucell8 = (unsigned)-(8 * ((ty0 - 1) * kPtW4 + (tx0 - 1)));
    asm volatile("" : "+s"(pitch8), "+s"(ucell8));
    double kunit = kCountUnit;                 // (set per pass, below)
    bool far = false;
    unsigned long long near_seen = 0;             // (scalar) home quads of this wave that hold a source that is not far

    // The cold branch of the far test: the home quads of the wave that hold a far source (ballot farq != 0).  Where the
    // tile's far sources land goes to tile_box (the wave's box first, ONE lane's atomics then: 64 lanes on one LDS word
    // serialise -- a camera pan, where every source was far before round 5, ran this kernel at 1.1 ms instead of 0.12).
    auto far_quads = [&](bool homeq, const f32x4 &a, const f32x4 &c, float sxf, float syf) __attribute__((always_inline)) {
        int bx0 = 0x7fffffff, bx1 = -1, by0 = 0x7fffffff, by1 = -1;
        bool nearj = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float fxv = a[j], fyv = c[j], x2 = (sxf + (float)j) + fxv, y2 = syf + fyv;
            const bool nr = fabsf(fxv - mxf) < (float)kReach && fabsf(fyv - myf) < (float)kReach;
            nearj = nearj || (homeq && nr);
            const bool valid = homeq && !nr && x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(W - 1) && y2 <= (float)(H - 1);
            if (valid) {               // where the tile's far sources land, for proj_owner_far (x2, y2 >= 0:
                bx0 = min(bx0, __float_as_int(x2));                       // non-negative floats order like their bits)
                bx1 = max(bx1, __float_as_int(x2));
                by0 = min(by0, __float_as_int(y2));
                by1 = max(by1, __float_as_int(y2));
            }
        }
        near_seen |= __builtin_amdgcn_ballot_w64(nearj);
        bx1 = -wave_min_i32(-bx1);
        if (bx1 >= 0) {                // wave-uniform: a valid far source
            far = true;
            bx0 = wave_min_i32(bx0);
            by0 = wave_min_i32(by0);
            by1 = -wave_min_i32(-by1);
            if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) {
                atomicMin(&tile_box[0], bx0);
                atomicMax(&tile_box[1], bx1);
                atomicMin(&tile_box[2], by0);
                atomicMax(&tile_box[3], by1);
            }
        }
    };
    // the tile's own quad of sources per lane (images that move as a whole only)
    f32x4 home_a = {0.f, 0.f, 0.f, 0.f}, home_c = home_a;
    auto home_issue = [&]() __attribute__((always_inline)) {
        const int hx = tx0 + 4 * (tid0 % 16), hy = ty0 + tid0 / 16;
        const bool homeq = hx < W && hy < H;
        const int rh = RAG ? tail_shift(hx, W) : 0;
        const unsigned offh = homeq ? 4u * (unsigned)(hy * s1h + hx - rh) : 0u;
        home_a = ld_cached4_u(flow_b, offh);
        home_c = ld_cached4_u(flow_b + s1c, offh);
    };
    auto home_test = [&]() __attribute__((always_inline)) {
        const int hx = tx0 + 4 * (tid0 % 16), hy = ty0 + tid0 / 16;
        const bool homeq = hx < W && hy < H;
        f32x4 a = home_a, c = home_c;
        if (RAG) {
            const int rh = tail_shift(hx, W);
            a = tail_fix(a, rh, kNaN);
            c = tail_fix(c, rh, kNaN);
        }
        float dmax = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) dmax = fmaxf(dmax, fmaxf(fabsf(a[j] - mxf), fabsf(c[j] - myf)));
        const unsigned long long farq = __builtin_amdgcn_ballot_w64(homeq && !(dmax < (float)kReach));   // (NaN: far, found not valid)
        near_seen |= __builtin_amdgcn_ballot_w64(homeq) & ~farq;
        if (farq != 0) far_quads(homeq, a, c, (float)hx, (float)hy);
    };
    // One pass of scan + splat for the motion in (mx, my).  MODE 0: SPEC's speculative pass -- m = 0 at compile time, the code
    // of round 4's kernel plus wave 0's posting and one look at the posted word; MODE 1: SPEC's second pass, for an image that
    // moves (shifted at compile time); MODE 2: round 5's single pass, the posted motion known when it starts.  (Two inlined
    // copies rather than a loop around one: as a loop the scan's registers were live around the back edge and 30 of them
    // spilled at the 64 the eight waves per SIMD leave; the cold copy costs instruction memory only.)
    auto scan_pass = [&](auto mode_tag) __attribute__((always_inline)) {   // (called twice: left alone, the compiler makes it a function)
    constexpr int MODE = decltype(mode_tag)::value;
    shifted = MODE == 0 ? false : MODE == 1 ? true : (mx | my) != 0;
    mxf = MODE == 0 ? 0.0f : (float)mx;
    myf = MODE == 0 ? 0.0f : (float)my;
    // The packed plane holds count * 2^20 + sum(mx - fx): the RESIDUAL of the image's motion, not -fx itself.  A source
    // that is not far has |fx - mx| < kReach whatever the pan, so |sum w S| < 2^19 holds under any shift (with -fx in the
    // plane a 47 x 24 block converging on a corner cell under a 216 px pan summed to 975 k and the split came out one count
    // off -- round-5 review).  One source adds (2^20 + mx) - fx: the same two instructions as before, mx folded into the
    // unit (exact: an integer below 2^13 on 2^20); the readout takes cnt * (2^20 + mx) off again.
    {   // (built from integers -- scalar instructions; a conversion and an add would be vector ones.  n = 2^20 + mx lies in
        // (2^19, 2^21): a double with exponent 20 or 19 and the integer's low bits at the top of its mantissa)
        const unsigned kunit_n = (unsigned)(1048576 + (MODE == 0 ? 0 : mx));
        const unsigned kunit_hi = kunit_n >= 1048576u ? (1043u << 20) | (kunit_n - 1048576u)        // 2^20 (1 + (n - 2^20) / 2^20)
                                                      : (1042u << 20) | ((kunit_n - 524288u) << 1);  // 2^19 (1 + (n - 2^19) / 2^19)
        kunit = __longlong_as_double((long long)((unsigned long long)kunit_hi << 32));
        asm volatile("" : "+s"(kunit));
    }
    if (MODE != 0 && __builtin_expect(shifted, 0)) {   // (scalar) the image moves as a whole: see the header
        // The tile's own 64 x TH sources are no longer (all) inside its scan: one quad per lane, tested for far sources here.
        // MODE 2 requests the scan's loads behind the home quad's and tests the quad while they fly (one more round trip under
        // a pan, not two); MODE 1's home quad was requested before the planes were cleared (home_issue, in front of the
        // barrier) and is tested -- and its registers are free -- before the scan's loads go out.
        if (MODE == 2) {
            home_issue();
            scan_issue(mx, my);
            home_test();
        } else {
            home_test();
            scan_issue(mx, my);
        }
    }

#pragma unroll
    for (int it = 0; it < kIts; it++) {
        if (SPEC && MODE == 0 && it == 1 && wave_index == 0) motion_post();   // (its first iteration's data has come back: so have its samples)
        if (SPEC && MODE == 0 && it == 2) {        // one look at the posted motion: a pan is not scanned to the end for m = 0
            typedef int i32x2 __attribute__((ext_vector_type(2)));
            const i32x2 pm = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(&tile_box[6]));   // (a fresh LDS read)
            const int pmx = __builtin_amdgcn_readfirstlane(pm[0]), pmy = __builtin_amdgcn_readfirstlane(pm[1]);
            if (pmx != kUnposted && (pmx | pmy) != 0) break;                  // (wave-uniform)
        }
        const bool lv = live[it];
        // dead slots (outside the image / the scan region) fail every window test: their row is NaN
        const int sy = sy0 + kRowsIt * it;
        const float syf = lv ? (float)sy : __int_as_float(0x7fc00000), sxf = (float)sx;
        // The quad lies inside the tile itself (tx0, the pad and sx are multiples of 4: all four sites or none) -- only
        // slots that can hold rows of the tile evaluate this (and the far-source test below) at all.
        if (RAG && !partial_it(it)) {             // the row's last quad: registers rotated back, sites past the row NaN
            fy[it] = tail_fix(fy[it], rt, kNaN);
            if (!far_it(it)) {
                fx[it] = tail_fix(fx[it], rt, kNaN);
                if (DEPTH) dd[it] = tail_fix(dd[it], rt, kNaN);
            }
        }
        const bool kHomeIt = kRowsIt * it < kReach + 1 + TH && kRowsIt * it + kRowsIt - 1 >= kReach + 1;       // folds: `it` is unrolled
        if (kHomeIt && !shifted) {
            // A far source (|f| >= kReach) whose home is this tile: the image is redone by proj_owner_far.  The hit test
            // below does NOT ask for |f| < kReach: an image without a valid far source has only near hits, which every
            // owner of their point sees (they lie inside its scan region); in an image WITH one the owners may disagree
            // -- and every tile of that image is recomputed anyway.  One max chain and one compare per quad; NaN motion
            // takes the branch too and is found not valid.
            const bool homeq = lv && (unsigned)(sy - ty0) < (unsigned)TH && (unsigned)(sx - tx0) < 64u;
            const f32x4 &a = fx[it], &c = fy[it];
            const float m = fmaxf(fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))),
                                  fmaxf(fmaxf(fabsf(c[0]), fabsf(c[1])), fmaxf(fabsf(c[2]), fabsf(c[3]))));
            const unsigned long long farq = __builtin_amdgcn_ballot_w64(homeq && !(m < (float)kReach));
            near_seen |= __builtin_amdgcn_ballot_w64(homeq) & ~farq;
            if (__builtin_expect(farq != 0, 0)) far_quads(homeq, a, c, sxf, syf);                         // wave-uniform, cold
        }
        if (partial_it(it)) {                  // (wave-uniform: most waves have no slot here at all)
            if (__builtin_amdgcn_ballot_w64(lv) == 0) continue;
            fy[it] = ld_cached4_u(flow_b + s1c, lv ? off0 + (unsigned)(4 * kRowsIt * it) * (unsigned)s1h : 0u);
            if (RAG) fy[it] = tail_fix(fy[it], rt, kNaN);
        }
        float y2[4];
        bool wy[4], rowany = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            y2[j] = syf + fy[it][j];
            wy[j] = (unsigned)(__float_as_int(y2[j]) - ylo_b) < yrange;
            rowany = rowany || wy[j];
        }
        // rows farther from the tile than the local motion: the whole wave leaves after the four y tests
        if (__builtin_amdgcn_ballot_w64(rowany) == 0) continue;
        f32x4 fxq = fx[it], ddq = dd[it];
        // (round 5 tried the near rows FIRST and the far rows' row tests early -- the j-th far iteration's behind the j-th near
        // one --, so that this request has the remaining near iterations to arrive in: the owner kernel at twice the
        // benchmark's motion stayed at 185 us (182), the benchmark's flow lost 3 % more -- the 28 us that motion costs are not
        // these round trips; profiles/r05_proj_scan_reordered_early_far_rows_ab.txt.  Reverted.)
        if (far_it(it) || partial_it(it)) {    // rare: requested only now (and consumed inside this branch)
            const unsigned off = lv ? off0 + (unsigned)(4 * kRowsIt * it) * (unsigned)s1h : 0u;
            fxq = ld_cached4_u(flow_b, off);
            if (DEPTH) ddq = ld_cached4_u(depth_b, lv ? offd0 + (unsigned)(4 * kRowsIt * it) * (unsigned)sdh : 0u);
            if (RAG) {
                fxq = tail_fix(fxq, rt, kNaN);
                if (DEPTH) ddq = tail_fix(ddq, rt, kNaN);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float fxv = fxq[j], fyv = fy[it][j];
            const float x2 = (sxf + (float)j) + fxv;           // (float)x + fx, as the reference rounds it
            if (wy[j] && (unsigned)(__float_as_int(x2) - xlo_b) < xrange) {
                const unsigned a = __umul24((unsigned)(int)y2[j], pitch8) + ((((unsigned)(int)x2) << 3) + ucell8);
                double *q = reinterpret_cast<double *>(reinterpret_cast<char *>(P) + a);
                if (DEPTH && FIX64) {          // (timing arm) x * 2^30 rounded to an integer: the low mantissa bits of x * 2^30 + 1.5 * 2^52
                    const float d = ddq[j];
                    const double kMagic = 6755399441055744.0, kScale = 1073741824.0;
                    const long long kBits = __double_as_longlong(kMagic);
                    unsigned long long *qu = reinterpret_cast<unsigned long long *>(q);
                    __hip_atomic_fetch_add(qu, (unsigned long long)(__double_as_longlong(__builtin_fma((double)(d * 1.0f), kScale, kMagic)) - kBits),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(qu + kPlane, (unsigned long long)(__double_as_longlong(__builtin_fma(-(double)(d * fxv), kScale, kMagic)) - kBits),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(qu + 2 * kPlane, (unsigned long long)(__double_as_longlong(__builtin_fma(-(double)(d * fyv), kScale, kMagic)) - kBits),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if (DEPTH) {            // my_lib_kernel.cu:2102-2114: v = -d * f, count += d
                    const float d = ddq[j];
                    lds_add_f64(q, (double)(d * 1.0f));
                    lds_add_f64(q + kPlane, -(double)(d * fxv));
                    lds_add_f64(q + 2 * kPlane, -(double)(d * fyv));
                } else {                       // one source: count += 1, sum(vx - mx) += mx - fx
                    lds_add_f64(q, kunit - (double)fxv);
                    lds_add_f64(q + kPlane, -(double)fyv);
                }
            }
        }
    }
    if (near_seen != 0 && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) tile_box[4] = 1;
    };
    if (SPEC) {
        scan_pass(std::integral_constant<int, 0>{});
        trace_mark_proj<TRACE>(3);             // scan + splat done (wave 0)
        __syncthreads();                       // every wave's points are in P (and the tile's far landing box in tile_box)
        trace_mark_proj<TRACE>(4);
        // wave 0 posted the image's motion before it came to the barrier
        mx = __builtin_amdgcn_readfirstlane(tile_box[6]);
        my = __builtin_amdgcn_readfirstlane(tile_box[7]);
        if (__builtin_expect((mx | my) != 0, 0)) {
            // (cold) the image moves as a whole: what the first pass splatted, found far and recorded means nothing -- start over
            home_issue();                      // (the tile's own sources: back by the time the planes are clear)
            zero_planes();
            tile_box_init();
            far = false;
            near_seen = 0;
            __syncthreads();
            scan_pass(std::integral_constant<int, 1>{});
            __syncthreads();
        }
    } else {
        if (MOT == 2) scan_pass(std::integral_constant<int, 0>{});
        else scan_pass(std::integral_constant<int, 2>{});
        trace_mark_proj<TRACE>(3);
        __syncthreads();
        trace_mark_proj<TRACE>(4);
    }
    if (far) {                                 // this image is redone by proj_owner_far.  The flag words are NOT cleared
        far_flag[b % kFlagWords] = nonce;      // before the call: "raised" = "holds this call's nonce" (launcher), so stale
        far_flag[kFlagWords] = nonce;          // or uninitialised words can at worst cause a needless redo, never a missed one
    }
    // (from here on the thread index is REBUILT from the wave's index, a scalar, and the lane's rank in the wave:
    // it need not live through the scan)
    const int tid = wave_index * kWave + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    // bounds[]: 8 words per tile -- the box its far sources land in (4), one unused, "the tile has sources that are not far"
    // (1), 2 unused; stamps[]: one word per tile, "a far source of another tile lands here": this call's nonce, stamped by
    // that tile (a dense array: proj_owner_far reads ALL of them, 64 consecutive words per wave and load)
    const int64_t tile_lin = ((int64_t)b * tiles_y + tc.ty) * tiles_x + tc.tx;
    if (tid < 6 && tid != 4) bounds[kFarWords * tile_lin + tid] = tile_box[tid < 4 ? tid : 4];
    if (__builtin_expect(tile_box[1] >= 0, 0)) {   // cold: stamp the tiles whose window -- x2 in [tx0 - 1, tx0 + 64), y2 in
        // [ty0 - 1, ty0 + TH) -- meets the box: proj_owner_far recomputes those.  Not this tile: it scanned its own sources.
        const int xa = (int)__int_as_float(tile_box[0]), xb = (int)__int_as_float(tile_box[1]) + 1;
        const int ya = (int)__int_as_float(tile_box[2]), yb = (int)__int_as_float(tile_box[3]) + 1;
        const int txa = xa / 64, txb = min(xb / 64, tiles_x - 1), tya = ya / TH, tyb = min(yb / TH, tiles_y - 1);
        const int nx = txb - txa + 1, n = nx * (tyb - tya + 1);
        for (int i = tid; i < n; i += NT) {
            const int ty = tya + i / nx, tx = txa + i % nx;
            // (not this tile when its scan covered its own sources: what they hit inside its window it splatted itself)
            if (tx != tc.tx || ty != tc.ty || shifted) stamps[((int64_t)b * tiles_y + ty) * tiles_x + tx] = nonce;
        }
    }

    // Every lane owns four cells of a row: 2x2 box sums of the points of columns c-1 .. c+3, rows cy-1 and cy (border
    // duplicates as weights 2, see proj_scatter_tiled), summed in DOUBLE -- exact, also for the packed plane:
    // sum_i w_i (count_i 2^20 + S_i) = (sum w count) 2^20 + sum w S with |sum w S| < 2^19 -- S the residual mx - fx: at most
    // 2401 sources that are not far reach a point, weights <= 4, |mx - fx| < kReach: 230 k; the tile's OWN far sources (m = 0
    // only: a tile is not stamped for those) add at most 4 * 32 * (1 + ... + 64) = 266 k -- then split and rounded to fp32
    // ONCE per cell.
    const int cx = tx0 + 4 * (tid % 16), cy = ty0 + tid / 16;
    const bool inb = cx < W && cy < H;            // (no early exit: the epilogue below has barriers)
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
            if (DEPTH && FIX64) {              // (timing arm) the cells hold integers: box sums in integers, one conversion per cell
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int sx0 = (cx + j == W - 1) ? 1 : 0, sy0 = (cy == H - 1) ? 1 : 0;
                    const long long t = (__double_as_longlong(top[j + 1]) << sx0) + __double_as_longlong(top[j]);
                    const long long u = (__double_as_longlong(bot[j + 1]) << sx0) + __double_as_longlong(bot[j]);
                    box[pl][j] = (double)((u << sy0) + t) * (1.0 / 1073741824.0);
                }
                continue;
            }
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
            } else {                           // A = count * 2^20 + sum(mx - fx): split exactly, the motion taken off again
                const double cnt = __builtin_rint(box[0][j] * (1.0 / kCountUnit));
                v0 = (float)cnt;
                v1 = (float)__builtin_fma(cnt, -kunit, box[0][j]);
                v2 = (float)box[1][j];
            }
            if (v0 > 0.0f) {                   // my_lib_kernel.cu:1730-1735; one reciprocal for both components
                // FlowProjection's count is a whole number >= 1: v_rcp_f32 (1 ulp, one instruction; the IEEE division is
                // ten, and this kernel is bound by its vector instruction count) -- within 2 ulp of the two divisions.
                // The depth operator's count is a sum of arbitrary depths (it may be denormal): the full division.
                const float inv = DEPTH ? 1.0f / v0 : __builtin_amdgcn_rcpf(v0);
                v1 = v1 * inv;
                v2 = v2 * inv;
            }
            oc[j] = v0;  ox[j] = v1;  oy[j] = v2;
        }
    }
    trace_mark_proj<TRACE>(10);                // (box sums done)
    if (ws.up)                                 // pass 3 (hole filling) follows: fill what the tile can, summaries, masks
        owner_fill_epilogue<TH, NT, TRACE, PENDT>(fl, reinterpret_cast<float *>(P), ws, tid, b, tc.tx, tc.ty, W, H, tiles_x, tiles_y, inb,
                                    ox, oy, oc);
    if (inb) {                                 // single-use streams: nothing of this launch reads them back from cache
        if (RAG) {
            const int rs = tail_shift(cx, W);
            st_tail4<true>(out + b * s1b + (int64_t)cy * s1h + cx, ox, rs);
            st_tail4<true>(out + b * s1b + s1c + (int64_t)cy * s1h + cx, oy, rs);
            st_tail4<true>(count + b * scb + (int64_t)cy * sch + cx, oc, rs);
        } else {
            st_stream4(out + b * s1b + (int64_t)cy * s1h + cx, ox);
            st_stream4(out + b * s1b + s1c + (int64_t)cy * s1h + cx, oy);
            st_stream4(count + b * scb + (int64_t)cy * sch + cx, oc);
        }
    }
    trace_mark_proj<TRACE>(5);
}
