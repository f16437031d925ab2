// memc_pk.hpp -- packed fixed-point LDS accumulation of three colour planes (the RGB backward passes).
#pragma once

#include "memc_tile.hpp"

namespace memc {

// ---------------------------------------------------------------------------------------------------------
// The image gradient of an RGB backward pass in ONE LDS round instead of three.
//
// What the round-1/2 kernel (fi_bwd_tiled_c3, now a measurement arm) spends after its HBM-bound phase 1 is LDS
// atomic work: 48 ds_add_f64 per site (16 taps x 3 colours) into ONE fp64 plane that the colours take in turn --
// adds, barrier, flush, barrier, three times.  ds_add_f32 is an emulated path on this chip (0.33 lane-ops/clk/CU),
// 64-bit integer adds are the fastest LDS atomic there is (9.0; ds_add_f64: 6.7), so the three colours of a
// contribution are accumulated as 23-bit fixed-point numbers packed into TWO 64-bit words per cell:
//
//     word A = n0 * 2^26 + (n2 >> 11)            word B = n1 * 2^26 + (n2 & 2047)
//
// with n_c = round(g_c * wq * tap * 2^(22 - e)), |n_c| <= 2^22, where 2^e bounds every contribution of the TILE:
// 2^e > (the tile's largest |gradoutput|) x (its largest |tap|), by less than a factor two (the bilinear weights
// are <= 1).  Integer adds are exact and associative: a cell's sums come out bit-identical whatever order the LDS
// retires them in (the reference's fp32 atomics do not), scaling the inputs by a power of two scales the result by
// exactly that power, and nothing can overflow -- one site puts at most 9 of its 16 taps into one cell (the image
// corner, where the clamp folds 3 x 3 window positions), so a tile adds at most 1024 x 9 < 2^14 numbers to a cell:
// |sum n0| < 2^36 (word A holds 38 signed bits above bit 26), |sum (n2 >> 11)| < 2^25 (26-bit signed field),
// sum (n2 & 2047) < 2^25 (26-bit unsigned field).  Every contribution is rounded once, to a multiple of 2^(e - 22)
// <= 2^-21 x (the tile's largest possible contribution) -- about fp32's own resolution of that largest contribution;
// a cell's error is at most (its number of contributions) x 2^(e - 23).  32 adds per site instead of 48, both planes
// flushed in one pass: one round of adds / barrier / flush.
// A tile whose gradoutput or taps are not all finite takes per-site global atomics instead (NaN / Inf then land
// exactly where the reference puts them); a tile whose bound is zero has nothing to add.
// ---------------------------------------------------------------------------------------------------------
struct PkAcc {
    static constexpr int kMagic = 0x4B400000;                                  // bits of 1.5 * 2^23
    static constexpr int kShift = 26, kSplit = 11;
};
// A plane has the band's geometry: r.h rows of r.pitch 64-bit slots (pitch 96, 80 or 64: a multiple of 16 slots, so
// every row starts on the same bank).  Column c of a row is stored at (c & 3) * (pitch / 4) + (c >> 2): cells 4 apart
// (the sites of neighbouring lanes) are adjacent 8-byte slots -- see AccT in memc_tile.hpp.
__device__ __forceinline__ int pk_col(int c, int quarter) { return (c & 3) * quarter + (c >> 2); }

__device__ __forceinline__ void lds_add_u64(unsigned long long *p, unsigned long long v)
{
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_u64
}

// exponent e with |v| < 2^e for the non-negative float whose bits are `bits` (finite, non-zero), kept inside
// [-100, 128] so that every power of two formed from it is a normal float
__device__ __forceinline__ int pk_exponent(int bits)
{
    int e;
    (void)frexpf(__int_as_float(bits), &e);
    return max(e, -100);
}

// One contribution: adds round(g_c * w) for the three colours to cell `slot` of the two planes.  g_c and w are the
// pre-scaled factors (|g_c| < 2^11, |w| <= 2^11: |product| <= 2^22).  fma(g, w, 1.5 * 2^23) leaves the rounded product
// in the mantissa: one instruction per colour for multiply + round-to-nearest-even + convert.
__device__ __forceinline__ void pk_add3(unsigned long long *accA, unsigned long long *accB, int slot, float g0, float g1,
                                        float g2, float w)
{
    const float magic = __int_as_float(PkAcc::kMagic);
    const int n0 = __float_as_int(fmaf(g0, w, magic)) - PkAcc::kMagic;
    const int n1 = __float_as_int(fmaf(g1, w, magic)) - PkAcc::kMagic;
    const int n2 = __float_as_int(fmaf(g2, w, magic)) - PkAcc::kMagic;
    const long long A = ((long long)n0 << PkAcc::kShift) + (long long)(n2 >> PkAcc::kSplit);
    const long long B = ((long long)n1 << PkAcc::kShift) | (long long)(n2 & ((1 << PkAcc::kSplit) - 1));
    lds_add_u64(accA + slot, (unsigned long long)A);
    lds_add_u64(accB + slot, (unsigned long long)B);
}

// Unpacks every cell of the band and adds its three colours to gradinput1 (row-coalesced global atomics: the boxes of
// neighbouring tiles overlap).  The 256 lanes walk the box's cells in row-major order, so a wave's 64 cells are one
// run of a row (256 contiguous bytes per colour) and, in the planes, 4 x 8 consecutive slots: conflict-free.
// inv = 2^(e_g + e_t - 22) as a double (the float may not exist).
template <int NT = 256>
__device__ __forceinline__ void pk_flush(const Region &r, const unsigned long long *accA,
                                                const unsigned long long *accB, double inv, float *gin1_b,
                                                int64_t s1c, int s1h)
{
    const unsigned tid = tid_now();
    const int w = max(r.w, 1), total = r.w * r.h;
    int row = tid / w, col = tid % w;                      // one run-time division per band
    const int drow = NT / w, dcol = NT % w;
    const uintptr_t b0 = pin_sgpr(gin1_b), b1 = pin_sgpr(gin1_b + s1c), b2 = pin_sgpr(gin1_b + 2 * s1c);
    constexpr int kBatch = 4;
#pragma unroll 1
    for (int base = 0; base < total; base += NT * kBatch) {
        long long A[kBatch], B[kBatch];
        unsigned off[kBatch];
        bool on[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            on[u] = row < r.h;
            const int slot = (on[u] ? row : 0) * r.pitch + pk_col(col, r.pitch >> 2);
            A[u] = (long long)accA[slot];
            B[u] = (long long)accB[slot];
            off[u] = 4u * (unsigned)((r.y0 + row) * s1h + r.x0 + col);
            col += dcol;
            row += drow + (col >= w ? 1 : 0);
            col -= col >= w ? w : 0;
        }
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            if (!on[u] || (A[u] | B[u]) == 0) continue;
            const long long c0 = (A[u] + (1LL << (PkAcc::kShift - 1))) >> PkAcc::kShift;
            const long long sa = A[u] - (c0 << PkAcc::kShift);
            const long long c1 = B[u] >> PkAcc::kShift;
            const long long lb = B[u] & ((1LL << PkAcc::kShift) - 1);
            const long long c2 = (sa << PkAcc::kSplit) + lb;
            const float v0 = (float)((double)c0 * inv), v1 = (float)((double)c1 * inv), v2 = (float)((double)c2 * inv);
            if (v0 != 0.0f) (void)__builtin_amdgcn_global_atomic_fadd_f32(reinterpret_cast<MEMC_GLOBAL float *>(b0 + off[u]), v0);
            if (v1 != 0.0f) (void)__builtin_amdgcn_global_atomic_fadd_f32(reinterpret_cast<MEMC_GLOBAL float *>(b1 + off[u]), v1);
            if (v2 != 0.0f) (void)__builtin_amdgcn_global_atomic_fadd_f32(reinterpret_cast<MEMC_GLOBAL float *>(b2 + off[u]), v2);
        }
    }
}

// largest value of a non-negative int over the wave (float bit patterns of |x| order like ints; NaN sorts last)
__device__ __forceinline__ int wave_max_i32(int v) { return -wave_min_i32(-v); }

// sum over the 64 lanes of a wave, returned in every lane
__device__ __forceinline__ float wave_sum_f32(float v)
{
    // row_shr:1,2,4,8 inside each row of 16 lanes (bound_ctrl: lanes without a source add 0)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));
    const int a = __builtin_amdgcn_readlane(__float_as_int(v), 15), b = __builtin_amdgcn_readlane(__float_as_int(v), 31);
    const int c = __builtin_amdgcn_readlane(__float_as_int(v), 47), d = __builtin_amdgcn_readlane(__float_as_int(v), 63);
    return (__int_as_float(a) + __int_as_float(b)) + (__int_as_float(c) + __int_as_float(d));
}

// ---------------------------------------------------------------------------------------------------------
// The tile's block exponent, round 4: from PER-SITE bounds, with outlier sites set aside.
//
// Rounds 3's exponent came from (the tile's largest |gradoutput|) x (its largest |tap|): one site with a gradient or a tap
// of 1e4 coarsened the fixed-point grid of the whole tile, and cells whose own gradient is O(0.01) lost their 1e-4 -- a
// coupling between sites that the reference's fp32 atomics do not have (my_lib_kernel.cu:1276-1288).  Now every site
// has its own bound s = (its largest |gradoutput|) x (its largest |tap|) >= each of its contributions, and the tile takes
//     B = min(max s, kPkOutlier x mean s)          over its valid sites with a finite, non-zero bound (a ROBUST mean:
//                                                  the median of the waves' means, each without its wave's largest s)
// as the bound of the PACKED sites: 2^e > B by less than a factor two, contributions are rounded once to a multiple of
// 2^(e - 22), a cell's error is at most (its contributions) x 2^(e - 23) <= n x 1.9e-6 x (the tile's MEAN site bound).
// Sites beyond B -- and sites with an Inf / NaN input -- add their image gradient with per-site global atomics, exactly as
// the reference does.  (By Markov fewer than 1 / 16 of a tile's sites can exceed 16 x the mean.)
// ---------------------------------------------------------------------------------------------------------
constexpr float kPkOutlier = 16.0f;
struct PkTile {
    float sa, sb;      // pre-scales of gradoutput and of the weights: |g sa| |w sb| <= 2^22 for every packed site
    double inv;        // 2^(e - 22): restores the sums
    float limit;       // B: a site is packed iff its bound is <= limit (never true for NaN; -1: no site is)
    int any;           // some packed site has something to add
};

// Before the workgroup's next barrier: the wave's share of (max s, max |g|, sum s, count) -> mx[4 wave .. 4 wave + 3].
// sbits[j] / gbits[j]: bit patterns of site j's bound and of its largest |gradoutput| (non-negative floats);
// valid: the lane's sites that take part at all; tbits: bit pattern of the lane's largest |weight factor| (the taps; 1.0
// where the weights are the bilinear ones alone) -- only its exponent travels, for the overflow guard of pk_tile_resolve.
__device__ __forceinline__ void pk_tile_publish(int *mx, unsigned tid, const int (&sbits)[4], const int (&gbits)[4], unsigned valid,
                                                int tbits)
{
    int M = 0, G = 0, cnt = 0;
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool use = ((valid >> j) & 1u) && sbits[j] != 0 && sbits[j] < 0x7F800000;
        M = max(M, use ? sbits[j] : 0);
        G = max(G, use ? gbits[j] : 0);
        sum += use ? __int_as_float(sbits[j]) : 0.0f;
        cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(use));
    }
    M = wave_max_i32(M);
    G = wave_max_i32(G);
    sum = wave_sum_f32(sum);
    const int T = wave_max_i32(tbits < 0x7F800000 ? tbits : 0);
    if ((tid & (kWave - 1)) == 0) {
        int *m = mx + 4 * (tid / kWave);
        m[0] = M;  m[1] = G;  m[2] = __float_as_int(sum);  m[3] = cnt | (T & ~0xFFFF);   // (cnt <= 256 per wave)
    }
}

// After that barrier (NW waves published).  Workgroup-uniform.
// The "mean" is robust against the very outliers it is there to find: every wave's mean leaves out the wave's largest
// bound, and the tile takes the (lower) MEDIAN of its waves' means -- one site of 1e4 among a thousand of 0.1 would lift
// the plain mean a hundredfold, and with it the grid of everything packed.
template <int NW>
__device__ __forceinline__ PkTile pk_tile_resolve(const int *mx)
{
    int M = 0, G = 0, T = 0, nmean = 0;
    int mean_bits[NW];                                     // bit patterns of the waves' trimmed means (non-negative floats)
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const int Mw = __builtin_amdgcn_readfirstlane(mx[4 * w]);
        const int cw = __builtin_amdgcn_readfirstlane(mx[4 * w + 3]);
        const float sw = __int_as_float(__builtin_amdgcn_readfirstlane(mx[4 * w + 2]));
        M = max(M, Mw);
        G = max(G, __builtin_amdgcn_readfirstlane(mx[4 * w + 1]));
        T = max(T, cw & ~0xFFFF);
        const int cnt = cw & 0xFFFF;
        // (sum - max can come out a rounding error below zero when the wave's largest bound IS the sum)
        const float mean = cnt > 1 ? fmaxf(sw - __int_as_float(Mw), 0.0f) / (float)(cnt - 1) : __int_as_float(Mw);
        mean_bits[w] = cnt > 0 ? __builtin_amdgcn_readfirstlane(__float_as_int(mean)) : -1;
        nmean += cnt > 0 ? 1 : 0;
    }
    PkTile t;
    t.sa = t.sb = 1.0f;  t.inv = 1.0;  t.limit = -1.0f;  t.any = 0;
    if (nmean == 0) return t;                              // nothing finite and non-zero: whatever is left is an outlier
    // lower median of the nmean valid means: the one with exactly (nmean - 1) / 2 valid means before it (ties by index)
    int med = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        int before = 0;
#pragma unroll
        for (int k = 0; k < NW; k++)
            before += (mean_bits[k] >= 0 && (mean_bits[k] < mean_bits[i] || (mean_bits[k] == mean_bits[i] && k < i))) ? 1 : 0;
        if (mean_bits[i] >= 0 && before == (nmean - 1) / 2) med = mean_bits[i];
    }
    const float B = fminf(__int_as_float(M), kPkOutlier * fmaxf(__int_as_float(med), 1.0e-37f));
    if (!(B > 0.0f) || !(B < 3.0e38f)) return t;
    int e, eg, et;
    (void)frexpf(B, &e);                                   // B < 2^e
    (void)frexpf(__int_as_float(G), &eg);
    (void)frexpf(__int_as_float(T), &et);
    // The factors are scaled separately -- |g sa| < 2^11 by construction, |w sb| <= 2^(11 + eg + et - e) -- and every one
    // of sa, sb and w sb must stay a normal float: magnitudes beyond that (a tile whose largest |gradoutput| x largest
    // |tap| exceeds its packed bound by 2^100, gradients below 2^-100) take per-site atomics for everything.
    if (eg < -100 || 11 - e + eg < -100 || 11 - e + eg > 100 || eg + et - e > 100) return t;
    t.sa = ldexpf(1.0f, 11 - eg);
    t.sb = ldexpf(1.0f, 11 - e + eg);
    t.inv = ldexp(1.0, e - 22);
    t.limit = B;
    t.any = 1;
    return t;
}

// the lane's sites that are packed / that take per-site atomics instead
__device__ __forceinline__ unsigned pk_packed_sites(const PkTile &t, const int (&sbits)[4], unsigned valid)
{
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (((valid >> j) & 1u) && sbits[j] != 0 && __int_as_float(sbits[j]) <= t.limit) m |= 1u << j;
    return m;
}
__device__ __forceinline__ unsigned pk_outlier_sites(const PkTile &t, const int (&sbits)[4], unsigned valid)
{
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (((valid >> j) & 1u) && sbits[j] != 0 && !(__int_as_float(sbits[j]) <= t.limit)) m |= 1u << j;
    return m;
}

}  // namespace memc
