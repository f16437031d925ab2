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

// Block exponents from the tile's largest |a| and |b| (bit patterns, both finite and non-zero): 2^(ea + eb) > max|a| x
// max|b|, off by less than a factor two -- the mantissas' product tells whether the sum of the two exponents is one too
// many.  sa = 2^(11 - ea), sb = 2^(11 - eb) pre-scale the factors, inv = 2^(ea + eb - 22) (a double: the float may not
// exist) restores the sums.
struct PkScale {
    float sa, sb;
    double inv;
};
__device__ __forceinline__ PkScale pk_scale(int abits, int bbits)
{
    const int ea = pk_exponent(abits);
    int eb = pk_exponent(bbits);
    int e0, e1;
    const float mm = frexpf(__int_as_float(abits), &e0) * frexpf(__int_as_float(bbits), &e1);
    if (mm < 0.4999f && eb > -100) eb -= 1;
    PkScale s;
    s.sa = ldexpf(1.0f, 11 - ea);
    s.sb = ldexpf(1.0f, 11 - eb);
    s.inv = ldexp(1.0, ea + eb - 22);
    return s;
}

}  // namespace memc
