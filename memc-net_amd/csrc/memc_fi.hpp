// memc_fi.hpp -- site geometry shared by the FilterInterpolation kernels (filter_interpolation.hip, fi_bwd_cn.hip).
#pragma once

#include "memc_tile.hpp"

namespace memc {

struct FiSite4 {          // geometry of a lane's four sites
    int ix[4], iy[4];
    float a[4], b[4];
    unsigned valid;       // bit j
};

// sites of this lane whose (clamped) window lies inside the band
__device__ __forceinline__ unsigned fi_covered(const Region &r, const FiSite4 &g, int W, int H)
{
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (((g.valid >> j) & 1) &&
            r.covers(max(g.ix[j] - 1, 0), min(g.ix[j] + 2, W - 1), max(g.iy[j] - 1, 0), min(g.iy[j] + 2, H - 1)))
            m |= 1u << j;
    return m;
}

}  // namespace memc
