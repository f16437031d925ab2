"""A SECOND, independent restatement of the operator semantics -- pure Python loops written from SURVEY.md
appendix A (the formulas recorded from my_lib.c / my_lib_kernel.cu), NOT from oracle/memc_oracle.c.  Test-only:
it cross-checks the C oracle on small inputs (tests/test_oracle_vs_spec.py), so that a slip in one restatement
has to be repeated in the other, written differently, to go unnoticed.  fp32 arithmetic via numpy scalars, same
operation order as the appendix states."""
import math

import numpy as np

F = np.float32


def _clamp(i, hi):
    return min(max(0, i), hi)


def fi_forward(x, flow, filt):
    """A.1"""
    B, C, H, W = x.shape
    fs = int(math.sqrt(float(filt.shape[1])))
    out = np.zeros_like(x)
    for b in range(B):
        for y in range(H):
            for xx in range(W):
                fx, fy = flow[b, 0, y, xx], flow[b, 1, y, xx]
                x2, y2 = F(xx) + fx, F(y) + fy
                valid = (x2 >= 0 and y2 >= 0 and x2 <= W - 1 and y2 <= H - 1
                         and abs(fx) < F(W / 2.0) and abs(fy) < F(H / 2.0))
                if not valid:
                    out[b, :, y, xx] = x[b, :, y, xx]
                    continue
                ix, iy = int(x2), int(y2)
                L, T = ix + 1 - fs // 2, iy + 1 - fs // 2
                R, Bm = L + fs, T + fs
                a, be = x2 - F(ix), y2 - F(iy)
                for c in range(C):
                    quad = {}
                    for name, rows, cols in (("TL", range(T, iy + 1), range(L, ix + 1)),
                                             ("TR", range(T, iy + 1), range(ix + 1, R)),
                                             ("BL", range(iy + 1, Bm), range(L, ix + 1)),
                                             ("BR", range(iy + 1, Bm), range(ix + 1, R))):
                        s = F(0)
                        for j in rows:
                            for i in cols:
                                s = F(s + F(x[b, c, _clamp(j, H - 1), _clamp(i, W - 1)]
                                            * filt[b, (j - T) * fs + (i - L), y, xx]))
                        quad[name] = s
                    one = F(1)
                    out[b, c, y, xx] = F(F(F(F((one - a) * (one - be)) * quad["TL"]) + F(F(a * (one - be)) * quad["TR"]))
                                         + F(F((one - a) * be) * quad["BL"])) + F(F(a * be) * quad["BR"])
    return out


def flow_projection_forward(flow, depth=None):
    """A.3 / A.4, passes 1 and 2 (no hole filling)."""
    B, _, H, W = flow.shape
    out = np.zeros_like(flow)
    count = np.zeros((B, 1, H, W), np.float32)
    for b in range(B):
        for y in range(H):
            for xx in range(W):
                fx, fy = flow[b, 0, y, xx], flow[b, 1, y, xx]
                x2, y2 = F(xx) + fx, F(y) + fy
                if not (x2 >= 0 and y2 >= 0 and x2 <= W - 1 and y2 <= H - 1):
                    continue
                L, T = int(x2), int(y2)
                R, Bm = min(L + 1, W - 1), min(T + 1, H - 1)
                d = F(1) if depth is None else depth[b, 0, y, xx]
                for yy, xc in ((T, L), (T, R), (Bm, L), (Bm, R)):          # duplicates on clamped edges are kept
                    out[b, 0, yy, xc] = F(out[b, 0, yy, xc] + F(-d * fx if depth is not None else -fx))
                    out[b, 1, yy, xc] = F(out[b, 1, yy, xc] + F(-d * fy if depth is not None else -fy))
                    count[b, 0, yy, xc] = F(count[b, 0, yy, xc] + d)
    pos = count[:, 0] > 0
    for k in range(2):
        out[:, k][pos] = out[:, k][pos] / count[:, 0][pos]
    return out, count


def fill_holes(out, count):
    """A.3 pass 3, including the dead downward search."""
    B, _, H, W = out.shape
    res = out.copy()
    for b in range(B):
        for y in range(H):
            for x in range(W):
                if count[b, 0, y, x] > 0:
                    continue
                col, lt = x, F(0)
                while lt == 0 and col - 1 >= 0:
                    col -= 1
                    lt = count[b, 0, y, col]
                lcol = col
                col, rt = x, F(0)
                while rt == 0 and col + 1 <= W - 1:
                    col += 1
                    rt = count[b, 0, y, col]
                rcol = col
                row, ut = y, F(0)
                while ut == 0 and row - 1 >= 0:
                    row -= 1
                    ut = count[b, 0, row, x]
                urow = row
                if lt + rt + ut <= 0:
                    continue
                fl, fr, fu = F(lt > 0), F(rt > 0), F(ut > 0)
                for k in range(2):
                    num = F(F(F(fl * out[b, k, y, lcol]) + F(fr * out[b, k, y, rcol])) + F(fu * out[b, k, urow, x]))
                    num = F(num + F(F(0) * out[b, k, y, x]))
                    res[b, k, y, x] = num / F(F(F(fl + fr) + fu) + F(0))
    return res


def interpolation_forward(x, flow):
    """A.5"""
    B, C, H, W = x.shape
    out = np.zeros_like(x)
    for b in range(B):
        for y in range(H):
            for xx in range(W):
                x2, y2 = F(xx) + flow[b, 0, y, xx], F(y) + flow[b, 1, y, xx]
                if not (x2 >= 0 and y2 >= 0 and x2 < W and y2 < H):
                    continue
                L, T = int(x2), int(y2)
                R, Bm = min(L + 1, W - 1), min(T + 1, H - 1)
                a, be = x2 - F(L), y2 - F(T)
                one = F(1)
                for c in range(C):
                    out[b, c, y, xx] = F(F(F(F((one - a) * (one - be)) * x[b, c, T, L]) + F(F(a * (one - be)) * x[b, c, T, R]))
                                         + F(F((one - a) * be) * x[b, c, Bm, L])) + F(F(a * be) * x[b, c, Bm, R])
    return out
