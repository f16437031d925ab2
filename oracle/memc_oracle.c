/*
 * memc_oracle.c -- CPU restatement of MEMC-Net's adaptive-warp / flow-projection operators.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The shipped path is the HIP
 * library under memc-net_amd/csrc and it has no CPU fallback.
 *
 * PARITY PINNED against outputs of the reference itself (see DESIGN.md "Oracle").  The reference holds no
 * tests, golden vectors or fixtures for this path, its Python wrappers cannot be imported
 * (torch.utils.ffi is gone) and its CPU file needs TH.h from PyTorch 0.2's TH library, which this image lacks --
 * but its GPU file, my_package/src/my_lib_kernel.cu, is self-contained and builds for gfx950 with the image's own
 * hipify-perl + hipcc (`make -C oracle ref` -> oracle/_ref/libmemc_ref_gpu.so).  Those kernels, run on an MI355X:
 *   - tests/golden/ref_gpu_*.npz hold their outputs for every entry point of the path (forward, backward,
 *     with and without hole filling); tests/test_golden_reference.py holds this file to them, on any machine;
 *   - tests/test_gpu_reference.py repeats the comparison live on the GPU box, and compares the HIP path with
 *     the reference kernels directly.
 * Also: the known answers SURVEY.md appendix A.7 recorded from the reference's C code (tests/test_oracle_kat.py)
 * and an independent pure-Python restatement (tests/test_oracle_vs_spec.py).
 *
 * Every function restates one reference function; citations are file:line under /root/reference.
 * Float arithmetic follows the reference's C expression order exactly (build with -ffp-contract=off:
 * the reference's gcc/x86-64 build has no fused multiply-add).
 *
 * Calling convention: plain pointers, sizes and ELEMENT strides {b, c, h, w}; every output / gradient
 * buffer is caller-allocated and caller-zeroed, exactly as the reference's Python layer does
 * (my_package/functions/FilterInterpolationLayer.py:26-29,46-48, FlowProjectionLayer.py:27-29,54).
 * Return value: 0 ok, -1 on a failed shape/stride check (my_lib.c:911-954 and siblings).
 *
 * Batch items are independent in every operator, so the outer batch loop may run under OpenMP (the two
 * forward gathers, whose output sites are all independent, also split over rows); the order of
 * accumulation inside one batch item is the reference's sequential order, so results do not depend on
 * the thread count.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef int64_t i64;

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int clampi(int v, int hi) { return imin(imax(0, v), hi); }

int memc_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * Plain bilinear warp.  my_lib.c:440-531 (Interpolation, channel must be 3) and :668-757
 * (InterpolationCh, any channel count); the two bodies are identical apart from that check.
 * ---------------------------------------------------------------------------------------------- */
static int bilinear_forward(int require_c3, int B, int C, int H, int W,
                            const float *in1, const i64 *s1, const float *flow, const i64 *s2,
                            float *out, const i64 *so)
{
    if (require_c3 && C != 3) return -1;                       /* my_lib.c:450 */
    if (s1[3] != 1 || s2[3] != 1) return -1;                    /* my_lib.c:474-475 */
    if (s1[0] != so[0] || s1[1] != so[1]) return -1;            /* my_lib.c:476-477 */
    /* every output site is independent in the forward gather: rows may run in parallel */
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++) {
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const i64 off = b * s1[0];
                float fx = flow[b * s2[0] + 0 * s2[1] + y * s2[2] + x];
                float fy = flow[b * s2[0] + 1 * s2[1] + y * s2[2] + x];
                float x2 = (float)x + fx, y2 = (float)y + fy;
                /* strict upper bound against w/h, my_lib.c:495 */
                if (x2 >= 0.0f && y2 >= 0.0f && x2 < (float)W && y2 < (float)H) {
                    int L = (int)x2, T = (int)y2;
                    int R = imin(L + 1, W - 1), Bm = imin(T + 1, H - 1);
                    float alpha = x2 - L, beta = y2 - T;
                    for (int c = 0; c < C; c++) {
                        const float *p = in1 + off + c * s1[1];
                        float TL = p[T * s1[2] + L], TR = p[T * s1[2] + R];
                        float BL = p[Bm * s1[2] + L], BR = p[Bm * s1[2] + R];
                        out[off + c * s1[1] + y * s1[2] + x] =              /* my_lib.c:510-514 */
                            (1 - alpha) * (1 - beta) * TL + alpha * (1 - beta) * TR +
                            (1 - alpha) * beta * BL + alpha * beta * BR;
                    }
                } else {
                    for (int c = 0; c < C; c++) out[off + c * s1[1] + y * s1[2] + x] = 0.0f;  /* :518-522 */
                }
            }
    }
    return 0;
}

/* my_lib.c:534-666 and :760-892 */
static int bilinear_backward(int require_c3, int B, int C, int H, int W,
                             const float *in1, const i64 *s1, const float *flow, const i64 *s2,
                             const float *gout, float *gin1, float *gin2)
{
    if (require_c3 && C != 3) return -1;
    if (s1[3] != 1 || s2[3] != 1) return -1;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; b++) {
        const i64 off = b * s1[0];
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                float fx = flow[b * s2[0] + 0 * s2[1] + y * s2[2] + x];
                float fy = flow[b * s2[0] + 1 * s2[1] + y * s2[2] + x];
                float x2 = (float)x + fx, y2 = (float)y + fy;
                if (!(x2 >= 0.0f && y2 >= 0.0f && x2 < (float)W && y2 < (float)H)) continue;
                int L = (int)x2, T = (int)y2;
                int R = imin(L + 1, W - 1), Bm = imin(T + 1, H - 1);
                float alpha = x2 - L, beta = y2 - T;
                for (int c = 0; c < C; c++) {                               /* my_lib.c:611-619 */
                    float g = gout[off + c * s1[1] + y * s1[2] + x];
                    float *q = gin1 + off + c * s1[1];
                    q[T * s1[2] + L] += g * (1 - alpha) * (1 - beta);
                    q[T * s1[2] + R] += g * alpha * (1 - beta);
                    q[Bm * s1[2] + L] += g * (1 - alpha) * beta;
                    q[Bm * s1[2] + R] += g * alpha * beta;
                }
                float gamma = Bm - y2;                                      /* clamped corner, :622 */
                float bot = 0;
                for (int c = 0; c < C; c++) {
                    const float *p = in1 + off + c * s1[1];
                    float t = 0.0f;
                    t += gamma * (p[T * s1[2] + R] - p[T * s1[2] + L]);
                    t += (1 - gamma) * (p[Bm * s1[2] + R] - p[Bm * s1[2] + L]);
                    bot += gout[off + c * s1[1] + y * s1[2] + x] * t;
                }
                gin2[b * s2[0] + 0 * s2[1] + y * s2[2] + x] = bot;          /* assignment, :637 */
                gamma = R - x2;                                             /* :640 */
                bot = 0;
                for (int c = 0; c < C; c++) {
                    const float *p = in1 + off + c * s1[1];
                    float t = 0.0f;
                    t += gamma * (p[Bm * s1[2] + L] - p[T * s1[2] + L]);
                    t += (1 - gamma) * (p[Bm * s1[2] + R] - p[T * s1[2] + R]);
                    bot += gout[off + c * s1[1] + y * s1[2] + x] * t;
                }
                gin2[b * s2[0] + 1 * s2[1] + y * s2[2] + x] = bot;          /* :655 */
            }
    }
    return 0;
}

int memc_oracle_interpolation_forward(int B, int C, int H, int W, const float *in1, const i64 *s1,
                                      const float *flow, const i64 *s2, float *out, const i64 *so)
{ return bilinear_forward(1, B, C, H, W, in1, s1, flow, s2, out, so); }

int memc_oracle_interpolation_backward(int B, int C, int H, int W, const float *in1, const i64 *s1,
                                       const float *flow, const i64 *s2, const float *gout,
                                       float *gin1, float *gin2)
{ return bilinear_backward(1, B, C, H, W, in1, s1, flow, s2, gout, gin1, gin2); }

int memc_oracle_interpolation_ch_forward(int B, int C, int H, int W, const float *in1, const i64 *s1,
                                         const float *flow, const i64 *s2, float *out, const i64 *so)
{ return bilinear_forward(0, B, C, H, W, in1, s1, flow, s2, out, so); }

int memc_oracle_interpolation_ch_backward(int B, int C, int H, int W, const float *in1, const i64 *s1,
                                          const float *flow, const i64 *s2, const float *gout,
                                          float *gin1, float *gin2)
{ return bilinear_backward(0, B, C, H, W, in1, s1, flow, s2, gout, gin1, gin2); }

/* ------------------------------------------------------------------------------------------------
 * FilterInterpolation: flow-displaced fs x fs window, per-output-pixel taps, four quadrant sums blended
 * bilinearly.  Forward my_lib.c:904-1079, backward :1082-1444.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int L, T, R, Bm;    /* window [L,R) x [T,Bm), unclamped */
    int ix, iy;         /* (int)x2, (int)y2 */
    float alpha, beta;
} fi_site;

/* validity test and window of one output site; my_lib.c:977-987 */
static inline int fi_locate(int x, int y, int W, int H, int fs, float fx, float fy, fi_site *s)
{
    float x2 = (float)x + fx, y2 = (float)y + fy;
    if (!(x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(W - 1) && y2 <= (float)(H - 1) &&
          fabs(fx) < (float)(W) / 2.0f && fabs(fy) < (float)(H) / 2.0f))
        return 0;
    s->ix = (int)x2;
    s->iy = (int)y2;
    s->L = s->ix + 1 - (int)(fs / 2);
    s->T = s->iy + 1 - (int)(fs / 2);
    s->R = s->L + fs;
    s->Bm = s->T + fs;
    s->alpha = x2 - (int)x2;
    s->beta = y2 - (int)y2;
    return 1;
}

/* one quadrant sum, rows [j0,j1] x cols [i0,i1] inclusive, j outer / i inner, accumulate from 0;
 * my_lib.c:994-1032.  img = one channel plane of input1, tap = filter taps at this output site
 * (tap stride = input3 channel stride). */
static inline float fi_quad(const float *img, i64 hs, int W, int H, const float *tap, i64 tap_cs,
                            int fs, const fi_site *s, int j0, int j1, int i0, int i1)
{
    float acc = 0.0f;
    for (int j = j0; j <= j1; j++) {
        int jj = clampi(j, H - 1);
        for (int i = i0; i <= i1; i++) {
            int ii = clampi(i, W - 1);
            acc += img[jj * hs + ii] * tap[((j - s->T) * fs + (i - s->L)) * tap_cs];
        }
    }
    return acc;
}

int memc_oracle_filter_interpolation_forward(int B, int C, int H, int W, int fs2,
                                             const float *in1, const i64 *s1,
                                             const float *flow, const i64 *s2,
                                             const float *filt, const i64 *s3,
                                             float *out, const i64 *so)
{
    const int fs = (int)sqrt((float)fs2);                         /* my_lib.c:925 */
    if (s1[3] != 1 || s2[3] != 1 || s3[3] != 1) return -1;          /* :950-952 */
    if (s1[0] != so[0] || s1[1] != so[1]) return -1;                /* :953-954 */
    /* every output site is independent in the forward gather: rows may run in parallel */
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++) {
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const i64 off = b * s1[0];
                float fx = flow[b * s2[0] + 0 * s2[1] + y * s2[2] + x];
                float fy = flow[b * s2[0] + 1 * s2[1] + y * s2[2] + x];
                fi_site s;
                if (fi_locate(x, y, W, H, fs, fx, fy, &s)) {
                    const float *tap = filt + b * s3[0] + y * s3[2] + x;
                    for (int c = 0; c < C; c++) {
                        const float *img = in1 + off + c * s1[1];
                        float TL = fi_quad(img, s1[2], W, H, tap, s3[1], fs, &s, s.T, s.iy, s.L, s.ix);
                        float TR = fi_quad(img, s1[2], W, H, tap, s3[1], fs, &s, s.T, s.iy, s.ix + 1, s.R - 1);
                        float BL = fi_quad(img, s1[2], W, H, tap, s3[1], fs, &s, s.iy + 1, s.Bm - 1, s.L, s.ix);
                        float BR = fi_quad(img, s1[2], W, H, tap, s3[1], fs, &s, s.iy + 1, s.Bm - 1, s.ix + 1, s.R - 1);
                        float alpha = s.alpha, beta = s.beta;
                        out[off + c * s1[1] + y * s1[2] + x] =              /* :1034-1038 */
                            (1 - alpha) * (1 - beta) * TL + alpha * (1 - beta) * TR +
                            (1 - alpha) * beta * BL + alpha * beta * BR;
                    }
                } else {
                    /* out-of-range site copies the input pixel (not zero), :1065-1069 */
                    for (int c = 0; c < C; c++)
                        out[off + c * s1[1] + y * s1[2] + x] = in1[off + c * s1[1] + y * s1[2] + x];
                }
            }
    }
    return 0;
}

/* scatter one quadrant's image and tap gradients; my_lib.c:1193-1252 */
static inline void fi_quad_grad(const float *img, float *gimg, i64 hs, int W, int H,
                                const float *tap, float *gtap, i64 tap_cs, int fs, const fi_site *s,
                                float wgrad, int j0, int j1, int i0, int i1)
{
    for (int j = j0; j <= j1; j++) {
        int jj = clampi(j, H - 1);
        for (int i = i0; i <= i1; i++) {
            int ii = clampi(i, W - 1);
            i64 t = ((j - s->T) * fs + (i - s->L)) * tap_cs;
            gimg[jj * hs + ii] += wgrad * tap[t];
            gtap[t] += wgrad * img[jj * hs + ii];
        }
    }
}

int memc_oracle_filter_interpolation_backward(int B, int C, int H, int W, int fs2,
                                              const float *in1, const i64 *s1,
                                              const float *flow, const i64 *s2,
                                              const float *filt, const i64 *s3,
                                              const float *gout,
                                              float *gin1, float *gin2, float *gin3)
{
    const int fs = (int)sqrt((float)fs2);                         /* my_lib.c:1106 */
    if (s1[3] != 1 || s2[3] != 1 || s3[3] != 1) return -1;          /* :1136-1138 */
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; b++) {
        const i64 off = b * s1[0];
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                float fx = flow[b * s2[0] + 0 * s2[1] + y * s2[2] + x];
                float fy = flow[b * s2[0] + 1 * s2[1] + y * s2[2] + x];
                fi_site s;
                if (!fi_locate(x, y, W, H, fs, fx, fy, &s)) continue;   /* no gradient at all, :1166 */
                const float alpha = s.alpha, beta = s.beta;
                const float *tap = filt + b * s3[0] + y * s3[2] + x;
                float *gtap = gin3 + b * s3[0] + y * s3[2] + x;
                /* steps 1 and 3: image and tap gradients, :1189-1267 */
                for (int c = 0; c < C; c++) {
                    const float *img = in1 + off + c * s1[1];
                    float *gimg = gin1 + off + c * s1[1];
                    float g = gout[off + c * s1[1] + y * s1[2] + x];
                    float TLg = g * (1 - alpha) * (1 - beta);
                    fi_quad_grad(img, gimg, s1[2], W, H, tap, gtap, s3[1], fs, &s, TLg, s.T, s.iy, s.L, s.ix);
                    float TRg = g * alpha * (1 - beta);
                    fi_quad_grad(img, gimg, s1[2], W, H, tap, gtap, s3[1], fs, &s, TRg, s.T, s.iy, s.ix + 1, s.R - 1);
                    float BLg = g * (1 - alpha) * beta;
                    fi_quad_grad(img, gimg, s1[2], W, H, tap, gtap, s3[1], fs, &s, BLg, s.iy + 1, s.Bm - 1, s.L, s.ix);
                    float BRg = g * alpha * beta;
                    fi_quad_grad(img, gimg, s1[2], W, H, tap, gtap, s3[1], fs, &s, BRg, s.iy + 1, s.Bm - 1, s.ix + 1, s.R - 1);
                }
                /* step 2: flow gradients, :1273-1414 */
                float gamma = 1.0f - beta;
                float botx = 0.0f;
                for (int c = 0; c < C; c++) {
                    const float *img = in1 + off + c * s1[1];
                    float g = gout[off + c * s1[1] + y * s1[2] + x];
                    float TL = fi_quad(img, s1[2], W, H, tap, s3[1], fs, &s, s.T, s.iy, s.L, s.ix);
                    float TR = fi_quad(img, s1[2], W, H, tap, s3[1], fs, &s, s.T, s.iy, s.ix + 1, s.R - 1);
                    float BL = fi_quad(img, s1[2], W, H, tap, s3[1], fs, &s, s.iy + 1, s.Bm - 1, s.L, s.ix);
                    float BR = fi_quad(img, s1[2], W, H, tap, s3[1], fs, &s, s.iy + 1, s.Bm - 1, s.ix + 1, s.R - 1);
                    float t = 0.0f;
                    t += gamma * (TR - TL);
                    t += (1.0f - gamma) * (BR - BL);
                    botx += g * t;
                }
                gin2[b * s2[0] + 0 * s2[1] + y * s2[2] + x] = botx;          /* assignment, :1341 */
                gamma = 1.0f - alpha;
                float boty = 0.0f;
                for (int c = 0; c < C; c++) {
                    const float *img = in1 + off + c * s1[1];
                    float g = gout[off + c * s1[1] + y * s1[2] + x];
                    float TL = fi_quad(img, s1[2], W, H, tap, s3[1], fs, &s, s.T, s.iy, s.L, s.ix);
                    float TR = fi_quad(img, s1[2], W, H, tap, s3[1], fs, &s, s.T, s.iy, s.ix + 1, s.R - 1);
                    float BL = fi_quad(img, s1[2], W, H, tap, s3[1], fs, &s, s.iy + 1, s.Bm - 1, s.L, s.ix);
                    float BR = fi_quad(img, s1[2], W, H, tap, s3[1], fs, &s, s.iy + 1, s.Bm - 1, s.ix + 1, s.R - 1);
                    float t = 0.0f;
                    t += gamma * (BL - TL);
                    t += (1.0f - gamma) * (BR - TR);
                    boty += g * t;
                }
                gin2[b * s2[0] + 1 * s2[1] + y * s2[2] + x] = boty;          /* :1414 */
            }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * FlowProjection / DepthFlowProjection.
 * Scatter + average: my_lib.c:1447-1547 (flow only), :1637-1749 (depth weighted).
 * Hole filling exists only in the CUDA file: my_lib_kernel.cu:1742-1836 (identical copy :2169-2264);
 * the CPU functions print "Not implemented" instead (my_lib.c:1539-1543).  It is restated here from the
 * .cu so the GPU-semantics result (what the networks actually get at inference,
 * FlowProjectionLayer.py:15) has a checker.
 * ---------------------------------------------------------------------------------------------- */
static void project_scatter(int b, int H, int W, const float *flow, const i64 *s1,
                            const float *depth, const i64 *sd,
                            float *count, const i64 *sc, float *out)
{
    const i64 off = b * s1[0];
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float fx = flow[off + 0 * s1[1] + y * s1[2] + x];
            float fy = flow[off + 1 * s1[1] + y * s1[2] + x];
            float x2 = (float)x + fx, y2 = (float)y + fy;
            /* no |f| < w/2 guard in this operator, my_lib.c:1501 */
            if (!(x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(W - 1) && y2 <= (float)(H - 1))) continue;
            int L = (int)x2, T = (int)y2;
            int R = imin(L + 1, W - 1), Bm = imin(T + 1, H - 1);
            float *ox = out + off + 0 * s1[1], *oy = out + off + 1 * s1[1];
            float *cn = count + b * sc[0];
            if (depth) {                                             /* my_lib.c:1706-1722 */
                float d = depth[b * sd[0] + y * sd[2] + x];
                ox[T * s1[2] + L] += -d * fx;  ox[T * s1[2] + R] += -d * fx;
                ox[Bm * s1[2] + L] += -d * fx; ox[Bm * s1[2] + R] += -d * fx;
                oy[T * s1[2] + L] += -d * fy;  oy[T * s1[2] + R] += -d * fy;
                oy[Bm * s1[2] + L] += -d * fy; oy[Bm * s1[2] + R] += -d * fy;
                cn[T * sc[2] + L] += d * 1.0f;  cn[T * sc[2] + R] += d * 1.0f;
                cn[Bm * sc[2] + L] += d * 1.0f; cn[Bm * sc[2] + R] += d * 1.0f;
            } else {                                                 /* my_lib.c:1507-1520 */
                /* when R==L or Bm==T the same cell is added twice, as in the reference */
                ox[T * s1[2] + L] += -fx;  ox[T * s1[2] + R] += -fx;
                ox[Bm * s1[2] + L] += -fx; ox[Bm * s1[2] + R] += -fx;
                oy[T * s1[2] + L] += -fy;  oy[T * s1[2] + R] += -fy;
                oy[Bm * s1[2] + L] += -fy; oy[Bm * s1[2] + R] += -fy;
                cn[T * sc[2] + L] += 1.0f;  cn[T * sc[2] + R] += 1.0f;
                cn[Bm * sc[2] + L] += 1.0f; cn[Bm * sc[2] + R] += 1.0f;
            }
        }
}

/* my_lib.c:1525-1536 */
static void project_average(int b, int H, int W, const i64 *s1, const float *count, const i64 *sc,
                            float *out)
{
    const i64 off = b * s1[0];
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float t = count[b * sc[0] + y * sc[2] + x];
            if (t > 0.0f) {
                out[off + 0 * s1[1] + y * s1[2] + x] /= t;
                out[off + 1 * s1[1] + y * s1[2] + x] /= t;
            }
        }
}

/* my_lib_kernel.cu:1776-1832.  The walk reads `count` and the averaged `out` of cells with count != 0;
 * it writes only cells with count <= 0, so the result does not depend on visiting order. */
static void project_fillhole(int b, int H, int W, const i64 *s1, const float *count, const i64 *sc,
                             float *out)
{
    const i64 off = b * s1[0];
    const float *cn = count + b * sc[0];
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            if (!(cn[y * sc[2] + x] <= 0.0f)) continue;
            int lo = x;  float lt = 0.0f;
            while (lt == 0.0f && lo - 1 >= 0) { lo--; lt = cn[y * sc[2] + lo]; }
            int ro = x;  float rt = 0.0f;
            while (rt == 0.0f && ro + 1 <= W - 1) { ro++; rt = cn[y * sc[2] + ro]; }
            int uo = y;  float ut = 0.0f;
            while (ut == 0.0f && uo - 1 >= 0) { uo--; ut = cn[uo * sc[2] + x]; }
            /* the downward search never runs in the reference: its loop condition is the assignment
             * `down_temp = 0.0f && ...` (my_lib_kernel.cu:1799), so down_temp stays 0 and
             * down_offset stays at the hole's own row. */
            int dn = y;  float dt = 0.0f;
            if (lt + rt + ut + dt <= 0.0f) continue;                         /* :1804-1807 */
            lt = (lt > 0.0f) ? 1 : 0;  rt = (rt > 0.0f) ? 1 : 0;
            ut = (ut > 0.0f) ? 1 : 0;  dt = (dt > 0.0f) ? 1 : 0;
            for (int k = 0; k < 2; k++) {                                    /* :1814-1831 */
                float *o = out + off + k * s1[1];
                o[y * s1[2] + x] = (lt * o[y * s1[2] + lo] + rt * o[y * s1[2] + ro] +
                                    ut * o[uo * s1[2] + x] + dt * o[dn * s1[2] + x]) /
                                   (lt + rt + ut + dt);
            }
        }
}

static int project_forward(int B, int C, int H, int W, const float *flow, const i64 *s1,
                           const float *depth, const i64 *sd,
                           float *count, const i64 *sc, float *out, const i64 *so, int fillhole)
{
    if (C != 2) return -1;                                           /* my_lib.c:1458 */
    if (s1[3] != 1 || sc[3] != 1) return -1;                          /* :1482,1485 */
    if (depth && sd[3] != 1) return -1;                               /* :1682 */
    if (s1[0] != so[0] || s1[1] != so[1]) return -1;                  /* :1483-1484 */
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; b++) {
        project_scatter(b, H, W, flow, s1, depth, sd, count, sc, out);
        project_average(b, H, W, s1, count, sc, out);
        if (fillhole) project_fillhole(b, H, W, s1, count, sc, out);
    }
    return 0;
}

/* fillhole: 0 = CPU-reference semantics; 1 = GPU-reference semantics (adds the .cu hole-fill pass) */
int memc_oracle_flow_projection_forward(int B, int C, int H, int W, const float *flow, const i64 *s1,
                                        float *count, const i64 *sc, float *out, const i64 *so,
                                        int fillhole)
{ return project_forward(B, C, H, W, flow, s1, NULL, NULL, count, sc, out, so, fillhole); }

int memc_oracle_depth_flow_projection_forward(int B, int C, int H, int W,
                                              const float *flow, const i64 *s1,
                                              const float *depth, const i64 *sd,
                                              float *count, const i64 *sc, float *out, const i64 *so,
                                              int fillhole)
{
    if (!depth) return -1;
    return project_forward(B, C, H, W, flow, s1, depth, sd, count, sc, out, so, fillhole);
}

/* my_lib.c:1549-1634 */
int memc_oracle_flow_projection_backward(int B, int C, int H, int W, const float *flow, const i64 *s1,
                                         const float *count, const i64 *sc, const float *gout,
                                         float *gin1)
{
    if (C != 2) return -1;
    if (s1[3] != 1 || sc[3] != 1) return -1;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; b++) {
        const i64 off = b * s1[0];
        const float *cn = count + b * sc[0];
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                float fx = flow[off + 0 * s1[1] + y * s1[2] + x];
                float fy = flow[off + 1 * s1[1] + y * s1[2] + x];
                float x2 = (float)x + fx, y2 = (float)y + fy;
                if (!(x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(W - 1) && y2 <= (float)(H - 1))) continue;
                int L = (int)x2, T = (int)y2;
                int R = imin(L + 1, W - 1), Bm = imin(T + 1, H - 1);
                for (int k = 0; k < 2; k++) {
                    const float *go = gout + off + k * s1[1];
                    float *g = gin1 + off + k * s1[1] + y * s1[2] + x;
                    *g += -go[T * s1[2] + L] / cn[T * sc[2] + L];
                    *g += -go[T * s1[2] + R] / cn[T * sc[2] + R];
                    *g += -go[Bm * s1[2] + L] / cn[Bm * sc[2] + L];
                    *g += -go[Bm * s1[2] + R] / cn[Bm * sc[2] + R];
                }
            }
    }
    return 0;
}

/* my_lib.c:1751-1878.  `out` is the forward pass's final output (after averaging). */
int memc_oracle_depth_flow_projection_backward(int B, int C, int H, int W,
                                               const float *flow, const i64 *s1,
                                               const float *depth, const i64 *sd,
                                               const float *count, const i64 *sc,
                                               const float *out, const float *gout,
                                               float *gin1, float *gin2)
{
    if (C != 2) return -1;
    if (s1[3] != 1 || sd[3] != 1 || sc[3] != 1) return -1;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; b++) {
        const i64 off = b * s1[0];
        const float *cn = count + b * sc[0];
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                float fx = flow[off + 0 * s1[1] + y * s1[2] + x];
                float fy = flow[off + 1 * s1[1] + y * s1[2] + x];
                float x2 = (float)x + fx, y2 = (float)y + fy;
                if (!(x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(W - 1) && y2 <= (float)(H - 1))) continue;
                int L = (int)x2, T = (int)y2;
                int R = imin(L + 1, W - 1), Bm = imin(T + 1, H - 1);
                float d = depth[b * sd[0] + y * sd[2] + x];
                for (int k = 0; k < 2; k++) {                               /* :1823-1841 */
                    const float *go = gout + off + k * s1[1];
                    float *g = gin1 + off + k * s1[1] + y * s1[2] + x;
                    *g += -go[T * s1[2] + L] * d / cn[T * sc[2] + L];
                    *g += -go[T * s1[2] + R] * d / cn[T * sc[2] + R];
                    *g += -go[Bm * s1[2] + L] * d / cn[Bm * sc[2] + L];
                    *g += -go[Bm * s1[2] + R] * d / cn[Bm * sc[2] + R];
                }
                float *gd = gin2 + b * sd[0] + y * sd[2] + x;               /* :1844-1869 */
                for (int k = 0; k < 2; k++) {
                    const float *go = gout + off + k * s1[1];
                    const float *o = out + off + k * s1[1];
                    float f = k ? fy : fx;
                    *gd += -go[T * s1[2] + L] / cn[T * sc[2] + L] * (f - o[T * s1[2] + L]);
                    *gd += -go[T * s1[2] + R] / cn[T * sc[2] + R] * (f - o[T * s1[2] + R]);
                    *gd += -go[Bm * s1[2] + L] / cn[Bm * sc[2] + L] * (f - o[Bm * s1[2] + L]);
                    *gd += -go[Bm * s1[2] + R] / cn[Bm * sc[2] + R] * (f - o[Bm * s1[2] + R]);
                }
            }
    }
    return 0;
}
