/*
 * orb_oracle.c -- CPU restatement of the reference ORB front-end (see orb_oracle.h header note:
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED at the OpenCV 3.1.0 boundary).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math (strict IEEE fp32).
 * All file:line citations are relative to /root/reference.
 */
#include "orb_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orb_pattern.inc"

/* ------------------------------------------------------------------------------------------------
 * OpenCV 3.1.0 scalar helpers
 * ---------------------------------------------------------------------------------------------- */

/* OpenCV 3.1.0: cvRound(double) = _mm_cvtsd_si32 / lrint => round half to even (default FE mode) */
int oro_cv_round(double v) { return (int)lrint(v); }
static int cv_floor(double v) { int i = (int)v; return i - (v < i); }
static int cv_ceil(double v) { int i = (int)v; return i + (v > i); }

/* OpenCV 3.1.0: core/src/mathfuncs.cpp fastAtan2 (degree polynomial, 0.3 deg max error) */
float oro_fast_atan2(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/*
 * src/ORBextractor.cc:109,114-115: angle = kpt.angle*factorPI; a = (float)cos(angle); b = (float)sin(angle)
 * (resolves to libm cosf/sinf).  libm's cosf/sinf differ between glibc versions and CPU ifunc
 * variants, so the canonical value is the CORRECTLY ROUNDED fp32 cosine/sine of the fp32 angle,
 * obtained here from the x87 80-bit cosl/sinl (a misrounding needs the true value within 2^-63 of
 * an fp32 tie).  DESIGN.md "canonical semantics".
 */
void oro_sincos_deg(float angle_deg, float *a_cos, float *b_sin)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = angle_deg * factorPI;
    *a_cos = (float)cosl((long double)angle);
    *b_sin = (float)sinl((long double)angle);
}

/* array form on radians (tools/verify_sincos.py: exhaustive check of the device routine) */
void oro_sincos_rad_array(const float *theta, float *c, float *s, long long n)
{
    for (long long i = 0; i < n; i++) {
        c[i] = (float)cosl((long double)theta[i]);
        s[i] = (float)sinl((long double)theta[i]);
    }
}

/* OpenCV 3.1.0: borderInterpolate(p, len, BORDER_REFLECT_101): -k -> k, len-1+k -> len-1-k */
int oro_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

/* ------------------------------------------------------------------------------------------------
 * A1: constructor (src/ORBextractor.cc:412-472)
 * ---------------------------------------------------------------------------------------------- */
const signed char *oro_pattern(void) { return ORO_PATTERN; }

int oro_extractor_init(oro_extractor *e, int nfeatures, float scale_factor, int nlevels,
                       int ini_th, int min_th)
{
    if (!e || nlevels < 1 || nlevels > ORO_MAX_LEVELS || nfeatures < 0) return -1;
    memset(e, 0, sizeof(*e));
    e->nfeatures = nfeatures;
    e->scale_factor = scale_factor;
    e->nlevels = nlevels;
    e->ini_th_fast = ini_th;
    e->min_th_fast = min_th;
    e->blur_mode = ORO_BLUR_SCALAR;

    /* :417-433 */
    e->scale[0] = 1.0f;
    e->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        e->scale[i] = e->scale[i - 1] * scale_factor;
        e->sigma2[i] = e->scale[i] * e->scale[i];
    }
    for (int i = 0; i < nlevels; i++) {
        e->inv_scale[i] = 1.0f / e->scale[i];
        e->inv_sigma2[i] = 1.0f / e->sigma2[i];
    }

    /* :437-448 */
    float factor = 1.0f / scale_factor;
    float nDesired = (float)nfeatures * (1 - factor) /
                     (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        e->quota[level] = oro_cv_round(nDesired);
        sum += e->quota[level];
        nDesired *= factor;
    }
    e->quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;

    /* :456-471 umax */
    int v, v0;
    int vmax = cv_floor(ORO_HALF_PATCH * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil(ORO_HALF_PATCH * sqrtf(2.f) / 2);
    const double hp2 = ORO_HALF_PATCH * ORO_HALF_PATCH;
    for (v = 0; v <= vmax; ++v) e->umax[v] = oro_cv_round(sqrt(hp2 - v * v));
    for (v = ORO_HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }

    /*
     * OpenCV 3.1.0: getGaussianKernel(7, 2, CV_32F) (imgproc/src/smooth.cpp), then
     * createSeparableLinearFilter's u8 path converts each 1-D kernel to CV_32S with scale 2^8.
     */
    {
        const int n = 7;
        const double sigmaX = 2.0;
        const double scale2X = -0.5 / (sigmaX * sigmaX);
        float cf[7];
        double s = 0;
        for (int i = 0; i < n; i++) {
            double x = i - (n - 1) * 0.5;
            double t = exp(scale2X * x * x);
            cf[i] = (float)t;
            s += cf[i];
        }
        s = 1. / s;
        for (int i = 0; i < n; i++) {
            cf[i] = (float)(cf[i] * s);
            e->gauss_k[i] = oro_cv_round((double)cf[i] * 256.0);
        }
    }
    return 0;
}

/* :1113-1114: size from the ORIGINAL image at every level */
void oro_level_size(const oro_extractor *e, int W, int H, int level, int *w, int *h)
{
    float scale = e->inv_scale[level];
    *w = oro_cv_round((float)W * scale);
    *h = oro_cv_round((float)H * scale);
}

/* ------------------------------------------------------------------------------------------------
 * OpenCV 3.1.0: cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1
 * (imgproc/src/imgwarp.cpp: resize -> resizeGeneric_<HResizeLinear<uchar,int,short,2048>,
 *  VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>>), portable C path, no IPP.
 * ---------------------------------------------------------------------------------------------- */
static short sat_short(float v)
{
    int iv = oro_cv_round(v);
    return (short)(iv < SHRT_MIN ? SHRT_MIN : iv > SHRT_MAX ? SHRT_MAX : iv);
}

static void resize_area_fast2(const uint8_t *src, int sstride, uint8_t *dst, int dw, int dh,
                              int dstride)
{
    /* OpenCV 3.1.0: INTER_LINEAR with an exact 2x decimation is rerouted to INTER_AREA
     * (ResizeAreaFastVec: (s00+s01+s10+s11+2)>>2). */
    for (int y = 0; y < dh; y++) {
        const uint8_t *s0 = src + (size_t)(2 * y) * sstride, *s1 = s0 + sstride;
        for (int x = 0; x < dw; x++)
            dst[(size_t)y * dstride + x] =
                (uint8_t)((s0[2 * x] + s0[2 * x + 1] + s1[2 * x] + s1[2 * x + 1] + 2) >> 2);
    }
}

void oro_resize_linear(const uint8_t *src, int sw, int sh, int sstride,
                       uint8_t *dst, int dw, int dh, int dstride)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;

    if (dw == sw && dh == sh) { /* resize() with equal sizes is a copy */
        for (int y = 0; y < dh; y++) memcpy(dst + (size_t)y * dstride, src + (size_t)y * sstride, (size_t)dw);
        return;
    }
    {
        int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);
        int is_area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
        if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
            resize_area_fast2(src, sstride, dst, dw, dh, dstride);
            return;
        }
    }

    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * (size_t)dw);
    int *yofs = (int *)malloc(sizeof(int) * (size_t)dh);
    short *ibeta = (short *)malloc(sizeof(short) * 2 * (size_t)dh);
    int *rows = (int *)malloc(sizeof(int) * 2 * (size_t)dw);

    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[dx * 2] = sat_short((1.f - fx) * 2048);
        ialpha[dx * 2 + 1] = sat_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = sat_short((1.f - fy) * 2048);
        ibeta[dy * 2 + 1] = sat_short(fy * 2048);
    }

    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy];
        for (int k = 0; k < 2; k++) {
            int sy = sy0 + k; /* clip(sy0 - ksize2 + 1 + k, 0, ssize.height) */
            sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
            const uint8_t *S = src + (size_t)sy * sstride;
            int *D = rows + (size_t)k * dw;
            for (int dx = 0; dx < dw; dx++) {
                int sx = xofs[dx];
                int s1 = sx + 1 < sw ? S[sx + 1] : 0; /* weight is 0 there */
                D[dx] = S[sx] * ialpha[dx * 2] + s1 * ialpha[dx * 2 + 1];
            }
        }
        short b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        const int *S0 = rows, *S1 = rows + dw;
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++)
            D[x] = (uint8_t)((((b0 * (S0[x] >> 4)) >> 16) + ((b1 * (S1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(yofs); free(ibeta); free(rows);
}

/* OpenCV 3.1.0: copyMakeBorder(..., BORDER_REFLECT_101) (src/ORBextractor.cc:1127,1132) */
void oro_copy_make_border101(const uint8_t *src, int w, int h, int sstride,
                             uint8_t *dst, int dstride, int border)
{
    for (int y = -border; y < h + border; y++) {
        const uint8_t *S = src + (size_t)oro_reflect101(y, h) * sstride;
        uint8_t *D = dst + (size_t)(y + border) * dstride;
        for (int x = -border; x < w + border; x++) D[x + border] = S[oro_reflect101(x, w)];
    }
}

/* ------------------------------------------------------------------------------------------------
 * OpenCV 3.1.0: GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) for CV_8UC1
 * (src/ORBextractor.cc:1088).  Separable fixed-point filter: row pass exact int32 with the x256
 * kernel; column pass FixedPtCastEx<int,uchar>(16): saturate_u8((sum + 32768) >> 16).  The x86 SSE2
 * column path (SymmColumnVec_32s8u, columns x < (w & ~3)) converts to float and rounds half to
 * even; the float sums are exact below 256, so the two differ only on exact .5 ties.
 * ---------------------------------------------------------------------------------------------- */
void oro_gaussian_blur7(const uint8_t *src, int w, int h, int sstride,
                        uint8_t *dst, int dstride, const int k[7], int mode)
{
    int *rowbuf = (int *)malloc(sizeof(int) * (size_t)w * (size_t)h);
    for (int y = 0; y < h; y++) {
        const uint8_t *S = src + (size_t)y * sstride;
        int *R = rowbuf + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = 0; i < 7; i++) s += k[i] * S[oro_reflect101(x + i - 3, w)];
            R[x] = s;
        }
    }
    int simd_cols = (mode == ORO_BLUR_X86_SIMD) ? (w & ~3) : 0;
    for (int y = 0; y < h; y++) {
        uint8_t *D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = 0; i < 7; i++) s += k[i] * rowbuf[(size_t)oro_reflect101(y + i - 3, h) * w + x];
            int v;
            if (x < simd_cols) {
                v = s >> 16;
                int rem = s & 0xFFFF;
                if (rem > 0x8000 || (rem == 0x8000 && (v & 1))) v++;
            } else {
                v = (s + 32768) >> 16;
            }
            D[x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
    free(rowbuf);
}

/* ------------------------------------------------------------------------------------------------
 * OpenCV 3.1.0: cv::FAST(img, kps, threshold, nonmax=true, TYPE_9_16)
 * (features2d/src/fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>), scalar path.
 * ---------------------------------------------------------------------------------------------- */
static const int FAST_OFFS[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static void make_offsets(int pixel[25], int stride)
{
    int k = 0;
    for (; k < 16; k++) pixel[k] = FAST_OFFS[k][0] + FAST_OFFS[k][1] * stride;
    for (; k < 25; k++) pixel[k] = pixel[k - 16];
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

static int corner_score16(const uint8_t *ptr, const int pixel[25], int threshold)
{
    const int N = 25;
    int k, v = ptr[0];
    short d[25];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);

    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = imin((int)d[k + 1], (int)d[k + 2]);
        a = imin(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = imin(a, (int)d[k + 4]);
        a = imin(a, (int)d[k + 5]);
        a = imin(a, (int)d[k + 6]);
        a = imin(a, (int)d[k + 7]);
        a = imin(a, (int)d[k + 8]);
        a0 = imax(a0, imin(a, (int)d[k]));
        a0 = imax(a0, imin(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = imax((int)d[k + 1], (int)d[k + 2]);
        b = imax(b, (int)d[k + 3]);
        b = imax(b, (int)d[k + 4]);
        b = imax(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = imax(b, (int)d[k + 6]);
        b = imax(b, (int)d[k + 7]);
        b = imax(b, (int)d[k + 8]);
        b0 = imin(b0, imax(b, (int)d[k]));
        b0 = imin(b0, imax(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

int oro_fast_score_pixel(const uint8_t *p, int stride)
{
    /* threshold-free: max over the 16 nine-arcs of min(d) and of min(-d), minus 1 */
    int d[25], best = INT_MIN;
    for (int k = 0; k < 25; k++) {
        int kk = k & 15;
        d[k] = (int)p[0] - (int)p[FAST_OFFS[kk][0] + FAST_OFFS[kk][1] * stride];
    }
    for (int s = 0; s < 16; s++) {
        int mn = INT_MAX, mx = INT_MIN;
        for (int j = 0; j < 9; j++) { mn = imin(mn, d[s + j]); mx = imax(mx, d[s + j]); }
        best = imax(best, imax(mn, -mx));
    }
    return best - 1;
}

int oro_fast9_16(const uint8_t *img, int stride, int cols, int rows, int threshold, int nonmax,
                 oro_cand *out, int cap)
{
    const int K = 8, N = 25;
    int i, j, k, pixel[25], nout = 0;
    make_offsets(pixel, stride);

    threshold = imin(imax(threshold, 0), 255);
    uint8_t threshold_tab[512];
    for (i = -255; i <= 255; i++)
        threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);

    if (cols <= 0 || rows <= 0) return 0;
    uint8_t *bufmem = (uint8_t *)calloc((size_t)cols * 3, 1);
    int *cpmem = (int *)malloc(sizeof(int) * 3 * ((size_t)cols + 1));
    uint8_t *buf[3] = {bufmem, bufmem + cols, bufmem + 2 * cols};
    int *cpbuf[3] = {cpmem + 1, cpmem + 1 + (cols + 1), cpmem + 1 + 2 * (cols + 1)};
    cpbuf[0][-1] = cpbuf[1][-1] = cpbuf[2][-1] = 0;

    for (i = 3; i < rows - 2; i++) {
        const uint8_t *ptr = img + (size_t)i * stride + 3;
        uint8_t *curr = buf[(i - 3) % 3];
        int *cornerpos = cpbuf[(i - 3) % 3];
        memset(curr, 0, (size_t)cols);
        int ncorners = 0;

        if (i < rows - 3) {
            for (j = 3; j < cols - 3; j++, ptr++) {
                int v = ptr[0];
                const uint8_t *tab = &threshold_tab[0] - v + 255;
                int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
                d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
                d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
                d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
                d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
                d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];

                if (d & 1) {
                    int vt = v - threshold, count = 0;
                    for (k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x < vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                if (nonmax) curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                                break;
                            }
                        } else
                            count = 0;
                    }
                }
                if (d & 2) {
                    int vt = v + threshold, count = 0;
                    for (k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x > vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                if (nonmax) curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                                break;
                            }
                        } else
                            count = 0;
                    }
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;

        const uint8_t *prev = buf[(i - 4 + 3) % 3];
        const uint8_t *pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];
        for (k = 0; k < ncorners; k++) {
            j = cornerpos[k];
            int score = prev[j];
            if (!nonmax ||
                (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] &&
                 score > pprev[j] && score > pprev[j + 1] && score > curr[j - 1] &&
                 score > curr[j] && score > curr[j + 1])) {
                if (nout >= cap) { free(bufmem); free(cpmem); return -1; }
                out[nout].x = j;
                out[nout].y = i - 1;
                out[nout].response = score;
                nout++;
            }
        }
    }
    free(bufmem);
    free(cpmem);
    return nout;
}

/* ------------------------------------------------------------------------------------------------
 * A2: ComputePyramid (src/ORBextractor.cc:1109-1137).  The 19-px reflect-101 border the reference
 * adds is never read on this path; levels here are interior only and
 * oro_copy_make_border101 reproduces the bordered buffer when a caller wants it.
 * ---------------------------------------------------------------------------------------------- */
void oro_compute_pyramid(const oro_extractor *e, const uint8_t *img, int W, int H, int stride,
                         uint8_t *const *levels)
{
    int pw = W, ph = H;
    for (int level = 0; level < e->nlevels; level++) {
        int w, h;
        oro_level_size(e, W, H, level, &w, &h);
        if (level == 0) {
            for (int y = 0; y < h; y++) memcpy(levels[0] + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
        } else {
            oro_resize_linear(levels[level - 1], pw, ph, pw, levels[level], w, h, w); /* :1123 */
        }
        pw = w; ph = h;
    }
}

/* ------------------------------------------------------------------------------------------------
 * A3: ComputeKeyPointsOctTree cell loop (src/ORBextractor.cc:767-831)
 * ---------------------------------------------------------------------------------------------- */
int oro_detect_level(const oro_extractor *e, const uint8_t *lvl, int w, int h, int stride,
                     oro_cand *out, int cap)
{
    const float W = 30;
    const int minBorderX = ORO_EDGE_THRESHOLD - 3;
    const int minBorderY = minBorderX;
    const int maxBorderX = w - ORO_EDGE_THRESHOLD + 3;
    const int maxBorderY = h - ORO_EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX);
    const float height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W);
    const int nRows = (int)(height / W);
    if (nCols <= 0 || nRows <= 0) return 0; /* reference: loops do not execute (wCell would be UB) */
    const int wCell = (int)ceilf(width / nCols);
    const int hCell = (int)ceilf(height / nRows);
    int n = 0;
    enum { CELL_CAP = 4096 };
    oro_cand *cell = (oro_cand *)malloc(sizeof(oro_cand) * CELL_CAP);

    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;

            const uint8_t *roi = lvl + (size_t)(int)iniY * stride + (int)iniX;
            int rc = (int)maxX - (int)iniX, rr = (int)maxY - (int)iniY;
            int nc = oro_fast9_16(roi, stride, rc, rr, e->ini_th_fast, 1, cell, CELL_CAP); /* :811 */
            if (nc == 0)
                nc = oro_fast9_16(roi, stride, rc, rr, e->min_th_fast, 1, cell, CELL_CAP); /* :816 */
            if (nc < 0) { free(cell); return -1; }
            for (int k = 0; k < nc; k++) {
                if (n >= cap) { free(cell); return -1; }
                out[n].x = cell[k].x + j * wCell; /* :824-825 */
                out[n].y = cell[k].y + i * hCell;
                out[n].response = cell[k].response;
                n++;
            }
        }
    }
    free(cell);
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * A4: DistributeOctTree + ExtractorNode::DivideNode (src/ORBextractor.cc:483-765)
 * std::list<ExtractorNode> restated as a doubly linked list over an append-only node arena, so a
 * node's arena index is its creation order; the reference's pointer tie-break (:629,:686) is taken
 * as creation order (what a never-reusing bump allocator yields).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int ulx, uly, urx, bry;   /* UL.x, UL.y, UR.x, BR.y (BL/BR.x/UR.y are redundant) */
    int *keys; int nkeys;     /* indices into cands, reference order */
    int no_more;
    int prev, next;           /* list links (-1 = none) */
} onode;

typedef struct {
    onode *nodes; int nnodes, capnodes;
    int head, tail, size;
} olist;

static int ol_new(olist *L)
{
    if (L->nnodes == L->capnodes) {
        L->capnodes = L->capnodes ? L->capnodes * 2 : 256;
        L->nodes = (onode *)realloc(L->nodes, sizeof(onode) * (size_t)L->capnodes);
    }
    onode *n = &L->nodes[L->nnodes];
    memset(n, 0, sizeof(*n));
    n->prev = n->next = -1;
    return L->nnodes++;
}
static void ol_push_front(olist *L, int id)
{
    L->nodes[id].prev = -1;
    L->nodes[id].next = L->head;
    if (L->head >= 0) L->nodes[L->head].prev = id; else L->tail = id;
    L->head = id;
    L->size++;
}
static void ol_push_back(olist *L, int id)
{
    L->nodes[id].next = -1;
    L->nodes[id].prev = L->tail;
    if (L->tail >= 0) L->nodes[L->tail].next = id; else L->head = id;
    L->tail = id;
    L->size++;
}
static int ol_erase(olist *L, int id) /* returns next */
{
    int p = L->nodes[id].prev, n = L->nodes[id].next;
    if (p >= 0) L->nodes[p].next = n; else L->head = n;
    if (n >= 0) L->nodes[n].prev = p; else L->tail = p;
    L->size--;
    return n;
}

/* DivideNode :483-539; creates 4 arena nodes (not yet linked); ids in c[4] */
static void divide_node(olist *L, int id, const oro_cand *cands, int c[4])
{
    for (int k = 0; k < 4; k++) c[k] = ol_new(L);
    onode *P = &L->nodes[id];
    const int halfX = (int)ceilf((float)(P->urx - P->ulx) / 2);
    const int halfY = (int)ceilf((float)(P->bry - P->uly) / 2);
    onode *n1 = &L->nodes[c[0]], *n2 = &L->nodes[c[1]], *n3 = &L->nodes[c[2]], *n4 = &L->nodes[c[3]];
    n1->ulx = P->ulx;          n1->uly = P->uly;          n1->urx = P->ulx + halfX; n1->bry = P->uly + halfY;
    n2->ulx = P->ulx + halfX;  n2->uly = P->uly;          n2->urx = P->urx;         n2->bry = P->uly + halfY;
    n3->ulx = P->ulx;          n3->uly = P->uly + halfY;  n3->urx = P->ulx + halfX; n3->bry = P->bry;
    n4->ulx = P->ulx + halfX;  n4->uly = P->uly + halfY;  n4->urx = P->urx;         n4->bry = P->bry;
    for (int k = 0; k < 4; k++) {
        L->nodes[c[k]].keys = (int *)malloc(sizeof(int) * (size_t)(P->nkeys ? P->nkeys : 1));
        L->nodes[c[k]].nkeys = 0;
    }
    const float midx = (float)n1->urx, midy = (float)n1->bry;
    for (int i = 0; i < P->nkeys; i++) {
        const oro_cand *kp = &cands[P->keys[i]];
        onode *dst;
        if ((float)kp->x < midx) dst = ((float)kp->y < midy) ? n1 : n3;
        else dst = ((float)kp->y < midy) ? n2 : n4;
        dst->keys[dst->nkeys++] = P->keys[i];
    }
    for (int k = 0; k < 4; k++)
        if (L->nodes[c[k]].nkeys == 1) L->nodes[c[k]].no_more = 1;
}

typedef struct { int size; int id; } size_id;
static int cmp_size_id(const void *a, const void *b)
{
    const size_id *x = (const size_id *)a, *y = (const size_id *)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

int oro_distribute_octree(const oro_cand *cands, int n, int minX, int maxX, int minY, int maxY,
                          int N, int32_t *out_idx, int cap)
{
    olist L;
    memset(&L, 0, sizeof(L));
    L.head = L.tail = -1;

    /* :545-565 */
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    if (nIni <= 0) return n == 0 ? 0 : -3; /* reference indexes an empty vector: UB */
    const float hX = (float)(maxX - minX) / nIni;
    int *ini = (int *)malloc(sizeof(int) * (size_t)nIni);
    for (int i = 0; i < nIni; i++) {
        int id = ol_new(&L);
        onode *ni = &L.nodes[id];
        ni->ulx = (int)(hX * (float)i);
        ni->uly = 0;
        ni->urx = (int)(hX * (float)(i + 1));
        ni->bry = maxY - minY;
        ni->keys = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
        ol_push_back(&L, id);
        ini[i] = id;
    }
    /* :568-572 */
    for (int i = 0; i < n; i++) {
        int b = (int)((float)cands[i].x / hX);
        if (b < 0 || b >= nIni) { b = b < 0 ? 0 : nIni - 1; } /* cannot happen for in-region points */
        onode *nd = &L.nodes[ini[b]];
        nd->keys[nd->nkeys++] = i;
    }
    free(ini);
    /* :574-587 */
    for (int lit = L.head; lit >= 0;) {
        onode *nd = &L.nodes[lit];
        if (nd->nkeys == 1) { nd->no_more = 1; lit = nd->next; }
        else if (nd->nkeys == 0) lit = ol_erase(&L, lit);
        else lit = nd->next;
    }

    int finish = 0;
    size_id *vSize = NULL, *vPrev = NULL;
    int nSize = 0, capSize = 0, capPrev = 0;
#define PUSH_SIZE(sz, idv) do { \
        if (nSize == capSize) { capSize = capSize ? capSize * 2 : 256; \
            vSize = (size_id *)realloc(vSize, sizeof(size_id) * (size_t)capSize); } \
        vSize[nSize].size = (sz); vSize[nSize].id = (idv); nSize++; } while (0)

    while (!finish) {
        int prevSize = L.size;
        int lit = L.head;
        int nToExpand = 0;
        nSize = 0;
        while (lit >= 0) {
            if (L.nodes[lit].no_more) { lit = L.nodes[lit].next; continue; }
            int c[4];
            divide_node(&L, lit, cands, c);
            for (int k = 0; k < 4; k++) { /* :623-662 */
                if (L.nodes[c[k]].nkeys > 0) {
                    ol_push_front(&L, c[k]);
                    if (L.nodes[c[k]].nkeys > 1) { nToExpand++; PUSH_SIZE(L.nodes[c[k]].nkeys, c[k]); }
                }
            }
            lit = ol_erase(&L, lit); /* :664 */
        }
        /* :671-740 */
        if (L.size >= N || L.size == prevSize) {
            finish = 1;
        } else if (L.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = L.size;
                if (capPrev < nSize) { capPrev = nSize; vPrev = (size_id *)realloc(vPrev, sizeof(size_id) * (size_t)(capPrev ? capPrev : 1)); }
                int nPrev = nSize;
                memcpy(vPrev, vSize, sizeof(size_id) * (size_t)nPrev);
                nSize = 0;
                qsort(vPrev, (size_t)nPrev, sizeof(size_id), cmp_size_id); /* :686 (size, pointer) */
                for (int j = nPrev - 1; j >= 0; j--) {
                    int c[4];
                    divide_node(&L, vPrev[j].id, cands, c);
                    for (int k = 0; k < 4; k++) {
                        if (L.nodes[c[k]].nkeys > 0) {
                            ol_push_front(&L, c[k]);
                            if (L.nodes[c[k]].nkeys > 1) PUSH_SIZE(L.nodes[c[k]].nkeys, c[k]);
                        }
                    }
                    ol_erase(&L, vPrev[j].id); /* :730 */
                    if (L.size >= N) break;    /* :732 */
                }
                if (L.size >= N || L.size == prevSize) finish = 1;
            }
        }
    }
#undef PUSH_SIZE

    /* :743-762 retain the best point per node, first wins on ties */
    int nout = 0, rc = 0;
    for (int lit = L.head; lit >= 0; lit = L.nodes[lit].next) {
        onode *nd = &L.nodes[lit];
        int best = nd->keys[0];
        int maxResponse = cands[best].response;
        for (int k = 1; k < nd->nkeys; k++) {
            if (cands[nd->keys[k]].response > maxResponse) {
                best = nd->keys[k];
                maxResponse = cands[best].response;
            }
        }
        if (nout >= cap) { rc = -2; break; }
        out_idx[nout++] = best;
    }
    for (int i = 0; i < L.nnodes; i++) free(L.nodes[i].keys);
    free(L.nodes); free(vSize); free(vPrev);
    return rc < 0 ? rc : nout;
}

/* ------------------------------------------------------------------------------------------------
 * A5: IC_Angle (src/ORBextractor.cc:79-106)
 * ---------------------------------------------------------------------------------------------- */
float oro_ic_angle(const uint8_t *lvl, int stride, int x, int y, const int umax[16])
{
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = lvl + (ptrdiff_t)y * stride + x;
    for (int u = -ORO_HALF_PATCH; u <= ORO_HALF_PATCH; ++u) m_10 += u * center[u];
    for (int v = 1; v <= ORO_HALF_PATCH; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return oro_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------------------------------------------------------------
 * A7: computeOrbDescriptor (src/ORBextractor.cc:110-149)
 * ---------------------------------------------------------------------------------------------- */
void oro_descriptor(const uint8_t *blurred, int stride, int x, int y, float angle_deg,
                    uint8_t desc[32])
{
    float a, b;
    oro_sincos_deg(angle_deg, &a, &b);
    const uint8_t *center = blurred + (ptrdiff_t)y * stride + x;
    const signed char *pat = ORO_PATTERN;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            float x0 = (float)pat[4 * k], y0 = (float)pat[4 * k + 1];
            float x1 = (float)pat[4 * k + 2], y1 = (float)pat[4 * k + 3];
            /* GET_VALUE(idx): center[cvRound(p.x*b + p.y*a)*step + cvRound(p.x*a - p.y*b)] */
            float r0a = x0 * b, r0b = y0 * a, c0a = x0 * a, c0b = y0 * b;
            float r1a = x1 * b, r1b = y1 * a, c1a = x1 * a, c1b = y1 * b;
            int t0 = center[oro_cv_round(r0a + r0b) * stride + oro_cv_round(c0a - c0b)];
            int t1 = center[oro_cv_round(r1a + r1b) * stride + oro_cv_round(c1a - c1b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------------------------------------
 * A8: operator() (src/ORBextractor.cc:1045-1107) + the rest of ComputeKeyPointsOctTree (:833-854)
 * ---------------------------------------------------------------------------------------------- */
int oro_extract(const oro_extractor *e, const uint8_t *img, int W, int H, int stride,
                oro_keypoint *kps, uint8_t *desc, int cap, int *n,
                uint8_t *const *level_out, int *n_per_level)
{
    if (!e || !n) return -1;
    *n = 0;
    if (!img || W <= 0 || H <= 0) return 0; /* :1048 empty image: silent return */
    const int L = e->nlevels;
    uint8_t *lv[ORO_MAX_LEVELS] = {0};
    int lw[ORO_MAX_LEVELS], lh[ORO_MAX_LEVELS];
    for (int l = 0; l < L; l++) {
        oro_level_size(e, W, H, l, &lw[l], &lh[l]);
        if (lw[l] <= 0 || lh[l] <= 0) return -3;
    }
    for (int l = 0; l < L; l++) lv[l] = (uint8_t *)malloc((size_t)lw[l] * lh[l]);
    oro_compute_pyramid(e, img, W, H, stride, lv);
    if (level_out)
        for (int l = 0; l < L; l++) memcpy(level_out[l], lv[l], (size_t)lw[l] * lh[l]);

    int rc = 0, total = 0;
    const int cand_cap = 1 << 20;
    oro_cand *cands = (oro_cand *)malloc(sizeof(oro_cand) * (size_t)cand_cap);
    int32_t *sel = (int32_t *)malloc(sizeof(int32_t) * (size_t)cand_cap); /* a node list never outgrows the candidates */
    for (int l = 0; l < L && rc == 0; l++) {
        const int w = lw[l], h = lh[l];
        const int minBorderX = ORO_EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = w - ORO_EDGE_THRESHOLD + 3, maxBorderY = h - ORO_EDGE_THRESHOLD + 3;
        int nc = oro_detect_level(e, lv[l], w, h, w, cands, cand_cap);
        if (nc < 0) { rc = -2; break; }
        int nk = 0;
        if (maxBorderY - minBorderY <= 0 || maxBorderX - minBorderX <= 0) {
            if (nc > 0) { rc = -3; break; }
        } else {
            nk = oro_distribute_octree(cands, nc, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                       e->quota[l], sel, cand_cap); /* :836 */
            if (nk < 0) { rc = nk; break; }
        }
        if (n_per_level) n_per_level[l] = nk;
        if (nk == 0) continue;
        if (total + nk > cap) { rc = -2; break; }

        const int scaledPatchSize = (int)(ORO_PATCH_SIZE * e->scale[l]); /* :839 */
        uint8_t *blur = (uint8_t *)malloc((size_t)w * h);
        oro_gaussian_blur7(lv[l], w, h, w, blur, w, e->gauss_k, e->blur_mode); /* :1087-1088 */
        for (int i = 0; i < nk; i++) {
            const oro_cand *c = &cands[sel[i]];
            oro_keypoint *kp = &kps[total + i];
            int px = c->x + minBorderX, py = c->y + minBorderY; /* :845-846 */
            kp->x = (float)px;
            kp->y = (float)py;
            kp->size = (float)scaledPatchSize;
            kp->response = (float)c->response;
            kp->octave = l;
            kp->class_id = -1;
            kp->angle = oro_ic_angle(lv[l], w, px, py, e->umax); /* :853-854 */
            oro_descriptor(blur, w, px, py, kp->angle, desc + (size_t)(total + i) * 32);
            if (l != 0) { /* :1097-1103 */
                float scale = e->scale[l];
                kp->x *= scale;
                kp->y *= scale;
            }
        }
        free(blur);
        total += nk;
    }
    free(cands); free(sel);
    for (int l = 0; l < L; l++) free(lv[l]);
    if (rc == 0) *n = total;
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * A9: ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1647-1663)
 * ---------------------------------------------------------------------------------------------- */
int oro_descriptor_distance(const uint8_t a[32], const uint8_t b[32])
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24);
    }
    return dist;
}

/* A10: best / second-best (src/ORBmatcher.cc:201-226; same shape at :566-596, :432-457 ...) */
void oro_best2(const uint8_t *q, int nq, const uint8_t *t, int nt,
               const int32_t *cand_off, const int32_t *cand_idx,
               int32_t *best_idx, int32_t *best_d, int32_t *second_d)
{
    for (int i = 0; i < nq; i++) {
        int bestDist1 = 256, bestIdx = -1, bestDist2 = 256;
        int lo = cand_off ? cand_off[i] : 0, hi = cand_off ? cand_off[i + 1] : nt;
        for (int c = lo; c < hi; c++) {
            int j = cand_off ? cand_idx[c] : c;
            int dist = oro_descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx = j; }
            else if (dist < bestDist2) { bestDist2 = dist; }
        }
        best_idx[i] = bestIdx; best_d[i] = bestDist1; second_d[i] = bestDist2;
    }
}

/* A11: ComputeThreeMaxima (src/ORBmatcher.cc:1601-1642) */
void oro_three_maxima(const int *histo, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* :236-244 */
int oro_rot_bin(float angle1, float angle2)
{
    const float factor = 1.0f / 30;
    float rot = angle1 - angle2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == 30) bin = 0;
    return bin;
}

int oro_rot_filter(const float *angle_q, const float *angle_t, int32_t *match12, int nq)
{
    int hist[30] = {0};
    int *bins = (int *)malloc(sizeof(int) * (size_t)(nq ? nq : 1));
    int nmatches = 0;
    for (int i = 0; i < nq; i++) {
        bins[i] = -1;
        if (match12[i] < 0) continue;
        bins[i] = oro_rot_bin(angle_q[i], angle_t[match12[i]]);
        hist[bins[i]]++;
        nmatches++;
    }
    int ind1 = -1, ind2 = -1, ind3 = -1;
    oro_three_maxima(hist, 30, &ind1, &ind2, &ind3);
    for (int i = 0; i < nq; i++) {
        if (bins[i] < 0) continue;
        if (bins[i] != ind1 && bins[i] != ind2 && bins[i] != ind3) { match12[i] = -1; nmatches--; }
    }
    free(bins);
    return nmatches;
}

int oro_match_dense(const uint8_t *q, const float *angle_q, int nq,
                    const uint8_t *t, const float *angle_t, int nt,
                    int th, float nnratio, int check_ori, int32_t *match12)
{
    int32_t *bi = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)(nq ? nq : 1));
    int32_t *bd = bi + nq, *sd = bd + nq;
    oro_best2(q, nq, t, nt, NULL, NULL, bi, bd, sd);
    int nmatches = 0;
    for (int i = 0; i < nq; i++) {
        match12[i] = -1;
        if (bd[i] <= th && (float)bd[i] < nnratio * (float)sd[i]) { match12[i] = bi[i]; nmatches++; } /* :228-232 */
    }
    free(bi);
    if (check_ori) nmatches = oro_rot_filter(angle_q, angle_t, match12, nq);
    return nmatches;
}

/* ------------------------------------------------------------------------------------------------
 * N1: Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea (src/Frame.cc:230-245, 327-392)
 * ---------------------------------------------------------------------------------------------- */
static int pos_in_grid(const oro_grid *g, const oro_keypoint *kp, int *px, int *py)
{
    *px = (int)roundf((kp->x - g->min_x) * g->inv_w);   /* :384 */
    *py = (int)roundf((kp->y - g->min_y) * g->inv_h);   /* :385 */
    if (*px < 0 || *px >= ORO_GRID_COLS || *py < 0 || *py >= ORO_GRID_ROWS) return 0;
    return 1;
}

void oro_grid_build(oro_grid *g, const oro_keypoint *kps_un, int n, float min_x, float max_x, float min_y, float max_y, int *items)
{
    g->min_x = min_x; g->max_x = max_x; g->min_y = min_y; g->max_y = max_y;
    g->inv_w = (float)ORO_GRID_COLS / (max_x - min_x);   /* :212 */
    g->inv_h = (float)ORO_GRID_ROWS / (max_y - min_y);   /* :213 */
    g->n = n; g->items = items;
    const int NC = ORO_GRID_COLS * ORO_GRID_ROWS;
    int *cnt = (int *)calloc((size_t)NC + 1, sizeof(int));
    int *cell = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    for (int i = 0; i < n; i++) {
        int px, py;
        cell[i] = pos_in_grid(g, &kps_un[i], &px, &py) ? px * ORO_GRID_ROWS + py : -1;
        if (cell[i] >= 0) cnt[cell[i]]++;
    }
    g->cell_start[0] = 0;
    for (int c = 0; c < NC; c++) g->cell_start[c + 1] = g->cell_start[c] + cnt[c];
    memset(cnt, 0, sizeof(int) * (size_t)NC);
    for (int i = 0; i < n; i++)      /* push_back in keypoint order (:237-244) */
        if (cell[i] >= 0) items[g->cell_start[cell[i]] + cnt[cell[i]]++] = i;
    free(cnt); free(cell);
}

int oro_features_in_area(const oro_grid *g, const oro_keypoint *kps_un, float x, float y, float r,
                         int minLevel, int maxLevel, int32_t *out, int cap)
{
    int n = 0;
    const int nMinCellX = imax(0, (int)floorf((x - g->min_x - r) * g->inv_w));
    if (nMinCellX >= ORO_GRID_COLS) return 0;
    const int nMaxCellX = imin(ORO_GRID_COLS - 1, (int)ceilf((x - g->min_x + r) * g->inv_w));
    if (nMaxCellX < 0) return 0;
    const int nMinCellY = imax(0, (int)floorf((y - g->min_y - r) * g->inv_h));
    if (nMinCellY >= ORO_GRID_ROWS) return 0;
    const int nMaxCellY = imin(ORO_GRID_ROWS - 1, (int)ceilf((y - g->min_y + r) * g->inv_h));
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * ORO_GRID_ROWS + iy;
            for (int j = g->cell_start[c]; j < g->cell_start[c + 1]; j++) {
                const oro_keypoint *kpUn = &kps_un[g->items[j]];
                if (bCheckLevels) {
                    if (kpUn->octave < minLevel) continue;
                    if (maxLevel >= 0 && kpUn->octave > maxLevel) continue;
                }
                const float distx = kpUn->x - x, disty = kpUn->y - y;
                if (fabsf(distx) < r && fabsf(disty) < r) {
                    if (n >= cap) return -1;
                    out[n++] = g->items[j];
                }
            }
        }
    }
    return n;
}

void oro_search_area_best2(const oro_grid *g, const oro_keypoint *kps_un, const uint8_t *train_desc, const uint8_t *skip,
                           const uint8_t *qdesc, const float *x, const float *y, const float *r,
                           const int32_t *min_level, const int32_t *max_level, int nq,
                           int32_t *best_idx, int32_t *best_d, int32_t *second_d)
{
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(g->n ? g->n : 1));
    for (int i = 0; i < nq; i++) {
        int nc = oro_features_in_area(g, kps_un, x[i], y[i], r[i], min_level[i], max_level[i], idx, g->n);
        int bestDist = 256, bestDist2 = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = idx[c];
            if (skip && skip[i2]) continue;
            const int dist = oro_descriptor_distance(qdesc + (size_t)i * 32, train_desc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = i2; }
            else if (dist < bestDist2) { bestDist2 = dist; }
        }
        best_idx[i] = bestIdx; best_d[i] = bestDist; second_d[i] = bestDist2;
    }
    free(idx);
}

/* ------------------------------------------------------------------------------------------------
 * N3: Frame::ComputeStereoMatches (src/Frame.cc:466-640)
 * ---------------------------------------------------------------------------------------------- */
static int cmp_pair_ii(const void *a, const void *b)
{
    const int *x = (const int *)a, *y = (const int *)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    return x[1] < y[1] ? -1 : (x[1] > y[1] ? 1 : 0);
}

void oro_stereo_matches(const oro_extractor *e, const oro_keypoint *kl, const uint8_t *dl, int N,
                        const oro_keypoint *kr, const uint8_t *dr, int Nr,
                        uint8_t *const *pyrL, uint8_t *const *pyrR, const int *lw, const int *lh,
                        float mb, float mbf, float *mvuRight, float *mvDepth)
{
    for (int i = 0; i < N; i++) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; }
    const int thOrbDist = (100 + 50) / 2;                       /* (TH_HIGH+TH_LOW)/2 :471 */
    const int nRows = lh[0];
    /* :476-494 row table: right keypoint iR is a candidate for every row in [floor(y-r), ceil(y+r)] */
    int *rcnt = (int *)calloc((size_t)nRows + 1, sizeof(int));
    for (int iR = 0; iR < Nr; iR++) {
        const float kpY = kr[iR].y, r = 2.0f * e->scale[kr[iR].octave];
        const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < nRows) rcnt[yi]++;
    }
    int *rstart = (int *)malloc(sizeof(int) * ((size_t)nRows + 1));
    rstart[0] = 0;
    for (int y = 0; y < nRows; y++) rstart[y + 1] = rstart[y] + rcnt[y];
    int *ritems = (int *)malloc(sizeof(int) * (size_t)(rstart[nRows] ? rstart[nRows] : 1));
    memset(rcnt, 0, sizeof(int) * (size_t)nRows);
    for (int iR = 0; iR < Nr; iR++) {
        const float kpY = kr[iR].y, r = 2.0f * e->scale[kr[iR].octave];
        const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < nRows) ritems[rstart[yi] + rcnt[yi]++] = iR;
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;        /* :497-499 */
    int *vDistIdx = (int *)malloc(sizeof(int) * 2 * (size_t)(N ? N : 1));
    int nd = 0;
    for (int iL = 0; iL < N; iL++) {
        const oro_keypoint *kpL = &kl[iL];
        const int levelL = kpL->octave;
        const float vL = kpL->y, uL = kpL->x;
        const int row = (int)vL;
        if (row < 0 || row >= nRows) continue;
        const int c0 = rstart[row], c1 = rstart[row + 1];
        if (c0 == c1) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = 100;                                     /* TH_HIGH :519 */
        int bestIdxR = 0;
        for (int c = c0; c < c1; c++) {
            const int iR = ritems[c];
            const oro_keypoint *kpR = &kr[iR];
            if (kpR->octave < levelL - 1 || kpR->octave > levelL + 1) continue;
            const float uR = kpR->x;
            if (uR >= minU && uR <= maxU) {
                const int dist = oro_descriptor_distance(dl + (size_t)iL * 32, dr + (size_t)iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {                             /* :549 subpixel match by correlation */
            const float uR0 = kr[bestIdxR].x;
            const float scaleFactor = e->inv_scale[kpL->octave];
            const float scaleduL = roundf(kpL->x * scaleFactor);
            const float scaledvL = roundf(kpL->y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, Lw = 5;
            const int lv = kpL->octave, W = lw[lv], H = lh[lv];
            const uint8_t *IL = pyrL[lv], *IR = pyrR[lv];
            const int cvL = (int)scaledvL, cuL = (int)scaleduL, cuR = (int)scaleduR0;
            int bestD = INT_MAX, bestincR = 0;
            float vDists[11];
            const float iniu = scaleduR0 + Lw - w, endu = scaleduR0 + Lw + w + 1;
            if (iniu < 0 || endu >= W) continue;                /* :575-577 */
            const int cL = IL[(size_t)cvL * W + cuL];
            for (int incR = -Lw; incR <= Lw; incR++) {
                /* the reference reads through its reflect-101 bordered buffer; rows are always inside */
                const int cR = IR[(size_t)oro_reflect101(cvL, H) * W + oro_reflect101(cuR + incR, W)];
                int sum = 0;
                for (int dy = -w; dy <= w; dy++)
                    for (int dx = -w; dx <= w; dx++) {
                        const int a = IL[(size_t)oro_reflect101(cvL + dy, H) * W + oro_reflect101(cuL + dx, W)] - cL;
                        const int b = IR[(size_t)oro_reflect101(cvL + dy, H) * W + oro_reflect101(cuR + incR + dx, W)] - cR;
                        sum += abs(a - b);
                    }
                const float dist = (float)sum;                  /* cv::norm(IL,IR,NORM_L1): exact integers */
                if (dist < (float)bestD) { bestD = (int)dist; bestincR = incR; }
                vDists[Lw + incR] = dist;
            }
            if (bestincR == -Lw || bestincR == Lw) continue;
            const float dist1 = vDists[Lw + bestincR - 1], dist2 = vDists[Lw + bestincR], dist3 = vDists[Lw + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = e->scale[kpL->octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {
                    disparity = (float)0.01;
                    bestuR = (float)((double)uL - 0.01);
                }
                mvDepth[iL] = mbf / disparity;
                mvuRight[iL] = bestuR;
                vDistIdx[2 * nd] = bestD; vDistIdx[2 * nd + 1] = iL; nd++;
            }
        }
    }
    if (nd > 0) {                                               /* :627-639 */
        qsort(vDistIdx, (size_t)nd, 2 * sizeof(int), cmp_pair_ii);
        const float median = (float)vDistIdx[2 * (nd / 2)];
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nd - 1; i >= 0; i--) {
            if ((float)vDistIdx[2 * i] < thDist) break;
            mvuRight[vDistIdx[2 * i + 1]] = -1;
            mvDepth[vDistIdx[2 * i + 1]] = -1;
        }
    }
    free(rcnt); free(rstart); free(ritems); free(vDistIdx);
}

/* ------------------------------------------------------------------------------------------------
 * N2: DBoW2 transform (TemplatedVocabulary.h:1127-1259), BowVector.cpp:34-84, FeatureVector.cpp:31-45,
 * ScoringObject.cpp:23-68
 * ---------------------------------------------------------------------------------------------- */
void oro_voc_transform_features(const oro_voc *v, const uint8_t *feat, int n, int levelsup,
                                int32_t *word_id, int32_t *node_id, double *weight)
{
    const int nid_level = v->L - levelsup;
    for (int i = 0; i < n; i++) {
        int final_id = 0, current_level = 0, nid = 0;       /* nid_level <= 0 -> root (:1228) */
        do {
            ++current_level;
            const int c0 = v->child_off[final_id], c1 = v->child_off[final_id + 1];
            final_id = v->child_ids[c0];
            double best_d = (double)oro_descriptor_distance(feat + (size_t)i * 32, v->desc + (size_t)final_id * 32);
            for (int c = c0 + 1; c < c1; c++) {
                const int id = v->child_ids[c];
                const double d = (double)oro_descriptor_distance(feat + (size_t)i * 32, v->desc + (size_t)id * 32);
                if (d < best_d) { best_d = d; final_id = id; }
            }
            if (current_level == nid_level) nid = final_id;
        } while (v->child_off[final_id + 1] > v->child_off[final_id]);
        word_id[i] = v->word_of[final_id];
        node_id[i] = nid;
        weight[i] = v->weight[final_id];
    }
}

typedef struct { int32_t key; int32_t idx; } key_idx;
static int cmp_key_idx(const void *a, const void *b)
{
    const key_idx *x = (const key_idx *)a, *y = (const key_idx *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

int oro_voc_bow_vector(const oro_voc *v, const int32_t *word_id, const double *weight, int n, int32_t *ids, double *vals)
{
    const int must = v->scoring != 5;          /* DOT_PRODUCT is the only scoring that does not normalise */
    const int l2 = v->scoring == 1;
    key_idx *ki = (key_idx *)malloc(sizeof(key_idx) * (size_t)(n ? n : 1));
    int m = 0;
    for (int i = 0; i < n; i++) if (weight[i] > 0) { ki[m].key = word_id[i]; ki[m].idx = i; m++; }
    qsort(ki, (size_t)m, sizeof(key_idx), cmp_key_idx);
    int o = 0;
    for (int j = 0; j < m; j++) {
        const double w = weight[ki[j].idx];
        if (o > 0 && ids[o - 1] == ki[j].key) {
            if (v->weighting == 0 || v->weighting == 1) vals[o - 1] += w;   /* addWeight; addIfNotExist keeps the first */
        } else { ids[o] = ki[j].key; vals[o] = w; o++; }
    }
    if ((v->weighting == 0 || v->weighting == 1) && o > 0 && !must) {
        const double nd = (double)o;
        for (int j = 0; j < o; j++) vals[j] /= nd;
    }
    if (must) {
        double norm = 0.0;
        if (!l2) for (int j = 0; j < o; j++) norm += fabs(vals[j]);
        else { for (int j = 0; j < o; j++) norm += vals[j] * vals[j]; norm = sqrt(norm); }
        if (norm > 0.0) for (int j = 0; j < o; j++) vals[j] /= norm;
    }
    free(ki);
    return o;
}

int oro_voc_feature_vector(const int32_t *node_id, const double *weight, int n, int32_t *node_ids, int32_t *off, int32_t *idx)
{
    key_idx *ki = (key_idx *)malloc(sizeof(key_idx) * (size_t)(n ? n : 1));
    int m = 0;
    for (int i = 0; i < n; i++) if (weight[i] > 0) { ki[m].key = node_id[i]; ki[m].idx = i; m++; }
    qsort(ki, (size_t)m, sizeof(key_idx), cmp_key_idx);
    int o = 0;
    off[0] = 0;
    for (int j = 0; j < m; j++) {
        if (o == 0 || node_ids[o - 1] != ki[j].key) { node_ids[o] = ki[j].key; o++; off[o] = off[o - 1]; }
        idx[off[o]++] = ki[j].idx;
    }
    free(ki);
    return o;
}

double oro_voc_score_l1(const int32_t *ids1, const double *vals1, int n1, const int32_t *ids2, const double *vals2, int n2)
{
    double score = 0;
    int i = 0, j = 0;
    while (i < n1 && j < n2) {
        if (ids1[i] == ids2[j]) { score += fabs(vals1[i] - vals2[j]) - fabs(vals1[i]) - fabs(vals2[j]); i++; j++; }
        else if (ids1[i] < ids2[j]) i++;
        else j++;
    }
    return -score / 2.0;
}

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) src/ORBmatcher.cc:159-288, with MapPoint pointers
 * replaced by key-frame feature indices (match_f[i] = KF feature matched to frame feature i, -1 = NULL) and
 * `pMP && !pMP->isBad()` by valid_kf[].  FeatureVectors as (ascending node ids, CSR lists), the std::map's order. */
int oro_search_by_bow(const uint8_t *desc_kf, const float *angle_kf, int n_kf, const uint8_t *valid_kf,
                      const int32_t *kf_node, const int32_t *kf_off, const int32_t *kf_idx, int kf_n,
                      const uint8_t *desc_f, const float *angle_f, int n_f,
                      const int32_t *f_node, const int32_t *f_off, const int32_t *f_idx, int f_n,
                      float nnratio, int check_orientation, int32_t *match_f)
{
    int nmatches = 0;
    int *rot_items = (int *)malloc(sizeof(int) * (size_t)(n_f > 0 ? n_f : 1) * 2);   /* (bin, frame feature) in push order */
    int nrot = 0;
    int hist[30];
    (void)n_kf;
    for (int i = 0; i < 30; i++) hist[i] = 0;
    for (int i = 0; i < n_f; i++) match_f[i] = -1;                                   /* :163 */
    int a = 0, b = 0;
    while (a < kf_n && b < f_n) {                                                    /* :180 */
        if (kf_node[a] == f_node[b]) {
            for (int ik = kf_off[a]; ik < kf_off[a + 1]; ik++) {                      /* :187 */
                const int realIdxKF = kf_idx[ik];
                if (valid_kf && !valid_kf[realIdxKF]) continue;                      /* :193-197 */
                const uint8_t *dKF = desc_kf + (size_t)realIdxKF * 32;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int jf = f_off[b]; jf < f_off[b + 1]; jf++) {                    /* :205 */
                    const int realIdxF = f_idx[jf];
                    if (match_f[realIdxF] >= 0) continue;                            /* :209 */
                    const int dist = oro_descriptor_distance(dKF, desc_f + (size_t)realIdxF * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= 50) {                                               /* TH_LOW :228 */
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {             /* :230 */
                        match_f[bestIdxF] = realIdxKF;
                        if (check_orientation) {
                            const int bin = oro_rot_bin(angle_kf[realIdxKF], angle_f[bestIdxF]);
                            rot_items[2 * nrot] = bin; rot_items[2 * nrot + 1] = bestIdxF; nrot++;
                            hist[bin]++;
                        }
                        nmatches++;
                    }
                }
            }
            a++; b++;
        } else if (kf_node[a] < f_node[b]) {
            a++;                                                                     /* lower_bound(Fit->first) :253 */
        } else {
            b++;
        }
    }
    if (check_orientation) {                                                         /* :266-284 */
        int ind1, ind2, ind3;
        oro_three_maxima(hist, 30, &ind1, &ind2, &ind3);
        for (int k = 0; k < nrot; k++) {
            const int bin = rot_items[2 * k];
            if (bin == ind1 || bin == ind2 || bin == ind3) continue;
            match_f[rot_items[2 * k + 1]] = -1;
            nmatches--;
        }
    }
    free(rot_items);
    return nmatches;
}

/* ---- Frame::UndistortKeyPoints / ComputeImageBounds, src/Frame.cc:404-463 ----
 * cv::undistortPoints of OpenCV 3.1.0 (imgproc/src/undistort.cpp, cvUndistortPoints): camera matrix and distortion
 * coefficients widened to double, x0 = (u - cx) / fx, then five times x <- (x0 - deltaX) * icdist with
 * icdist = 1 / (1 + k1 r2 + k2 r2^2 + k3 r2^3) and the tangential deltas, finally u' = fx x + cx stored as float. */
void oro_undistort_points(float *xy, int n, float fx, float fy, float cx, float cy, const float dist[5])
{
    if (dist[0] == 0.0f) return;                       /* :406-410 */
    const double Fx = fx, Fy = fy, Cx = cx, Cy = cy;
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
    for (int i = 0; i < n; i++) {
        const double xn = ((double)xy[2 * i] - Cx) * (1.0 / Fx), yn = ((double)xy[2 * i + 1] - Cy) * (1.0 / Fy);
        double x = xn, y = yn;
        for (int it = 0; it < 5; it++) {
            const double r2 = x * x + y * y;
            const double icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
            const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
            const double dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
            x = (xn - dx) * icdist;
            y = (yn - dy) * icdist;
        }
        xy[2 * i] = (float)(Fx * x + Cx);
        xy[2 * i + 1] = (float)(Fy * y + Cy);
    }
}

void oro_image_bounds(int width, int height, float fx, float fy, float cx, float cy, const float dist[5], float bounds[4])
{
    float c[8] = {0.f, 0.f, (float)width, 0.f, 0.f, (float)height, (float)width, (float)height};
    if (dist[0] != 0.0f) {
        oro_undistort_points(c, 4, fx, fy, cx, cy, dist);
        bounds[0] = c[0] < c[4] ? c[0] : c[4]; bounds[1] = c[2] > c[6] ? c[2] : c[6];
        bounds[2] = c[1] < c[3] ? c[1] : c[3]; bounds[3] = c[5] > c[7] ? c[5] : c[7];
    } else {
        bounds[0] = 0.0f; bounds[1] = (float)width; bounds[2] = 0.0f; bounds[3] = (float)height;
    }
}

/* ---- ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:405-520 ---- */
int oro_search_for_initialization(const oro_keypoint *kps1, const uint8_t *desc1, int n1,
                                  const oro_grid *g2, const oro_keypoint *kps2, const uint8_t *desc2, int n2,
                                  float *prev_matched, int window_size, float nnratio, int check_orientation, int32_t *matches12)
{
    int nmatches = 0;
    int *matched_dist = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));       /* vMatchedDistance :415 */
    int *matches21 = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));          /* vnMatches21 :416 */
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    int *rot_bin = (int *)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1) * 2);         /* rotHist as (bin, i1) in push order */
    int nrot = 0, hist[30];
    for (int i = 0; i < 30; i++) hist[i] = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;                                    /* :408 */
    for (int i = 0; i < n2; i++) { matched_dist[i] = 2147483647; matches21[i] = -1; }
    for (int i1 = 0; i1 < n1; i1++) {                                                  /* :418 */
        const int level1 = kps1[i1].octave;
        if (level1 > 0) continue;                                                      /* :422 */
        const int nc = oro_features_in_area(g2, kps2, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window_size,
                                            level1, level1, cand, n2);                 /* :425 */
        if (nc <= 0) continue;
        const uint8_t *d1 = desc1 + (size_t)i1 * 32;
        int bestDist = 2147483647, bestDist2 = 2147483647, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {                                                 /* :436 */
            const int i2 = cand[c];
            const int dist = oro_descriptor_distance(d1, desc2 + (size_t)i2 * 32);
            if (matched_dist[i2] <= dist) continue;                                    /* :444 */
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= 50) {                                                          /* TH_LOW :459 */
            if ((float)bestDist < (float)bestDist2 * nnratio) {                        /* :461 */
                if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2;
                matches21[bestIdx2] = i1;
                matched_dist[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) {
                    const int bin = oro_rot_bin(kps1[i1].angle, kps2[bestIdx2].angle);
                    rot_bin[2 * nrot] = bin; rot_bin[2 * nrot + 1] = i1; nrot++;
                    hist[bin]++;                                                       /* rotHist[bin].size(), stale entries included */
                }
            }
        }
    }
    if (check_orientation) {                                                           /* :489-510 */
        int ind1, ind2, ind3;
        oro_three_maxima(hist, 30, &ind1, &ind2, &ind3);
        for (int k = 0; k < nrot; k++) {
            const int bin = rot_bin[2 * k], idx1 = rot_bin[2 * k + 1];
            if (bin == ind1 || bin == ind2 || bin == ind3) continue;
            if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)                                                    /* :513-516 */
        if (matches12[i1] >= 0) { prev_matched[2 * i1] = kps2[matches12[i1]].x; prev_matched[2 * i1 + 1] = kps2[matches12[i1]].y; }
    free(matched_dist); free(matches21); free(cand); free(rot_bin);
    return nmatches;
}

/* ---- ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono), src/ORBmatcher.cc:1328-1470 ---- */
/* OpenCV 3.1.0: cv::Mat algebra on 3x3 * 3x1 float matrices.  `R*x + t` is one cv::gemm(R, x, 1, t, 1, dst, 0) call (MatOp_GEMM), and
 * cv::gemm (modules/core/src/matmul.cpp) has a small-matrix path for `flags == 0 && 2 <= len && len <= 4 && (len == d_size.width ||
 * len == d_size.height)`: for CV_32F the three products are summed IN FLOAT, left to right, `float t0 = a[0]*b[0] + a[1]*b[b_step] +
 * a[2]*b[b_step*2]`, and the element is `(float)(t0*alpha + c[0]*beta)` with alpha, beta doubles.  Unary minus on a transpose
 * expression (`-Rcw.t()*tcw`) materialises the transpose (MatOp::subtract(Scalar, MatExpr) assigns the expression) and scales by
 * alpha = -1, so that product runs through the same path with flags == 0.  (Rounds 1-2 restated these products with double
 * accumulation -- GEMMSingleMul's arithmetic, which only non-small or transposed-flag products reach.) */
float oro_gemm_row(const float *T, int row, const float *x)      /* (R x + t)[row] of a row-major 4x4 [R|t] */
{
    const float t0 = T[4 * row] * x[0] + T[4 * row + 1] * x[1] + T[4 * row + 2] * x[2];
    return (float)((double)t0 * 1.0 + (double)T[4 * row + 3] * 1.0);
}
void oro_camera_center(const float *T, float Ow[3])                /* -R^T t: KeyFrame::SetPose's Ow (src/KeyFrame.cc), :303, :990, :1342, :1478 */
{
    for (int k = 0; k < 3; k++) {
        const float t0 = T[k] * T[3] + T[4 + k] * T[7] + T[8 + k] * T[11];
        Ow[k] = (float)((double)t0 * -1.0 + 0.0 * 0.0);       /* no C operand: c = zerof, beta = 0 (a zero sum comes out as +0) */
    }
}
int oro_search_by_projection_last(int n_last, const uint8_t *has_point, const float *xw, const uint8_t *mp_desc, const int32_t *mp_obs,
                                  const oro_keypoint *kps_last, const float *Tcw, const float *Tlw,
                                  float fx, float fy, float cx, float cy, float mb, float mbf, const float bounds[4],
                                  const float *scale_factors, const oro_grid *g, const oro_keypoint *kps_cur, const uint8_t *desc_cur,
                                  const float *u_right, int n_cur, float th, int mono, int check_orientation,
                                  int32_t *cur_obs, int32_t *cur_match)
{
    int nmatches = 0, nrot = 0, cap_rot = n_last > 0 ? n_last : 1, hist[30];
    int *rot = (int *)malloc(sizeof(int) * 2 * (size_t)cap_rot);      /* rotHist as (bin, bestIdx2) in push order */
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_cur > 0 ? n_cur : 1));
    for (int i = 0; i < 30; i++) hist[i] = 0;
    for (int i = 0; i < n_cur; i++) cur_match[i] = -1;
    /* twc = -Rcw^T tcw (:1342); tlc = Rlw twc + tlw (:1347): only the z component is used */
    float twc[3], tlc2;
    oro_camera_center(Tcw, twc);
    tlc2 = oro_gemm_row(Tlw, 2, twc);
    const int forward = tlc2 > mb && !mono, backward = -tlc2 > mb && !mono;     /* :1349-1350 */
    for (int i = 0; i < n_last; i++) {
        if (!has_point[i]) continue;                                   /* :1356-1358 */
        const float *X = xw + 3 * (size_t)i;
        const float xc = oro_gemm_row(Tcw, 0, X), yc = oro_gemm_row(Tcw, 1, X), zc = oro_gemm_row(Tcw, 2, X);
        const float invzc = (float)(1.0 / zc);                         /* :1366 */
        if (invzc < 0) continue;
        const float u = fx * xc * invzc + cx, v = fy * yc * invzc + cy;
        if (u < bounds[0] || u > bounds[1]) continue;
        if (v < bounds[2] || v > bounds[3]) continue;
        const int nLastOctave = kps_last[i].octave;
        const float radius = th * scale_factors[nLastOctave];          /* :1381 */
        int nc;
        if (forward) nc = oro_features_in_area(g, kps_cur, u, v, radius, nLastOctave, -1, cand, n_cur);
        else if (backward) nc = oro_features_in_area(g, kps_cur, u, v, radius, 0, nLastOctave, cand, n_cur);
        else nc = oro_features_in_area(g, kps_cur, u, v, radius, nLastOctave - 1, nLastOctave + 1, cand, n_cur);
        if (nc <= 0) continue;
        const uint8_t *dMP = mp_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {                                 /* :1400 */
            const int i2 = cand[c];
            if (cur_obs[i2] > 0) continue;                             /* a point with observations keeps its place :1403-1405 */
            if (u_right && u_right[i2] > 0) {                          /* :1407-1413 */
                const float ur = u - mbf * invzc;
                const float er = fabsf(ur - u_right[i2]);
                if (er > radius) continue;
            }
            const int dist = oro_descriptor_distance(dMP, desc_cur + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= 100) {                                         /* TH_HIGH :1426 */
            cur_obs[bestIdx2] = mp_obs[i];
            cur_match[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                const int bin = oro_rot_bin(kps_last[i].angle, kps_cur[bestIdx2].angle);
                if (nrot == cap_rot) { cap_rot *= 2; rot = (int *)realloc(rot, sizeof(int) * 2 * (size_t)cap_rot); }
                rot[2 * nrot] = bin; rot[2 * nrot + 1] = bestIdx2; nrot++;
                hist[bin]++;
            }
        }
    }
    if (check_orientation) {                                           /* :1447-1466: one decrement per entry, even for a repeated i2 */
        int ind1, ind2, ind3;
        oro_three_maxima(hist, 30, &ind1, &ind2, &ind3);
        for (int k = 0; k < nrot; k++) {
            const int bin = rot[2 * k];
            if (bin != ind1 && bin != ind2 && bin != ind3) { cur_obs[rot[2 * k + 1]] = -1; cur_match[rot[2 * k + 1]] = -1; nmatches--; }
        }
    }
    free(rot); free(cand);
    return nmatches;
}

/* ---- ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), src/ORBmatcher.cc:45-125 (+ RadiusByViewingCos :127-133) ---- */
int oro_search_by_projection_map(int n_mp, const uint8_t *in_view, const float *proj_x, const float *proj_y, const float *proj_xr,
                                 const int32_t *pred_level, const float *view_cos, const uint8_t *mp_desc, const int32_t *mp_obs,
                                 const float *scale_factors, const oro_grid *g, const oro_keypoint *kps_cur, const uint8_t *desc_cur,
                                 const float *u_right, int n_cur, float th, float nnratio, int32_t *cur_obs, int32_t *cur_match)
{
    int nmatches = 0;
    const int bFactor = th != 1.0;                                     /* :49 */
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_cur > 0 ? n_cur : 1));
    for (int i = 0; i < n_cur; i++) cur_match[i] = -1;
    for (int iMP = 0; iMP < n_mp; iMP++) {
        if (!in_view[iMP]) continue;                                   /* :54-58 */
        const int nPredictedLevel = pred_level[iMP];
        float r = view_cos[iMP] > 0.998 ? 2.5f : 4.0f;                  /* RadiusByViewingCos (double 0.998 against a float) */
        if (bFactor) r *= th;
        const float rad = r * scale_factors[nPredictedLevel];
        const int nc = oro_features_in_area(g, kps_cur, proj_x[iMP], proj_y[iMP], rad, nPredictedLevel - 1, nPredictedLevel, cand, n_cur);
        if (nc <= 0) continue;
        const uint8_t *dMP = mp_desc + (size_t)iMP * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            if (cur_obs[idx] > 0) continue;                            /* :85-87 */
            if (u_right && u_right[idx] > 0) {                         /* :89-94 */
                const float er = fabsf(proj_xr[iMP] - u_right[idx]);
                if (er > rad) continue;
            }
            const int dist = oro_descriptor_distance(dMP, desc_cur + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kps_cur[idx].octave; bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = kps_cur[idx].octave; bestDist2 = dist;
            }
        }
        if (bestDist <= 100) {                                         /* TH_HIGH :115 */
            if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
            cur_obs[bestIdx] = mp_obs[iMP];
            cur_match[bestIdx] = iMP;
            nmatches++;
        }
    }
    free(cand);
    return nmatches;
}

/* ---- ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist),
 * src/ORBmatcher.cc:1472-1599, Tracking::Relocalization's second-chance matcher (src/Tracking.cc:1459,1473).
 * Per key-frame feature i: usable[i] = vpMPs[i] && !isBad() && !sAlreadyFound.count(pMP); xw = GetWorldPos(); min_dist_inv /
 * max_dist_inv = GetMin/MaxDistanceInvariance(); mf_max_distance = the MapPoint's mfMaxDistance (PredictScale, src/MapPoint.cc:402-417);
 * mp_desc = GetDescriptor(); kf_angle[i] = pKF->mvKeysUn[i].angle.  Current frame: Tcw row-major 4x4, grid g of kps_cur (mvKeysUn),
 * log_scale_factor = mfLogScaleFactor, cur_has_point[i2] = (mvpMapPoints[i2] != NULL) in/out, cur_match[i2] out = i or -1.
 * cv::Mat algebra as cv::gemm's small-matrix path (oro_gemm_row: float accumulation, then alpha / beta in double); cv::norm of a float vector accumulates in double;
 * log() of a float resolves to logf in the reference's translation unit (decision). */
int oro_predict_scale(float mf_max_distance, float current_dist, float log_scale_factor, int n_levels)
{
    const float ratio = mf_max_distance / current_dist;                 /* src/MapPoint.cc:407 */
    int nScale = (int)ceilf(logf(ratio) / log_scale_factor);           /* :410 */
    if (nScale < 0) nScale = 0;
    else if (nScale >= n_levels) nScale = n_levels - 1;
    return nScale;
}
int oro_search_by_projection_kf(int n_kf, const uint8_t *usable, const float *xw, const float *min_dist_inv, const float *max_dist_inv,
                                const float *mf_max_distance, const uint8_t *mp_desc, const float *kf_angle, const float *Tcw,
                                float fx, float fy, float cx, float cy, const float bounds[4], const float *scale_factors, int nlevels,
                                float log_scale_factor, const oro_grid *g, const oro_keypoint *kps_cur, const uint8_t *desc_cur, int n_cur,
                                float th, int orb_dist, int check_orientation, uint8_t *cur_has_point, int32_t *cur_match)
{
    int nmatches = 0, nrot = 0, hist[30];
    int *rot = (int *)malloc(sizeof(int) * 2 * (size_t)(n_kf > 0 ? n_kf : 1));
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_cur > 0 ? n_cur : 1));
    for (int i = 0; i < 30; i++) hist[i] = 0;
    for (int i = 0; i < n_cur; i++) cur_match[i] = -1;
    float Ow[3];                                                        /* Ow = -Rcw^T tcw, :1478 */
    oro_camera_center(Tcw, Ow);
    for (int i = 0; i < n_kf; i++) {
        if (!usable[i]) continue;                                      /* :1492-1494 */
        const float *X = xw + 3 * (size_t)i;
        const float xc = oro_gemm_row(Tcw, 0, X), yc = oro_gemm_row(Tcw, 1, X), zc = oro_gemm_row(Tcw, 2, X);
        const float invzc = (float)(1.0 / zc);                         /* :1502 (no sign check in this variant) */
        const float u = fx * xc * invzc + cx, v = fy * yc * invzc + cy;
        if (u < bounds[0] || u > bounds[1]) continue;
        if (v < bounds[2] || v > bounds[3]) continue;
        double nn = 0;                                                  /* cv::norm(x3Dw - Ow), :1513-1514 */
        for (int k = 0; k < 3; k++) { const float po = X[k] - Ow[k]; nn += (double)po * (double)po; }
        const float dist3D = (float)sqrt(nn);
        if (dist3D < min_dist_inv[i] || dist3D > max_dist_inv[i]) continue;    /* :1520 */
        const int nPredictedLevel = oro_predict_scale(mf_max_distance[i], dist3D, log_scale_factor, nlevels);
        const float radius = th * scale_factors[nPredictedLevel];      /* :1526 */
        const int nc = oro_features_in_area(g, kps_cur, u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1, cand, n_cur);
        if (nc <= 0) continue;
        const uint8_t *dMP = mp_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {                                 /* :1538-1554 */
            const int i2 = cand[c];
            if (cur_has_point[i2]) continue;
            const int dist = oro_descriptor_distance(dMP, desc_cur + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= orb_dist && bestIdx2 >= 0) {                   /* :1556; with ORBdist >= 256 the reference would index [-1] */
            cur_has_point[bestIdx2] = 1;
            cur_match[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                const int bin = oro_rot_bin(kf_angle[i], kps_cur[bestIdx2].angle);
                rot[2 * nrot] = bin; rot[2 * nrot + 1] = bestIdx2; nrot++;
                hist[bin]++;
            }
        }
    }
    if (check_orientation) {                                           /* :1577-1596 */
        int ind1, ind2, ind3;
        oro_three_maxima(hist, 30, &ind1, &ind2, &ind3);
        for (int k = 0; k < nrot; k++) {
            const int bin = rot[2 * k];
            if (bin != ind1 && bin != ind2 && bin != ind3) { cur_has_point[rot[2 * k + 1]] = 0; cur_match[rot[2 * k + 1]] = -1; nmatches--; }
        }
    }
    free(rot); free(cand);
    return nmatches;
}
