// orbp.cc -- host-side pose solvers behind include/orbp.h (EPnP RANSAC + motion-only pose optimisation).
// Plain C++, fp64, no device code: SURVEY keeps these "tiny dense solves" on the host.  Each function names the
// reference lines it follows; the linear algebra the reference borrows from OpenCV / Eigen is written here.
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/orbp.h"
#include "../../include/orbx.h"

static thread_local char g_perr[256] = "";
static int pfail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_perr, sizeof g_perr, fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char *orbp_last_error(void) { return g_perr; }

// -------------------------------------------------------------------------------------------------
// Small dense linear algebra (row-major)
// -------------------------------------------------------------------------------------------------
// One-sided (Hestenes) Jacobi SVD of A (m x n, m >= n): A = U diag(W) V^T, W descending, U m x n, V n x n.
// Columns of U that belong to a zero singular value are left zero; V is always orthonormal.
static void jacobi_svd(const double *A, int m, int n, double *U, double *W, double *V)
{
    std::vector<double> B(A, A + (size_t)m * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = i == j;
    for (int sweep = 0; sweep < 60; sweep++) {
        bool rotated = false;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double a = 0, b = 0, g = 0;
                for (int i = 0; i < m; i++) {
                    const double x = B[i * n + p], y = B[i * n + q];
                    a += x * x; b += y * y; g += x * y;
                }
                if (std::fabs(g) <= DBL_EPSILON * std::sqrt(a * b) || g == 0.0) continue;
                rotated = true;
                const double zeta = (b - a) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < m; i++) {
                    const double x = B[i * n + p], y = B[i * n + q];
                    B[i * n + p] = c * x - s * y;
                    B[i * n + q] = s * x + c * y;
                }
                for (int i = 0; i < n; i++) {
                    const double x = V[i * n + p], y = V[i * n + q];
                    V[i * n + p] = c * x - s * y;
                    V[i * n + q] = s * x + c * y;
                }
            }
        if (!rotated) break;
    }
    std::vector<double> w(n);
    std::vector<int> order(n);
    for (int j = 0; j < n; j++) {
        double s = 0;
        for (int i = 0; i < m; i++) s += B[i * n + j] * B[i * n + j];
        w[j] = std::sqrt(s);
        order[j] = j;
    }
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return w[x] > w[y]; });
    std::vector<double> Vs((size_t)n * n);
    for (int k = 0; k < n; k++) {
        const int j = order[k];
        W[k] = w[j];
        for (int i = 0; i < m; i++) U[i * n + k] = w[j] > 0 ? B[i * n + j] / w[j] : 0.0;
        for (int i = 0; i < n; i++) Vs[i * n + k] = V[i * n + j];
    }
    memcpy(V, Vs.data(), sizeof(double) * n * n);
}

// x = pinv(A) b through the SVD, singular values below 2 eps sum(w) dropped (cvSolve / cvInvert with CV_SVD).
static void svd_solve(const double *A, int m, int n, const double *b, int nrhs, double *x)
{
    std::vector<double> U((size_t)m * n), W(n), V((size_t)n * n);
    jacobi_svd(A, m, n, U.data(), W.data(), V.data());
    double thr = 0;
    for (int i = 0; i < n; i++) thr += W[i];
    thr *= 2 * DBL_EPSILON;
    for (int r = 0; r < nrhs; r++) {
        for (int i = 0; i < n; i++) x[i * nrhs + r] = 0;
        for (int k = 0; k < n; k++) {
            if (!(W[k] > thr)) continue;
            double s = 0;
            for (int i = 0; i < m; i++) s += U[i * n + k] * b[i * nrhs + r];
            s /= W[k];
            for (int i = 0; i < n; i++) x[i * nrhs + r] += V[i * n + k] * s;
        }
    }
}

static inline double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline double dist2_3(const double *a, const double *b)
{
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

// Householder QR least squares for the 6 x 4 Gauss-Newton system (PnPsolver.cc:878-957: scaled columns, R's diagonal
// kept aside).  Returns false on a zero column, which leaves x untouched like the reference's early return.
static bool qr_solve_6x4(double *A, double *b, double *x)
{
    const int nr = 6, nc = 4;
    double a1[4], a2[4];
    for (int k = 0; k < nc; k++) {
        double eta = 0;
        for (int i = k; i < nr; i++) eta = std::max(eta, std::fabs(A[i * nc + k]));
        if (eta == 0) return false;
        double sum = 0;
        const double inv_eta = 1. / eta;
        for (int i = k; i < nr; i++) { A[i * nc + k] *= inv_eta; sum += A[i * nc + k] * A[i * nc + k]; }
        double sigma = std::sqrt(sum);
        if (A[k * nc + k] < 0) sigma = -sigma;
        A[k * nc + k] += sigma;
        a1[k] = sigma * A[k * nc + k];
        a2[k] = -eta * sigma;
        for (int j = k + 1; j < nc; j++) {
            double s = 0;
            for (int i = k; i < nr; i++) s += A[i * nc + k] * A[i * nc + j];
            const double tau = s / a1[k];
            for (int i = k; i < nr; i++) A[i * nc + j] -= tau * A[i * nc + k];
        }
    }
    for (int j = 0; j < nc; j++) {          // b <- Q^T b
        double tau = 0;
        for (int i = j; i < nr; i++) tau += A[i * nc + j] * b[i];
        tau /= a1[j];
        for (int i = j; i < nr; i++) b[i] -= tau * A[i * nc + j];
    }
    x[nc - 1] = b[nc - 1] / a2[nc - 1];     // back substitution
    for (int i = nc - 2; i >= 0; i--) {
        double s = 0;
        for (int j = i + 1; j < nc; j++) s += A[i * nc + j] * x[j];
        x[i] = (b[i] - s) / a2[i];
    }
    return true;
}

// -------------------------------------------------------------------------------------------------
// EPnP (PnPsolver.cc:346-876)
// -------------------------------------------------------------------------------------------------
namespace {
struct Epnp {
    double fu, fv, uc, vc;
    int n = 0;
    std::vector<double> pws, us, alphas, pcs;
    double cws[4][3], ccs[4][3];

    void reset() { n = 0; pws.clear(); us.clear(); }
    void add(double X, double Y, double Z, double u, double v)
    {
        pws.push_back(X); pws.push_back(Y); pws.push_back(Z);
        us.push_back(u); us.push_back(v);
        n++;
    }

    void choose_control_points()             // :346-384
    {
        for (int j = 0; j < 3; j++) cws[0][j] = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
        for (int j = 0; j < 3; j++) cws[0][j] /= n;
        double c[9] = {0};
        for (int i = 0; i < n; i++) {
            double d[3];
            for (int j = 0; j < 3; j++) d[j] = pws[3 * i + j] - cws[0][j];
            for (int r = 0; r < 3; r++)
                for (int s = 0; s < 3; s++) c[3 * r + s] += d[r] * d[s];
        }
        double U[9], W[3], V[9];
        jacobi_svd(c, 3, 3, U, W, V);        // symmetric PSD: V holds the principal axes
        for (int i = 1; i < 4; i++) {
            // The sign of a principal axis is the SVD routine's choice (OpenCV's in the reference) and, with noisy
            // points, changes which local minimum the betas reach.  Fixed here: largest component positive.
            int big = 0;
            for (int j = 1; j < 3; j++)
                if (std::fabs(V[3 * j + (i - 1)]) > std::fabs(V[3 * big + (i - 1)])) big = j;
            const double sgn = V[3 * big + (i - 1)] < 0 ? -1.0 : 1.0;
            const double k = sgn * std::sqrt(W[i - 1] / n);
            for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * V[3 * j + (i - 1)];
        }
    }

    void compute_barycentric_coordinates()   // :386-412
    {
        double cc[9], ci[9], eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int i = 0; i < 3; i++)
            for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
        svd_solve(cc, 3, 3, eye, 3, ci);     // pseudo-inverse (cvInvert CV_SVD)
        alphas.resize((size_t)4 * n);
        for (int i = 0; i < n; i++) {
            const double *pi = &pws[3 * i];
            double *a = &alphas[4 * i];
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) + ci[3 * j + 2] * (pi[2] - cws[0][2]);
            a[0] = 1.0f - a[1] - a[2] - a[3];
        }
    }

    void compute_ccs(const double *betas, const double *vt)    // :431-444; vt row r = r-th singular vector
    {
        for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0;
        for (int i = 0; i < 4; i++) {
            const double *v = vt + 12 * (11 - i);
            for (int j = 0; j < 4; j++)
                for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
        }
    }

    void compute_pcs()                        // :446-456
    {
        pcs.resize((size_t)3 * n);
        for (int i = 0; i < n; i++) {
            const double *a = &alphas[4 * i];
            for (int j = 0; j < 3; j++)
                pcs[3 * i + j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
        }
    }

    void solve_for_sign()                     // :610-624
    {
        if (pcs[2] < 0.0) {
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
            for (size_t i = 0; i < pcs.size(); i++) pcs[i] = -pcs[i];
        }
    }

    double reprojection_error(const double R[3][3], const double t[3]) const     // :528-546
    {
        double sum = 0.0;
        for (int i = 0; i < n; i++) {
            const double *pw = &pws[3 * i];
            const double Xc = dot3(R[0], pw) + t[0], Yc = dot3(R[1], pw) + t[1];
            const double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
            const double ue = uc + fu * Xc * inv_Zc, ve = vc + fv * Yc * inv_Zc;
            const double u = us[2 * i], v = us[2 * i + 1];
            sum += std::sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
        }
        return sum / n;
    }

    void estimate_R_and_t(double R[3][3], double t[3]) const     // :548-601 (absolute orientation, Arun et al.)
    {
        double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++) { pc0[j] += pcs[3 * i + j]; pw0[j] += pws[3 * i + j]; }
        for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
        double abt[9] = {0};
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) abt[3 * j + k] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i + k] - pw0[k]);
        double U[9], W[3], V[9];
        jacobi_svd(abt, 3, 3, U, W, V);
        if (!(W[2] > 2 * DBL_EPSILON * (W[0] + W[1] + W[2]))) {
            // rank-deficient (coplanar points): complete U to an orthonormal basis.  In OpenCV's Jacobi SVD the third
            // left vector is what rounding noise leaves after orthogonalisation against the other two, i.e. +-(u1 x u2)
            // with a sign nobody chose; the reference is implementation-defined here.
            const double a[3] = {U[0], U[3], U[6]}, b[3] = {U[1], U[4], U[7]};
            U[2] = a[1] * b[2] - a[2] * b[1]; U[5] = a[2] * b[0] - a[0] * b[2]; U[8] = a[0] * b[1] - a[1] * b[0];
        }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) R[i][j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
        const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                           R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
        if (det < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }
        for (int j = 0; j < 3; j++) t[j] = pc0[j] - dot3(R[j], pw0);
    }

    double compute_R_and_t(const double *vt, const double *betas, double R[3][3], double t[3])     // :626-637
    {
        compute_ccs(betas, vt);
        compute_pcs();
        solve_for_sign();
        estimate_R_and_t(R, t);
        return reprojection_error(R, t);
    }

    // betas10 = [B11 B12 B22 B13 B23 B33 B14 B24 B34 B44]; the three linearisations :639-757
    static void sub_solve(const double *L, const double *rho, const int *cols, int nc, double *out)
    {
        double A[6 * 5];
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < nc; j++) A[i * nc + j] = L[10 * i + cols[j]];
        svd_solve(A, 6, nc, rho, 1, out);
    }
    static void find_betas_approx_1(const double *L, const double *rho, double *betas)
    {
        const int cols[4] = {0, 1, 3, 6};
        double b4[4];
        sub_solve(L, rho, cols, 4, b4);
        if (b4[0] < 0) {
            betas[0] = std::sqrt(-b4[0]);
            betas[1] = -b4[1] / betas[0]; betas[2] = -b4[2] / betas[0]; betas[3] = -b4[3] / betas[0];
        } else {
            betas[0] = std::sqrt(b4[0]);
            betas[1] = b4[1] / betas[0]; betas[2] = b4[2] / betas[0]; betas[3] = b4[3] / betas[0];
        }
    }
    static void find_betas_approx_2(const double *L, const double *rho, double *betas)
    {
        const int cols[3] = {0, 1, 2};
        double b3[3];
        sub_solve(L, rho, cols, 3, b3);
        if (b3[0] < 0) {
            betas[0] = std::sqrt(-b3[0]);
            betas[1] = (b3[2] < 0) ? std::sqrt(-b3[2]) : 0.0;
        } else {
            betas[0] = std::sqrt(b3[0]);
            betas[1] = (b3[2] > 0) ? std::sqrt(b3[2]) : 0.0;
        }
        if (b3[1] < 0) betas[0] = -betas[0];
        betas[2] = 0.0; betas[3] = 0.0;
    }
    static void find_betas_approx_3(const double *L, const double *rho, double *betas)
    {
        const int cols[5] = {0, 1, 2, 3, 4};
        double b5[5];
        sub_solve(L, rho, cols, 5, b5);
        if (b5[0] < 0) {
            betas[0] = std::sqrt(-b5[0]);
            betas[1] = (b5[2] < 0) ? std::sqrt(-b5[2]) : 0.0;
        } else {
            betas[0] = std::sqrt(b5[0]);
            betas[1] = (b5[2] > 0) ? std::sqrt(b5[2]) : 0.0;
        }
        if (b5[1] < 0) betas[0] = -betas[0];
        betas[2] = b5[3] / betas[0];
        betas[3] = 0.0;
    }

    static void compute_L_6x10(const double *vt, double *L)      // :760-801
    {
        const double *v[4] = {vt + 12 * 11, vt + 12 * 10, vt + 12 * 9, vt + 12 * 8};
        double dv[4][6][3];
        for (int i = 0; i < 4; i++) {
            int a = 0, b = 1;
            for (int j = 0; j < 6; j++) {
                for (int k = 0; k < 3; k++) dv[i][j][k] = v[i][3 * a + k] - v[i][3 * b + k];
                if (++b > 3) { a++; b = a + 1; }
            }
        }
        for (int i = 0; i < 6; i++) {
            double *row = L + 10 * i;
            row[0] = dot3(dv[0][i], dv[0][i]);
            row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
            row[2] = dot3(dv[1][i], dv[1][i]);
            row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
            row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
            row[5] = dot3(dv[2][i], dv[2][i]);
            row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
            row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
            row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
            row[9] = dot3(dv[3][i], dv[3][i]);
        }
    }

    static void gauss_newton(const double *L, const double *rho, double *be)      // :813-876
    {
        for (int k = 0; k < 5; k++) {
            double A[24], b[6], x[4] = {0, 0, 0, 0};
            for (int i = 0; i < 6; i++) {
                const double *r = L + 10 * i;
                double *a = A + 4 * i;
                a[0] = 2 * r[0] * be[0] + r[1] * be[1] + r[3] * be[2] + r[6] * be[3];
                a[1] = r[1] * be[0] + 2 * r[2] * be[1] + r[4] * be[2] + r[7] * be[3];
                a[2] = r[3] * be[0] + r[4] * be[1] + 2 * r[5] * be[2] + r[8] * be[3];
                a[3] = r[6] * be[0] + r[7] * be[1] + r[8] * be[2] + 2 * r[9] * be[3];
                b[i] = rho[i] - (r[0] * be[0] * be[0] + r[1] * be[0] * be[1] + r[2] * be[1] * be[1] + r[3] * be[0] * be[2] +
                                 r[4] * be[1] * be[2] + r[5] * be[2] * be[2] + r[6] * be[0] * be[3] + r[7] * be[1] * be[3] +
                                 r[8] * be[2] * be[3] + r[9] * be[3] * be[3]);
            }
            if (!qr_solve_6x4(A, b, x)) continue;    // singular column: the reference leaves X as it was
            for (int i = 0; i < 4; i++) be[i] += x[i];
        }
    }

    double compute_pose(double R[3][3], double t[3])       // :458-508
    {
        choose_control_points();
        compute_barycentric_coordinates();
        double mtm[144] = {0};
        for (int i = 0; i < n; i++) {        // rows of M (:414-429) accumulated straight into M^T M
            const double *as = &alphas[4 * i];
            const double u = us[2 * i], v = us[2 * i + 1];
            double m1[12], m2[12];
            for (int j = 0; j < 4; j++) {
                m1[3 * j] = as[j] * fu; m1[3 * j + 1] = 0.0; m1[3 * j + 2] = as[j] * (uc - u);
                m2[3 * j] = 0.0; m2[3 * j + 1] = as[j] * fv; m2[3 * j + 2] = as[j] * (vc - v);
            }
            for (int r = 0; r < 12; r++)
                for (int c = 0; c < 12; c++) mtm[12 * r + c] += m1[r] * m1[c] + m2[r] * m2[c];
        }
        double U[144], W[12], V[144], vt[144];
        jacobi_svd(mtm, 12, 12, U, W, V);    // symmetric PSD: eigenvectors = columns of V, eigenvalues descending
        for (int r = 0; r < 12; r++)
            for (int c = 0; c < 12; c++) vt[12 * r + c] = V[12 * c + r];
        double L[60], rho[6];
        compute_L_6x10(vt, L);
        rho[0] = dist2_3(cws[0], cws[1]); rho[1] = dist2_3(cws[0], cws[2]); rho[2] = dist2_3(cws[0], cws[3]);
        rho[3] = dist2_3(cws[1], cws[2]); rho[4] = dist2_3(cws[1], cws[3]); rho[5] = dist2_3(cws[2], cws[3]);
        double Betas[4][4], err[4], Rs[4][3][3], ts[4][3];
        find_betas_approx_1(L, rho, Betas[1]); gauss_newton(L, rho, Betas[1]); err[1] = compute_R_and_t(vt, Betas[1], Rs[1], ts[1]);
        find_betas_approx_2(L, rho, Betas[2]); gauss_newton(L, rho, Betas[2]); err[2] = compute_R_and_t(vt, Betas[2], Rs[2], ts[2]);
        find_betas_approx_3(L, rho, Betas[3]); gauss_newton(L, rho, Betas[3]); err[3] = compute_R_and_t(vt, Betas[3], Rs[3], ts[3]);
        int N = 1;
        if (err[2] < err[1]) N = 2;
        if (err[3] < err[N]) N = 3;
        memcpy(R, Rs[N], sizeof(double) * 9);
        memcpy(t, ts[N], sizeof(double) * 3);
        return err[N];
    }
};
}   // namespace

extern "C" double orbp_epnp(int n, const double *pws, const double *us, double fu, double fv, double uc, double vc,
                            double *R, double *t)
{
    if (n < 4 || !pws || !us || !R || !t) { pfail(ORBX_E_INVALID, "orbp_epnp: n >= 4 and non-NULL buffers"); return -1.0; }
    Epnp e;
    e.fu = fu; e.fv = fv; e.uc = uc; e.vc = vc;
    for (int i = 0; i < n; i++) e.add(pws[3 * i], pws[3 * i + 1], pws[3 * i + 2], us[2 * i], us[2 * i + 1]);
    double Rm[3][3];
    const double err = e.compute_pose(Rm, t);
    memcpy(R, Rm, sizeof Rm);
    return err;
}

// -------------------------------------------------------------------------------------------------
// RANSAC driver (PnPsolver.cc:66-344)
// -------------------------------------------------------------------------------------------------
static int libc_rand(void *) { return rand(); }

struct orbp_pnp {
    Epnp e;
    int N = 0;
    std::vector<float> p2d, sigma2, p3d, max_error;
    std::vector<uint8_t> inl_i, inl_best, inl_refined;
    double Ri[3][3], ti[3];
    double prob = 0.99; int min_inliers = 8, max_its = 300, min_set = 4; float eps = 0.4f;
    int n_inl_i = 0, n_iter = 0, n_best = 0, n_refined = 0;
    float best_T[16], refined_T[16];
    int (*rnd)(void *) = libc_rand; void *rnd_ctx = nullptr; double rnd_max = RAND_MAX;

    int random_int(int lo, int hi)            // DUtils::Random::RandomInt, Random.cpp:47-50
    {
        const int d = hi - lo + 1;
        return int(((double)rnd(rnd_ctx) / (rnd_max + 1.0)) * d) + lo;
    }
    void pose_to_T(float *T) const            // :206-212
    {
        for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1.f : 0.f;
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) T[4 * i + j] = (float)Ri[i][j];
            T[4 * i + 3] = (float)ti[i];
        }
    }
    void check_inliers()                      // :312-344, with its float/double mix
    {
        n_inl_i = 0;
        for (int i = 0; i < N; i++) {
            const float X = p3d[3 * i], Y = p3d[3 * i + 1], Z = p3d[3 * i + 2];
            const float Xc = (float)(Ri[0][0] * X + Ri[0][1] * Y + Ri[0][2] * Z + ti[0]);
            const float Yc = (float)(Ri[1][0] * X + Ri[1][1] * Y + Ri[1][2] * Z + ti[1]);
            const float invZc = (float)(1 / (Ri[2][0] * X + Ri[2][1] * Y + Ri[2][2] * Z + ti[2]));
            const double ue = e.uc + e.fu * Xc * invZc;
            const double ve = e.vc + e.fv * Yc * invZc;
            const float dx = (float)(p2d[2 * i] - ue), dy = (float)(p2d[2 * i + 1] - ve);
            const float error2 = dx * dx + dy * dy;
            inl_i[i] = error2 < max_error[i];
            n_inl_i += inl_i[i];
        }
    }
    void add_corr(int idx) { e.add(p3d[3 * idx], p3d[3 * idx + 1], p3d[3 * idx + 2], p2d[2 * idx], p2d[2 * idx + 1]); }
    bool refine()                             // :260-309
    {
        e.reset();
        for (int i = 0; i < N; i++)
            if (inl_best[i]) add_corr(i);
        e.compute_pose(Ri, ti);
        check_inliers();
        n_refined = n_inl_i;
        inl_refined = inl_i;
        if (n_inl_i > min_inliers) { pose_to_T(refined_T); return true; }
        return false;
    }
};

extern "C" int orbp_pnp_set_ransac_parameters(orbp_pnp *s, double probability, int min_inliers, int max_iterations,
                                              int min_set, float epsilon, float th2)
{
    if (!s) return pfail(ORBX_E_INVALID, "solver is NULL");
    if (min_set < 4) return pfail(ORBX_E_INVALID, "min_set %d < 4: EPnP needs four points", min_set);
    s->prob = probability; s->min_inliers = min_inliers; s->max_its = max_iterations; s->eps = epsilon; s->min_set = min_set;
    const int N = s->N;                        // :128-157
    int nmin = (int)(N * s->eps);
    if (nmin < s->min_inliers) nmin = s->min_inliers;
    if (nmin < min_set) nmin = min_set;
    s->min_inliers = nmin;
    if (s->eps < (float)s->min_inliers / N) s->eps = (float)s->min_inliers / N;
    int nit;
    if (s->min_inliers == N) nit = 1;
    else {
        // double -> int of a NaN / infinite / out-of-range quotient (epsilon > 1 when N < minInliers, epsilon ~ 0) is what
        // x86's cvttsd2si makes of it in the reference build: INT_MIN, which the clamp below turns into one iteration
        const double v = std::ceil(std::log(1 - s->prob) / std::log(1 - std::pow((double)s->eps, 3)));
        nit = (v >= -2147483648.0 && v <= 2147483647.0) ? (int)v : INT_MIN;
    }
    s->max_its = std::max(1, std::min(nit, s->max_its));
    for (int i = 0; i < N; i++) s->max_error[i] = s->sigma2[i] * th2;
    return 0;
}

extern "C" int orbp_pnp_create(orbp_pnp **out, int n, const float *p2d, const float *sigma2, const float *p3d,
                               float fx, float fy, float cx, float cy)
{
    if (!out) return pfail(ORBX_E_INVALID, "out is NULL");
    *out = nullptr;
    if (n < 0 || (n > 0 && (!p2d || !sigma2 || !p3d))) return pfail(ORBX_E_INVALID, "bad correspondences");
    orbp_pnp *s = new orbp_pnp();
    s->N = n;
    s->p2d.assign(p2d, p2d + 2 * (size_t)n); s->sigma2.assign(sigma2, sigma2 + n); s->p3d.assign(p3d, p3d + 3 * (size_t)n);
    s->max_error.resize(n); s->inl_i.assign(n, 0); s->inl_best.assign(n, 0); s->inl_refined.assign(n, 0);
    s->e.fu = fx; s->e.fv = fy; s->e.uc = cx; s->e.vc = cy;
    orbp_pnp_set_ransac_parameters(s, 0.99, 8, 300, 4, 0.4f, 5.991f);     // the constructor's SetRansacParameters() (:108)
    *out = s;
    return 0;
}

extern "C" void orbp_pnp_destroy(orbp_pnp *s) { delete s; }

extern "C" void orbp_pnp_set_rand(orbp_pnp *s, int (*fn)(void *), void *ctx, int rand_max)
{
    if (!s) return;
    if (fn) { s->rnd = fn; s->rnd_ctx = ctx; s->rnd_max = rand_max; }
    else { s->rnd = libc_rand; s->rnd_ctx = nullptr; s->rnd_max = RAND_MAX; }
}

extern "C" void orbp_pnp_get_ransac_state(const orbp_pnp *s, int *min_inliers, int *max_its, float *epsilon, int *iterations_done)
{
    if (!s) return;
    if (min_inliers) *min_inliers = s->min_inliers;
    if (max_its) *max_its = s->max_its;
    if (epsilon) *epsilon = s->eps;
    if (iterations_done) *iterations_done = s->n_iter;
}

extern "C" int orbp_pnp_iterate(orbp_pnp *s, int n_iterations, int *no_more, uint8_t *inliers, int *n_inliers, float *Tcw)
{
    if (!s || !no_more || !inliers || !n_inliers || !Tcw) return pfail(ORBX_E_INVALID, "NULL argument");
    *no_more = 0; *n_inliers = 0;              // :168-170
    const int N = s->N;
    if (N < s->min_inliers) { *no_more = 1; return 0; }
    std::vector<int> avail;
    int cur = 0;
    while (s->n_iter < s->max_its || cur < n_iterations) {
        cur++; s->n_iter++;
        s->e.reset();
        avail.resize(N);
        for (int i = 0; i < N; i++) avail[i] = i;
        for (int i = 0; i < s->min_set; ++i) {         // draw without replacement (:193-204)
            const int r = s->random_int(0, (int)avail.size() - 1);
            s->add_corr(avail[r]);
            avail[r] = avail.back();
            avail.pop_back();
        }
        s->e.compute_pose(s->Ri, s->ti);
        s->check_inliers();
        if (s->n_inl_i >= s->min_inliers) {
            if (s->n_inl_i > s->n_best) {
                s->inl_best = s->inl_i;
                s->n_best = s->n_inl_i;
                s->pose_to_T(s->best_T);
            }
            if (s->refine()) {                         // note: refines on the best set so far, not on this draw's
                *n_inliers = s->n_refined;
                memcpy(inliers, s->inl_refined.data(), N);
                memcpy(Tcw, s->refined_T, sizeof(float) * 16);
                return 1;
            }
        }
    }
    if (s->n_iter >= s->max_its) {
        *no_more = 1;
        if (s->n_best >= s->min_inliers) {
            *n_inliers = s->n_best;
            memcpy(inliers, s->inl_best.data(), N);
            memcpy(Tcw, s->best_T, sizeof(float) * 16);
            return 1;
        }
    }
    return 0;
}

extern "C" int orbp_pnp_find(orbp_pnp *s, uint8_t *inliers, int *n_inliers, float *Tcw)
{
    int flag = 0;
    if (!s) return pfail(ORBX_E_INVALID, "solver is NULL");
    return orbp_pnp_iterate(s, s->max_its, &flag, inliers, n_inliers, Tcw);
}

// -------------------------------------------------------------------------------------------------
// Motion-only pose optimisation (Optimizer.cc:239-451 on g2o)
// -------------------------------------------------------------------------------------------------
namespace {
struct Quat { double w, x, y, z; };
struct Se3 { Quat r; double t[3]; };

static Quat quat_from_R(const double R[9])    // Eigen's Quaterniond(Matrix3d) (Shepperd), as g2o::SE3Quat(R, t) uses
{
    Quat q;
    double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        tr = std::sqrt(tr + 1.0);
        q.w = 0.5 * tr;
        tr = 0.5 / tr;
        q.x = (R[7] - R[5]) * tr; q.y = (R[2] - R[6]) * tr; q.z = (R[3] - R[1]) * tr;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double v[3];
        tr = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        v[i] = 0.5 * tr;
        tr = 0.5 / tr;
        q.w = (R[3 * k + j] - R[3 * j + k]) * tr;
        v[j] = (R[3 * j + i] + R[3 * i + j]) * tr;
        v[k] = (R[3 * k + i] + R[3 * i + k]) * tr;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}
static void quat_normalize(Quat &q)           // se3quat.h:280-285
{
    if (q.w < 0) { q.w = -q.w; q.x = -q.x; q.y = -q.y; q.z = -q.z; }
    const double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    q.w /= n; q.x /= n; q.y /= n; q.z /= n;
}
static Quat quat_mul(const Quat &a, const Quat &b)
{
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
static void quat_rotate(const Quat &q, const double v[3], double out[3])      // Eigen: v + w uv + q.vec x uv, uv = 2 q.vec x v
{
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
    out[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
    out[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
static void quat_to_R(const Quat &q, double R[9])
{
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static void se3_map(const Se3 &T, const double x[3], double out[3])
{
    quat_rotate(T.r, x, out);
    out[0] += T.t[0]; out[1] += T.t[1]; out[2] += T.t[2];
}
static Se3 se3_exp(const double u[6])          // se3quat.h:223-257 (omega first, then upsilon)
{
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9], R[9], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) { R[i] = (i % 4 == 0) + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta);
        const double c = (theta - std::sin(theta)) / std::pow(theta, 3);
        for (int i = 0; i < 9; i++) { R[i] = (i % 4 == 0) + a * O[i] + b * O2[i]; V[i] = (i % 4 == 0) + b * O[i] + c * O2[i]; }
    }
    Se3 T;
    T.r = quat_from_R(R);
    quat_normalize(T.r);
    for (int i = 0; i < 3; i++) T.t[i] = V[3 * i] * up[0] + V[3 * i + 1] * up[1] + V[3 * i + 2] * up[2];
    return T;
}
static Se3 se3_mul(const Se3 &a, const Se3 &b)  // se3quat.h:104-110
{
    Se3 r = a;
    double rt[3];
    quat_rotate(a.r, b.t, rt);
    r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
    r.r = quat_mul(a.r, b.r);
    quat_normalize(r.r);
    return r;
}

// 6x6 symmetric positive definite solve (the reference: Eigen LDLT inside g2o::LinearSolverDense)
static bool ldlt_solve6(const double H[36], const double b[6], double x[6])
{
    double L[36] = {0}, D[6];
    for (int j = 0; j < 6; j++) {
        double d = H[6 * j + j];
        for (int k = 0; k < j; k++) d -= L[6 * j + k] * L[6 * j + k] * D[k];
        if (!(std::fabs(d) > 0) || !std::isfinite(d)) return false;
        D[j] = d;
        L[6 * j + j] = 1;
        for (int i = j + 1; i < 6; i++) {
            double s = H[6 * i + j];
            for (int k = 0; k < j; k++) s -= L[6 * i + k] * L[6 * j + k] * D[k];
            L[6 * i + j] = s / d;
        }
    }
    double y[6];
    for (int i = 0; i < 6; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[6 * i + k] * y[k]; y[i] = s; }
    for (int i = 0; i < 6; i++) y[i] /= D[i];
    for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[6 * k + i] * x[k]; x[i] = s; }
    return true;
}

struct Edge {
    int idx; bool stereo;
    double obs[3], xw[3], info;
    double err[3];         // _error: refreshed only where g2o refreshes it
    double pc[3];          // the point in the camera frame at the estimate compute_error last saw
    bool robust; int level;
};

struct PoseProblem {
    double fx, fy, cx, cy, bf;
    std::vector<Edge> edges;
    std::vector<int> active;
    Se3 est;
    double delta_mono, delta_stereo;

    void compute_error(Edge &e) const         // types_six_dof_expmap.h:153-157,184-188; .cpp:290-306
    {
        double *p = e.pc;
        se3_map(est, e.xw, p);
        if (!e.stereo) {
            e.err[0] = e.obs[0] - (p[0] / p[2] * fx + cx);
            e.err[1] = e.obs[1] - (p[1] / p[2] * fy + cy);
            e.err[2] = 0;
        } else {
            const float invz = 1.0f / (float)p[2];
            const double u = p[0] * invz * fx + cx;
            e.err[0] = e.obs[0] - u;
            e.err[1] = e.obs[1] - (p[1] * invz * fy + cy);
            e.err[2] = e.obs[2] - (u - bf * invz);
        }
    }
    static double chi2(const Edge &e) { return (e.err[0] * e.err[0] + e.err[1] * e.err[1] + e.err[2] * e.err[2]) * e.info; }
    void huber(const Edge &e, double c2, double rho[2]) const      // robust_kernel_impl.cpp:78-91
    {
        const double delta = e.stereo ? delta_stereo : delta_mono, dsqr = delta * delta;
        if (c2 <= dsqr) { rho[0] = c2; rho[1] = 1.; }
        else { const double s = std::sqrt(c2); rho[0] = 2 * s * delta - dsqr; rho[1] = delta / s; }
    }
    void compute_active_errors() { for (int k : active) compute_error(edges[k]); }
    double active_robust_chi2() const
    {
        double sum = 0;
        for (int k : active) {
            const Edge &e = edges[k];
            const double c2 = chi2(e);
            if (e.robust) { double rho[2]; huber(e, c2, rho); sum += rho[0]; }
            else sum += c2;
        }
        return sum;
    }
    // linearizeOplus (.cpp:266-288,335-364) + constructQuadraticForm (base_unary_edge.hpp:43-72).  Called right after
    // compute_active_errors() at the same estimate, so the camera-frame points cached there are the ones g2o maps again.
    void build_system(double H[36], double b[6]) const
    {
        double Hu[21] = {0};
        for (int i = 0; i < 6; i++) b[i] = 0;
        for (int k : active) {
            const Edge &e = edges[k];
            const double x = e.pc[0], y = e.pc[1], invz = 1.0 / e.pc[2], invz_2 = invz * invz;
            double J[3][6];
            J[0][0] = x * y * invz_2 * fx; J[0][1] = -(1 + (x * x * invz_2)) * fx; J[0][2] = y * invz * fx;
            J[0][3] = -invz * fx; J[0][4] = 0; J[0][5] = x * invz_2 * fx;
            J[1][0] = (1 + y * y * invz_2) * fy; J[1][1] = -x * y * invz_2 * fy; J[1][2] = -x * invz * fy;
            J[1][3] = 0; J[1][4] = -invz * fy; J[1][5] = y * invz_2 * fy;
            if (e.stereo) {
                J[2][0] = J[0][0] - bf * y * invz_2; J[2][1] = J[0][1] + bf * x * invz_2; J[2][2] = J[0][2];
                J[2][3] = J[0][3]; J[2][4] = 0; J[2][5] = J[0][5] - bf * invz_2;
            }
            double w = 1.0;
            if (e.robust) { double rho[2]; huber(e, chi2(e), rho); w = rho[1]; }
            const double wi = w * e.info;
            // b -= rho' J^T Omega e and H += J^T (rho' Omega) J, upper triangle; written out per row count so the
            // compiler sees fixed trip counts (the sums keep the order 0 + row0 + row1 (+ row2))
            double wJ[3][6];
            for (int r = 0; r < 6; r++) { wJ[0][r] = J[0][r] * wi; wJ[1][r] = J[1][r] * wi; }
            if (!e.stereo) {
                int u = 0;
                for (int r = 0; r < 6; r++) {
                    const double g = J[0][r] * e.info * e.err[0] + J[1][r] * e.info * e.err[1];
                    b[r] -= w * g;
                    for (int c = r; c < 6; c++, u++) Hu[u] += wJ[0][r] * J[0][c] + wJ[1][r] * J[1][c];
                }
            } else {
                for (int r = 0; r < 6; r++) wJ[2][r] = J[2][r] * wi;
                int u = 0;
                for (int r = 0; r < 6; r++) {
                    const double g = J[0][r] * e.info * e.err[0] + J[1][r] * e.info * e.err[1] + J[2][r] * e.info * e.err[2];
                    b[r] -= w * g;
                    for (int c = r; c < 6; c++, u++) Hu[u] += wJ[0][r] * J[0][c] + wJ[1][r] * J[1][c] + wJ[2][r] * J[2][c];
                }
            }
        }
        int u = 0;
        for (int r = 0; r < 6; r++)
            for (int c = r; c < 6; c++, u++) H[6 * r + c] = H[6 * c + r] = Hu[u];
    }

    // OptimizationAlgorithmLevenberg::solve driven by SparseOptimizer::optimize (sparse_optimizer.cpp:354-420)
    void optimize(int iterations)
    {
        double lambda = 0, ni = 2;
        int n_bad = 0;
        // g2o recomputes the active errors and their robust chi2 at the top of every iteration; right after an accepted
        // step they are exactly what the trial left on the edges, so that pass is skipped (same values, same sum)
        bool fresh = false;
        double carried = 0;
        for (int it = 0; it < iterations; it++) {
            if (!fresh) { compute_active_errors(); carried = active_robust_chi2(); }
            double current = carried, temp = current;
            const double ini = current;
            double H[36], b[6], x[6];
            build_system(H, b);
            if (it == 0) {
                double mx = 0;
                for (int j = 0; j < 6; j++) mx = std::max(std::fabs(H[7 * j]), mx);
                lambda = 1e-5 * mx; ni = 2; n_bad = 0;
            }
            double rho = 0;
            int qmax = 0;
            do {
                const Se3 backup = est;
                double Hl[36];
                memcpy(Hl, H, sizeof Hl);
                for (int j = 0; j < 6; j++) Hl[7 * j] += lambda;
                for (int j = 0; j < 6; j++) x[j] = 0;
                const bool ok2 = ldlt_solve6(Hl, b, x);
                est = se3_mul(se3_exp(x), est);
                compute_active_errors();
                temp = active_robust_chi2();
                if (!ok2) temp = DBL_MAX;
                rho = current - temp;
                double scale = 0;
                for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && std::isfinite(temp)) {
                    double alpha = 1. - std::pow((2 * rho - 1), 3);
                    alpha = std::min(alpha, 2. / 3.);
                    lambda *= std::max(1. / 3., alpha);
                    ni = 2;
                    current = temp;
                    fresh = true; carried = temp;
                } else {
                    lambda *= ni;
                    ni *= 2;
                    est = backup;          // the edges keep the trial's errors, as in g2o
                    fresh = false;
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (qmax == 10 || rho == 0) return;
            if ((ini - current) * 1e3 < ini) n_bad++; else n_bad = 0;
            if (n_bad >= 3) return;
        }
    }
};
}   // namespace

extern "C" int orbp_pose_optimization(int n, const float *obs, const float *u_right, const float *inv_sigma2, const float *xw,
                                      float fx, float fy, float cx, float cy, float bf, float *Tcw, uint8_t *outlier)
{
    if (n < 0 || !Tcw || !outlier || (n > 0 && (!obs || !inv_sigma2 || !xw))) return pfail(ORBX_E_INVALID, "bad arguments");
    PoseProblem P;
    P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.bf = bf;
    P.delta_mono = (float)std::sqrt(5.991); P.delta_stereo = (float)std::sqrt(7.815);     // const float delta* (:270-271)
    double R0[9], t0[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R0[3 * i + j] = Tcw[4 * i + j]; t0[i] = Tcw[4 * i + 3]; }
    Se3 init;
    init.r = quat_from_R(R0);
    quat_normalize(init.r);
    memcpy(init.t, t0, sizeof t0);
    P.est = init;
    P.edges.resize(n);
    for (int i = 0; i < n; i++) {
        Edge &e = P.edges[i];
        e.idx = i; e.stereo = u_right && u_right[i] >= 0;
        e.obs[0] = obs[2 * i]; e.obs[1] = obs[2 * i + 1]; e.obs[2] = e.stereo ? u_right[i] : 0;
        for (int j = 0; j < 3; j++) e.xw[j] = xw[3 * i + j];
        e.info = inv_sigma2[i];
        e.err[0] = e.err[1] = e.err[2] = 0;
        e.robust = true; e.level = 0;
        outlier[i] = 0;
    }
    if (n < 3) return 0;                        // :363-364
    const float chi2_mono = 5.991f, chi2_stereo = 7.815f;
    int n_bad = 0;
    for (int it = 0; it < 4; it++) {
        P.est = init;                           // every round restarts from pFrame->mTcw (:375)
        P.active.clear();
        for (int i = 0; i < n; i++)
            if (P.edges[i].level == 0) P.active.push_back(i);
        if (!P.active.empty()) P.optimize(10);
        n_bad = 0;
        for (int i = 0; i < n; i++) {
            Edge &e = P.edges[i];
            if (outlier[i]) P.compute_error(e);
            const float c2 = (float)PoseProblem::chi2(e);
            if (c2 > (e.stereo ? chi2_stereo : chi2_mono)) { outlier[i] = 1; e.level = 1; n_bad++; }
            else { outlier[i] = 0; e.level = 0; }
            if (it == 2) e.robust = false;
        }
        if (n < 10) break;                      // optimizer.edges().size() < 10 (:443-444)
    }
    double R[9];
    quat_to_R(P.est.r, R);
    for (int i = 0; i < 16; i++) Tcw[i] = (i % 5 == 0) ? 1.f : 0.f;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Tcw[4 * i + j] = (float)R[3 * i + j];
        Tcw[4 * i + 3] = (float)P.est.t[i];
    }
    return n - n_bad;
}
