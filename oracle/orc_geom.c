/*
 * orc_geom.c — CPU ORACLE (test infrastructure): SE(3) algebra, linear stereo
 * triangulation, pose-only Levenberg-Marquardt and Schur-complement local
 * bundle adjustment.
 *
 * Restates for the reference call sites
 *   include/StereoVisionSLAM/algorithm.h:10-87   triangulation()  (Eigen bdcSvd)
 *   src/frontend.cpp:394-558                      EstimateCurrentPose (g2o)
 *   src/backend.cpp:22-164                        Backend::Optimize   (g2o)
 *   include/StereoVisionSLAM/g2o_types.h:25-229   vertices / edges
 * the published algorithms of the un-vendored libraries that do the arithmetic
 * (NOT in /root/reference):
 *   g2o   core/optimization_algorithm_levenberg.cpp  (solve, computeLambdaInit,
 *         computeScale), core/block_solver.hpp (buildSystem, Schur solve),
 *         core/base_{unary,binary}_edge.hpp (constructQuadraticForm, numeric
 *         Jacobian delta=1e-9), core/robust_kernel_impl.cpp (Huber),
 *         solvers/dense/linear_solver_dense.h (Eigen LDLT)
 *   Sophus se3.hpp / so3.hpp  (exp, log, product, inverse, action)
 *   Eigen 3.4 SVD of a 4x4 (bdcSvd falls back to Jacobi below 16 columns)
 * PARITY UNPINNED (see svs_oracle.h).
 */
#include "svs_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <stddef.h>

/* ================================================================== */
/* SE(3): T = [qx qy qz qw tx ty tz]                                   */

static void quat_mul(const double a[4], const double b[4], double c[4])
{
    double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    c[0] = aw * bx + ax * bw + ay * bz - az * by;
    c[1] = aw * by + ay * bw + az * bx - ax * bz;
    c[2] = aw * bz + az * bw + ax * by - ay * bx;
    c[3] = aw * bw - ax * bx - ay * by - az * bz;
}

static void quat_rot(const double q[4], const double v[3], double o[3])
{
    /* Eigen QuaternionBase::_transformVector */
    double ux = q[1] * v[2] - q[2] * v[1];
    double uy = q[2] * v[0] - q[0] * v[2];
    double uz = q[0] * v[1] - q[1] * v[0];
    ux += ux; uy += uy; uz += uz;
    o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

void orc_se3_identity(double T[7])
{
    T[0] = T[1] = T[2] = 0; T[3] = 1; T[4] = T[5] = T[6] = 0;
}

void orc_se3_act(const double T[7], const double p[3], double o[3])
{
    quat_rot(T, p, o);
    o[0] += T[4]; o[1] += T[5]; o[2] += T[6];
}

void orc_se3_mul(const double A[7], const double B[7], double C[7])
{
    double q[4], t[3];
    quat_mul(A, B, q);
    /* Sophus SO3::operator*: first-order renormalisation */
    double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (n2 != 1.0) {
        double s = 2.0 / (1.0 + n2);
        q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
    }
    quat_rot(A, B + 4, t);
    C[0] = q[0]; C[1] = q[1]; C[2] = q[2]; C[3] = q[3];
    C[4] = A[4] + t[0]; C[5] = A[5] + t[1]; C[6] = A[6] + t[2];
}

void orc_se3_inv(const double T[7], double Ti[7])
{
    double q[4] = { -T[0], -T[1], -T[2], T[3] };
    double nt[3] = { -T[4], -T[5], -T[6] }, t[3];
    quat_rot(q, nt, t);
    Ti[0] = q[0]; Ti[1] = q[1]; Ti[2] = q[2]; Ti[3] = q[3];
    Ti[4] = t[0]; Ti[5] = t[1]; Ti[6] = t[2];
}

static void quat_to_R(const double q[4], double R[9])
{
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

#define SOPHUS_EPS 1e-10

void orc_se3_exp(const double xi[6], double T[7])
{
    /* Sophus SE3::exp(upsilon, omega) */
    const double *u = xi, *om = xi + 3;
    double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    double theta, imag, real;
    if (th2 < SOPHUS_EPS * SOPHUS_EPS) {
        theta = 0;
        double th4 = th2 * th2;
        imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
    } else {
        theta = sqrt(th2);
        double half = 0.5 * theta;
        imag = sin(half) / theta;
        real = cos(half);
    }
    T[0] = imag * om[0]; T[1] = imag * om[1]; T[2] = imag * om[2]; T[3] = real;
    double V[9];
    if (theta < SOPHUS_EPS) {
        quat_to_R(T, V);
    } else {
        double a = (1.0 - cos(theta)) / th2;
        double b = (theta - sin(theta)) / (th2 * theta);
        double O[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
        double O2[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
        for (int i = 0; i < 9; ++i) V[i] = a * O[i] + b * O2[i];
        V[0] += 1; V[4] += 1; V[8] += 1;
    }
    T[4] = V[0] * u[0] + V[1] * u[1] + V[2] * u[2];
    T[5] = V[3] * u[0] + V[4] * u[1] + V[5] * u[2];
    T[6] = V[6] * u[0] + V[7] * u[1] + V[8] * u[2];
}

void orc_se3_log(const double T[7], double xi[6])
{
    /* Sophus SO3::logAndTheta + SE3::log */
    double n2 = T[0] * T[0] + T[1] * T[1] + T[2] * T[2], w = T[3];
    double two_atan;
    if (n2 < SOPHUS_EPS * SOPHUS_EPS) {
        double w2 = w * w;
        two_atan = 2.0 / w - (2.0 / 3.0) * n2 / (w * w2);
    } else {
        double n = sqrt(n2);
        if (fabs(w) < SOPHUS_EPS) two_atan = (w > 0 ? M_PI : -M_PI) / n;
        else two_atan = 2.0 * atan(n / w) / n;
    }
    double om[3] = { two_atan * T[0], two_atan * T[1], two_atan * T[2] };
    double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    double O[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
    double O2[9], Vi[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
    double c;
    if (fabs(theta) < SOPHUS_EPS) c = 1.0 / 12.0;
    else {
        double half = 0.5 * theta;
        c = (1.0 - theta * cos(half) / (2.0 * sin(half))) / (theta * theta);
    }
    for (int i = 0; i < 9; ++i) Vi[i] = -0.5 * O[i] + c * O2[i];
    Vi[0] += 1; Vi[4] += 1; Vi[8] += 1;
    const double *t = T + 4;
    xi[0] = Vi[0] * t[0] + Vi[1] * t[1] + Vi[2] * t[2];
    xi[1] = Vi[3] * t[0] + Vi[4] * t[1] + Vi[5] * t[2];
    xi[2] = Vi[6] * t[0] + Vi[7] * t[1] + Vi[8] * t[2];
    xi[3] = om[0]; xi[4] = om[1]; xi[5] = om[2];
}

/* ================================================================== */
/* triangulation: 4x4 DLT, one-sided Jacobi SVD                        */

static void svd4_jacobi(double A[16], double V[16], double sv[4])
{
    /* Hestenes one-sided Jacobi on the columns of A (4x4, row-major);
     * on exit columns are sorted by decreasing norm, V accumulates rotations. */
    for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 4; ++i) {
                    al += A[i * 4 + p] * A[i * 4 + p];
                    be += A[i * 4 + q] * A[i * 4 + q];
                    ga += A[i * 4 + p] * A[i * 4 + q];
                }
                if (ga == 0.0 || fabs(ga) <= 1e-300 + 2.3e-16 * sqrt(al * be)) continue;
                rotated = 1;
                double zeta = (be - al) / (2.0 * ga);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 4; ++i) {
                    double ap = A[i * 4 + p], aq = A[i * 4 + q];
                    A[i * 4 + p] = c * ap - s * aq;
                    A[i * 4 + q] = s * ap + c * aq;
                    double vp = V[i * 4 + p], vq = V[i * 4 + q];
                    V[i * 4 + p] = c * vp - s * vq;
                    V[i * 4 + q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int i = 0; i < 4; ++i) s += A[i * 4 + j] * A[i * 4 + j];
        sv[j] = sqrt(s);
    }
    /* sort descending (selection sort, swapping columns of V) */
    for (int j = 0; j < 3; ++j) {
        int m = j;
        for (int k = j + 1; k < 4; ++k) if (sv[k] > sv[m]) m = k;
        if (m != j) {
            double t = sv[j]; sv[j] = sv[m]; sv[m] = t;
            for (int i = 0; i < 4; ++i) {
                double v = V[i * 4 + j]; V[i * 4 + j] = V[i * 4 + m]; V[i * 4 + m] = v;
            }
        }
    }
}

int orc_triangulate_dlt(const double ext_l[7], const double ext_r[7],
                        const double pl[3], const double pr[3], double out[3])
{
    /* algorithm.h:62-86: rows x*m2 - m0, y*m2 - m1 with m = pose.matrix3x4() */
    double A[16], V[16], sv[4];
    const double *ext[2] = { ext_l, ext_r };
    const double *pt[2] = { pl, pr };
    for (int c = 0; c < 2; ++c) {
        double R[9];
        quat_to_R(ext[c], R);
        double m[12] = { R[0], R[1], R[2], ext[c][4], R[3], R[4], R[5], ext[c][5],
                         R[6], R[7], R[8], ext[c][6] };
        for (int j = 0; j < 4; ++j) {
            A[(2 * c) * 4 + j] = pt[c][0] * m[8 + j] - m[j];
            A[(2 * c + 1) * 4 + j] = pt[c][1] * m[8 + j] - m[4 + j];
        }
    }
    svd4_jacobi(A, V, sv);
    double w = V[3 * 4 + 3];
    out[0] = V[0 * 4 + 3] / w; out[1] = V[1 * 4 + 3] / w; out[2] = V[2 * 4 + 3] / w;
    return (sv[3] / sv[2] < 1e-2) ? 1 : 0;
}

void orc_triangulate(int n, const double cam_l[4], const double ext_l[7],
                     const double cam_r[4], const double ext_r[7],
                     const float *uv_l, const float *uv_r, const double T_wc[7],
                     double zmax, double *out_xyz, uint8_t *out_ok)
{
    for (int i = 0; i < n; ++i) {
        /* Camera::pixel2camera (src/camera.cpp:58-72), depth = 1 */
        double pl[3] = { ((double)uv_l[2 * i] - cam_l[2]) * 1.0 / cam_l[0],
                         ((double)uv_l[2 * i + 1] - cam_l[3]) * 1.0 / cam_l[1], 1.0 };
        double pr[3] = { ((double)uv_r[2 * i] - cam_r[2]) * 1.0 / cam_r[0],
                         ((double)uv_r[2 * i + 1] - cam_r[3]) * 1.0 / cam_r[1], 1.0 };
        double p[3];
        int ok = orc_triangulate_dlt(ext_l, ext_r, pl, pr, p);
        ok = ok && (p[2] > 0) && (zmax <= 0 || p[2] <= zmax);
        double pw[3];
        orc_se3_act(T_wc, p, pw);
        out_xyz[3 * i] = pw[0]; out_xyz[3 * i + 1] = pw[1]; out_xyz[3 * i + 2] = pw[2];
        out_ok[i] = (uint8_t)(ok ? 1 : 0);
    }
}

/* ================================================================== */
/* dense LDLT with diagonal pivoting (Eigen::LDLT), n <= ORC_MAXN       */
#define ORC_MAXN 192

static int ldlt_solve(int n, const double *Hin, const double *b, double *x)
{
    /* returns 1 if H is positive (semi)definite and the system was solved */
    double *L = (double *)malloc(sizeof(double) * (size_t)n * n);
    double D[ORC_MAXN];
    int perm[ORC_MAXN];
    for (int i = 0; i < n; ++i) {
        perm[i] = i;
        for (int j = 0; j < n; ++j) L[i * n + j] = Hin[i * n + j];
    }
    int positive = 1;
    for (int k = 0; k < n; ++k) {
        int piv = k; double best = fabs(L[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(L[i * n + i]) > best) { best = fabs(L[i * n + i]); piv = i; }
        if (piv != k) {
            /* symmetric swap of rows/cols k and piv (full storage) */
            for (int j = 0; j < n; ++j) { double t = L[k * n + j]; L[k * n + j] = L[piv * n + j]; L[piv * n + j] = t; }
            for (int i = 0; i < n; ++i) { double t = L[i * n + k]; L[i * n + k] = L[i * n + piv]; L[i * n + piv] = t; }
            int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
        }
        double dk = L[k * n + k];
        for (int j = 0; j < k; ++j) dk -= L[k * n + j] * L[k * n + j] * D[j];
        D[k] = dk;
        if (dk < 0) positive = 0;
        for (int i = k + 1; i < n; ++i) {
            double v = L[i * n + k];
            for (int j = 0; j < k; ++j) v -= L[i * n + j] * L[k * n + j] * D[j];
            L[i * n + k] = (dk != 0.0) ? v / dk : 0.0;
        }
    }
    if (!positive) { free(L); return 0; }
    double y[ORC_MAXN];
    for (int i = 0; i < n; ++i) {
        double v = b[perm[i]];
        for (int j = 0; j < i; ++j) v -= L[i * n + j] * y[j];
        y[i] = v;
    }
    for (int i = 0; i < n; ++i) y[i] = (D[i] != 0.0) ? y[i] / D[i] : 0.0;
    for (int i = n - 1; i >= 0; --i) {
        double v = y[i];
        for (int j = i + 1; j < n; ++j) v -= L[j * n + i] * y[j];
        y[i] = v;
    }
    for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
    free(L);
    return 1;
}

/* g2o RobustKernelHuber::robustify */
static inline void huber(double e2, double delta, double rho[3])
{
    double dsqr = delta * delta;
    if (e2 <= dsqr) { rho[0] = e2; rho[1] = 1.0; rho[2] = 0.0; }
    else {
        double sq = sqrt(e2);
        rho[0] = 2 * sq * delta - dsqr;
        rho[1] = delta / sq;
        rho[2] = -0.5 * rho[1] / e2;
    }
}

/* ================================================================== */
/* pose-only LM  (src/frontend.cpp:394-558)                            */

static void po_error(const double cam[4], const double T[7], const double P[3],
                     const float uv[2], double e[2])
{
    /* EdgeProjectionPoseOnly::computeError (g2o_types.h:118-130):
     * pos_pixel = K * (T * P); pos_pixel /= pos_pixel[2]; e = meas - pixel */
    double pc[3];
    orc_se3_act(T, P, pc);
    double px = cam[0] * pc[0] + cam[2] * pc[2];
    double py = cam[1] * pc[1] + cam[3] * pc[2];
    double pz = pc[2];
    e[0] = (double)uv[0] - px / pz;
    e[1] = (double)uv[1] - py / pz;
}

static void po_jacobian(const double cam[4], const double T[7], const double P[3],
                        double J[12])
{
    /* EdgeProjectionPoseOnly::linearizeOplus (g2o_types.h:132-163) */
    double pc[3];
    orc_se3_act(T, P, pc);
    double fx = cam[0], fy = cam[1];
    double X = pc[0], Y = pc[1], Z = pc[2];
    double Zinv = 1.0 / (Z + 1e-18), Zinv2 = Zinv * Zinv;
    J[0] = -fx * Zinv; J[1] = 0; J[2] = fx * X * Zinv2; J[3] = fx * X * Y * Zinv2;
    J[4] = -fx - fx * X * X * Zinv2; J[5] = fx * Y * Zinv;
    J[6] = 0; J[7] = -fy * Zinv; J[8] = fy * Y * Zinv2; J[9] = fy + fy * Y * Y * Zinv2;
    J[10] = -fy * X * Y * Zinv2; J[11] = -fy * X * Zinv;
}

static double po_compute_errors(int n, const double cam[4], const double T[7],
                                const double *xyz, const float *uv,
                                const uint8_t *active, int robust, double *err)
{
    double chi = 0;
    for (int i = 0; i < n; ++i) {
        if (!active[i]) continue;
        po_error(cam, T, xyz + 3 * i, uv + 2 * i, err + 2 * i);
        double e2 = err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1];
        if (robust) { double r[3]; huber(e2, 1.0, r); chi += r[0]; }
        else chi += e2;
    }
    return chi;
}

/* trace sink of orc_pose_only_trace (NULL otherwise): records as in orc_local_ba_trace, t[0] = 16 round + iteration */
static double *po_trace; static int po_trace_cap, po_trace_n, po_round;

static void po_optimize(int n, const double cam[4], double T[7], const double *xyz,
                        const float *uv, const uint8_t *active, int robust,
                        int iters, double *err)
{
    int nact = 0;
    for (int i = 0; i < n; ++i) nact += active[i] ? 1 : 0;
    if (nact == 0) return; /* SparseOptimizer::optimize: _ivMap.size()==0 -> -1 */
    double lambda = 0, ni = 2;
    for (int it = 0; it < iters; ++it) {
        double currentChi = po_compute_errors(n, cam, T, xyz, uv, active, robust, err);
        double tempChi = currentChi;
        double H[36], b[6];
        memset(H, 0, sizeof(H)); memset(b, 0, sizeof(b));
        for (int i = 0; i < n; ++i) {
            if (!active[i]) continue;
            double J[12];
            po_jacobian(cam, T, xyz + 3 * i, J);
            const double *e = err + 2 * i;
            double w = 1.0;
            if (robust) { double r[3]; huber(e[0] * e[0] + e[1] * e[1], 1.0, r); w = r[1]; }
            for (int a = 0; a < 6; ++a) {
                b[a] -= w * (J[a] * e[0] + J[6 + a] * e[1]);
                for (int c = 0; c < 6; ++c)
                    H[a * 6 + c] += w * (J[a] * J[c] + J[6 + a] * J[6 + c]);
            }
        }
        if (it == 0) {
            double md = 0;
            for (int a = 0; a < 6; ++a) if (fabs(H[a * 7]) > md) md = fabs(H[a * 7]);
            lambda = (orc_whatif[1] ? 1e-3 : 1e-5) * md; ni = 2;
        }
        double rho = 0; int qmax = 0;
        double x[6] = { 0, 0, 0, 0, 0, 0 };
        do {
            double Tb[7]; memcpy(Tb, T, sizeof(Tb));
            double Hl[36]; memcpy(Hl, H, sizeof(Hl));
            for (int a = 0; a < 6; ++a) Hl[a * 7] += lambda;
            int ok2 = ldlt_solve(6, Hl, b, x);
            double dT[7], Tn[7];
            orc_se3_exp(x, dT);
            orc_se3_mul(dT, T, Tn);
            memcpy(T, Tn, sizeof(Tn));
            tempChi = po_compute_errors(n, cam, T, xyz, uv, active, robust, err);
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            double scale = 0;
            for (int a = 0; a < 6; ++a) scale += x[a] * (lambda * x[a] + b[a]);
            if (!orc_whatif[2]) scale += 1e-3;
            rho /= scale;
            if (po_trace && po_trace_n < po_trace_cap) {
                double *t = po_trace + (size_t)ORC_TRACE_REC * po_trace_n++;
                t[0] = 16 * po_round + it; t[1] = lambda; t[2] = currentChi; t[3] = tempChi; t[4] = rho;
                t[5] = (rho > 0 && isfinite(tempChi)) ? 1.0 : 0.0;
            }
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow(2 * rho - 1, 3);
                if (alpha > 2. / 3.) alpha = 2. / 3.;
                double sf = alpha < 1. / 3. ? 1. / 3. : alpha;
                lambda *= sf; ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                memcpy(T, Tb, sizeof(Tb));
                if (!isfinite(lambda)) break;
            }
            ++qmax;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) break;
    }
}

int orc_pose_only(int n, const double cam[4], double pose[7], const double *xyz,
                  const float *uv, uint8_t *outlier, double chi2_th, int rounds,
                  int iters)
{
    double T0[7], T[7];
    memcpy(T0, pose, sizeof(T0)); memcpy(T, pose, sizeof(T));
    double *err = (double *)calloc((size_t)(2 * n + 2), sizeof(double));
    uint8_t *active = (uint8_t *)malloc((size_t)n + 1);
    for (int i = 0; i < n; ++i) outlier[i] = 0;
    int robust = 1, cnt_outlier = 0;
    for (int r = 0; r < rounds; ++r) {
        memcpy(T, T0, sizeof(T));                       /* :485 */
        po_round = r;
        for (int i = 0; i < n; ++i) active[i] = outlier[i] ? 0 : 1; /* level 0 */
        po_optimize(n, cam, T, xyz, uv, active, robust, iters, err);
        cnt_outlier = 0;
        for (int i = 0; i < n; ++i) {
            if (outlier[i]) po_error(cam, T, xyz + 3 * i, uv + 2 * i, err + 2 * i); /* :498-501 */
            double chi2 = err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1];
            if (chi2 > chi2_th) { outlier[i] = 1; ++cnt_outlier; }
            else outlier[i] = 0;
        }
        if (r == 2) robust = 0;                          /* :518-523 */
    }
    memcpy(pose, T, sizeof(T));
    free(err); free(active);
    return n - cnt_outlier;
}

int orc_pose_only_trace(int n, const double cam[4], double pose[7], const double *xyz,
                        const float *uv, uint8_t *outlier, double chi2_th, int rounds,
                        int iters, double *trace, int trace_cap, int *trace_n)
{
    po_trace = trace; po_trace_cap = trace_cap; po_trace_n = 0;
    int r = orc_pose_only(n, cam, pose, xyz, uv, outlier, chi2_th, rounds, iters);
    po_trace = NULL;
    if (trace_n) *trace_n = po_trace_n;
    return r;
}

/* ================================================================== */
/* local BA (src/backend.cpp:22-164)                                   */

typedef struct {
    const double *cam[2];
    const double *ext[2];
} ba_cams;

static void ba_error(const ba_cams *c, int cam, const double T[7], const double P[3],
                     const float uv[2], double e[2])
{
    /* EdgeProjection::computeError (g2o_types.h:200-216) */
    double q[3], p[3];
    orc_se3_act(T, P, q);
    orc_se3_act(c->ext[cam], q, p);
    const double *K = c->cam[cam];
    double px = K[0] * p[0] + K[2] * p[2];
    double py = K[1] * p[1] + K[3] * p[2];
    e[0] = (double)uv[0] - px / p[2];
    e[1] = (double)uv[1] - py / p[2];
}

static void ba_jac_analytic(const ba_cams *c, int cam, const double T[7],
                            const double P[3], double Jp[12], double Jl[6])
{
    double q[3], p[3], Re[9], R[9];
    orc_se3_act(T, P, q);
    orc_se3_act(c->ext[cam], q, p);
    quat_to_R(c->ext[cam], Re);
    quat_to_R(T, R);
    const double *K = c->cam[cam];
    double X = p[0], Y = p[1], Z = p[2];
    double zi = 1.0 / Z, zi2 = zi * zi;
    /* de/dp (2x3) */
    double E[6] = { -K[0] * zi, 0, K[0] * X * zi2, 0, -K[1] * zi, K[1] * Y * zi2 };
    /* dp/dxi = Re * [I | -q^] */
    double A[18];
    double qh[9] = { 0, q[2], -q[1], -q[2], 0, q[0], q[1], -q[0], 0 }; /* -q^ */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[i * 6 + j] = Re[i * 3 + j];
            A[i * 6 + 3 + j] = Re[i * 3] * qh[j] + Re[i * 3 + 1] * qh[3 + j] + Re[i * 3 + 2] * qh[6 + j];
        }
    for (int r = 0; r < 2; ++r)
        for (int j = 0; j < 6; ++j)
            Jp[r * 6 + j] = E[r * 3] * A[j] + E[r * 3 + 1] * A[6 + j] + E[r * 3 + 2] * A[12 + j];
    /* dp/dP = Re * R */
    double M[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            M[i * 3 + j] = Re[i * 3] * R[j] + Re[i * 3 + 1] * R[3 + j] + Re[i * 3 + 2] * R[6 + j];
    for (int r = 0; r < 2; ++r)
        for (int j = 0; j < 3; ++j)
            Jl[r * 3 + j] = E[r * 3] * M[j] + E[r * 3 + 1] * M[3 + j] + E[r * 3 + 2] * M[6 + j];
}

static void ba_jac_numeric(const ba_cams *c, int cam, const double T[7],
                           const double P[3], const float uv[2], double Jp[12],
                           double Jl[6])
{
    /* g2o BaseBinaryEdge::linearizeOplus(): central differences, delta 1e-9 */
    const double delta = 1e-9, scalar = 1.0 / (2 * delta);
    for (int d = 0; d < 6; ++d) {
        double add[6] = { 0, 0, 0, 0, 0, 0 }, dT[7], Tp[7], e1[2], e2[2];
        add[d] = delta;
        orc_se3_exp(add, dT); orc_se3_mul(dT, T, Tp);
        ba_error(c, cam, Tp, P, uv, e1);
        add[d] = -delta;
        orc_se3_exp(add, dT); orc_se3_mul(dT, T, Tp);
        ba_error(c, cam, Tp, P, uv, e2);
        Jp[d] = scalar * (e1[0] - e2[0]);
        Jp[6 + d] = scalar * (e1[1] - e2[1]);
    }
    for (int d = 0; d < 3; ++d) {
        double Pp[3] = { P[0], P[1], P[2] }, e1[2], e2[2];
        Pp[d] = P[d] + delta;
        ba_error(c, cam, T, Pp, uv, e1);
        Pp[d] = P[d] - delta;
        ba_error(c, cam, T, Pp, uv, e2);
        Jl[d] = scalar * (e1[0] - e2[0]);
        Jl[3 + d] = scalar * (e1[1] - e2[1]);
    }
}

/* test hooks: the Jacobians as the oracle evaluates them, for the symbolic (sympy) pin in
 * tests/test_oracle_pins.py.  mode 0 analytic, 1 numeric (g2o's central differences). */
void orc_ba_jacobian(const double cam[4], const double ext[7], const double T[7], const double P[3],
                     const float uv[2], int mode, double Jp[12], double Jl[6])
{
    ba_cams c;
    c.cam[0] = c.cam[1] = cam; c.ext[0] = c.ext[1] = ext;
    if (mode == 0) ba_jac_analytic(&c, 0, T, P, Jp, Jl);
    else ba_jac_numeric(&c, 0, T, P, uv, Jp, Jl);
}
void orc_po_jacobian(const double cam[4], const double T[7], const double P[3], double J[12])
{
    po_jacobian(cam, T, P, J);
}

static void inv3(const double A[9], double Ai[9])
{
    double c0 = A[4] * A[8] - A[5] * A[7];
    double c1 = A[5] * A[6] - A[3] * A[8];
    double c2 = A[3] * A[7] - A[4] * A[6];
    double det = A[0] * c0 + A[1] * c1 + A[2] * c2;
    double id = 1.0 / det;
    Ai[0] = c0 * id; Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id; Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    Ai[3] = c1 * id; Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id; Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    Ai[6] = c2 * id; Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id; Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

/* LM trajectory hook (test infrastructure): one record of ORC_TRACE_REC doubles per LM trial —
 * iteration, lambda of the trial, chi2 before, chi2 of the trial state (DBL_MAX: solve failed),
 * rho, accepted (1/0) — so that a GPU kernel can be compared trial by trial, rejected ones included. */
int orc_local_ba_trace(const double cam_l[4], const double ext_l[7],
                 const double cam_r[4], const double ext_r[7], int nkf,
                 double *poses, int nlm, double *pts, int nobs,
                 const int *obs_kf, const int *obs_lm,
                 const uint8_t *obs_is_right, const float *obs_uv,
                 double huber_delta, int iters, int jac_mode, double *edge_chi2,
                 double *trace, int trace_cap, int *trace_n);

int orc_local_ba(const double cam_l[4], const double ext_l[7],
                 const double cam_r[4], const double ext_r[7], int nkf,
                 double *poses, int nlm, double *pts, int nobs,
                 const int *obs_kf, const int *obs_lm,
                 const uint8_t *obs_is_right, const float *obs_uv,
                 double huber_delta, int iters, int jac_mode, double *edge_chi2)
{
    return orc_local_ba_trace(cam_l, ext_l, cam_r, ext_r, nkf, poses, nlm, pts, nobs, obs_kf, obs_lm, obs_is_right,
                              obs_uv, huber_delta, iters, jac_mode, edge_chi2, NULL, 0, NULL);
}

int orc_local_ba_trace(const double cam_l[4], const double ext_l[7],
                 const double cam_r[4], const double ext_r[7], int nkf,
                 double *poses, int nlm, double *pts, int nobs,
                 const int *obs_kf, const int *obs_lm,
                 const uint8_t *obs_is_right, const float *obs_uv,
                 double huber_delta, int iters, int jac_mode, double *edge_chi2,
                 double *trace, int trace_cap, int *trace_n)
{
    int ntrace = 0;
    if (trace_n) *trace_n = 0;
    ba_cams cams; cams.cam[0] = cam_l; cams.cam[1] = cam_r; cams.ext[0] = ext_l; cams.ext[1] = ext_r;
    if (nobs <= 0 || nkf <= 0 || nlm <= 0 || 6 * nkf > ORC_MAXN) return 0;
    /* active vertices: those with at least one edge (g2o initializeOptimization) */
    int *kf_act = (int *)calloc((size_t)nkf, sizeof(int));
    int *lm_act = (int *)calloc((size_t)nlm, sizeof(int));
    for (int e = 0; e < nobs; ++e) { kf_act[obs_kf[e]] = 1; lm_act[obs_lm[e]] = 1; }
    int *kf_idx = (int *)malloc(sizeof(int) * (size_t)nkf);
    int na = 0;
    for (int k = 0; k < nkf; ++k) kf_idx[k] = kf_act[k] ? na++ : -1;
    int np = 6 * na;
    /* (kf,lm) -> Hpl block index */
    int *blk = (int *)malloc(sizeof(int) * (size_t)nkf * nlm);
    for (size_t i = 0; i < (size_t)nkf * nlm; ++i) blk[i] = -1;
    int nblk = 0;
    int *eblk = (int *)malloc(sizeof(int) * (size_t)nobs);
    for (int e = 0; e < nobs; ++e) {
        size_t key = (size_t)obs_kf[e] * nlm + obs_lm[e];
        if (blk[key] < 0) blk[key] = nblk++;
        eblk[e] = blk[key];
    }
    int *blk_kf = (int *)malloc(sizeof(int) * (size_t)nblk);
    int *blk_lm = (int *)malloc(sizeof(int) * (size_t)nblk);
    for (int e = 0; e < nobs; ++e) { blk_kf[eblk[e]] = obs_kf[e]; blk_lm[eblk[e]] = obs_lm[e]; }
    /* per-landmark block lists */
    int *lm_start = (int *)calloc((size_t)nlm + 1, sizeof(int));
    for (int b = 0; b < nblk; ++b) lm_start[blk_lm[b] + 1]++;
    for (int j = 0; j < nlm; ++j) lm_start[j + 1] += lm_start[j];
    int *lm_blocks = (int *)malloc(sizeof(int) * (size_t)nblk);
    int *fill = (int *)calloc((size_t)nlm, sizeof(int));
    for (int k = 0; k < nkf; ++k)          /* ascending pose index inside a landmark */
        for (int b = 0; b < nblk; ++b)
            if (blk_kf[b] == k) { int j = blk_lm[b]; lm_blocks[lm_start[j] + fill[j]++] = b; }

    double *err = (double *)calloc((size_t)nobs * 2, sizeof(double));
    double *Hpp = (double *)malloc(sizeof(double) * (size_t)np * np);
    double *S = (double *)malloc(sizeof(double) * (size_t)np * np);
    double *bp = (double *)malloc(sizeof(double) * (size_t)np);
    double *bs = (double *)malloc(sizeof(double) * (size_t)np);
    double *Hll = (double *)malloc(sizeof(double) * (size_t)nlm * 9);
    double *Dinv = (double *)malloc(sizeof(double) * (size_t)nlm * 9);
    double *bl = (double *)malloc(sizeof(double) * (size_t)nlm * 3);
    double *W = (double *)malloc(sizeof(double) * (size_t)nblk * 18);
    double *xp = (double *)calloc((size_t)np, sizeof(double));
    double *xl = (double *)calloc((size_t)nlm * 3, sizeof(double));
    double *poses_b = (double *)malloc(sizeof(double) * (size_t)nkf * 7);
    double *pts_b = (double *)malloc(sizeof(double) * (size_t)nlm * 3);

#define BA_ERRORS(chi_out) do { \
        double chi_ = 0; \
        for (int e = 0; e < nobs; ++e) { \
            ba_error(&cams, obs_is_right[e] ? 1 : 0, poses + 7 * obs_kf[e], pts + 3 * obs_lm[e], \
                     obs_uv + 2 * e, err + 2 * e); \
            double r_[3]; \
            huber(err[2 * e] * err[2 * e] + err[2 * e + 1] * err[2 * e + 1], (orc_whatif[7] ? sqrt(huber_delta) : huber_delta), r_); \
            chi_ += r_[0]; \
        } \
        (chi_out) = chi_; } while (0)

    double lambda = 0, ni = 2;
    int it_done = 0;
    for (int it = 0; it < iters; ++it) {
        double currentChi, tempChi;
        BA_ERRORS(currentChi);
        tempChi = currentChi;
        /* buildSystem */
        memset(Hpp, 0, sizeof(double) * (size_t)np * np);
        memset(bp, 0, sizeof(double) * (size_t)np);
        memset(Hll, 0, sizeof(double) * (size_t)nlm * 9);
        memset(bl, 0, sizeof(double) * (size_t)nlm * 3);
        memset(W, 0, sizeof(double) * (size_t)nblk * 18);
        for (int e = 0; e < nobs; ++e) {
            int k = obs_kf[e], j = obs_lm[e], cam = obs_is_right[e] ? 1 : 0;
            double Jp[12], Jl[6];
            if (jac_mode == 1) ba_jac_numeric(&cams, cam, poses + 7 * k, pts + 3 * j, obs_uv + 2 * e, Jp, Jl);
            else ba_jac_analytic(&cams, cam, poses + 7 * k, pts + 3 * j, Jp, Jl);
            const double *er = err + 2 * e;
            double r[3];
            huber(er[0] * er[0] + er[1] * er[1], (orc_whatif[7] ? sqrt(huber_delta) : huber_delta), r);
            double w = r[1];
            int pk = 6 * kf_idx[k];
            for (int a = 0; a < 6; ++a) {
                bp[pk + a] -= w * (Jp[a] * er[0] + Jp[6 + a] * er[1]);
                for (int c = 0; c < 6; ++c)
                    Hpp[(size_t)(pk + a) * np + pk + c] += w * (Jp[a] * Jp[c] + Jp[6 + a] * Jp[6 + c]);
                for (int c = 0; c < 3; ++c)
                    W[(size_t)eblk[e] * 18 + a * 3 + c] += w * (Jp[a] * Jl[c] + Jp[6 + a] * Jl[3 + c]);
            }
            for (int a = 0; a < 3; ++a) {
                bl[3 * j + a] -= w * (Jl[a] * er[0] + Jl[3 + a] * er[1]);
                for (int c = 0; c < 3; ++c)
                    Hll[9 * j + a * 3 + c] += w * (Jl[a] * Jl[c] + Jl[3 + a] * Jl[3 + c]);
            }
        }
        if (it == 0) {
            double md = 0;
            for (int a = 0; a < np; ++a) if (fabs(Hpp[(size_t)a * np + a]) > md) md = fabs(Hpp[(size_t)a * np + a]);
            for (int j = 0; j < nlm; ++j) if (lm_act[j])
                for (int a = 0; a < 3; ++a) if (fabs(Hll[9 * j + a * 4]) > md) md = fabs(Hll[9 * j + a * 4]);
            lambda = (orc_whatif[1] ? 1e-3 : 1e-5) * md; ni = 2;
        }
        double rho = 0; int qmax = 0;
        do {
            memcpy(poses_b, poses, sizeof(double) * (size_t)nkf * 7);
            memcpy(pts_b, pts, sizeof(double) * (size_t)nlm * 3);
            /* Schur complement with lambda on every diagonal block */
            memcpy(S, Hpp, sizeof(double) * (size_t)np * np);
            for (int a = 0; a < np; ++a) S[(size_t)a * np + a] += lambda;
            memcpy(bs, bp, sizeof(double) * (size_t)np);
            for (int j = 0; j < nlm; ++j) {
                if (!lm_act[j]) continue;
                double D[9]; memcpy(D, Hll + 9 * j, sizeof(D));
                D[0] += lambda; D[4] += lambda; D[8] += lambda;
                double *Di = Dinv + 9 * j;
                inv3(D, Di);
                double db[3];
                for (int a = 0; a < 3; ++a)
                    db[a] = Di[a * 3] * bl[3 * j] + Di[a * 3 + 1] * bl[3 * j + 1] + Di[a * 3 + 2] * bl[3 * j + 2];
                for (int u = lm_start[j]; u < lm_start[j + 1]; ++u) {
                    int b1 = lm_blocks[u]; int p1 = 6 * kf_idx[blk_kf[b1]];
                    const double *W1 = W + (size_t)b1 * 18;
                    double BD[18];
                    for (int a = 0; a < 6; ++a)
                        for (int c = 0; c < 3; ++c)
                            BD[a * 3 + c] = W1[a * 3] * Di[c] + W1[a * 3 + 1] * Di[3 + c] + W1[a * 3 + 2] * Di[6 + c];
                    for (int a = 0; a < 6; ++a)
                        bs[p1 + a] -= W1[a * 3] * db[0] + W1[a * 3 + 1] * db[1] + W1[a * 3 + 2] * db[2];
                    for (int v = lm_start[j]; v < lm_start[j + 1]; ++v) {
                        int b2 = lm_blocks[v]; int p2 = 6 * kf_idx[blk_kf[b2]];
                        const double *W2 = W + (size_t)b2 * 18;
                        for (int a = 0; a < 6; ++a)
                            for (int c = 0; c < 6; ++c)
                                S[(size_t)(p1 + a) * np + p2 + c] -=
                                    BD[a * 3] * W2[c * 3] + BD[a * 3 + 1] * W2[c * 3 + 1] + BD[a * 3 + 2] * W2[c * 3 + 2];
                    }
                }
            }
            int ok2 = ldlt_solve(np, S, bs, xp);
            if (ok2) {
                /* back-substitution: xl = Dinv * (bl - W^T xp) */
                for (int j = 0; j < nlm; ++j) {
                    if (!lm_act[j]) { xl[3 * j] = xl[3 * j + 1] = xl[3 * j + 2] = 0; continue; }
                    double c3[3] = { bl[3 * j], bl[3 * j + 1], bl[3 * j + 2] };
                    for (int u = lm_start[j]; u < lm_start[j + 1]; ++u) {
                        int b1 = lm_blocks[u]; int p1 = 6 * kf_idx[blk_kf[b1]];
                        const double *W1 = W + (size_t)b1 * 18;
                        for (int a = 0; a < 6; ++a)
                            for (int c = 0; c < 3; ++c) c3[c] -= W1[a * 3 + c] * xp[p1 + a];
                    }
                    const double *Di = Dinv + 9 * j;
                    for (int a = 0; a < 3; ++a)
                        xl[3 * j + a] = Di[a * 3] * c3[0] + Di[a * 3 + 1] * c3[1] + Di[a * 3 + 2] * c3[2];
                }
            }
            /* update (g2o applies x even if the solve failed; x is then stale and
             * the step is rejected and popped, so the net effect is none) */
            if (ok2) {
                for (int k = 0; k < nkf; ++k) {
                    if (!kf_act[k]) continue;
                    double dT[7], Tn[7];
                    orc_se3_exp(xp + 6 * kf_idx[k], dT);
                    orc_se3_mul(dT, poses + 7 * k, Tn);
                    memcpy(poses + 7 * k, Tn, sizeof(Tn));
                }
                for (int j = 0; j < nlm; ++j) {
                    if (!lm_act[j]) continue;
                    pts[3 * j] += xl[3 * j]; pts[3 * j + 1] += xl[3 * j + 1]; pts[3 * j + 2] += xl[3 * j + 2];
                }
            }
            BA_ERRORS(tempChi);
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            double scale = 0;
            if (ok2) {
                for (int a = 0; a < np; ++a) scale += xp[a] * (lambda * xp[a] + bp[a]);
                for (int j = 0; j < nlm; ++j) if (lm_act[j])
                    for (int a = 0; a < 3; ++a) scale += xl[3 * j + a] * (lambda * xl[3 * j + a] + bl[3 * j + a]);
            }
            if (!orc_whatif[2]) scale += 1e-3;
            rho /= scale;
            if (trace && ntrace < trace_cap) {
                double *t = trace + (size_t)ORC_TRACE_REC * ntrace++;
                t[0] = it; t[1] = lambda; t[2] = currentChi; t[3] = tempChi; t[4] = rho;
                t[5] = (rho > 0 && isfinite(tempChi)) ? 1.0 : 0.0;
            }
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow(2 * rho - 1, 3);
                if (alpha > 2. / 3.) alpha = 2. / 3.;
                double sf = alpha < 1. / 3. ? 1. / 3. : alpha;
                lambda *= sf; ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                memcpy(poses, poses_b, sizeof(double) * (size_t)nkf * 7);
                memcpy(pts, pts_b, sizeof(double) * (size_t)nlm * 3);
                if (!isfinite(lambda)) break;
            }
            ++qmax;
        } while (rho < 0 && qmax < 10);
        ++it_done;
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) break;
    }
    for (int e = 0; e < nobs; ++e)
        edge_chi2[e] = err[2 * e] * err[2 * e] + err[2 * e + 1] * err[2 * e + 1];

    free(kf_act); free(lm_act); free(kf_idx); free(blk); free(eblk); free(blk_kf); free(blk_lm);
    free(lm_start); free(lm_blocks); free(fill); free(err); free(Hpp); free(S); free(bp); free(bs);
    free(Hll); free(Dinv); free(bl); free(W); free(xp); free(xl); free(poses_b); free(pts_b);
    if (trace_n) *trace_n = ntrace;
    return it_done;
}
