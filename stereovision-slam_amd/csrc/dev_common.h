// dev_common.h — shared device helpers for the gfx950 kernels.
// All float code in this library is compiled with -ffp-contract=off: the
// integer/f32 paths (pyramids, LK, GFTT) are bit-exact against the declared
// operation order of the oracle, so no FMA contraction is allowed.
#pragma once
// The f64 LM code opts into FMA contraction through this macro.  -DSVS_NO_CONTRACT (parity debugging, tests/ate_bias.py)
// builds it without: no contraction, and the explicit __builtin_fma calls become a multiplication and an addition.
#ifdef SVS_NO_CONTRACT
#define SVS_CONTRACT_FAST _Pragma("clang fp contract(off)")
#else
#define SVS_CONTRACT_FAST _Pragma("clang fp contract(fast)")
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#ifdef SVS_NO_CONTRACT
#define __builtin_fma(a, b, c) ((a) * (b) + (c))      // (after the runtime's headers: only this library's kernels)
#endif

#define SVS_WAVE 64
#define SVS_BORDER 16          // stored REFLECT_101 border of every pyramid level
#define SVS_LEVELS 4

struct PyrGeom {
    int w[SVS_LEVELS], h[SVS_LEVELS];
    int pitch[SVS_LEVELS];          // bytes per padded row
    size_t ofs[SVS_LEVELS];         // byte offset of the padded level inside a slot
    size_t slot_bytes;
    int nlevels;
};

__device__ __forceinline__ int reflect101(int p, int len)
{
    // cv::borderInterpolate(BORDER_REFLECT_101)
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

// pixel (0,0) of level l in a slot
__device__ __forceinline__ const uint8_t *lvl_origin(const uint8_t *slot, const PyrGeom &g, int l)
{
    return slot + g.ofs[l] + (size_t)SVS_BORDER * g.pitch[l] + SVS_BORDER;
}
__device__ __forceinline__ uint8_t *lvl_origin(uint8_t *slot, const PyrGeom &g, int l)
{
    return slot + g.ofs[l] + (size_t)SVS_BORDER * g.pitch[l] + SVS_BORDER;
}

// ---- wave reductions (all lanes end with the total) -------------------
// DPP row operations (quad_perm xor1, xor2, row_half_mirror, row_mirror) give every
// lane the sum of its 16-lane row in 4 VALU steps with no LDS traffic; the four row
// totals are then combined through v_readlane (SGPR broadcast).  Requires a fully
// active wave.  The tree is fixed, so f64 sums are deterministic.
#define SVS_DPP_XOR1 0xB1        // quad_perm:[1,0,3,2]
#define SVS_DPP_XOR2 0x4E        // quad_perm:[2,3,0,1]
#define SVS_DPP_HALF_MIRROR 0x141
#define SVS_DPP_MIRROR 0x140

// bound_ctrl:1 (a lane without a source reads 0): the same values as `old = 0` gives, but the compiler no longer has to write
// that 0 into the destination first — one v_mov less per v_mov_dpp, i.e. 2 instead of 4 instructions per f64 exchange in every
// butterfly of k_local_ba / k_ba_ll / k_pose_only and per wave shift of k_gftt_eig3 (round 5)
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ long long dpp_i64(long long v)
{
    int lo = (int)(v & 0xffffffffll), hi = (int)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ long long readlane_i64(long long v, int lane)
{
    int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), lane);
    int hi = __builtin_amdgcn_readlane((int)(v >> 32), lane);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// sum over each 16-lane row (every lane of the row gets it)
__device__ __forceinline__ double row_sum_f64(double v)
{
    v += dpp_f64<SVS_DPP_XOR1>(v);
    v += dpp_f64<SVS_DPP_XOR2>(v);
    v += dpp_f64<SVS_DPP_HALF_MIRROR>(v);
    v += dpp_f64<SVS_DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ int row_sum_i32(int v)     // every lane: sum over its 16-lane DPP row
{
    v += dpp_i32<SVS_DPP_XOR1>(v);
    v += dpp_i32<SVS_DPP_XOR2>(v);
    v += dpp_i32<SVS_DPP_HALF_MIRROR>(v);
    v += dpp_i32<SVS_DPP_MIRROR>(v);
    return v;
}
// int32 lanes whose 16-lane partial sums fit int32 but whose total may not: row sums on the
// VALU (4 fused DPP adds), the last 4-term sum in 64-bit scalar arithmetic
__device__ __forceinline__ long long wave_sum_i32_wide(int v)
{
    v = row_sum_i32(v);
    return ((long long)__builtin_amdgcn_readlane(v, 0) + (long long)__builtin_amdgcn_readlane(v, 16)) +
           ((long long)__builtin_amdgcn_readlane(v, 32) + (long long)__builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += dpp_i32<SVS_DPP_XOR1>(v);
    v += dpp_i32<SVS_DPP_XOR2>(v);
    v += dpp_i32<SVS_DPP_HALF_MIRROR>(v);
    v += dpp_i32<SVS_DPP_MIRROR>(v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
           (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
    v += dpp_i64<SVS_DPP_XOR1>(v);
    v += dpp_i64<SVS_DPP_XOR2>(v);
    v += dpp_i64<SVS_DPP_HALF_MIRROR>(v);
    v += dpp_i64<SVS_DPP_MIRROR>(v);
    return (readlane_i64(v, 0) + readlane_i64(v, 16)) + (readlane_i64(v, 32) + readlane_i64(v, 48));
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
    v = row_sum_f64(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ double wave_max_f64(double v)
{
    v = fmax(v, dpp_f64<SVS_DPP_XOR1>(v));
    v = fmax(v, dpp_f64<SVS_DPP_XOR2>(v));
    v = fmax(v, dpp_f64<SVS_DPP_HALF_MIRROR>(v));
    v = fmax(v, dpp_f64<SVS_DPP_MIRROR>(v));
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// Bit-for-bit twin of host/se3.h SE3::act and Camera::project (no contraction here): the
// LK start guess of a tracked map point must equal the host's float, because LK is integer
// arithmetic downstream of it.
__device__ __forceinline__ void d_project_exact(const double *T, const double *K, const double *x, double *uv)
{
    const double *q = T;
    double ux = q[1] * x[2] - q[2] * x[1];
    double uy = q[2] * x[0] - q[0] * x[2];
    double uz = q[0] * x[1] - q[1] * x[0];
    ux += ux; uy += uy; uz += uz;
    double p0 = x[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    double p1 = x[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    double p2 = x[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
    p0 += T[4]; p1 += T[5]; p2 += T[6];
    uv[0] = K[0] * p0 / p2 + K[2];
    uv[1] = K[1] * p1 / p2 + K[3];
}

// The f64 geometry below is compared to the oracle at a stated tolerance, not bit for bit,
// so FMA contraction is allowed for it (halves the f64 op count); the integer / f32 code
// above and in k_pyramid/k_lk/k_gftt stays strictly un-contracted.
SVS_CONTRACT_FAST
// ---- SE(3), Sophus layout qx qy qz qw tx ty tz (mirrors oracle/orc_geom.c) ----
__device__ __forceinline__ void d_quat_rot(const double *q, const double *v, double *o)
{
    double ux = q[1] * v[2] - q[2] * v[1];
    double uy = q[2] * v[0] - q[0] * v[2];
    double uz = q[0] * v[1] - q[1] * v[0];
    ux += ux; uy += uy; uz += uz;
    o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}
__device__ __forceinline__ void d_se3_act(const double *T, const double *p, double *o)
{
    d_quat_rot(T, p, o);
    o[0] += T[4]; o[1] += T[5]; o[2] += T[6];
}
__device__ __forceinline__ void d_quat_to_R(const double *q, double *R)
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
__device__ __forceinline__ void d_se3_mul(const double *A, const double *B, double *C)
{
    double ax = A[0], ay = A[1], az = A[2], aw = A[3];
    double bx = B[0], by = B[1], bz = B[2], bw = B[3];
    double q0 = aw * bx + ax * bw + ay * bz - az * by;
    double q1 = aw * by + ay * bw + az * bx - ax * bz;
    double q2 = aw * bz + az * bw + ax * by - ay * bx;
    double q3 = aw * bw - ax * bx - ay * by - az * bz;
    double n2 = q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
    if (n2 != 1.0) { double s = 2.0 / (1.0 + n2); q0 *= s; q1 *= s; q2 *= s; q3 *= s; }
    double t[3];
    d_quat_rot(A, B + 4, t);
    C[0] = q0; C[1] = q1; C[2] = q2; C[3] = q3;
    C[4] = A[4] + t[0]; C[5] = A[5] + t[1]; C[6] = A[6] + t[2];
}
// sin and cos for the rotation steps of an LM update: |x| < 0.5 almost always, where the Taylor
// sums below are exact to the last place or two (truncation x^19/19!, x^18/18! < 1e-21); the
// library routine (argument reduction, several branches) is only the fallback.
// 1 / x: hardware estimate + two Newton steps (<= 1 ulp of the correctly rounded quotient).  The IEEE division expands to
// a dozen dependent instructions; the LM kernels have several on the serial critical path of every trial.  0, inf, NaN
// and denormals (estimate inf) keep the raw estimate, i.e. what the division returns.
// -DSVS_IEEE_DIV (build.py: SVS_IEEE_DIV=1) is a parity-debugging build: every estimate-plus-Newton reciprocal / root of the
// LM kernels (here, d_huber, d_ldlt6's pivots, the BA Cholesky's pivots) becomes the correctly rounded IEEE operation again,
// so that a trajectory can be compared with the CPU oracle without the <= 1 ulp differences of the fast forms.
__device__ __forceinline__ double d_rcp1(double x)
{
#ifdef SVS_IEEE_DIV
    return 1.0 / x;
#endif
    const double r0 = __builtin_amdgcn_rcp(x);
    double r = __builtin_fma(__builtin_fma(-x, r0, 1.0), r0, r0);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return (fabs(r0) < __builtin_inf() && r0 != 0.0) ? r : r0;
}
__device__ __forceinline__ void d_sincos_small(double x, double &s, double &c)
{
    if (fabs(x) < 0.5) {
        const double z = x * x;
        double ps = -1.0 / 355687428096000.0;                 // -1/17!
        ps = ps * z + 1.0 / 1307674368000.0;                  //  1/15!
        ps = ps * z - 1.0 / 6227020800.0;                     // -1/13!
        ps = ps * z + 1.0 / 39916800.0;                       //  1/11!
        ps = ps * z - 1.0 / 362880.0;                         // -1/9!
        ps = ps * z + 1.0 / 5040.0;                           //  1/7!
        ps = ps * z - 1.0 / 120.0;                            // -1/5!
        ps = ps * z + 1.0 / 6.0;                              //  1/3!
        s = x - x * z * ps;
        double pc = 1.0 / 20922789888000.0;                   //  1/16!
        pc = pc * z - 1.0 / 87178291200.0;                    // -1/14!
        pc = pc * z + 1.0 / 479001600.0;                      //  1/12!
        pc = pc * z - 1.0 / 3628800.0;                        // -1/10!
        pc = pc * z + 1.0 / 40320.0;                          //  1/8!
        pc = pc * z - 1.0 / 720.0;                            // -1/6!
        pc = pc * z + 1.0 / 24.0;                             //  1/4!
        c = 1.0 - z * (0.5 - z * pc);
    } else {
        s = sin(x); c = cos(x);
    }
}
__device__ __forceinline__ void d_se3_exp(const double *xi, double *T)
{
    const double EPS = 1e-10;
    const double *u = xi, *om = xi + 3;
    double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    double theta, imag, real;
    if (th2 < EPS * EPS) {
        theta = 0;
        double th4 = th2 * th2;
        imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
    } else {
        theta = sqrt(th2);
        double sh, ch;
        d_sincos_small(0.5 * theta, sh, ch);
        imag = sh * d_rcp1(theta);
        real = ch;
    }
    T[0] = imag * om[0]; T[1] = imag * om[1]; T[2] = imag * om[2]; T[3] = real;
    double V[9];
    if (theta < EPS) {
        d_quat_to_R(T, V);
    } else {
        double st, ct;
        d_sincos_small(theta, st, ct);
        double a = (1.0 - ct) * d_rcp1(th2);
        double b = (theta - st) * d_rcp1(th2 * theta);
        double O[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double o2 = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
                V[i * 3 + j] = a * O[i * 3 + j] + b * o2;
            }
        V[0] += 1; V[4] += 1; V[8] += 1;
    }
    T[4] = V[0] * u[0] + V[1] * u[1] + V[2] * u[2];
    T[5] = V[3] * u[0] + V[4] * u[1] + V[5] * u[2];
    T[6] = V[6] * u[0] + V[7] * u[1] + V[8] * u[2];
}

// g2o RobustKernelHuber::robustify -> (rho, rho')
__device__ __forceinline__ void d_huber(double e2, double delta, double &rho0, double &rho1)
{
    double dsqr = delta * delta;
    if (e2 <= dsqr) { rho0 = e2; rho1 = 1.0; }
#ifndef SVS_IEEE_DIV
    else if (e2 < 1e300) {
        // 1 / sqrt(e2): hardware estimate + two Newton steps, one correction of the root (e2 > delta^2: normal range)
        double y = __builtin_amdgcn_rsq(e2);
        y = y * (1.5 - 0.5 * e2 * y * y);
        y = y * (1.5 - 0.5 * e2 * y * y);
        double sq = e2 * y;
        sq = __builtin_fma(__builtin_fma(-sq, sq, e2), 0.5 * y, sq);
        rho0 = 2 * sq * delta - dsqr; rho1 = delta * y;
    }
#endif
    else { double sq = sqrt(e2); rho0 = 2 * sq * delta - dsqr; rho1 = delta / sq; }
}
#pragma clang fp contract(off)

// LM trajectory test hook (svslam_lm_trace): per job [0] = trials recorded, records of LM_TRACE_REC doubles from
// [8]: iteration (pose-only: 16 round + iteration), lambda of the trial, chi2 before, chi2 of the trial state,
// rho, accepted — the layout of the oracle's orc_*_trace.  Written by thread 0 of the job's workgroup.
#define LM_TRACE_REC 6
#define LM_TRACE_CAP 408
#define LM_TRACE_STRIDE (8 + LM_TRACE_REC * LM_TRACE_CAP)
__device__ __forceinline__ void lm_trace_put(double *trace_all, int job, double it, double lambda, double chi0, double chi1,
                                             double rho, bool accepted)
{
    double *t = trace_all + (size_t)job * LM_TRACE_STRIDE;
    const int n = (int)t[0];
    if (n >= LM_TRACE_CAP) return;
    double *r = t + 8 + LM_TRACE_REC * n;
    r[0] = it; r[1] = lambda; r[2] = chi0; r[3] = chi1; r[4] = rho; r[5] = accepted ? 1.0 : 0.0;
    t[0] = (double)(n + 1);
}
// the records with lo <= iteration < hi once more, iteration + add (a pose-only round that repeats the previous one)
__device__ __forceinline__ void lm_trace_replay(double *trace_all, int job, int lo, int hi, int add)
{
    double *t = trace_all + (size_t)job * LM_TRACE_STRIDE;
    const int n = (int)t[0];
    int m = n;
    for (int i = 0; i < n && m < LM_TRACE_CAP; ++i) {
        const double *r = t + 8 + LM_TRACE_REC * i;
        if (r[0] < lo || r[0] >= hi) continue;
        double *w = t + 8 + LM_TRACE_REC * m;
        w[0] = r[0] + add; w[1] = r[1]; w[2] = r[2]; w[3] = r[3]; w[4] = r[4]; w[5] = r[5];
        ++m;
    }
    t[0] = (double)m;
}
