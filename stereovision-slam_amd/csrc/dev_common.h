// dev_common.h — shared device helpers for the gfx950 kernels.
// All float code in this library is compiled with -ffp-contract=off: the
// integer/f32 paths (pyramids, LK, GFTT) are bit-exact against the declared
// operation order of the oracle, so no FMA contraction is allowed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
    // fixed butterfly order: deterministic
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { double u = __shfl_xor(v, o, 64); v = u > v ? u : v; }
    return v;
}

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
        double half = 0.5 * theta;
        imag = sin(half) / theta;
        real = cos(half);
    }
    T[0] = imag * om[0]; T[1] = imag * om[1]; T[2] = imag * om[2]; T[3] = real;
    double V[9];
    if (theta < EPS) {
        d_quat_to_R(T, V);
    } else {
        double a = (1.0 - cos(theta)) / th2;
        double b = (theta - sin(theta)) / (th2 * theta);
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
    else { double sq = sqrt(e2); rho0 = 2 * sq * delta - dsqr; rho1 = delta / sq; }
}
