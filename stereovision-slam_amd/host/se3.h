// se3.h — host-side SE(3) in the memory layout of Sophus::SE3d
// (unit quaternion x,y,z,w + translation).  Stands in for the Sophus calls the
// reference makes in its host code: SE3d::operator*, inverse(), log()
// (src/frontend.cpp:604,655,685; src/map.cpp:92,108; src/backend.cpp:245).
#pragma once
#include <cmath>
#include <cstring>

namespace svs {

struct SE3 {
    double v[7]; // qx qy qz qw tx ty tz
    SE3() { v[0] = v[1] = v[2] = 0; v[3] = 1; v[4] = v[5] = v[6] = 0; }
    explicit SE3(const double *p) { std::memcpy(v, p, sizeof(v)); }

    static void rot(const double *q, const double *x, double *o)
    {
        double ux = q[1] * x[2] - q[2] * x[1];
        double uy = q[2] * x[0] - q[0] * x[2];
        double uz = q[0] * x[1] - q[1] * x[0];
        ux += ux; uy += uy; uz += uz;
        o[0] = x[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
        o[1] = x[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
        o[2] = x[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
    }
    void act(const double *p, double *o) const
    {
        rot(v, p, o);
        o[0] += v[4]; o[1] += v[5]; o[2] += v[6];
    }
    SE3 operator*(const SE3 &b) const
    {
        const double *A = v, *B = b.v;
        SE3 c;
        double ax = A[0], ay = A[1], az = A[2], aw = A[3];
        double bx = B[0], by = B[1], bz = B[2], bw = B[3];
        double q0 = aw * bx + ax * bw + ay * bz - az * by;
        double q1 = aw * by + ay * bw + az * bx - ax * bz;
        double q2 = aw * bz + az * bw + ax * by - ay * bx;
        double q3 = aw * bw - ax * bx - ay * by - az * bz;
        double n2 = q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
        if (n2 != 1.0) { double s = 2.0 / (1.0 + n2); q0 *= s; q1 *= s; q2 *= s; q3 *= s; }
        double t[3];
        rot(A, B + 4, t);
        c.v[0] = q0; c.v[1] = q1; c.v[2] = q2; c.v[3] = q3;
        c.v[4] = A[4] + t[0]; c.v[5] = A[5] + t[1]; c.v[6] = A[6] + t[2];
        return c;
    }
    SE3 inverse() const
    {
        SE3 r;
        r.v[0] = -v[0]; r.v[1] = -v[1]; r.v[2] = -v[2]; r.v[3] = v[3];
        double nt[3] = { -v[4], -v[5], -v[6] };
        rot(r.v, nt, r.v + 4);
        return r;
    }
    // norm of the 6-vector log (Map::RemoveOldKeyframe, src/map.cpp:108)
    double log_norm() const
    {
        const double EPS = 1e-10;
        double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], w = v[3];
        double two_atan;
        if (n2 < EPS * EPS) {
            double w2 = w * w;
            two_atan = 2.0 / w - (2.0 / 3.0) * n2 / (w * w2);
        } else {
            double n = std::sqrt(n2);
            if (std::fabs(w) < EPS) two_atan = (w > 0 ? M_PI : -M_PI) / n;
            else two_atan = 2.0 * std::atan(n / w) / n;
        }
        double om[3] = { two_atan * v[0], two_atan * v[1], two_atan * v[2] };
        double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
        double O[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
        double O2[9], Vi[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
        double c;
        if (std::fabs(theta) < EPS) c = 1.0 / 12.0;
        else {
            double half = 0.5 * theta;
            c = (1.0 - theta * std::cos(half) / (2.0 * std::sin(half))) / (theta * theta);
        }
        for (int i = 0; i < 9; ++i) Vi[i] = -0.5 * O[i] + c * O2[i];
        Vi[0] += 1; Vi[4] += 1; Vi[8] += 1;
        const double *t = v + 4;
        double u0 = Vi[0] * t[0] + Vi[1] * t[1] + Vi[2] * t[2];
        double u1 = Vi[3] * t[0] + Vi[4] * t[1] + Vi[5] * t[2];
        double u2 = Vi[6] * t[0] + Vi[7] * t[1] + Vi[8] * t[2];
        return std::sqrt(u0 * u0 + u1 * u1 + u2 * u2 + theta * theta);
    }
};

// Camera (src/camera.cpp): pinhole + rig->camera extrinsic
struct Camera {
    double fx = 0, fy = 0, cx = 0, cy = 0, baseline = 0;
    SE3 pose; // rig -> camera
    void k4(double *o) const { o[0] = fx; o[1] = fy; o[2] = cx; o[3] = cy; }
    // world2pixel (src/camera.cpp:74-80): camera2pixel(pose_ * T_c_w * p_w)
    void world2pixel(const double *pw, const SE3 &T_cw, double *uv) const
    {
        project(pose * T_cw, pw, uv); // pose_ * T_c_w * p_w associates left to right
    }
    // the same with the product pose_ * T_c_w formed once by the caller (loops over points)
    void project(const SE3 &T_cam_w, const double *pw, double *uv) const
    {
        double p[3];
        T_cam_w.act(pw, p);
        uv[0] = fx * p[0] / p[2] + cx;
        uv[1] = fy * p[1] / p[2] + cy;
    }
};

} // namespace svs
