/*
 * synth_traj.h — ground-truth rig trajectory of the synthetic stream and the
 * per-frame view parameters handed to the renderer (synth_scene.h).
 * Host-side, double precision; shared by the CPU and the HIP generator so both
 * render exactly the same views.
 */
#ifndef SVS_SYNTH_TRAJ_H
#define SVS_SYNTH_TRAJ_H
#include "synth_scene.h"

/* KITTI-00 calibration (from memory), full resolution */
#define SVS_KITTI00_FX 718.856
#define SVS_KITTI00_CX 607.1928
#define SVS_KITTI00_CY 185.2157
#define SVS_KITTI00_BASELINE 0.537166

static inline void svs_quat_from_R(const double R[9], double q[4])
{
    double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double s = sqrt(tr + 1.0) * 2;
        q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
        q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
        q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s;
    } else {
        double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
        q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s;
    }
}

/* rig (left camera) -> world: rotation R_wc (row-major) and centre C */
static inline void svs_synth_rig(uint32_t seed, int frame, double R[9], double C[3])
{
    double k = (double)frame;
    double ph1 = (double)(svs_hash3(seed, 1u, 7u) & 0xFFFF) * (6.283185307179586 / 65536.0);
    double ph2 = (double)(svs_hash3(seed, 2u, 7u) & 0xFFFF) * (6.283185307179586 / 65536.0);
    double ph3 = (double)(svs_hash3(seed, 3u, 7u) & 0xFFFF) * (6.283185307179586 / 65536.0);
    C[0] = 1.3 * sin(0.031 * k + ph1);
    C[1] = 0.06 * sin(0.11 * k + ph2);
    C[2] = 0.85 * k + 1.5 * sin(0.05 * k + ph3);
    double dxdk = 1.3 * 0.031 * cos(0.031 * k + ph1);
    double dzdk = 0.85 + 1.5 * 0.05 * cos(0.05 * k + ph3);
    double yaw = atan2(dxdk, dzdk) + 0.02 * sin(0.027 * k + ph2);
    double pitch = 0.012 * sin(0.09 * k + ph3);
    double roll = 0.010 * sin(0.07 * k + ph1);
    double cy = cos(yaw), sy = sin(yaw), cp = cos(pitch), sp = sin(pitch), cr = cos(roll), sr = sin(roll);
    /* R = Ry(yaw) * Rx(pitch) * Rz(roll), camera axes: x right, y down, z forward */
    double Ry[9] = { cy, 0, sy, 0, 1, 0, -sy, 0, cy };
    double Rx[9] = { 1, 0, 0, 0, cp, -sp, 0, sp, cp };
    double Rz[9] = { cr, -sr, 0, sr, cr, 0, 0, 0, 1 };
    double M[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            M[i * 3 + j] = Ry[i * 3] * Rx[j] + Ry[i * 3 + 1] * Rx[3 + j] + Ry[i * 3 + 2] * Rx[6 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[i * 3 + j] = M[i * 3] * Rz[j] + M[i * 3 + 1] * Rz[3 + j] + M[i * 3 + 2] * Rz[6 + j];
}

/* ground-truth T_cw of frame `frame` relative to frame 0 (T_c(k) <- c(0)),
 * Sophus layout qx qy qz qw tx ty tz */
static inline void svs_synth_gt_pose(uint32_t seed, int frame, double T[7])
{
    double R0[9], C0[3], Rk[9], Ck[3];
    svs_synth_rig(seed, 0, R0, C0);
    svs_synth_rig(seed, frame, Rk, Ck);
    /* T_k<-0 : R = Rk^T R0 ; t = Rk^T (C0 - Ck) */
    double Rr[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Rr[i * 3 + j] = Rk[0 * 3 + i] * R0[0 * 3 + j] + Rk[1 * 3 + i] * R0[1 * 3 + j] + Rk[2 * 3 + i] * R0[2 * 3 + j];
    double d[3] = { C0[0] - Ck[0], C0[1] - Ck[1], C0[2] - Ck[2] };
    svs_quat_from_R(Rr, T);
    for (int i = 0; i < 3; ++i) T[4 + i] = Rk[0 * 3 + i] * d[0] + Rk[1 * 3 + i] * d[1] + Rk[2 * 3 + i] * d[2];
}

/* cam = fx, fy, cx, cy of the rendered (already decimated) image */
static inline void svs_synth_views(uint32_t seed, int frame, const double cam[4],
                                   double baseline, svs_synth_view *vl,
                                   svs_synth_view *vr)
{
    double R[9], C[3];
    svs_synth_rig(seed, frame, R, C);
    svs_synth_view *v[2] = { vl, vr };
    for (int c = 0; c < 2; ++c) {
        v[c]->fx = (float)cam[0]; v[c]->fy = (float)cam[1];
        v[c]->cx = (float)cam[2]; v[c]->cy = (float)cam[3];
        for (int i = 0; i < 9; ++i) v[c]->R[i] = (float)R[i];
        double off = c ? baseline : 0.0; /* right camera centre = C + R * (b,0,0) */
        v[c]->C[0] = C[0] + R[0] * off;
        v[c]->C[1] = C[1] + R[3] * off;
        v[c]->C[2] = C[2] + R[6] * off;
        v[c]->seed = seed;
        v[c]->noise_seed = svs_hash3(seed, (uint32_t)frame, (uint32_t)(c + 11));
        v[c]->scale = 1.f;
    }
}

#endif
