// k_geom.h — stereo triangulation and pose-only Levenberg-Marquardt (f64).
// Replaces slam::triangulation() (reference include/StereoVisionSLAM/
// algorithm.h:10-87, called at src/frontend.cpp:165-174, 277-295) and the g2o
// problem of Frontend::EstimateCurrentPose (src/frontend.cpp:394-558).
// Mirrors oracle/orc_geom.c (same formulas, same LM control flow); the f64
// edge sums use a fixed wave butterfly instead of the oracle's sequential
// order, so agreement is to rounding (tolerances in tests/).
#pragma once
#include "dev_common.h"
#include "k_rt.h"
// f64 LM kernels: tolerance-level parity, FMA contraction allowed (see dev_common.h)
SVS_CONTRACT_FAST

// ------------------------------------------------------------------ triangulation
struct TriJob { int pt_ofs, npts; double T_wc[7]; double zmax; };
struct TriCams { double cam_l[4], ext_l[7], cam_r[4], ext_r[7]; };

// (Round 5: the rotation without IEEE divisions / roots — one root, one reciprocal, one reciprocal root by estimate + Newton —
// was built: 22 -> 19.5 us for one launch, results equal to rounding.  Not kept: it changes the last digit of a landmark, and a
// pipeline run that differs in a last digit parts ways with the CPU twin as early as one at pose_xtol 1e-9 does (4.4), for
// 0.2 % of the step.  The launch geometry below was the triangulation's real cost.)
// No FMA contraction inside the SVD (round 5): the oracle's arithmetic operation for operation — IEEE products, sums, quotients
// and roots round the same on both sides, so singular values and vectors are the oracle's bit for bit — and, what made it
// urgent, its CONVERGENCE: with contracted dot products the off-diagonal term of a converged pair sat a hair above the
// 2.3e-16 threshold in some lanes, the rotation it asked for changed nothing, and those lanes — hence their waves — ran all
// 60 sweeps instead of 4-6: 210 us per wave, ~320 us per launch of 440 keyframes instead of ~30.
#pragma clang fp contract(off)
__device__ inline void d_svd4_jacobi(double *A, double *V, double *sv)
{
    for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        bool changed = false;
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
                    const double ap1 = c * ap - s * aq, aq1 = s * ap + c * aq;
                    A[i * 4 + p] = ap1; A[i * 4 + q] = aq1;
                    double vp = V[i * 4 + p], vq = V[i * 4 + q];
                    const double vp1 = c * vp - s * vq, vq1 = s * vp + c * vq;
                    V[i * 4 + p] = vp1; V[i * 4 + q] = vq1;
                    changed |= (ap1 != ap) | (aq1 != aq) | (vp1 != vp) | (vq1 != vq);
                }
            }
        // A sweep that left every entry of A and V as it was is a fixed point: the remaining sweeps would repeat it bit for
        // bit, so leaving here changes no output.  It is what an EXACTLY singular system does (a stereo match whose row equals
        // the left feature's to the last bit makes two rows of A equal: the zero column's gamma is rounding noise above a zero
        // threshold, the rotation it asks for is the identity) — 1 match in ~10 000, but its wave ran all 60 sweeps, ~200 us,
        // and was the launch: k_triangulate 213 us per launch of ~440 keyframes (round 5: found by printing from the kernel).
        if (!rotated || !changed) break;
    }
    for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int i = 0; i < 4; ++i) s += A[i * 4 + j] * A[i * 4 + j];
        sv[j] = sqrt(s);
    }
    for (int j = 0; j < 3; ++j) {
        int m = j;
        for (int k = j + 1; k < 4; ++k) if (sv[k] > sv[m]) m = k;
        if (m != j) {
            double t = sv[j]; sv[j] = sv[m]; sv[m] = t;
            for (int i = 0; i < 4; ++i) { double v = V[i * 4 + j]; V[i * 4 + j] = V[i * 4 + m]; V[i * 4 + m] = v; }
        }
    }
}

SVS_CONTRACT_FAST

__global__ void __launch_bounds__(64)
k_triangulate(const TriJob *jobs, TriCams cams, const float2 *uv_l, const float2 *uv_r,
              double *out_xyz, uint8_t *out_ok)
{
    // grid (jobs, 64-point blocks of the capacity): consecutive workgroup ids — which the dispatcher deals round-robin to the
    // 8 XCDs — are different JOBS' blocks.  The other way round (blocks of a job consecutive) the capacity of 512 points = 8
    // blocks per job put block b of every job on XCD b, and a keyframe's ~160 points (3 blocks) on 3 of the 8 XCDs (round 5).
    const TriJob &jb = jobs[blockIdx.x];
    const int i = blockIdx.y * blockDim.x + threadIdx.x;
    if (i >= jb.npts) return;
    const int pt = jb.pt_ofs + i;
    const float2 l = uv_l[pt], r = uv_r[pt];
    // Camera::pixel2camera, depth 1
    double pl[2] = { ((double)l.x - cams.cam_l[2]) * 1.0 / cams.cam_l[0], ((double)l.y - cams.cam_l[3]) * 1.0 / cams.cam_l[1] };
    double pr[2] = { ((double)r.x - cams.cam_r[2]) * 1.0 / cams.cam_r[0], ((double)r.y - cams.cam_r[3]) * 1.0 / cams.cam_r[1] };
    double A[16], V[16], sv[4], R[9];
    d_quat_to_R(cams.ext_l, R);
    {
        double m[12] = { R[0], R[1], R[2], cams.ext_l[4], R[3], R[4], R[5], cams.ext_l[5], R[6], R[7], R[8], cams.ext_l[6] };
        for (int j = 0; j < 4; ++j) { A[j] = pl[0] * m[8 + j] - m[j]; A[4 + j] = pl[1] * m[8 + j] - m[4 + j]; }
    }
    d_quat_to_R(cams.ext_r, R);
    {
        double m[12] = { R[0], R[1], R[2], cams.ext_r[4], R[3], R[4], R[5], cams.ext_r[5], R[6], R[7], R[8], cams.ext_r[6] };
        for (int j = 0; j < 4; ++j) { A[8 + j] = pr[0] * m[8 + j] - m[j]; A[12 + j] = pr[1] * m[8 + j] - m[4 + j]; }
    }
    d_svd4_jacobi(A, V, sv);
    const double w = V[15];
    double p[3] = { V[3] / w, V[7] / w, V[11] / w };
    bool ok = (sv[3] / sv[2] < 1e-2) && (p[2] > 0) && (jb.zmax <= 0 || p[2] <= jb.zmax);
    double pw[3];
    d_se3_act(jb.T_wc, p, pw);
    out_xyz[3 * pt] = pw[0]; out_xyz[3 * pt + 1] = pw[1]; out_xyz[3 * pt + 2] = pw[2];
    out_ok[pt] = ok ? 1 : 0;
}

// ------------------------------------------------------------------ pose-only LM
struct PoseJob { int pt_ofs, npts; double pose[7]; int n_inlier; int pad; };

// 6x6 SPD solve (H + lambda I) x = b by an unpivoted, fully unrolled LDL^T kept in
// registers (no dynamically indexed arrays -> no scratch).  Eigen's LDLT pivots; for
// the positive definite damped system both give the solution to rounding.  Returns
// false when a pivot is not positive (g2o: solver failure -> step rejected).
// Executed redundantly by every lane on wave-uniform data.
__device__ __forceinline__ bool d_ldlt6(const double *H, const double *b, double *x)
{
    double L[6][6], D[6], Dinv[6], y[6];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double dk = H[k * 6 + k];
#pragma unroll
        for (int j = 0; j < k; ++j) dk -= L[k][j] * L[k][j] * D[j];
        D[k] = dk;
        ok = ok && (dk > 0);
        // reciprocal of the pivot: hardware estimate + two Newton steps (<= 1 ulp from the quotient; the six divisions sit
        // on the serial critical path of every LM trial).  A non-positive / denormal pivot gives inf or NaN here like the
        // division would downstream: the trial is rejected by the finiteness test on its chi2 either way.
#ifdef SVS_IEEE_DIV
        double inv = 1.0 / dk;
#else
        double inv = __builtin_amdgcn_rcp(dk);
        inv = __builtin_fma(__builtin_fma(-dk, inv, 1.0), inv, inv);
        inv = __builtin_fma(__builtin_fma(-dk, inv, 1.0), inv, inv);
#endif
        Dinv[k] = inv;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            double v = H[i * 6 + k];
#pragma unroll
            for (int j = 0; j < k; ++j) v -= L[i][j] * L[k][j] * D[j];
            L[i][k] = v * inv;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = b[i];
#pragma unroll
        for (int j = 0; j < i; ++j) v -= L[i][j] * y[j];
        y[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = y[i] * Dinv[i];       // the reciprocals of the pivots are already there
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double v = y[i];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) v -= L[j][i] * y[j];
        y[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = y[i];
    return ok;
}

#define PO_MAX_EDGES 512       // per job: 64 * PO_WAVES threads x (8 / PO_WAVES) register slots

// (Rt: the pose as rotation matrix (9, row major) + translation (3) — po_pose_table — so that the rig-frame point is nine
// FMAs per edge instead of the quaternion sandwich's two dozen operations; round 5)
__device__ __forceinline__ void po_pose_table(const double *T, double *Rt)
{
    d_quat_to_R(T, Rt);
    Rt[9] = T[4]; Rt[10] = T[5]; Rt[11] = T[6];
}
__device__ __forceinline__ void po_act(const double *Rt, const double *P, double *pc)
{
#pragma unroll
    for (int r = 0; r < 3; ++r) pc[r] = Rt[3 * r] * P[0] + Rt[3 * r + 1] * P[1] + Rt[3 * r + 2] * P[2] + Rt[9 + r];
}
__device__ __forceinline__ void po_error(const double *cam, const double *Rt, const double *P,
                                         double u, double v, double &e0, double &e1)
{
    double pc[3];
    po_act(Rt, P, pc);
    double px = cam[0] * pc[0] + cam[2] * pc[2];
    double py = cam[1] * pc[1] + cam[3] * pc[2];
    const double iz = d_rcp1(pc[2]);            // one reciprocal instead of two divisions (<= 1 ulp apart)
    e0 = u - px * iz;
    e1 = v - py * iz;
}

// ---- block sum of 32 f64 values per thread -------------------------------------------------------
// The normal equations of one LM step are 28 sums (21 + 6 + chi2) over all edges.  Reducing them one
// by one costs 28 x (4 DPP steps + 4 readlanes) ~ 640 VALU instructions per step, most of the wave's
// time.  Here the 16 lanes of a DPP row run a recursive-halving butterfly instead: at each of 4 steps
// a lane hands half of its values to its partner and accumulates the partner's copy of the half it
// keeps (16 + 8 + 4 + 2 = 30 adds), ending with the row totals of 2 of the 32 values.  The partners
// are the four single-instruction DPP permutations (xor 1, xor 2, row_half_mirror, row_mirror); the
// mirrors flip several lane bits at once, so the half a lane keeps at step k is selected by
//   s0 = b0^b2, s1 = b1^b2, s2 = b2^b3, s3 = b3   (b = bits of the lane id)
// which flips under the step's own permutation and is invariant under the later ones.  Row totals
// meet in LDS ([row of the block][32]); after one barrier every wave adds the rows in a fixed order
// and broadcasts the 28 totals with v_readlane, so all threads of the block get bit-identical sums.
template <int CTRL, int HALF>
__device__ __forceinline__ void po_bfly(double *v, bool sel)
{
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const double lo = v[i], hi = v[i + HALF];
        const double keep = sel ? hi : lo, send = sel ? lo : hi;
        v[i] = keep + dpp_f64<CTRL>(send);
    }
}

#ifndef PO_BCAST_W1
#define PO_BCAST_W1 0
#endif
#ifndef PO_BCAST_LDS        // A/B knob: 1 = the totals reach every lane through LDS broadcast reads, 0 = through v_readlane (SGPRs)
#define PO_BCAST_LDS 1
#endif
template <int WAVES>
__device__ __forceinline__ void po_block_sum32(double *v, double *buf, double *bcast, int tid)
{
    const int lane = tid & 63;
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    const bool s0 = b0 != b2, s1 = b1 != b2, s2 = b2 != b3, s3 = b3;
    po_bfly<SVS_DPP_XOR1, 16>(v, s0);
    po_bfly<SVS_DPP_XOR2, 8>(v, s1);
    po_bfly<SVS_DPP_HALF_MIRROR, 4>(v, s2);
    po_bfly<SVS_DPP_MIRROR, 2>(v, s3);
    const int code = (s0 ? 8 : 0) + (s1 ? 4 : 0) + (s2 ? 2 : 0) + (s3 ? 1 : 0);
    double2 *row = reinterpret_cast<double2 *>(buf + 32 * (tid >> 4));
    row[code] = make_double2(v[0], v[1]);
    __syncthreads();
    double tot = 0;
#pragma unroll
    for (int r = 0; r < 4 * WAVES; ++r) tot += buf[32 * r + (lane & 31)];
    if (PO_BCAST_LDS && (WAVES > 1 || PO_BCAST_W1)) {
    // the 28 totals to every lane: the wave parks them in its own 256 bytes of LDS and reads them back at wave-uniform
    // addresses (14 broadcast ds_read_b128).  They then live in VGPRs: v_readlane would put them into SGPRs, of which the
    // 6x6 algebra below wants more than there are (141 scalar spills = v_writelane / v_readlane pairs, and a v_mov for
    // every second operand of an f64 instruction).  Same values either way.
    double *mine = bcast + 32 * (tid >> 6);
    if (lane < 32) mine[lane] = tot;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        const double2 q = reinterpret_cast<const double2 *>(mine)[i];
        v[2 * i] = q.x; v[2 * i + 1] = q.y;
    }
    __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = readlane_f64(tot, i);
    }
}

template <int WAVES>
__device__ __forceinline__ double po_block_sum1(double x, double *buf, int tid)
{
    x = row_sum_f64(x);
    if ((tid & 15) == 0) buf[tid >> 4] = x;
    __syncthreads();
    double tot = 0;
#pragma unroll
    for (int r = 0; r < 4 * WAVES; ++r) tot += buf[r];
    return tot;
}

template <int WAVES>
__device__ __forceinline__ int po_block_sum_i32(int x, int *buf, int tid)
{
    x = wave_sum_i32(x);
    if (WAVES == 1) return x;
    __syncthreads();                 // previous readers of buf are done
    if ((tid & 63) == 0) buf[tid >> 6] = x;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int r = 0; r < WAVES; ++r) tot += buf[r];
    return tot;
}

// One block of WAVES wavefronts per job.  Edge e of the job lives in thread e % (64*WAVES),
// register slot e / (64*WAVES).  WAVES = 1 is the throughput shape (thousands of jobs per launch,
// least total work); WAVES = 4 the latency shape (a few jobs: one slot per thread up to 256 edges,
// the per-step critical path about half as long; the wave-uniform 6x6 algebra is simply repeated
// by every wave).  The two shapes sum in different orders, so they agree to rounding, not bit for bit.
// status_in (optional): edges whose status_in==0 are not part of the problem
// (used by the fused tracking path: LK failures / points without a map point).
#ifdef PO_NUM_VGPR          // A/B knob: cap of the kernel's unified registers (amdgpu_num_vgpr takes half of it on gfx90a+)
#define PO_VGPR_ATTR __attribute__((amdgpu_num_vgpr((PO_NUM_VGPR) / 2)))
#else
// (no cap: the kernel has to FIT two waves per SIMD — <= 256 unified registers — by itself.  The one-wave shape is a chain of
// dependent f64 instructions and runs a quarter slower with the SIMD to itself: 2048 copies of tools/po_trace.py's job (60
// trials) 277 us at 272 registers — what the kernel had grown to when the parameter tolerance was added — against 218 us at
// 228.  Forcing the limit (amdgpu_waves_per_eu) made the compiler spill two registers to scratch, and the FUSED instantiation
// then disagreed with the CPU twin in the pipeline tests — not pursued; the last trial's pose went to LDS instead: 228
// registers, nothing spilled.)
#define PO_VGPR_ATTR
#endif
// FUSED (resident tracking): the kernel also does what stands before and after the optimisation in a tracked frame — the
// survivor filter of TrackLastFrame (status && inside the image, src/frontend.cpp:361-371; an edge iff it also carries a map
// point, :443-444; k_track_filter as a kernel of its own) on the way in, and the hand-over of the survivors to the stream's
// other buffer (k_rt.h: rt_finish_wave) on the way out.  Same operations on the same values: results identical to the chain
// of three launches, two dependent launches (~10 us each for a lone camera) fewer.
struct PoFuse { uint8_t *status; const uint8_t *has_mp; int w, h; RtJob *rt; RtStore rs; float2 *out_xy; int *out_mp; };
template <int WAVES, bool FUSED>
__global__ void __launch_bounds__(64 * WAVES) PO_VGPR_ATTR
k_pose_only(PoseJob *jobs, const double *cam4, const double *xyz, const float2 *uv,
            const uint8_t *edge_valid, uint8_t *outlier, double chi2_th, int rounds, int iters, double *trace, PoFuse fz, double xtol)
{
    constexpr int NT = 64 * WAVES, SLOTS = PO_MAX_EDGES / NT;
    __shared__ __attribute__((aligned(16))) double s_red[4 * WAVES * 32];
    __shared__ __attribute__((aligned(16))) double s_bcast[WAVES * 32];
    __shared__ __attribute__((aligned(16))) double s_te[WAVES][8];
    __shared__ double s_one[2][4 * WAVES];
    __shared__ int s_int[WAVES];
    PoseJob &jb = jobs[blockIdx.x];                    // (may be pinned host memory, svslam_hip.hip:dpz — every field is read once)
    const int tid = threadIdx.x;
    const int n = jb.npts, jb_pt_ofs = jb.pt_ofs;
#ifdef PO_PROF     // development: clock ticks (100 MHz) per phase as two extra trace records
    long long po_t = wall_clock64(), po_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#define PO_TICK(i) do { long long t_ = wall_clock64(); po_acc[i] += t_ - po_t; po_t = t_; } while (0)
#define PO_TICKV(i, val) do { asm volatile("" :: "v"(val)); PO_TICK(i); } while (0)      // the phase's result is there first
#else
#define PO_TICK(i) do { } while (0)
#define PO_TICKV(i, val) do { } while (0)
#endif
    const double cam[4] = { cam4[0], cam4[1], cam4[2], cam4[3] };
    double T0[7], T[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { T0[i] = jb.pose[i]; T[i] = T0[i]; }

    // Per-edge state.  The first RS slots of a lane (edges e = s NT + tid, s < RS: 192 edges of a one-wave job, more than a
    // tracking frame has) keep position and pixel in registers; the slots beyond re-read them from memory in every pass (a
    // frame with more edges pays L2 loads, every other frame gets 80 registers back: with the residuals below, 335 -> under
    // 256 unified registers, i.e. TWO waves per SIMD — the wave is a chain of dependent f64 instructions at ~8 cycles each and
    // a second wave on the SIMD fills the gaps: 2048 jobs 357 -> 2xx us).  Valid / outlier flags are one bit per slot.
    // The residuals of the last evaluated trial, which the classification reads (g2o keeps the errors of its last
    // computeActiveErrors(), also those of a rejected trial), are not kept per edge: the pose of that trial is (Te), and the
    // classification re-evaluates po_error there — the same function on the same inputs, bit-identical.
    constexpr int RS = SLOTS < 3 ? SLOTS : 3;
    double P[RS][3], mu[RS], mv[RS];
    unsigned vmask = 0, omask = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int e = s * NT + tid;
        bool v = e < n;
        if (FUSED) {
            if (v) {
                const int pt = jb_pt_ofs + e;
                const float2 q = uv[pt];
                bool ok = fz.status[pt] != 0;
                if (q.y < 0.f || q.y >= (float)fz.h || q.x < 0.f || q.x >= (float)fz.w) ok = false;
                fz.status[pt] = ok ? 1 : 0;
                v = ok && fz.has_mp[pt] != 0;
            }
        } else if (v && edge_valid && !edge_valid[jb_pt_ofs + e]) v = false;
        vmask |= (v ? 1u : 0u) << s;
        if (s < RS) {
            if (e < n) {
                const int pt = jb_pt_ofs + e;
                P[s][0] = xyz[3 * pt]; P[s][1] = xyz[3 * pt + 1]; P[s][2] = xyz[3 * pt + 2];
                const float2 m = uv[pt];
                mu[s] = (double)m.x; mv[s] = (double)m.y;
            } else { P[s][0] = P[s][1] = 0; P[s][2] = 1; mu[s] = mv[s] = 0; }
        }
    }
    // position and pixel of the lane's edge in slot s (s a compile-time constant in the unrolled loops; only called for valid edges)
    auto edge = [&](int s, double *Pq, double &u_, double &v_) {
        if (s < RS) { Pq[0] = P[s < RS ? s : 0][0]; Pq[1] = P[s < RS ? s : 0][1]; Pq[2] = P[s < RS ? s : 0][2]; u_ = mu[s < RS ? s : 0]; v_ = mv[s < RS ? s : 0]; }
        else {
            const int pt = jb_pt_ofs + s * NT + tid;
            Pq[0] = xyz[3 * pt]; Pq[1] = xyz[3 * pt + 1]; Pq[2] = xyz[3 * pt + 2];
            const float2 m = uv[pt];
            u_ = (double)m.x; v_ = (double)m.y;
        }
    };
    // pose of the last evaluated LM trial: only the classification at the end of a round reads it, so it waits in LDS (the
    // wave's own 64 bytes, written by lane 0), not in 14 registers through the whole trial loop
    double *s_Te = s_te[tid >> 6];
    bool have_eval = false;
    bool robust = true;
    int cnt_outlier = 0, n_edges = 0, one = 0;
    n_edges = po_block_sum_i32<WAVES>(__builtin_popcount(vmask), s_int, tid);

    // A round restarts from T0 with the outlier flags of the previous classification and the robust flag: when that
    // classification changed no flag and the robust flag is the same, the round repeats the previous one operation for
    // operation (g2o recomputes lambda_0 from scratch in every optimize(), src/frontend.cpp:482-493) and ends in the
    // same pose, the same residuals and the same classification — it is not executed again.  In steady tracking rounds
    // 1 and 2 see the same outlier set, i.e. three rounds run instead of four; bit-identical results by construction.
    bool same_as_prev = false;
    for (int r = 0; r < rounds; ++r) {
        if (same_as_prev) {                    // (the flags stay unchanged again, so a further round repeats too)
            if (trace && tid == 0) lm_trace_replay(trace, blockIdx.x, 16 * (r - 1), 16 * r, 16);
            if (r == 2) { robust = false; same_as_prev = false; }
            continue;
        }
        // no trial of THIS round has been evaluated yet: a round that ends before its first trial (the parameter tolerance at
        // it == 0) classifies its active edges at the errors g2o's computeActiveErrors() took at the start estimate T0, not at
        // the previous round's last trial pose
        have_eval = false;
#pragma unroll
        for (int i = 0; i < 7; ++i) T[i] = T0[i];
        int nact = 0;
        nact = po_block_sum_i32<WAVES>(__builtin_popcount(vmask & ~omask), s_int, tid);
        if (nact > 0) {
            double lambda = 0, ni = 2;
            for (int it = 0; it < iters; ++it) {
                // errors + chi2 + normal equations at T: acc[0..20] upper triangle of H, [21..26] b, [27] chi2
                double acc[32];
                PO_TICK(5);
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = 0;
                double Rt[12];
                po_pose_table(T, Rt);
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    if (!(((vmask & ~omask) >> s) & 1)) continue;
                    double Pq[3], mu_, mv_;
                    edge(s, Pq, mu_, mv_);
                    double pc[3];
                    po_act(Rt, Pq, pc);
                    double X = pc[0], Y = pc[1], Z = pc[2];
                    double px = cam[0] * X + cam[2] * Z, py = cam[1] * Y + cam[3] * Z;
                    const double iz = d_rcp1(Z);
                    double ex = mu_ - px * iz, ey = mv_ - py * iz;
                    double e2 = ex * ex + ey * ey, w = 1.0, rho = e2;
                    if (robust) d_huber(e2, 1.0, rho, w);
                    acc[27] += rho;
                    const double Ze = Z + 1e-18;                       // g2o_types.h:159 (== Z unless |Z| < ~0.01)
                    double Zinv = Ze == Z ? iz : d_rcp1(Ze), Zinv2 = Zinv * Zinv;
                    double fx = cam[0], fy = cam[1];
                    double J0[6] = { -fx * Zinv, 0, fx * X * Zinv2, fx * X * Y * Zinv2, -fx - fx * X * X * Zinv2, fx * Y * Zinv };
                    double J1[6] = { 0, -fy * Zinv, fy * Y * Zinv2, fy + fy * Y * Y * Zinv2, -fy * X * Y * Zinv2, -fy * X * Zinv };
                    // H += w J^T J, b -= w J^T e by nested FMAs into the accumulators (round 5: `acc += w * (a * b + c * d)` was
                    // mul, fma, fma per entry), the products with the two structural zeros of the Jacobian (J0[1], J1[0]:
                    // g2o_types.h:159-162) left out: 50 instead of 81 instructions per edge
                    int k = 0;
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        const double w0 = w * J0[a], w1 = w * J1[a];       // (w0 of a = 1 and w1 of a = 0 are never used)
#pragma unroll
                        for (int c = a; c < 6; ++c) {
                            if (a != 0 && c != 0) acc[k] = __builtin_fma(w1, J1[c], acc[k]);        // row 1 of J: zero in column 0
                            if (a != 1 && c != 1) acc[k] = __builtin_fma(w0, J0[c], acc[k]);        // row 0 of J: zero in column 1
                            ++k;
                        }
                        if (a != 0) acc[21 + a] = __builtin_fma(-w1, ey, acc[21 + a]);
                        if (a != 1) acc[21 + a] = __builtin_fma(-w0, ex, acc[21 + a]);
                    }
                }
                PO_TICKV(0, acc[0] + acc[27]);
                po_block_sum32<WAVES>(acc, s_red, s_bcast, tid);
                PO_TICKV(1, acc[27] + acc[0]);
                double currentChi = acc[27];
                double H[36], b[6];
                {
                    int k = 0;
#pragma unroll
                    for (int a = 0; a < 6; ++a)
#pragma unroll
                        for (int c = a; c < 6; ++c) { H[a * 6 + c] = acc[k]; H[c * 6 + a] = acc[k]; ++k; }
#pragma unroll
                    for (int a = 0; a < 6; ++a) b[a] = acc[21 + a];
                }
                if (it == 0) {
                    double md = 0;
#pragma unroll
                    for (int a = 0; a < 6; ++a) md = fmax(md, fabs(H[a * 7]));
                    lambda = 1e-5 * md; ni = 2;
                }
                double rho = 0; int qmax = 0;
                double x[6] = { 0, 0, 0, 0, 0, 0 };
                do {
                    double Tb[7];
#pragma unroll
                    for (int i = 0; i < 7; ++i) Tb[i] = T[i];
                    double Hl[36];
#pragma unroll
                    for (int i = 0; i < 36; ++i) Hl[i] = H[i];
#pragma unroll
                    for (int a = 0; a < 6; ++a) Hl[a * 7] += lambda;
                    PO_TICK(5);
                    bool ok2 = d_ldlt6(Hl, b, x);
                    PO_TICKV(2, x[0] + x[5]);
                    // Parameter tolerance (xtol > 0; svslam_set_pose_only_xtol): the FIRST trial of an iteration whose damping is
                    // not above g2o's initial one for the present H (1e-5 max diag) is (nearly) the Newton step; when no component of it reaches xtol
                    // (metres / radians) the round stands at a stationary point of its cost and ends here, the step not taken.
                    // g2o has no such test (optimization_algorithm_levenberg.cpp: it goes on for the iterations asked for,
                    // src/frontend.cpp:487) — its remaining trials move the pose by rounding noise (|x| ~ 1e-13) and accept or
                    // reject on the sign of that noise; what is skipped is bounded by ~2 xtol, far inside the LM tolerances.
                    // (no state of its own: "not above the initial damping" is measured on the present H, and leaving the
                    // trial loop with rho still 0 ends the round through g2o's own rho == 0 exit below)
                    if (xtol > 0 && qmax == 0 && ok2) {
                        double mx = 0, md = 0;
#pragma unroll
                        for (int a = 0; a < 6; ++a) { mx = fmax(mx, fabs(x[a])); md = fmax(md, fabs(H[a * 7])); }
                        if (mx <= xtol && lambda <= 1e-5 * md) break;
                    }
                    double dT[7], Tn[7];
                    d_se3_exp(x, dT);
                    d_se3_mul(dT, T, Tn);
#pragma unroll
                    for (int i = 0; i < 7; ++i) T[i] = Tn[i];
                    if ((tid & 63) == 0) {
#pragma unroll
                        for (int i = 0; i < 7; ++i) s_Te[i] = Tn[i];
                    }
                    have_eval = true;
                    PO_TICKV(3, T[0] + T[6]);
                    double tchi = 0;
                    double Rtn[12];
                    po_pose_table(T, Rtn);
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) {
                        if (!(((vmask & ~omask) >> s) & 1)) continue;
                        double Pq[3], mu_, mv_, ea, eb;
                        edge(s, Pq, mu_, mv_);
                        po_error(cam, Rtn, Pq, mu_, mv_, ea, eb);
                        double e2 = ea * ea + eb * eb, w, rr = e2;
                        if (robust) d_huber(e2, 1.0, rr, w);
                        tchi += rr;
                    }
                    PO_TICKV(4, tchi);
                    // two alternating buffers: a rejected trial writes again before the next barrier
                    double tempChi = po_block_sum1<WAVES>(tchi, s_one[one], tid);
                    PO_TICKV(6, tempChi);
                    one ^= 1;
                    if (!ok2) tempChi = 1.7976931348623157e308;
                    rho = currentChi - tempChi;
                    double scale = 0;
#pragma unroll
                    for (int a = 0; a < 6; ++a) scale += x[a] * (lambda * x[a] + b[a]);
                    scale += 1e-3;
                    rho *= d_rcp1(scale);
                    if (trace && tid == 0) lm_trace_put(trace, blockIdx.x, 16 * r + it, lambda, currentChi, tempChi, rho, rho > 0 && isfinite(tempChi));
                    if (rho > 0 && isfinite(tempChi)) {
                        double t = 2 * rho - 1;
                        double alpha = 1. - t * t * t;
                        alpha = fmin(alpha, 2. / 3.);
                        double sf = fmax(1. / 3., alpha);
                        lambda *= sf; ni = 2; currentChi = tempChi;
                    } else {
                        lambda *= ni; ni *= 2;
#pragma unroll
                        for (int i = 0; i < 7; ++i) T[i] = Tb[i];
                        if (!isfinite(lambda)) break;
                    }
                    ++qmax;
                    PO_TICKV(7, lambda + rho);
                } while (rho < 0 && qmax < 10);
                if (qmax == 10 || rho == 0 || !isfinite(lambda)) break;
            }
        }
        // classify (src/frontend.cpp:495-525)
        int co = 0;
        double RtT[12], RtE[12], Te[7] = { 0, 0, 0, 1, 0, 0, 0 };
        if (have_eval) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 7; ++i) Te[i] = s_Te[i];
        } else {
#pragma unroll
            for (int i = 0; i < 7; ++i) Te[i] = T[i];          // (T is the round's start estimate then)
        }
        po_pose_table(T, RtT); po_pose_table(Te, RtE);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            if (!((vmask >> s) & 1)) continue;
            const bool was = (omask >> s) & 1;
            double Pq[3], mu_, mv_, ea = 0, eb = 0;
            edge(s, Pq, mu_, mv_);
            // an outlier of this round: its error at the round's result; an active edge: at the last evaluated trial
            if (was) po_error(cam, RtT, Pq, mu_, mv_, ea, eb);
            else po_error(cam, RtE, Pq, mu_, mv_, ea, eb);
            double chi2 = ea * ea + eb * eb;
            const bool now = chi2 > chi2_th;
            co += (now ? 1 : 0) + (now != was ? 1 << 16 : 0);           // outliers | flags changed << 16 (<= 512 edges)
            omask = (omask & ~(1u << s)) | ((now ? 1u : 0u) << s);
        }
        co = po_block_sum_i32<WAVES>(co, s_int, tid);
        cnt_outlier = co & 0xffff;
        same_as_prev = (co >> 16) == 0 && r != 2;                       // (round 3 drops the robust kernel: never a repeat)
        if (r == 2) robust = false;
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        int e = s * NT + tid;
        if (e < n) outlier[jb_pt_ofs + e] = ((vmask & omask) >> s) & 1;
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) jb.pose[i] = T[i];
        jb.n_inlier = n_edges - cnt_outlier;
#ifdef PO_PROF
        PO_TICK(5);
        if (trace) {
            lm_trace_put(trace, blockIdx.x, -1, (double)po_acc[0], (double)po_acc[1], (double)po_acc[2], (double)po_acc[3], false);
            lm_trace_put(trace, blockIdx.x, -2, (double)po_acc[4], (double)po_acc[5], (double)po_acc[6], (double)po_acc[7], false);
        }
#endif
    }
#undef PO_TICK
#undef PO_TICKV
    if (FUSED) {
        __syncthreads();            // the outlier flags of all threads, the filtered status bytes
        if (tid < 64) rt_finish_wave(fz.rt[blockIdx.x], fz.rs, uv, fz.status, outlier, xyz, fz.out_xy, fz.out_mp, tid);
    }
}

// ------------------------------------------------------------------ fused-track filter
// After LK: a point survives TrackLastFrame iff status && inside the image
// (src/frontend.cpp:361-371); it becomes a pose-only edge iff it also carries a
// map point (:443-444).  One block per job; writes the per-job survivor count.
struct LkJobView { int prev_slot, next_slot, pt_ofs, npts; };
__global__ void __launch_bounds__(256)
k_track_filter(const LkJobView *jobs, const float2 *next_xy, uint8_t *status, const uint8_t *has_mp,
               uint8_t *edge_valid, int *n_tracked, int w, int h)
{
    __shared__ int scnt[4];
    const LkJobView jb = jobs[blockIdx.x];
    int cnt = 0;
    for (int i = threadIdx.x; i < jb.npts; i += 256) {
        const int pt = jb.pt_ofs + i;
        const float2 p = next_xy[pt];
        bool ok = status[pt] != 0;
        if (p.y < 0.f || p.y >= (float)h || p.x < 0.f || p.x >= (float)w) ok = false;
        status[pt] = ok ? 1 : 0;
        edge_valid[pt] = (ok && has_mp[pt]) ? 1 : 0;
        cnt += ok ? 1 : 0;
    }
    cnt = wave_sum_i32(cnt);
    if ((threadIdx.x & 63) == 0) scnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) n_tracked[blockIdx.x] = scnt[0] + scnt[1] + scnt[2] + scnt[3];
}

#pragma clang fp contract(off)
// ---------------------------------------------------------------- device-resident tracking
// (RtJob / RtStore, the gather and the finish bodies: k_rt.h)
__global__ void __launch_bounds__(256)
k_rt_gather(const RtJob *jobs, RtStore rs, const double *cam, float2 *prev_xy, float2 *next_xy, uint8_t *has_mp, double *xyz)
{
    const RtJob jb = jobs[blockIdx.y];                 // a copy: the array may be pinned host memory
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < jb.npts) rt_gather_point(jb, rs, cam, i, prev_xy, next_xy, has_mp, xyz);
}

__global__ void __launch_bounds__(64)
k_rt_finish(RtJob *jobs, RtStore rs, const float2 *next_xy, const uint8_t *status, const uint8_t *outlier,
            const double *xyz, float2 *out_xy, int *out_mp)
{
    rt_finish_wave(jobs[blockIdx.x], rs, next_xy, status, outlier, xyz, out_xy, out_mp, threadIdx.x);
}

// host -> resident buffer (after a keyframe changed the frame's feature list or its map points)
struct RtUpJob { int stream, ofs, count, dst_buf; };
__global__ void __launch_bounds__(256)
k_rt_store(const RtUpJob *jobs, RtStore rs, const float2 *xy, const int *mp, const double *xyz)
{
    const RtUpJob jb = jobs[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= jb.count) return;
    const size_t d = (size_t)jb.stream * rs.max_pts + i, s = (size_t)jb.ofs + i;
    rs.xy[jb.dst_buf][d] = xy[s];
    rs.mp[jb.dst_buf][d] = mp[s];
    rs.xyz[jb.dst_buf][3 * d] = xyz[3 * s]; rs.xyz[jb.dst_buf][3 * d + 1] = xyz[3 * s + 1]; rs.xyz[jb.dst_buf][3 * d + 2] = xyz[3 * s + 2];
}
