// k_ba.h — local bundle adjustment: Levenberg-Marquardt with Schur complement,
// one persistent 1024-thread workgroup per problem, no host round trip per
// iteration.  Replaces optimizer.initializeOptimization(); optimizer.optimize(10)
// of Backend::Optimize (reference src/backend.cpp:22-164): g2o BlockSolver_6_3 +
// LinearSolverDense + OptimizationAlgorithmLevenberg, VertexXYZ marginalised,
// EdgeProjection (g2o_types.h:176-229) with Huber(delta).  Mirrors the control
// flow of oracle/orc_geom.c:orc_local_ba; Jacobians are analytic (the reference
// lets g2o differentiate numerically; see DESIGN.md for the tolerance).
//
// Data flow per LM iteration (all sums in fixed order -> deterministic):
//   edge pass      e, rho', Jp(2x6), Jl(2x3) per edge            (thread / edge)
//   landmark pass  Hll_j, bl_j, W_kj = sum w Jp^T Jl             (thread / landmark)
//   pose pass      Hpp_k, bp_k                                   (wave / pose)
//   per trial:  Dinv_j, Y_kj = W_kj Dinv_j                        (thread / landmark)
//               S_ab = Hpp+lambda - sum_j Y_aj W_bj^T             (wave / pose pair)
//               Cholesky + solves on the 6K x 6K system in LDS    (one wave)
//               back-substitution, update, new errors, rho test
// The reduced camera system is 60x60 f64 at K=10: MFMA does not apply.
#pragma once
#include "dev_common.h"
#include <vector>
#include <algorithm>

#define BA_THREADS 1024
#define BA_WAVES (BA_THREADS / 64)
#define BA_MAX_NP 192

struct BaJob { int kf_ofs, nkf, lm_ofs, nlm, obs_ofs, nobs, iters_done, reserved; };
struct BaCams { double cam[2][4]; double ext[2][7]; };

struct BaDev {               // device-side job descriptor (built on the host)
    int kf_ofs, nkf, lm_ofs, nlm, obs_ofs, nobs;
    int nblk, na;            // unique (kf,lm) blocks, active poses
    int aux_ofs;             // offset into the int aux buffer
    int iters_done;
};

struct BaWork {              // per-job HBM scratch, strided by the context limits
    int max_kf = 0, max_lm = 0, max_obs = 0;
    double *err = nullptr;   // [2*max_obs]
    double *Jp = nullptr;    // [12*max_obs]
    double *Jl = nullptr;    // [6*max_obs]
    double *wgt = nullptr;   // [max_obs]
    double *W = nullptr;     // [18*max_obs]
    double *Y = nullptr;     // [18*max_obs]
    double *Hpp = nullptr;   // [36*max_kf]
    double *bp = nullptr;    // [6*max_kf]
    double *Hll = nullptr;   // [9*max_lm]
    double *Dinv = nullptr;  // [9*max_lm]
    double *bl = nullptr;    // [3*max_lm]
    double *db = nullptr;    // [3*max_lm]
    double *xl = nullptr;    // [3*max_lm]
    double *poses_b = nullptr; // [7*max_kf]
    double *pts_b = nullptr;   // [3*max_lm]
    int *tbl = nullptr;      // [max_kf*max_lm]
    void *all = nullptr;
};

static inline hipError_t ba_work_alloc(BaWork &w, int jobs, int max_kf, int max_lm, int max_obs)
{
    w.max_kf = max_kf; w.max_lm = max_lm; w.max_obs = max_obs;
    if (max_kf <= 0 || max_lm <= 0 || max_obs <= 0) return hipSuccess;
    size_t J = jobs;
    size_t nd = J * ((size_t)max_obs * (2 + 12 + 6 + 1 + 18 + 18) + (size_t)max_kf * (36 + 6 + 7) +
                     (size_t)max_lm * (9 + 9 + 3 + 3 + 3 + 3));
    size_t ni = J * (size_t)max_kf * max_lm;
    hipError_t e = hipMalloc(&w.all, nd * sizeof(double) + ni * sizeof(int));
    if (e != hipSuccess) return e;
    double *p = static_cast<double *>(w.all);
    w.err = p; p += J * 2 * max_obs;
    w.Jp = p; p += J * 12 * max_obs;
    w.Jl = p; p += J * 6 * max_obs;
    w.wgt = p; p += J * max_obs;
    w.W = p; p += J * 18 * max_obs;
    w.Y = p; p += J * 18 * max_obs;
    w.Hpp = p; p += J * 36 * max_kf;
    w.bp = p; p += J * 6 * max_kf;
    w.poses_b = p; p += J * 7 * max_kf;
    w.Hll = p; p += J * 9 * max_lm;
    w.Dinv = p; p += J * 9 * max_lm;
    w.bl = p; p += J * 3 * max_lm;
    w.db = p; p += J * 3 * max_lm;
    w.xl = p; p += J * 3 * max_lm;
    w.pts_b = p; p += J * 3 * max_lm;
    w.tbl = reinterpret_cast<int *>(p);
    return hipSuccess;
}
static inline void ba_work_free(BaWork &w) { if (w.all) (void)hipFree(w.all); w.all = nullptr; }

// ---------------------------------------------------------------- host-side structure
// aux layout per job (ints):
//   lm_estart[nlm+1] | lm_edges[nobs] | kf_estart[nkf+1] | kf_edges[nobs] |
//   eblk[nobs] | lm_bstart[nlm+1] | blk_kf[nblk<=nobs, padded to nobs] | kf_pidx[nkf] | act_kf[nkf]
static inline size_t ba_aux_ints(int nkf, int nlm, int nobs)
{
    return (size_t)(nlm + 1) + nobs + (nkf + 1) + nobs + nobs + (nlm + 1) + nobs + nkf + nkf;
}

static inline void ba_build_aux(const BaJob &j, const int *obs_kf, const int *obs_lm, int *aux, BaDev &d)
{
    const int nkf = j.nkf, nlm = j.nlm, nobs = j.nobs;
    int *lm_estart = aux;
    int *lm_edges = lm_estart + nlm + 1;
    int *kf_estart = lm_edges + nobs;
    int *kf_edges = kf_estart + nkf + 1;
    int *eblk = kf_edges + nobs;
    int *lm_bstart = eblk + nobs;
    int *blk_kf = lm_bstart + nlm + 1;
    int *kf_pidx = blk_kf + nobs;
    int *act_kf = kf_pidx + nkf;
    const int *okf = obs_kf + j.obs_ofs, *olm = obs_lm + j.obs_ofs;
    // edges sorted by (lm, kf, edge id): counting sort by lm after stable sort by kf
    std::vector<int> order(nobs);
    for (int e = 0; e < nobs; ++e) order[e] = e;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (olm[a] != olm[b]) return olm[a] < olm[b];
        return okf[a] < okf[b];
    });
    for (int i = 0; i <= nlm; ++i) lm_estart[i] = 0;
    for (int e = 0; e < nobs; ++e) lm_estart[olm[e] + 1]++;
    for (int i = 0; i < nlm; ++i) lm_estart[i + 1] += lm_estart[i];
    int nblk = 0;
    for (int i = 0; i <= nlm; ++i) lm_bstart[i] = 0;
    int prev_lm = -1, prev_kf = -1;
    for (int i = 0; i < nobs; ++i) {
        int e = order[i];
        lm_edges[i] = e;
        if (olm[e] != prev_lm || okf[e] != prev_kf) {
            blk_kf[nblk] = okf[e];
            lm_bstart[olm[e] + 1]++;
            ++nblk;
            prev_lm = olm[e]; prev_kf = okf[e];
        }
        eblk[e] = nblk - 1;
    }
    for (int i = 0; i < nlm; ++i) lm_bstart[i + 1] += lm_bstart[i];
    // edges by pose (edge id ascending inside a pose)
    for (int i = 0; i <= nkf; ++i) kf_estart[i] = 0;
    for (int e = 0; e < nobs; ++e) kf_estart[okf[e] + 1]++;
    for (int i = 0; i < nkf; ++i) kf_estart[i + 1] += kf_estart[i];
    std::vector<int> fill(kf_estart, kf_estart + nkf);
    for (int e = 0; e < nobs; ++e) kf_edges[fill[okf[e]]++] = e;
    int na = 0;
    for (int k = 0; k < nkf; ++k) {
        if (kf_estart[k + 1] > kf_estart[k]) { kf_pidx[k] = na; act_kf[na] = k; ++na; }
        else kf_pidx[k] = -1;
    }
    for (int k = na; k < nkf; ++k) act_kf[k] = -1;
    d.kf_ofs = j.kf_ofs; d.nkf = nkf; d.lm_ofs = j.lm_ofs; d.nlm = nlm; d.obs_ofs = j.obs_ofs; d.nobs = nobs;
    d.nblk = nblk; d.na = na; d.iters_done = 0;
}

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ void ba_project(const BaCams &c, int cam, const double *T, const double *P,
                                           double *q, double *p)
{
    d_se3_act(T, P, q);
    d_se3_act(c.ext[cam], q, p);
}

__device__ __forceinline__ double block_sum(double v, double *red, int tid)
{
    // wave butterfly, then the 16 wave partials summed in fixed order
    v = wave_sum_f64(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int i = 0; i < BA_WAVES; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ double block_max(double v, double *red, int tid)
{
    v = wave_max_f64(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int i = 1; i < BA_WAVES; ++i) s = fmax(s, red[i]);
    return s;
}

__device__ __forceinline__ void d_inv3(const double *A, double *Ai)
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

__global__ void __launch_bounds__(BA_THREADS)
k_local_ba(BaDev *jobs, const BaCams *camsp, double *poses_all, double *pts_all, const int *obs_kf_all,
           const int *obs_lm_all, const uint8_t *obs_right_all, const float2 *obs_uv_all, const int *aux_all,
           BaWork wk, double delta, int iters, double *edge_chi2_all)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS carve (all dynamic): S[np*np] | bs[np] | xp[np] | red[16] | flags
    const int job = blockIdx.x;
    BaDev &jd = jobs[job];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nkf = jd.nkf, nlm = jd.nlm, nobs = jd.nobs, na = jd.na, np = 6 * jd.na;
    if (nobs <= 0 || na <= 0) { if (tid == 0) jd.iters_done = 0; return; }
    const int ld = np + 1;   // odd row stride (in doubles): spreads LDS banks
    double *S = reinterpret_cast<double *>(smem);
    double *bs = S + (size_t)np * ld;
    double *xp = bs + np;
    double *red = xp + np;
    int *iflag = reinterpret_cast<int *>(red + BA_WAVES);

    const BaCams cams = *camsp;
    double *poses = poses_all + (size_t)jd.kf_ofs * 7;
    double *pts = pts_all + (size_t)jd.lm_ofs * 3;
    const int *okf = obs_kf_all + jd.obs_ofs, *olm = obs_lm_all + jd.obs_ofs;
    const uint8_t *oright = obs_right_all + jd.obs_ofs;
    const float2 *ouv = obs_uv_all + jd.obs_ofs;
    double *edge_chi2 = edge_chi2_all + jd.obs_ofs;
    const int *aux = aux_all + jd.aux_ofs;
    const int *lm_estart = aux;
    const int *lm_edges = lm_estart + nlm + 1;
    const int *kf_estart = lm_edges + nobs;
    const int *kf_edges = kf_estart + nkf + 1;
    const int *eblk = kf_edges + nobs;
    const int *lm_bstart = eblk + nobs;
    const int *blk_kf = lm_bstart + nlm + 1;
    const int *kf_pidx = blk_kf + nobs;
    const int *act_kf = kf_pidx + nkf;

    const size_t J = job;
    double *err = wk.err + J * 2 * wk.max_obs;
    double *Jp = wk.Jp + J * 12 * wk.max_obs;
    double *Jl = wk.Jl + J * 6 * wk.max_obs;
    double *wgt = wk.wgt + J * wk.max_obs;
    double *W = wk.W + J * 18 * wk.max_obs;
    double *Y = wk.Y + J * 18 * wk.max_obs;
    double *Hpp = wk.Hpp + J * 36 * wk.max_kf;
    double *bp = wk.bp + J * 6 * wk.max_kf;
    double *Hll = wk.Hll + J * 9 * wk.max_lm;
    double *Dinv = wk.Dinv + J * 9 * wk.max_lm;
    double *bl = wk.bl + J * 3 * wk.max_lm;
    double *db = wk.db + J * 3 * wk.max_lm;
    double *xl = wk.xl + J * 3 * wk.max_lm;
    double *poses_b = wk.poses_b + J * 7 * wk.max_kf;
    double *pts_b = wk.pts_b + J * 3 * wk.max_lm;
    int *tbl = wk.tbl + J * (size_t)wk.max_kf * wk.max_lm;

    // (kf,lm) -> block table
    for (int i = tid; i < nkf * nlm; i += BA_THREADS) tbl[i] = -1;
    __syncthreads();
    for (int j = tid; j < nlm; j += BA_THREADS)
        for (int b = lm_bstart[j]; b < lm_bstart[j + 1]; ++b) tbl[blk_kf[b] * nlm + j] = b;
    __syncthreads();

    auto compute_errors = [&]() -> double {
        double chi = 0;
        for (int e = tid; e < nobs; e += BA_THREADS) {
            const int cam = oright[e] ? 1 : 0;
            double q[3], p[3];
            ba_project(cams, cam, poses + 7 * okf[e], pts + 3 * olm[e], q, p);
            const double *K = cams.cam[cam];
            double px = K[0] * p[0] + K[2] * p[2], py = K[1] * p[1] + K[3] * p[2];
            double ex = (double)ouv[e].x - px / p[2], ey = (double)ouv[e].y - py / p[2];
            err[2 * e] = ex; err[2 * e + 1] = ey;
            double r0, r1;
            d_huber(ex * ex + ey * ey, delta, r0, r1);
            chi += r0;
        }
        return block_sum(chi, red, tid);
    };

    double lambda = 0, ni = 2;
    int it_done = 0;
    for (int it = 0; it < iters; ++it) {
        double currentChi = compute_errors();
        double tempChi = currentChi;
        // ---- buildSystem: edge pass
        for (int e = tid; e < nobs; e += BA_THREADS) {
            const int cam = oright[e] ? 1 : 0;
            const double *T = poses + 7 * okf[e];
            double q[3], p[3], Re[9], R[9];
            ba_project(cams, cam, T, pts + 3 * olm[e], q, p);
            d_quat_to_R(cams.ext[cam], Re);
            d_quat_to_R(T, R);
            const double *K = cams.cam[cam];
            double X = p[0], Yc = p[1], Z = p[2];
            double zi = 1.0 / Z, zi2 = zi * zi;
            double E[6] = { -K[0] * zi, 0, K[0] * X * zi2, 0, -K[1] * zi, K[1] * Yc * zi2 };
            double qh[9] = { 0, q[2], -q[1], -q[2], 0, q[0], q[1], -q[0], 0 };
            double A[18];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    A[i * 6 + j] = Re[i * 3 + j];
                    A[i * 6 + 3 + j] = Re[i * 3] * qh[j] + Re[i * 3 + 1] * qh[3 + j] + Re[i * 3 + 2] * qh[6 + j];
                }
            double *jp = Jp + 12 * (size_t)e, *jl = Jl + 6 * (size_t)e;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    jp[r * 6 + j] = E[r * 3] * A[j] + E[r * 3 + 1] * A[6 + j] + E[r * 3 + 2] * A[12 + j];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double m0 = Re[0] * R[j] + Re[1] * R[3 + j] + Re[2] * R[6 + j];
                    double m1 = Re[3] * R[j] + Re[4] * R[3 + j] + Re[5] * R[6 + j];
                    double m2 = Re[6] * R[j] + Re[7] * R[3 + j] + Re[8] * R[6 + j];
                    jl[r * 3 + j] = E[r * 3] * m0 + E[r * 3 + 1] * m1 + E[r * 3 + 2] * m2;
                }
            double r0, r1;
            d_huber(err[2 * e] * err[2 * e] + err[2 * e + 1] * err[2 * e + 1], delta, r0, r1);
            wgt[e] = r1;
        }
        __syncthreads();
        // ---- landmark pass: Hll, bl, W blocks
        for (int j = tid; j < nlm; j += BA_THREADS) {
            double h[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, b3[3] = { 0, 0, 0 };
            int ebeg = lm_estart[j], eend = lm_estart[j + 1];
            int i = ebeg;
            while (i < eend) {
                const int blk = eblk[lm_edges[i]];
                double wacc[18];
#pragma unroll
                for (int t = 0; t < 18; ++t) wacc[t] = 0;
                while (i < eend && eblk[lm_edges[i]] == blk) {
                    const int e = lm_edges[i];
                    const double *jp = Jp + 12 * (size_t)e, *jl = Jl + 6 * (size_t)e;
                    const double w = wgt[e], ex = err[2 * e], ey = err[2 * e + 1];
#pragma unroll
                    for (int a = 0; a < 6; ++a)
#pragma unroll
                        for (int c = 0; c < 3; ++c) wacc[a * 3 + c] += w * (jp[a] * jl[c] + jp[6 + a] * jl[3 + c]);
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        b3[a] -= w * (jl[a] * ex + jl[3 + a] * ey);
#pragma unroll
                        for (int c = 0; c < 3; ++c) h[a * 3 + c] += w * (jl[a] * jl[c] + jl[3 + a] * jl[3 + c]);
                    }
                    ++i;
                }
#pragma unroll
                for (int t = 0; t < 18; ++t) W[18 * (size_t)blk + t] = wacc[t];
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) Hll[9 * j + t] = h[t];
            bl[3 * j] = b3[0]; bl[3 * j + 1] = b3[1]; bl[3 * j + 2] = b3[2];
        }
        // ---- pose pass: Hpp (block diagonal), bp   (wave per active pose)
        for (int a = wv; a < na; a += BA_WAVES) {
            const int k = act_kf[a];
            double acc[27];
#pragma unroll
            for (int t = 0; t < 27; ++t) acc[t] = 0;
            for (int i = kf_estart[k] + lane; i < kf_estart[k + 1]; i += 64) {
                const int e = kf_edges[i];
                const double *jp = Jp + 12 * (size_t)e;
                const double w = wgt[e], ex = err[2 * e], ey = err[2 * e + 1];
                int t = 0;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
#pragma unroll
                    for (int c = r; c < 6; ++c) { acc[t] += w * (jp[r] * jp[c] + jp[6 + r] * jp[6 + c]); ++t; }
                }
#pragma unroll
                for (int r = 0; r < 6; ++r) acc[21 + r] -= w * (jp[r] * ex + jp[6 + r] * ey);
            }
#pragma unroll
            for (int t = 0; t < 27; ++t) acc[t] = wave_sum_f64(acc[t]);
            if (lane == 0) {
                int t = 0;
                for (int r = 0; r < 6; ++r)
                    for (int c = r; c < 6; ++c) { Hpp[36 * a + r * 6 + c] = acc[t]; Hpp[36 * a + c * 6 + r] = acc[t]; ++t; }
                for (int r = 0; r < 6; ++r) bp[6 * a + r] = acc[21 + r];
            }
        }
        __syncthreads();
        if (it == 0) {
            double md = 0;
            for (int i = tid; i < np; i += BA_THREADS) md = fmax(md, fabs(Hpp[36 * (i / 6) + (i % 6) * 7]));
            for (int i = tid; i < 3 * nlm; i += BA_THREADS)
                if (lm_estart[i / 3 + 1] > lm_estart[i / 3]) md = fmax(md, fabs(Hll[9 * (i / 3) + (i % 3) * 4]));
            md = block_max(md, red, tid);
            lambda = 1e-5 * md; ni = 2;
        }
        double rho = 0; int qmax = 0;
        do {
            // backup
            for (int i = tid; i < 7 * nkf; i += BA_THREADS) poses_b[i] = poses[i];
            for (int i = tid; i < 3 * nlm; i += BA_THREADS) pts_b[i] = pts[i];
            // Dinv, db, Y
            for (int j = tid; j < nlm; j += BA_THREADS) {
                if (lm_estart[j + 1] == lm_estart[j]) continue;
                double D[9], Di[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) D[t] = Hll[9 * j + t];
                D[0] += lambda; D[4] += lambda; D[8] += lambda;
                d_inv3(D, Di);
#pragma unroll
                for (int t = 0; t < 9; ++t) Dinv[9 * j + t] = Di[t];
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    db[3 * j + a] = Di[a * 3] * bl[3 * j] + Di[a * 3 + 1] * bl[3 * j + 1] + Di[a * 3 + 2] * bl[3 * j + 2];
                for (int b = lm_bstart[j]; b < lm_bstart[j + 1]; ++b) {
                    const double *w1 = W + 18 * (size_t)b;
                    double *y = Y + 18 * (size_t)b;
#pragma unroll
                    for (int a = 0; a < 6; ++a)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            y[a * 3 + c] = w1[a * 3] * Di[c] + w1[a * 3 + 1] * Di[3 + c] + w1[a * 3 + 2] * Di[6 + c];
                }
            }
            // S = blockdiag(Hpp) + lambda I ; bs = bp
            for (int i = tid; i < np * np; i += BA_THREADS) {
                int r = i / np, c = i - r * np;
                double v = 0;
                if (r / 6 == c / 6) v = Hpp[36 * (r / 6) + (r % 6) * 6 + (c % 6)];
                if (r == c) v += lambda;
                S[(size_t)r * ld + c] = v;
            }
            __syncthreads();
            // pose-pair pass: S_ab -= sum_j Y_aj W_bj^T ; bs_a = bp_a - sum_j W_aj db_j
            const int npairs = na * (na + 1) / 2;
            for (int pidx = wv; pidx < npairs + na; pidx += BA_WAVES) {
                if (pidx < npairs) {
                    int a = 0, rem = pidx;
                    while (rem >= na - a) { rem -= na - a; ++a; }
                    const int b = a + rem;
                    const int *ta = tbl + act_kf[a] * nlm, *tb = tbl + act_kf[b] * nlm;
                    double acc[36];
#pragma unroll
                    for (int t = 0; t < 36; ++t) acc[t] = 0;
                    for (int j = lane; j < nlm; j += 64) {
                        const int i1 = ta[j], i2 = tb[j];
                        if (i1 < 0 || i2 < 0) continue;
                        const double *y = Y + 18 * (size_t)i1, *w2 = W + 18 * (size_t)i2;
#pragma unroll
                        for (int r = 0; r < 6; ++r)
#pragma unroll
                            for (int c = 0; c < 6; ++c)
                                acc[r * 6 + c] += y[r * 3] * w2[c * 3] + y[r * 3 + 1] * w2[c * 3 + 1] + y[r * 3 + 2] * w2[c * 3 + 2];
                    }
#pragma unroll
                    for (int t = 0; t < 36; ++t) acc[t] = wave_sum_f64(acc[t]);
                    if (lane == 0) {
                        for (int r = 0; r < 6; ++r)
                            for (int c = 0; c < 6; ++c) {
                                S[(size_t)(6 * a + r) * ld + 6 * b + c] -= acc[r * 6 + c];
                                if (a != b) S[(size_t)(6 * b + c) * ld + 6 * a + r] -= acc[r * 6 + c];
                            }
                    }
                } else {
                    const int a = pidx - npairs;
                    const int *ta = tbl + act_kf[a] * nlm;
                    double acc[6] = { 0, 0, 0, 0, 0, 0 };
                    for (int j = lane; j < nlm; j += 64) {
                        const int i1 = ta[j];
                        if (i1 < 0) continue;
                        const double *w1 = W + 18 * (size_t)i1;
#pragma unroll
                        for (int r = 0; r < 6; ++r)
                            acc[r] += w1[r * 3] * db[3 * j] + w1[r * 3 + 1] * db[3 * j + 1] + w1[r * 3 + 2] * db[3 * j + 2];
                    }
#pragma unroll
                    for (int r = 0; r < 6; ++r) acc[r] = wave_sum_f64(acc[r]);
                    if (lane == 0)
                        for (int r = 0; r < 6; ++r) bs[6 * a + r] = bp[6 * a + r] - acc[r];
                }
            }
            __syncthreads();
            // Cholesky S = L L^T (lower, in place) and the two triangular solves: wave 0.
            if (wv == 0) {
                int ok = 1;
                for (int k = 0; k < np; ++k) {
                    // left-looking: column k
                    for (int i = k + lane; i < np; i += 64) {
                        double v = S[(size_t)i * ld + k];
                        for (int m = 0; m < k; ++m) v -= S[(size_t)i * ld + m] * S[(size_t)k * ld + m];
                        S[(size_t)i * ld + k] = v; // un-normalised
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    double d = S[(size_t)k * ld + k];
                    if (!(d > 0)) { ok = 0; break; }
                    double sd = sqrt(d);
                    for (int i = k + lane; i < np; i += 64) S[(size_t)i * ld + k] = (i == k) ? sd : S[(size_t)i * ld + k] / sd;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                }
                if (ok) {
                    // forward: L y = bs  (y overwrites xp)
                    for (int i = lane; i < np; i += 64) xp[i] = bs[i];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    for (int k = 0; k < np; ++k) {
                        double yk = xp[k] / S[(size_t)k * ld + k];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        for (int i = k + lane; i < np; i += 64) {
                            if (i == k) xp[i] = yk;
                            else xp[i] -= S[(size_t)i * ld + k] * yk;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    }
                    // backward: L^T x = y
                    for (int k = np - 1; k >= 0; --k) {
                        double xk = xp[k] / S[(size_t)k * ld + k];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        for (int i = lane; i <= k; i += 64) {
                            if (i == k) xp[i] = xk;
                            else xp[i] -= S[(size_t)k * ld + i] * xk;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    }
                }
                if (lane == 0) iflag[0] = ok;
            }
            __syncthreads();
            const int ok2 = iflag[0];
            double scale_part = 0;
            if (ok2) {
                // back-substitution + landmark update
                for (int j = tid; j < nlm; j += BA_THREADS) {
                    if (lm_estart[j + 1] == lm_estart[j]) continue;
                    double c3[3] = { bl[3 * j], bl[3 * j + 1], bl[3 * j + 2] };
                    for (int b = lm_bstart[j]; b < lm_bstart[j + 1]; ++b) {
                        const double *w1 = W + 18 * (size_t)b;
                        const double *x6 = xp + 6 * kf_pidx[blk_kf[b]];
#pragma unroll
                        for (int a = 0; a < 6; ++a)
#pragma unroll
                            for (int c = 0; c < 3; ++c) c3[c] -= w1[a * 3 + c] * x6[a];
                    }
                    const double *Di = Dinv + 9 * j;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        double x = Di[a * 3] * c3[0] + Di[a * 3 + 1] * c3[1] + Di[a * 3 + 2] * c3[2];
                        xl[3 * j + a] = x;
                        pts[3 * j + a] += x;
                        scale_part += x * (lambda * x + bl[3 * j + a]);
                    }
                }
                for (int a = tid; a < na; a += BA_THREADS) {
                    const int k = act_kf[a];
                    double dT[7], Tn[7], x6[6];
#pragma unroll
                    for (int t = 0; t < 6; ++t) { x6[t] = xp[6 * a + t]; scale_part += x6[t] * (lambda * x6[t] + bp[6 * a + t]); }
                    d_se3_exp(x6, dT);
                    d_se3_mul(dT, poses + 7 * k, Tn);
#pragma unroll
                    for (int t = 0; t < 7; ++t) poses[7 * k + t] = Tn[t];
                }
            }
            double scale = block_sum(scale_part, red, tid);
            __syncthreads();
            tempChi = compute_errors();
            if (!ok2) tempChi = 1.7976931348623157e308;
            rho = currentChi - tempChi;
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(tempChi)) {
                double t = 2 * rho - 1;
                double alpha = 1. - t * t * t;
                alpha = fmin(alpha, 2. / 3.);
                double sf = fmax(1. / 3., alpha);
                lambda *= sf; ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                for (int i = tid; i < 7 * nkf; i += BA_THREADS) poses[i] = poses_b[i];
                for (int i = tid; i < 3 * nlm; i += BA_THREADS) pts[i] = pts_b[i];
                __syncthreads();
                if (!isfinite(lambda)) break;
            }
            ++qmax;
        } while (rho < 0 && qmax < 10);
        ++it_done;
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) break;
    }
    __syncthreads();
    for (int e = tid; e < nobs; e += BA_THREADS) edge_chi2[e] = err[2 * e] * err[2 * e] + err[2 * e + 1] * err[2 * e + 1];
    if (tid == 0) jd.iters_done = it_done;
}

static inline size_t ba_lds_bytes(int max_kf)
{
    size_t np = 6 * (size_t)max_kf;
    return (np * (np + 1) + 2 * np + BA_WAVES) * sizeof(double) + 64;
}
