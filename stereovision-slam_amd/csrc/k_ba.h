// k_ba.h — local bundle adjustment: Levenberg-Marquardt with Schur complement,
// one persistent 512-thread workgroup per problem, no host round trip per
// iteration.  Replaces optimizer.initializeOptimization(); optimizer.optimize(10)
// of Backend::Optimize (reference src/backend.cpp:22-164): g2o BlockSolver_6_3 +
// LinearSolverDense + OptimizationAlgorithmLevenberg, VertexXYZ marginalised,
// EdgeProjection (g2o_types.h:176-229) with Huber(delta).  Mirrors the control
// flow of oracle/orc_geom.c:orc_local_ba; Jacobians are analytic (the reference
// lets g2o differentiate numerically; see DESIGN.md for the tolerance).
//
// Data flow (every sum has a fixed order -> deterministic).  Nothing but the edge records, the
// poses and the landmark positions is read from global memory inside an LM iteration: the 6x3
// blocks W, Hll, Hll^-1 and bl live in LDS one landmark tile at a time and are recomputed from
// the records whenever they are needed (a Jacobian is ~70 FMA; a round trip of the 18 doubles
// of a block costs 9 L1 transactions per lane, and the kernel is bound by those: PMC
// TCP_TOTAL_CACHE_ACCESSES ~ 27 per VMEM instruction in the version that stored W).
//   per LM iteration:
//     pose pass      16-lane rows over the pose-major edge records -> Hpp_k, bp_k (LDS)
//   per LM trial, for each tile of <= BA_TILE landmarks / blocks (landmarks are renumbered by
//   descending block count, a tile is a run of them):
//     tile pass      thread / landmark walks its edge records: residual, rho', Jp, Jl ->
//                    Hll_j, bl_j, (Hll_j + lambda I)^-1 and the blocks W_kj -> LDS
//     Schur pass     16-lane row / pose pair over the tile's block pairs (listed by the host,
//                    tile-local indices): S_ab -= (W_aj Dinv_j) W_bj^T, bs_a -= W_aj Dinv_j bl_j
//   then Cholesky + triangular solves of the 6K x 6K system in LDS (whole workgroup),
//   back-substitution (thread / landmark, Jacobians recomputed:
//   dl = -Dinv sum Jl^T w (r + Jp dp)), update, error pass, rho test.
// Rotations / translations of the active poses and both cameras live in an LDS table that
// is rebuilt whenever the state changes, so a projection costs no global gather.
// The reduced camera system is 60x60 f64 at K=10: MFMA does not apply.
#pragma once
#include "dev_common.h"
#include <vector>
#include <algorithm>
#include <cstring>

// f64 LM algebra, compared with the oracle at stated tolerances: FMA contraction allowed
SVS_CONTRACT_FAST

#ifndef BA_THREADS
#define BA_THREADS 512
#endif
#ifndef BA_MIN_WAVES_PER_SIMD
#define BA_MIN_WAVES_PER_SIMD 2
#endif
#ifndef BA_LDS_LIMIT
#define BA_LDS_LIMIT (160 * 1024)   // A/B knob: leave LDS to co-resident kernels (no gain measured)
#endif
#define BA_WAVES (BA_THREADS / 64)
#define BA_ROWS (BA_THREADS / 16)
#define BA_MAX_NP 192
#define BA_PIT_CAP 2048          // block-pair items of a tile staged in LDS (more are read from global)
#define BA_TILE_MAX 480             // landmarks and (pose, landmark) blocks per LDS tile of the Schur sweep
#define BB_MAXKEYS 33               // keys of the landmark renumbering sort (k_ba_build.h keeps [key][thread] counters in LDS)

struct BaJob { int kf_ofs, nkf, lm_ofs, nlm, obs_ofs, nobs, iters_done, reserved; };
struct BaCams { double cam[2][4]; double ext[2][7]; };

// edge record, 16 bytes: measurement, landmark | (active pose << 1 | camera) << 24, block id
struct BaRec { float u, v; int lmkc; int blk; };
#define BA_LM_MASK 0x00ffffff

struct BaDev {               // device-side job descriptor (built on the host)
    int kf_ofs, nkf, lm_ofs, nlm, obs_ofs, nobs;
    int nblk, na;            // unique (kf,lm) blocks, active poses
    int ncontrib;            // (Y,W) block pairs
    int ntile;               // LDS tiles of the Schur sweep
    int aux_ofs;             // offset into the int aux buffer
    int iters_done;
    int rec_ofs;             // offset (records) of this job's 2*nobs records: landmark-major, then pose-major
    int lay_nblk, lay_na, lay_ntile;   // counts the aux LAYOUT was reserved for: the actual ones (host build) or
                                       // their upper bounds (device build, k_ba_build.h)
    int nmv;                 // landmarks [0, nmv) (internal numbering) go through the LDS tiles; the single-view
                             // landmarks behind them are grouped by pose (sv_start) and folded in by the row pass
    int reserved;
    int lm_base;             // low-latency shards (k_ba_split): the packed edges name landmarks of the PARENT problem;
                             // this shard owns [lm_base, lm_base + nlm) of them
    int shmask;              // low-latency shards: bit v set = shard v of the problem has edges (takes part in the exchanges)
    int ntrial;              // out: LM trials executed (accepted + rejected); with ncontrib (block pairs of the Schur
                             // complement) the accounting of the compute-side roofline (bench.py)
};

struct BaWork {              // per-job HBM scratch, strided by the context limits
    int max_kf = 0, max_lm = 0, max_obs = 0;
    double *err = nullptr;   // [2*max_obs]  residuals of the last evaluation, landmark-major edge order
    double *poses_b = nullptr; // [7*max_kf]
    double *poses_a = nullptr; // [7*max_kf]  low-latency shards: the shard's own copy of the current poses
    double *pts_b = nullptr;   // [3*max_lm]
    double *pts_i = nullptr;   // [3*max_lm]  landmark positions in the internal numbering
    void *all = nullptr;
};

static inline hipError_t ba_work_alloc(BaWork &w, int jobs, int max_kf, int max_lm, int max_obs)
{
    w.max_kf = max_kf; w.max_lm = max_lm; w.max_obs = max_obs;
    if (max_kf <= 0 || max_lm <= 0 || max_obs <= 0) return hipSuccess;
    size_t J = jobs;
    size_t nd = J * ((size_t)max_obs * 2 + (size_t)max_kf * 14 + (size_t)max_lm * 6);
    hipError_t e = hipMalloc(&w.all, nd * sizeof(double));
    if (e != hipSuccess) return e;
    double *p = static_cast<double *>(w.all);
    w.err = p; p += J * 2 * max_obs;
    w.poses_b = p; p += J * 7 * max_kf;
    w.poses_a = p; p += J * 7 * max_kf;
    w.pts_b = p; p += J * 3 * max_lm;
    w.pts_i = p; p += J * 3 * max_lm;
    return hipSuccess;
}
static inline void ba_work_free(BaWork &w) { if (w.all) (void)hipFree(w.all); w.all = nullptr; }

// ---------------------------------------------------------------- host-side structure
// aux layout per job (ints), offsets from ba_aux_layout():
//   lm_estart[nlm+1] lm_edges[nobs] kf_estart[nkf+1] lm_orig[nlm]
//   lm_bstart[nlm+1] blk_kf[nblk] blk_lm[nblk] blk_es[nblk+1] kf_pidx[nkf] act_kf[nkf] sv_start[nkf+1]
//   tile_lm[ntile+1]  pcs[ntile*npairs+1]  pitem[ncontrib] (y | w << 10 | landmark << 20, tile-local)
struct BaAuxLayout {
    size_t lm_estart, lm_edges, kf_estart, lm_orig, lm_bstart, blk_kf, blk_lm, blk_es, kf_pidx, act_kf, sv_start;
    size_t tile_lm, pcs, pitem, total;
};
__host__ __device__ inline BaAuxLayout ba_aux_layout(int nkf, int nlm, int nobs, int nblk, int na, int ncontrib, int ntile)
{
    BaAuxLayout L;
    size_t o = 0;
    L.lm_estart = o; o += (size_t)nlm + 1;
    L.lm_edges = o; o += nobs;
    L.kf_estart = o; o += (size_t)nkf + 1;
    L.lm_orig = o; o += nlm;
    L.lm_bstart = o; o += (size_t)nlm + 1;
    L.blk_kf = o; o += nblk;
    L.blk_lm = o; o += nblk;
    L.blk_es = o; o += (size_t)nblk + 1;
    L.kf_pidx = o; o += nkf;
    L.act_kf = o; o += nkf;
    L.sv_start = o; o += (size_t)nkf + 1;
    L.tile_lm = o; o += (size_t)ntile + 1;
    L.pcs = o; o += (size_t)ntile * ((size_t)na * (na + 1) / 2) + 1;
    L.pitem = o; o += ncontrib;
    L.total = o;
    return L;
}

__host__ __device__ inline int ba_pair_index(int a, int b, int na) { return a * na - a * (a - 1) / 2 + (b - a); }

struct BaHostStruct {        // scratch reused across jobs
    std::vector<int> lm_estart, lm_edges, kf_estart, lm_orig, lm_new, srt, ostart, lm_bstart, blk_kf, blk_lm, kf_pidx,
        act_kf, tile_lm, pcs, pitem, fill, bpa, lkf, sv_start, blk_es;
    std::vector<BaRec> recs;     // [0,nobs) landmark-major (= lm_edges order), [nobs,2nobs) pose-major
    int nblk = 0, na = 0, ncontrib = 0, ntile = 0, nmv = 0;

    // Returns false if an edge index is out of range.  Two passes over the edges when they
    // arrive landmark-major with keyframes ascending inside a landmark (the order the host
    // pipeline gathers them in, src/backend.cpp:83-160); otherwise they are sorted first.
    bool build(const BaJob &j, const int *obs_kf, const int *obs_lm, const uint8_t *obs_right, const float *obs_uv,
               int tile_cap)
    {
        const int nkf = j.nkf, nlm = j.nlm, nobs = j.nobs;
        const int *okf = obs_kf + j.obs_ofs, *olm = obs_lm + j.obs_ofs;
        const uint8_t *ori = obs_right + j.obs_ofs;
        const float *ouv = obs_uv + 2 * (size_t)j.obs_ofs;
        bool sorted = true;
        for (int e = 0; e < nobs; ++e) {
            const int k = okf[e], l = olm[e];
            if (k < 0 || k >= nkf || l < 0 || l >= nlm) return false;
            if (e && !((olm[e - 1] < l) || (olm[e - 1] == l && okf[e - 1] <= k))) sorted = false;
        }
        srt.resize(nobs);
        for (int e = 0; e < nobs; ++e) srt[e] = e;
        if (!sorted)
            std::stable_sort(srt.begin(), srt.end(), [&](int a, int b) {
                if (olm[a] != olm[b]) return olm[a] < olm[b];
                return okf[a] < okf[b];
            });
        // Landmarks are renumbered by descending block count (stable), so the lanes of a wave in
        // the thread-per-landmark passes walk equally many blocks: most landmarks of a local
        // window are seen from one keyframe only, a few from all of them.  The single-view ones are
        // further grouped by their keyframe (sv_start: landmark range per keyframe), so that their
        // Schur contributions — which touch nothing but S(a, a) and bs(a) — can be summed by rows of lanes
        // dealt to poses, without LDS tiles; [0, nmv) are the landmarks with two or more blocks.  Landmarks
        // without edges come last.  (Too many keys for the device build's LDS counters: no grouping,
        // nmv = nlm, every landmark goes through the tiles.)
        ostart.assign((size_t)nlm + 1, 0);          // edge ranges in caller numbering (srt order)
        lm_new.assign(nlm, 0);                       // first: blocks per landmark
        lkf.assign(nlm, 0);                          // keyframe of a landmark's (last) block
        {
            int prev_lm = -1, prev_kf = -1;
            for (int i = 0; i < nobs; ++i) {
                const int e = srt[i], k = okf[e], l = olm[e];
                ostart[l + 1]++;
                if (l != prev_lm || k != prev_kf) { lm_new[l]++; lkf[l] = k; prev_lm = l; prev_kf = k; }
            }
        }
        for (int i = 0; i < nlm; ++i) ostart[i + 1] += ostart[i];
        sv_start.assign((size_t)nkf + 1, nlm);
        {
            int maxc = 0;
            for (int l = 0; l < nlm; ++l) maxc = std::max(maxc, lm_new[l]);
            const bool grouped = maxc >= 1 && maxc + nkf <= BB_MAXKEYS;
            const int nkeys = grouped ? maxc + nkf : maxc + 1;
            auto key = [&](int l) {
                const int c = lm_new[l];
                if (!grouped) return maxc - c;
                return c >= 2 ? maxc - c : c == 1 ? maxc - 1 + lkf[l] : maxc - 1 + nkf;
            };
            fill.assign((size_t)nkeys + 1, 0);        // counting sort, stable
            for (int l = 0; l < nlm; ++l) fill[(size_t)key(l) + 1]++;
            for (int c = 0; c < nkeys; ++c) fill[(size_t)c + 1] += fill[c];
            if (grouped) for (int k = 0; k <= nkf; ++k) sv_start[k] = fill[(size_t)maxc - 1 + k];
            nmv = grouped ? sv_start[0] : nlm;
            lm_orig.resize(nlm);
            for (int l = 0; l < nlm; ++l) { const int nid = fill[(size_t)key(l)]++; lm_orig[nid] = l; }
        }
        // pass 1 (landmark-major, new numbering): edge ranges, blocks, records (keyframe still raw)
        lm_edges.resize(nobs);
        lm_estart.assign((size_t)nlm + 1, 0);
        lm_bstart.assign((size_t)nlm + 1, 0);
        kf_estart.assign((size_t)nkf + 1, 0);
        blk_kf.resize(nobs); blk_lm.resize(nobs); blk_es.resize((size_t)nobs + 1);     // upper bound, trimmed below
        recs.resize(2 * (size_t)nobs);
        {
            int i = 0, nb = 0;
            BaRec *__restrict rl = recs.data();
            int *__restrict le = lm_edges.data(), *__restrict ke = kf_estart.data();
            int *__restrict bk = blk_kf.data(), *__restrict bm = blk_lm.data(), *__restrict be = blk_es.data();
            int *__restrict les = lm_estart.data(), *__restrict lbs = lm_bstart.data();
            const int *__restrict lo = lm_orig.data(), *__restrict os = ostart.data(), *__restrict sr = srt.data();
            for (int jn = 0; jn < nlm; ++jn) {
                const int l = lo[jn];
                int prev_kf = -1;
                for (int q = os[l]; q < os[l + 1]; ++q, ++i) {
                    const int e = sr[q], k = okf[e];
                    le[i] = e;
                    ke[k + 1]++;
                    if (k != prev_kf) { bk[nb] = k; bm[nb] = jn; be[nb] = i; ++nb; prev_kf = k; }
                    BaRec r;
                    r.u = ouv[2 * e]; r.v = ouv[2 * e + 1];
                    r.lmkc = (k << 1) | (ori[e] ? 1 : 0);      // completed in pass 2
                    r.blk = nb - 1;
                    rl[i] = r;
                }
                les[jn + 1] = i;
                lbs[jn + 1] = nb;
            }
            nblk = nb;
            be[nb] = nobs;
        }
        blk_kf.resize(nblk); blk_lm.resize(nblk); blk_es.resize((size_t)nblk + 1);
        for (int i = 0; i < nkf; ++i) kf_estart[i + 1] += kf_estart[i];
        kf_pidx.assign(nkf, -1); act_kf.assign(nkf, -1);
        na = 0;
        for (int k = 0; k < nkf; ++k)
            if (kf_estart[k + 1] > kf_estart[k]) { kf_pidx[k] = na; act_kf[na] = k; ++na; }
        // pass 2: finish the records, scatter the pose-major copy (landmark-ascending inside a pose)
        fill.assign(kf_estart.begin(), kf_estart.begin() + nkf);
        {
            BaRec *__restrict rl = recs.data();
            BaRec *__restrict rp = recs.data() + nobs;
            int *__restrict fl = fill.data();
            const int *__restrict bl_ = blk_lm.data();
            const int *__restrict pidx = kf_pidx.data();
            for (int i = 0; i < nobs; ++i) {
                BaRec r = rl[i];
                const int k = r.lmkc >> 1, cam = r.lmkc & 1;
                r.lmkc = bl_[r.blk] | (((pidx[k] << 1) | cam) << 24);
                rl[i] = r;
                rp[fl[k]++] = r;
            }
        }
        // LDS tiles: consecutive landmarks (internal order) with at most tile_cap landmarks and
        // tile_cap blocks (a landmark seen from every keyframe must fit: tile_cap >= nkf); inside a tile the (Y,W) block pairs are listed per pose pair, landmark
        // ascending, as tile-local indices packed into one int
        tile_lm.clear(); tile_lm.push_back(0);
        {
            int nl_t = 0, nb_t = 0;
            for (int l = 0; l < nmv; ++l) {
                const int k = lm_bstart[l + 1] - lm_bstart[l];
                if (nl_t + 1 > tile_cap || nb_t + k > tile_cap) { tile_lm.push_back(l); nl_t = 0; nb_t = 0; }
                ++nl_t; nb_t += k;
            }
            if (nmv > 0) tile_lm.push_back(nmv);
        }
        ntile = (int)tile_lm.size() - 1;
        const int npairs = na * (na + 1) / 2;
        bpa.resize(nblk);
        for (int b = 0; b < nblk; ++b) bpa[b] = kf_pidx[blk_kf[b]];
        pcs.assign((size_t)ntile * npairs + 1, 0);
        {
            const int *__restrict lbs = lm_bstart.data(), *__restrict pa = bpa.data();
            int *__restrict pc = pcs.data();
            for (int t = 0; t < ntile; ++t)
                for (int l = tile_lm[t]; l < tile_lm[t + 1]; ++l) {
                    const int b0 = lbs[l], b1 = lbs[l + 1];
                    for (int u = b0; u < b1; ++u) {
                        const int pu = pa[u];
                        const int base = t * npairs + pu * na - pu * (pu - 1) / 2 - pu;
                        for (int v = u; v < b1; ++v) pc[base + pa[v] + 1]++;
                    }
                }
        }
        for (size_t q = 0; q < (size_t)ntile * npairs; ++q) pcs[q + 1] += pcs[q];
        ncontrib = pcs[(size_t)ntile * npairs];
        pitem.resize(ncontrib);
        fill.assign(pcs.begin(), pcs.begin() + (size_t)ntile * npairs);
        {
            const int *__restrict lbs = lm_bstart.data(), *__restrict pa = bpa.data();
            int *__restrict f1 = fill.data(), *__restrict pi = pitem.data();
            for (int t = 0; t < ntile; ++t) {
                const int l0 = tile_lm[t], bt0 = lbs[l0];
                for (int l = l0; l < tile_lm[t + 1]; ++l) {
                    const int b0 = lbs[l], b1 = lbs[l + 1];
                    for (int u = b0; u < b1; ++u) {
                        const int pu = pa[u];
                        const int base = t * npairs + pu * na - pu * (pu - 1) / 2 - pu;
                        for (int v = u; v < b1; ++v) pi[f1[base + pa[v]]++] = (u - bt0) | ((v - bt0) << 10) | ((l - l0) << 20);
                    }
                }
            }
        }
        return true;
    }
    size_t aux_ints(const BaJob &j) const { return ba_aux_layout(j.nkf, j.nlm, j.nobs, nblk, na, ncontrib, ntile).total; }
    void write(const BaJob &j, int *aux, BaRec *rec_out, BaDev &d) const
    {
        if (j.nobs) std::memcpy(rec_out, recs.data(), sizeof(BaRec) * 2 * (size_t)j.nobs);
        BaAuxLayout L = ba_aux_layout(j.nkf, j.nlm, j.nobs, nblk, na, ncontrib, ntile);
        auto cp = [&](size_t off, const std::vector<int> &v, size_t n) { if (n) std::memcpy(aux + off, v.data(), n * sizeof(int)); };
        cp(L.lm_estart, lm_estart, (size_t)j.nlm + 1); cp(L.lm_edges, lm_edges, j.nobs);
        cp(L.kf_estart, kf_estart, (size_t)j.nkf + 1); cp(L.lm_orig, lm_orig, j.nlm);
        cp(L.lm_bstart, lm_bstart, (size_t)j.nlm + 1);
        cp(L.blk_kf, blk_kf, nblk); cp(L.blk_lm, blk_lm, nblk); cp(L.blk_es, blk_es, (size_t)nblk + 1);
        cp(L.kf_pidx, kf_pidx, j.nkf); cp(L.act_kf, act_kf, j.nkf); cp(L.sv_start, sv_start, (size_t)j.nkf + 1);
        cp(L.tile_lm, tile_lm, (size_t)ntile + 1);
        cp(L.pcs, pcs, (size_t)ntile * ((size_t)na * (na + 1) / 2) + 1); cp(L.pitem, pitem, ncontrib);
        d.kf_ofs = j.kf_ofs; d.nkf = j.nkf; d.lm_ofs = j.lm_ofs; d.nlm = j.nlm; d.obs_ofs = j.obs_ofs; d.nobs = j.nobs;
        d.nblk = nblk; d.na = na; d.ncontrib = ncontrib; d.ntile = ntile; d.iters_done = 0; d.rec_ofs = 0;
        d.lay_nblk = nblk; d.lay_na = na; d.lay_ntile = ntile; d.nmv = nmv; d.reserved = 0; d.lm_base = 0; d.shmask = 0; d.ntrial = 0;
    }
};

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ double block_sum(double v, double *red, int tid)
{
    // wave DPP tree, then the 16 wave partials summed in fixed order
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

// Accumulations of two / three products: nested FMAs into the accumulator (round 5).  `x += a * b + c * d` compiles to
// mul, fmac, add — the sum of the products is rounded before it meets x; the nested form is one instruction shorter per
// entry (there are ~70 such entries per linearised edge and 42 per Schur item) and rounds once less.
#define BA_ACC2(x, a, b, c, d) x = __builtin_fma(a, b, __builtin_fma(c, d, x))
#define BA_ACC3(x, a, b, c, d, e, f) x = __builtin_fma(a, b, __builtin_fma(c, d, __builtin_fma(e, f, x)))
#define BA_SUB2(x, a, b, c, d) x = __builtin_fma(-(a), b, __builtin_fma(-(c), d, x))

// 6x3 blocks (18 doubles = 144 B, 16-B aligned) moved as 9 x 16-byte accesses
__device__ __forceinline__ void ld_block18(const double *p, double *o)
{
    const double2 *q = reinterpret_cast<const double2 *>(p);
#pragma unroll
    for (int t = 0; t < 9; ++t) { double2 v = q[t]; o[2 * t] = v.x; o[2 * t + 1] = v.y; }
}
__device__ __forceinline__ void st_block18(double *p, const double *o)
{
    double2 *q = reinterpret_cast<double2 *>(p);
#pragma unroll
    for (int t = 0; t < 9; ++t) q[t] = make_double2(o[2 * t], o[2 * t + 1]);
}

// The LM loop nest of k_local_ba_t<0> keeps every phase of a trial inside two loops; the compiler hoists the per-thread
// address arithmetic of all phases (row / lane indices, pointers into part, PTab, the record arrays ...) out of the
// nest and then spills it: 165 VGPRs of scratch, and every reload is followed by s_waitcnt vmcnt(0), which also
// drains the global loads in flight.  Each phase therefore derives its thread index from an opaque copy: nothing
// computed from it can be hoisted, live ranges end with the phase.
__device__ __forceinline__ int ba_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
#define BA_PHASE_TID const int tid = ba_opaque(tid0), lane = tid & 63, wv = tid >> 6; (void)lane; (void)wv

// optional cycle profile of job 0 (thread 0): index = phase
#define BA_PROF_N 12
#define BA_PROF(i) do { if (prof && tid == 0) { long long t_ = wall_clock64(); prof[i] += t_ - tprev; tprev = t_; } } while (0)

// LDS pose table entry: R (9, row major) then t (3); cameras: Re (9), te (3), K (4)
#define BA_PT 12
#define BA_CT 16

// projection of landmark X seen from table pose PT through table camera CT
struct BaProj { double q[3], p[3], zi, ex, ey; };
// EID (round 5): both cameras' extrinsic rotations are the identity — the reference's rig always is (Camera::pose_ of the right
// camera is a pure translation, src/dataset.cpp:63-77).  The products with the zeros and ones of Re are then left out; every
// remaining operation keeps the operands and the fused / unfused form it has in the general code (a term with a zero factor
// adds an exact zero there), so the results are the general code's bit for bit, at ~3/4 of its arithmetic.
template <bool EID>
__device__ __forceinline__ void ba_project(const double *PT, const double *CT, const double *X, float u, float v, BaProj &o)
{
#pragma unroll
    for (int r = 0; r < 3; ++r) o.q[r] = PT[3 * r] * X[0] + PT[3 * r + 1] * X[1] + PT[3 * r + 2] * X[2] + PT[9 + r];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        o.p[r] = EID ? o.q[r] + CT[9 + r] : CT[3 * r] * o.q[0] + CT[3 * r + 1] * o.q[1] + CT[3 * r + 2] * o.q[2] + CT[9 + r];
    o.zi = 1.0 / o.p[2];
    const double px = CT[12] * o.p[0] + CT[14] * o.p[2], py = CT[13] * o.p[1] + CT[15] * o.p[2];
    o.ex = (double)u - px * o.zi; o.ey = (double)v - py * o.zi;
}
// M = d(e)/d(p) * Re (2x3) and the pose Jacobian Jp = M [I | -q^] (2x6), g2o_types.h:188-215
template <bool EID>
__device__ __forceinline__ void ba_jac_pose(const double *CT, const BaProj &o, double *M, double *jp)
{
    const double zi2 = o.zi * o.zi;
    const double e00 = -CT[12] * o.zi, e02 = CT[12] * o.p[0] * zi2, e11 = -CT[13] * o.zi, e12 = CT[13] * o.p[1] * zi2;
    if (EID) {
        // M = [e00 0 e02; 0 e11 e12]
        M[0] = e00; M[1] = 0; M[2] = e02; M[3] = 0; M[4] = e11; M[5] = e12;
        jp[0] = e00; jp[1] = 0; jp[2] = e02;
        jp[3] = e02 * o.q[1];
        jp[4] = e00 * o.q[2] - e02 * o.q[0];
        jp[5] = -(e00 * o.q[1]);
        jp[6] = 0; jp[7] = e11; jp[8] = e12;
        jp[9] = e12 * o.q[1] - e11 * o.q[2];
        jp[10] = -(e12 * o.q[0]);
        jp[11] = e11 * o.q[0];
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        M[c] = e00 * CT[c] + e02 * CT[6 + c];
        M[3 + c] = e11 * CT[3 + c] + e12 * CT[6 + c];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const double m0 = M[3 * r], m1 = M[3 * r + 1], m2 = M[3 * r + 2];
        jp[6 * r + 0] = m0; jp[6 * r + 1] = m1; jp[6 * r + 2] = m2;
        jp[6 * r + 3] = m2 * o.q[1] - m1 * o.q[2];
        jp[6 * r + 4] = m0 * o.q[2] - m2 * o.q[0];
        jp[6 * r + 5] = m1 * o.q[0] - m0 * o.q[1];
    }
}

// one edge linearised: residual, robust weight, Jp (2x6), Jl = M R (2x3)
struct BaLin { double ex, ey, w, rho, jp[12], jl[6]; };
template <bool EID>
__device__ __forceinline__ void ba_linearize(const double *PT, const double *CT, const double *X, float u, float v,
                                             double delta, BaLin &L)
{
    BaProj o;
    ba_project<EID>(PT, CT, X, u, v, o);
    L.ex = o.ex; L.ey = o.ey;
    d_huber(o.ex * o.ex + o.ey * o.ey, delta, L.rho, L.w);
    double M[6];
    ba_jac_pose<EID>(CT, o, M, L.jp);
    if (EID) {
        // the general form is fma(M2, R2c, fma(M0, R0c, round(M1 R1c))): with M1 = 0 (row 0) the inner term is round(M0 R0c),
        // with M0 = 0 (row 1) it is round(M1 R1c)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            L.jl[c] = __builtin_fma(M[2], PT[6 + c], M[0] * PT[c]);
            L.jl[3 + c] = __builtin_fma(M[5], PT[6 + c], M[4] * PT[3 + c]);
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            L.jl[3 * r + c] = M[3 * r] * PT[c] + M[3 * r + 1] * PT[3 + c] + M[3 * r + 2] * PT[6 + c];
}
// The pose normal equations of one edge: acc[0..20] += upper triangle of Jp^T w Jp, acc[21..26] -= Jp^T w e.
// EID: the products with Jp's two zeros (row 0 column 1, row 1 column 0) are exact zeros and are left out.
template <bool EID>
__device__ __forceinline__ void ba_acc_pose(double *acc, const double *jp, double w, double ex, double ey)
{
    int t = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const double w0 = w * jp[r], w1 = w * jp[6 + r];
#pragma unroll
        for (int c = r; c < 6; ++c) {
            if (EID && r == 0) { if (c != 1) acc[t] = __builtin_fma(w0, jp[c], acc[t]); }
            else if (EID && r == 1) acc[t] = __builtin_fma(w1, jp[6 + c], acc[t]);
            else BA_ACC2(acc[t], w0, jp[c], w1, jp[6 + c]);
            ++t;
        }
        if (EID && r == 0) acc[21] = __builtin_fma(-w0, ex, acc[21]);
        else if (EID && r == 1) acc[22] = __builtin_fma(-w1, ey, acc[22]);
        else BA_SUB2(acc[21 + r], w0, ex, w1, ey);
    }
}
// W (6x3) += Jp^T w Jl of one edge, wl = w Jl.  EID: Jp's row 0 has a zero in column 1, its row 1 in column 0 — the
// products with them are exact zeros and are left out (the nested-FMA form adds them to the accumulator one by one)
template <bool EID>
__device__ __forceinline__ void ba_acc_w(double *ww, const double *jp, double wl0, double wl1, double wl2, double wl3, double wl4, double wl5)
{
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const double p0 = jp[r], p1 = jp[6 + r];
        if (EID && r == 0) {
            ww[0] = __builtin_fma(p0, wl0, ww[0]); ww[1] = __builtin_fma(p0, wl1, ww[1]); ww[2] = __builtin_fma(p0, wl2, ww[2]);
        } else if (EID && r == 1) {
            ww[3] = __builtin_fma(p1, wl3, ww[3]); ww[4] = __builtin_fma(p1, wl4, ww[4]); ww[5] = __builtin_fma(p1, wl5, ww[5]);
        } else {
            BA_ACC2(ww[r * 3 + 0], p0, wl0, p1, wl3);
            BA_ACC2(ww[r * 3 + 1], p0, wl1, p1, wl4);
            BA_ACC2(ww[r * 3 + 2], p0, wl2, p1, wl5);
        }
    }
}
// Recursive-halving butterfly over the 16 lanes of a DPP row (k_geom.h:po_bfly): 32 values per lane in, the
// row totals of values 2 code, 2 code + 1 out in v[0], v[1] (code = 8 s0 + 4 s1 + 2 s2 + s3; 30 adds, not 32 x 4)
template <int CTRL, int HALF>
__device__ __forceinline__ void ba_bfly(double *v, bool sel)
{
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const double lo = v[i], hi = v[i + HALF];
        const double keep = sel ? hi : lo, send = sel ? lo : hi;
        v[i] = keep + dpp_f64<CTRL>(send);
    }
}
__device__ __forceinline__ int ba_row_sum32(double *v, int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    const bool s0 = b0 != b2, s1 = b1 != b2, s2 = b2 != b3, s3 = b3;
    ba_bfly<SVS_DPP_XOR1, 16>(v, s0);
    ba_bfly<SVS_DPP_XOR2, 8>(v, s1);
    ba_bfly<SVS_DPP_HALF_MIRROR, 4>(v, s2);
    ba_bfly<SVS_DPP_MIRROR, 2>(v, s3);
    return (s0 ? 8 : 0) + (s1 ? 4 : 0) + (s2 ? 2 : 0) + (s3 ? 1 : 0);
}

// One Schur task of a tile: S(a, b) rows 2 rg, 2 rg + 1 (and the same rows of bs when a == b) lose
// sum_items Y W_b^T with Y = W_a (Hll + lambda I)^-1, the items striding over the LANES (8 or 16)
// lanes of a DPP row.  (Wider groups for the diagonal pairs — 32 / 64 lanes with row_bcast
// reductions — pushed the kernel into VGPR spills and were slower.)
template <int LANES>
__device__ __forceinline__ void ba_schur_task(int a, int b2_, int rg, int c0, int c1, int gl, int it0,
                                              const int *Pit, const int *__restrict__ pitem, const double *Wt,
                                              const double *Dl, const double *Bl, double *S, double *bs, int ld)
{
    const bool diag = a == b2_;
    double acc[12], accb[2];
#pragma unroll
    for (int z = 0; z < 12; ++z) acc[z] = 0;
    accb[0] = accb[1] = 0;
    for (int c = c0 + gl; c < c1; c += LANES) {
        const int it3 = (c - it0 < BA_PIT_CAP) ? Pit[c - it0] : pitem[c];
        const int by = it3 & 1023, bw = (it3 >> 10) & 1023, lq = it3 >> 20;
        double yy[6], ww[18];
        {
            const double *wy = Wt + 18 * by + 6 * rg;       // rows 2 rg, 2 rg + 1 of W_y
            const double *Di = Dl + 6 * lq;
            const double d00 = Di[0], d01 = Di[1], d02 = Di[2], d11 = Di[3], d12 = Di[4], d22 = Di[5];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const double x0 = wy[r * 3], x1 = wy[r * 3 + 1], x2 = wy[r * 3 + 2];
                yy[r * 3 + 0] = x0 * d00 + x1 * d01 + x2 * d02;
                yy[r * 3 + 1] = x0 * d01 + x1 * d11 + x2 * d12;
                yy[r * 3 + 2] = x0 * d02 + x1 * d12 + x2 * d22;
            }
        }
        if (diag) {
            const double g0 = Bl[3 * lq], g1 = Bl[3 * lq + 1], g2 = Bl[3 * lq + 2];
            BA_ACC3(accb[0], yy[0], g0, yy[1], g1, yy[2], g2);
            BA_ACC3(accb[1], yy[3], g0, yy[4], g1, yy[5], g2);
        }
        {
            const double *w2 = Wt + 18 * bw;
#pragma unroll
            for (int z = 0; z < 18; ++z) ww[z] = w2[z];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int cc = 0; cc < 6; ++cc)
                BA_ACC3(acc[r * 6 + cc], yy[r * 3], ww[cc * 3], yy[r * 3 + 1], ww[cc * 3 + 1], yy[r * 3 + 2], ww[cc * 3 + 2]);
    }
    // The 14 sums of the group's lanes by a recursive-halving butterfly (round 5; every sum used to walk its own 3- or 4-step
    // DPP tree: 14 x 3 f64 adds in DPP form + a 14-way select per retiring lane).  At each step a lane keeps the half of its
    // values that `sel` names and adds the partner's copy of it — 8 + 4 + 2 (+ 1) adds, the lane ends with the totals of
    // entries 2 c, 2 c + 1 (8 lanes) or of entry c (16 lanes), c from the lane's selector bits.  A value still meets its
    // partners in the order xor 1, xor 2, half mirror (, mirror) and fp addition is commutative: the same association tree as
    // before, bit-identical sums.
    double v[16];
#pragma unroll
    for (int z = 0; z < 12; ++z) v[z] = acc[z];
    v[12] = accb[0]; v[13] = accb[1]; v[14] = 0; v[15] = 0;
    const bool b0 = gl & 1, b1 = gl & 2, b2 = gl & 4, b3 = gl & 8;
    // selectors invariant under the later steps' permutations (half mirror flips bits 0..2, mirror bits 0..3)
    const bool s0 = b0 != b2, s1 = b1 != b2, s2 = LANES == 16 ? (b2 != b3) : b2, s3 = b3;
    ba_bfly<SVS_DPP_XOR1, 8>(v, s0);
    ba_bfly<SVS_DPP_XOR2, 4>(v, s1);
    ba_bfly<SVS_DPP_HALF_MIRROR, 2>(v, s2);
    if (LANES == 16) ba_bfly<SVS_DPP_MIRROR, 1>(v, s3);
    const int code = (s0 ? 8 : 0) + (s1 ? 4 : 0) + (s2 ? 2 : 0) + ((LANES == 16 && s3) ? 1 : 0);
#pragma unroll
    for (int g = 0; g < 16 / LANES; ++g) {
        const double mine = v[g];
        const int e = code + g;
        if (e < 12) {
            const int r = 2 * rg + e / 6, cc = e % 6;
            S[(size_t)(6 * a + r) * ld + 6 * b2_ + cc] -= mine;
            if (!diag) S[(size_t)(6 * b2_ + cc) * ld + 6 * a + r] -= mine;
        } else if (e < 14 && diag) bs[6 * a + 2 * rg + (e - 12)] -= mine;
    }
}

// ---- the reduced camera system: factor and solve, shared by the batch kernel and the low-latency kernels.
// S (np x np, row stride ld, plus one spare row np for the right-hand side) holds the system with lambda on its
// diagonal, bs the right-hand side; on return xp holds the solution and the result says whether S was positive definite.
// Whole workgroup; ends with a barrier.
// prof: optional phase clocks (slot 7 = the factorisation part).
//
// Blocked (6x6 = one pose) right-looking Cholesky S = L L^T in LDS; the right-hand side rides along as row np, so L y = bs
// comes out of the same sweep.  The kernel is bound by the f64 instructions of its busiest wave (one f64 instruction per
// 3.2 ns for a wave alone on its SIMD, 4.9 ns when the SIMD's other wave is busy too: profiles/r3_ubench_valu_issue_rates.txt),
// and the factorisation of a diagonal block is a chain of ~100 of them.  So (round 4):
//   * ONE wave factors the diagonal block and solves the panel below it (every thread used to factor the block redundantly —
//     the second wave of each SIMD only slowed the first), the other seven take the trailing update;
//   * look-ahead: wave 0 updates block column k + 1 with column k first and goes straight on to factor / solve column k + 1
//     while the others update the columns behind it — one barrier per block column instead of two, the chain of column k + 1
//     hidden behind the trailing update of column k;
//   * the back-substitution L^T x = y by blocks: wave 1 inverts every diagonal factor on the side (linv), the solve then takes
//     a 6-term product per block instead of six dependent readlane -> multiply -> FMA steps (60 -> 10 serial steps at K = 10).
__device__ __forceinline__ int ba_chol_solve(double *S, const int ld, const int np, const int na, const double *bs, double *xp,
                                             int *iflag, const int tid0, long long *prof, const long long tprev)
{
    // The inverse of diagonal factor kb is kept inside S, in a 6x6 block above the diagonal that the factorisation never
    // touches (the Cholesky reads and writes the lower triangle only): right of its own diagonal block, or — for the last
    // block — at the top of its block column.  (na >= 3, else the scalar back-substitution below runs.)
    auto linv_at = [&](const int kb) -> double * { return kb + 1 < na ? S + (size_t)(6 * kb) * ld + 6 * kb + 6 : S + 6 * kb; };
    double *rhs = S + (size_t)np * ld;             // extra row: bs on entry, y on exit
    double *invd = xp;                             // 1 / L_kk (xp is free until the back-substitution)
    // wave 0: factor diagonal block kb, solve the panel below it (rows up to the right-hand side's), leave the factor's rows,
    // the inverse pivots and the verdict in LDS
    auto factor_panel = [&](const int kb, const int lane) {
        const int c0 = 6 * kb;
        double Ld[6][6], inv[6];
        int ok = 1;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) Ld[r][c] = S[(size_t)(c0 + r) * ld + c0 + c];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double d = Ld[c][c];
#pragma unroll
            for (int m = 0; m < c; ++m) d -= Ld[c][m] * Ld[c][m];
            if (!(d > 0)) ok = 0;
            // rsqrt: hardware estimate + 2 Newton steps
#ifdef SVS_IEEE_DIV
            double y = 1.0 / sqrt(d);
#else
            double y = __builtin_amdgcn_rsq(d);
            y = y * (1.5 - 0.5 * d * y * y);
            y = y * (1.5 - 0.5 * d * y * y);
#endif
            inv[c] = y;
            Ld[c][c] = d * y;
#pragma unroll
            for (int r = c + 1; r < 6; ++r) {
                double v = Ld[r][c];
#pragma unroll
                for (int m = 0; m < c; ++m) v -= Ld[r][m] * Ld[c][m];
                Ld[r][c] = v * y;
            }
        }
        if (lane == 0) iflag[0] = ok;
        if (!ok) return;                           // (uniform: every lane factors the same block)
        for (int i = c0 + 6 + lane; i <= np; i += 64) {
            double *ri = S + (size_t)i * ld + c0;
            double x[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) x[c] = ri[c];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double v = x[c];
#pragma unroll
                for (int m = 0; m < c; ++m) v -= x[m] * Ld[c][m];
                x[c] = v * inv[c];
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) ri[c] = x[c];
        }
        if (lane < 6) {                            // factor rows of the diagonal block, inverse pivots
            double *ro = S + (size_t)(c0 + lane) * ld + c0;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double v = Ld[0][c];
#pragma unroll
                for (int r = 1; r < 6; ++r) v = (lane == r) ? Ld[r][c] : v;
                ro[c] = c > lane ? 0.0 : v;
            }
            double v = inv[0];
#pragma unroll
            for (int c = 1; c < 6; ++c) v = (lane == c) ? inv[c] : v;
            invd[c0 + lane] = v;
        }
    };
    // trailing update of (row i, block column jb) with block column kb:
    //    S[i][6jb + c'] -= sum_c L[i][c0 + c] * L[6jb + c'][c0 + c]
    auto trail = [&](const int kb, const int i, const int jb) {
        const int c0 = 6 * kb;
        const double *pi = S + (size_t)i * ld + c0;
        double x[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) x[c] = pi[c];
        double *u = S + (size_t)i * ld + 6 * jb;
        const double *P = S + (size_t)(6 * jb) * ld + c0;
#pragma unroll
        for (int cp = 0; cp < 6; ++cp) {
            const double *pr = P + (size_t)cp * ld;
            u[cp] -= x[0] * pr[0] + x[1] * pr[1] + x[2] * pr[2] + x[3] * pr[3] + x[4] * pr[4] + x[5] * pr[5];
        }
    };
    {
        BA_PHASE_TID;
        for (int i = tid; i < np; i += BA_THREADS) rhs[i] = bs[i];
        if (tid == 0) iflag[0] = 1;
        __syncthreads();
        if (wv == 0) factor_panel(0, lane);
        __syncthreads();
    }
    int ok = 1;
#ifdef BA_CHOL_PROF
    long long cp_t = wall_clock64();
#define CP_TICK(i) do { if (prof && tid == 0) { long long t_ = wall_clock64(); prof[i] += t_ - cp_t; cp_t = t_; } } while (0)
#else
#define CP_TICK(i) do { } while (0)
#endif
    for (int kb = 0; kb < na; ++kb) {
        BA_PHASE_TID;
        ok = iflag[0];
        if (!ok) break;                            // uniform
        const int c0 = 6 * kb;
        CP_TICK(12);
        if (wv == 0) {
            if (kb + 1 < na) {
                for (int i = c0 + 6 + lane; i <= np; i += 64) trail(kb, i, kb + 1);      // rows inside the next diagonal block: its upper entries are never read
                CP_TICK(13);
                factor_panel(kb + 1, lane);
                CP_TICK(14);
            }
        } else {
            // block columns behind the next one, lower part only, one thread per (row, block column)
            const int nbm = na - kb - 2, nrows = np + 1 - (c0 + 12);
            for (int t = tid - 64; t < nrows * nbm; t += BA_THREADS - 64) {
                const int i = c0 + 12 + t / nbm, jb = kb + 2 + t % nbm;
                const int jb_last = (i < np) ? i / 6 : na - 1;
                if (jb > jb_last) continue;
                trail(kb, i, jb);
            }
            if (wv == 1 && lane < 6 && na >= 3) {
                // inverse of the factor of diagonal block kb, column `lane` of it per lane: Li[c][c] = 1 / L[c][c],
                // Li[r][c] = -(sum_{m = c}^{r - 1} L[r][m] Li[m][c]) / L[r][r]; stored full (zeros above the diagonal)
                const double *Lk = S + (size_t)c0 * ld + c0;
                double li[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    double sacc = 0;
#pragma unroll
                    for (int m = 0; m < r; ++m) sacc += Lk[(size_t)r * ld + m] * li[m];      // li[m] = 0 for m < lane
                    const double piv = invd[c0 + r];
                    li[r] = r < lane ? 0.0 : (r == lane ? piv : -sacc * piv);
                }
                double *Lw = linv_at(kb);
#pragma unroll
                for (int r = 0; r < 6; ++r) Lw[(size_t)r * ld + lane] = li[r];
            }
        }
        __syncthreads();
        CP_TICK(15);
    }
#undef CP_TICK
    const int wv = ba_opaque(tid0) >> 6;
    if (wv == 0) {
        BA_PHASE_TID;
        if (prof && tid == 0) { long long t_ = wall_clock64(); prof[7] += t_ - tprev; }
        if (ok && np <= 64 && na >= 3) {
            // rhs holds y; L^T x = y by blocks from the last: x_blk = Li^T y_blk (lane of unknown r: column r of Li against the
            // block's six y, fetched with v_readlane), then every lane in front of the block subtracts the block's columns of L^T
            double y = lane < np ? rhs[lane] : 0.0;
            for (int kb = na - 1; kb >= 0; --kb) {
                const int c0 = 6 * kb, r = lane - c0;
                const bool inblk = r >= 0 && r < 6;
                const double *Lr = linv_at(kb);
                double a[6], lr[6], yb[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    a[c] = inblk ? Lr[(size_t)c * ld + r] : 0.0;                        // Li[c][r]
                    lr[c] = lane < c0 ? S[(size_t)(c0 + c) * ld + lane] : 0.0;           // L[c0 + c][lane]
                }
#pragma unroll
                for (int c = 0; c < 6; ++c) yb[c] = readlane_f64(y, c0 + c);
                double xv = 0;
#pragma unroll
                for (int c = 0; c < 6; ++c) xv += a[c] * yb[c];
                y = inblk ? xv : y;
                double xb[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) xb[c] = readlane_f64(y, c0 + c);
#pragma unroll
                for (int c = 0; c < 6; ++c) y -= lr[c] * xb[c];
            }
            if (lane < np) xp[lane] = y;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        } else if (ok) {
            // rhs holds y; back-substitution L^T x = y with the stored inverse pivots.  x lives in registers
            // (lane i: x_i, x_{i+64}, x_{i+128}), the pivot comes over v_readlane, row k of L is requested one
            // step ahead: the serial chain per unknown is readlane -> multiply -> FMA, no LDS round trip.
            double ivk[3], xr[3], srow[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int i = lane + 64 * c;
                ivk[c] = i < np ? invd[i] : 0.0;
                xr[c] = i < np ? rhs[i] : 0.0;
                srow[c] = i < np ? S[(size_t)(np - 1) * ld + i] : 0.0;
            }
            for (int k = np - 1; k >= 0; --k) {
                const int kc = k >> 6, kl = k & 63;
                double crow[3], nxt[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    crow[c] = srow[c];
                    const int i = lane + 64 * c;
                    nxt[c] = (k > 0 && i < k) ? S[(size_t)(k - 1) * ld + i] : 0.0;
                }
                const double xsel = kc == 0 ? xr[0] : kc == 1 ? xr[1] : xr[2];
                const double isel = kc == 0 ? ivk[0] : kc == 1 ? ivk[1] : ivk[2];
                const double xk = readlane_f64(xsel, kl) * readlane_f64(isel, kl);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int i = lane + 64 * c;
                    if (i < k) xr[c] -= crow[c] * xk;
                    else if (i == k) xr[c] = xk;
                    srow[c] = nxt[c];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) { const int i = lane + 64 * c; if (i < np) xp[i] = xr[c]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
        if (lane == 0) iflag[0] = ok;
    }
    __syncthreads();
    return iflag[0];
}

// Shared-map BA (BASELINE config 5): the same kernel cut at the points where ranks have to meet.  Every rank
// holds all K poses and a shard of the landmarks with their edges; MODE 1 runs ONE piece of an LM trial per
// launch and hands the host what must be summed over the ranks (RCCL all-reduce, host/shared_ba in
// stereovision-slam_amd/shared_ba.py drives g2o's LM control flow exactly as the loop below does):
//   phase 1  diag(Hpp) (to be summed) and the largest landmark diagonal (to be max-ed): lambda_0
//   phase 2  pose pass + Schur sweep of the local landmarks at the given lambda -> S, bs, bp, chi2 partials
//   phase 3  the REDUCED system comes back: Cholesky, back-substitution of the local landmarks, update,
//            errors -> chi2 partial of the trial state, the landmark / pose parts of the rho denominator
//   phase 4  reject: restore the backup      phase 5  finalise: per-edge chi2, positions in caller numbering
// io layout per job (doubles): S[np*np] | bs[np] | bp[np] | hdiag[np] | scalars[8]
//   scalars: 0 chi2, 1 landmark diagonal max, 2 cholesky ok, 3 scale (landmarks), 4 scale (poses), 5 chi2 of the trial
struct SbaArgs { int phase, first; double lambda; double *io; double *trace; int add_lambda;      // trace: svslam_lm_trace test hook (MODE 0 / 2);
                                                                   // add_lambda: phase 3 adds lambda I to the reduced system itself (svslam_sba_solve)
                 double *xch; unsigned int *cnt; BaDev *parents; size_t xch_stride;     // MODE 2 (low latency): exchange area, arrival counters, the problems
                 long long ll_timeout; };   // MODE 2: a shard that waits longer than this at one exchange (ticks of the 100 MHz wall clock) gives the problem up
#define SBA_IO_DOUBLES(np) ((size_t)(np) * (np) + 3 * (size_t)(np) + 8)

// Low-latency BA (MODE 2): ONE problem over LLW workgroups.  The phase cut is the shared-map one (MODE 1) — a shard holds all K
// poses and a contiguous range of the landmarks with their edges (k_ba_split deals them by cost, k_ba_build builds every
// shard's structure with all keyframes active) — but the whole LM loop stays in one launch and what ranks all-reduce over RCCL
// the shards exchange through L2 / HBM inside the launch:
//   sync 0 (once)     diag(Hpp) partials, largest landmark diagonal, chi2 of the start state -> lambda_0, currentChi
//   sync A (per trial) the shard's partial reduced system: lower triangle of S = Hpp_w - sum_own W Dinv W^T, bs, bp; every shard
//                     then adds the partials IN SHARD ORDER (bit-identical sums everywhere), adds lambda I, factors and solves
//                     redundantly — no second hop to broadcast the 6K unknowns
//   sync B (per trial) chi2 of the trial state and the landmark part of the rho denominator
// Protocol (MI355X_MICROARCH.md, inter-workgroup visibility; cdna_hip_programming.md Guideline 16, R1 in its counter form):
// payloads are written with agent-scope relaxed atomic stores (write-through `sc1`), every storing wave drains vmcnt, one lane
// adds 1 to the problem's monotonic arrival counter and polls it relaxed; payloads are read with agent-scope relaxed atomic
// loads (`sc1`: served past the CU's L1), so no release / acquire fence is needed.  A buffer written before sync X is next
// written after the following sync Y, which nobody passes before everyone has read: A and B protect each other's buffers.
// All shards of a problem must be resident (the host keeps problems x LLW far below the CU count); a shard that waits longer
// than SbaArgs::ll_timeout at ONE exchange (round 6: a wall-clock limit, 2 ms by default — an exchange takes microseconds when
// the shards are resident; rounds 4-5 counted ~2 M polls, about a second) sets the problem's abort word, every shard then
// returns with iters_done = -1 and the host repeats the problem with the batch solver.
#define LL_SLAB(np) ((size_t)(np) * (np) + 2 * (size_t)(np))      // lower triangle of S in an np x np frame | bs | bp
#define LL_X0(np) ((size_t)(np) + 2)                              // diag(Hpp) | landmark diagonal max | chi2
#define LL_XB 4                                                   // chi2 of the trial | rho denominator (landmarks)
__host__ __device__ inline size_t ll_xch_doubles(int np, int llw) { return (size_t)llw * (LL_SLAB(np) + LL_X0(np) + LL_XB); }
#define LL_CNT_WORDS 4
#define LL_MAX_W 16
// A shard's edge records (both orders) and landmark positions (current and trial state) stay in LDS for the whole LM loop
// when they fit these capacities (a shard of a K = 10 window over 8 workgroups: ~500 edges, ~220 landmarks); a larger shard
// reads them from global memory like the batch kernel.  Then nothing inside an LM trial but the exchanges and a few index
// arrays leaves the CU.
#ifndef LL_ECAP
#define LL_ECAP 704
#endif
#ifndef LL_LCAP
#define LL_LCAP 448
#endif
typedef __attribute__((ext_vector_type(4))) unsigned int ll_u4;
#define LL_SC1 16                                                 // aux bit of the raw buffer builtins: sc1 (agent scope: past L1, write-through)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ll_rsrc(double *base, size_t doubles)
{
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(doubles * sizeof(double)), 0x00020000);
}
__device__ __forceinline__ void ll_st2(__amdgpu_buffer_rsrc_t r, size_t idx, double a, double b)      // doubles idx, idx + 1 (idx even)
{
    ll_u4 v;
    v.x = (unsigned)__double2loint(a); v.y = (unsigned)__double2hiint(a); v.z = (unsigned)__double2loint(b); v.w = (unsigned)__double2hiint(b);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(idx * sizeof(double)), 0, LL_SC1);
}
__device__ __forceinline__ ll_u4 ll_ld2(__amdgpu_buffer_rsrc_t r, size_t idx)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)(idx * sizeof(double)), 0, LL_SC1);
}
__device__ __forceinline__ double ll_lo(ll_u4 v) { return __hiloint2double((int)v.y, (int)v.x); }
__device__ __forceinline__ double ll_hi(ll_u4 v) { return __hiloint2double((int)v.w, (int)v.z); }
__device__ __forceinline__ void ll_st(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ll_ld(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}

// A/B knob (tools/ab.sh): cap the kernel's registers below the 256 its two waves per SIMD may use, so that a wave of another
// kernel fits beside them on the SIMD (amdgpu_num_vgpr takes half of the unified count on gfx90a+)
#ifdef BA_NUM_VGPR
#define BA_VGPR_ATTR __attribute__((amdgpu_num_vgpr(BA_NUM_VGPR)))
#else
#define BA_VGPR_ATTR
#endif
template <int MODE, int LLW, bool EID = false>
__global__ void __launch_bounds__(BA_THREADS, BA_MIN_WAVES_PER_SIMD) BA_VGPR_ATTR
k_local_ba_t(BaDev *jobs, const BaCams *camsp, double *poses_all, double *pts_all, const BaRec *recs_all,
             const int *aux_all, BaWork wk, double delta, int iters, double *edge_chi2_all, long long *prof_all,
             int tile_cap, SbaArgs sba)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int job = blockIdx.x;
    BaDev &jd = jobs[job];
    const int tid0 = threadIdx.x;
    const int tid = tid0, lane = tid & 63, wv = tid >> 6;
    const int nkf = jd.nkf, nlm = jd.nlm, nobs = jd.nobs, na = jd.na, np = 6 * jd.na, nblk = jd.nblk;
    if (nobs <= 0 || na <= 0) { if (tid == 0) jd.iters_done = 0; return; }
    if (MODE == 2 && jd.reserved >= 2) return;             // 3: dropped by the test hook (k_ll_test_drop); 2: every shard of the problem fits the resident layout: k_ba_ll (k_ba_ll.h) takes it
    long long *prof = (prof_all && job == 0) ? prof_all : nullptr;
    long long tprev = prof ? wall_clock64() : 0;
    // LDS carve (all dynamic): S[(np+1)*(np+1)] | bs[np] | xp[np] | Hpp[36*na] | bp[np] | red[W] | PT[12 na] | CT[32] | flag
    const int ld = np + 1;   // odd row stride (in doubles): spreads LDS banks
    double *S = reinterpret_cast<double *>(smem);
    double *bs = S + (size_t)(np + 1) * ld;    // row np of S carries the right-hand side
    double *xp = bs + np;
    double *Hpp = xp + np;
    double *bp = Hpp + 36 * na;
    double *bpt = bp + np;                         // MODE 2: bp summed over the shards (bp itself stays this shard's partial)
    double *red = bpt + np;
    double *PTab = red + BA_WAVES;
    double *CTab = PTab + BA_PT * na;
    double *PTab2 = CTab + 2 * BA_CT;              // pose table of the trial state (errors of the trial, see the back-substitution)
    // Hpp / bp of the TRIAL state (MODE 0: copied over Hpp / bp when the trial is accepted): in the reduced system's
    // storage, which is dead between the back-substitution and the next trial's S initialisation
    double *Hpp2 = S;
    double *bp2 = Hpp2 + 36 * na;
    double *part = PTab2 + BA_PT * na;             // [BA_ROWS][32] row partial sums (pose pass, single-view pass: 28 used)
    double *Wt = part + 32 * BA_ROWS;              // [tile_cap][18] blocks of the current tile
    double *Dl = Wt + 18 * tile_cap;               // [tile_cap][6]  (Hll + lambda I)^-1, symmetric
    double *Bl = Dl + 6 * tile_cap;                // [tile_cap][3]  bl
    int *Pcs = reinterpret_cast<int *>(Bl + 3 * tile_cap);   // [npairs + 1] item ranges of the current tile
    int *Pit = Pcs + (BA_MAX_NP / 6) * (BA_MAX_NP / 6 + 1) / 2 + 1;   // [BA_PIT_CAP] its items
    int *iflag = Pit + BA_PIT_CAP;
    // MODE 2 only (ba_lds_bytes_ll): pose of each block of the tile, the shard's records, its positions
    int *ll_ba = iflag + 16;                                                     // [tile_cap]
    BaRec *ll_rec = reinterpret_cast<BaRec *>(ll_ba + ((tile_cap + 3) & ~3));   // [2 LL_ECAP]
    double *ll_pts = reinterpret_cast<double *>(ll_rec + 2 * LL_ECAP);          // [2][3 LL_LCAP]

    const BaCams &cams = *camsp;
    double *poses = poses_all + (size_t)jd.kf_ofs * 7;
    double *pts_io = pts_all + (size_t)jd.lm_ofs * 3;     // caller numbering
    double *edge_chi2 = edge_chi2_all + jd.obs_ofs;
    const BaRec *recL = recs_all + jd.rec_ofs;       // landmark-major
    const BaRec *recP = recL + nobs;                   // pose-major
    const bool ll_res = MODE == 2 && nobs <= LL_ECAP && nlm <= LL_LCAP;          // the shard lives in LDS
    const int *aux = aux_all + jd.aux_ofs;
    const int ntile = jd.ntile;
    const BaAuxLayout AL = ba_aux_layout(nkf, nlm, nobs, jd.lay_nblk, jd.lay_na, jd.ncontrib, jd.lay_ntile);
    const int *lm_estart = aux + AL.lm_estart, *lm_edges = aux + AL.lm_edges;
    const int *kf_estart = aux + AL.kf_estart;
    const int *lm_bstart = aux + AL.lm_bstart;
    const int *blk_kf = aux + AL.blk_kf, *blk_lm = aux + AL.blk_lm, *blk_es = aux + AL.blk_es;
    const int *kf_pidx = aux + AL.kf_pidx, *act_kf = aux + AL.act_kf, *sv_start = aux + AL.sv_start;
    const int *tile_lm = aux + AL.tile_lm, *pcs = aux + AL.pcs, *pitem = aux + AL.pitem;

    const size_t J = job;
    double *err = wk.err + J * 2 * wk.max_obs;
    double *poses_b = wk.poses_b + J * 7 * wk.max_kf;
    double *poses_a = wk.poses_a + J * 7 * wk.max_kf;
    double *pts_b = wk.pts_b + J * 3 * wk.max_lm;
    double *pts = wk.pts_i + J * 3 * wk.max_lm;            // internal numbering (see BaHostStruct::build)
    const int *lm_orig = aux + AL.lm_orig;
    // The state is double-buffered (MODE 0): an LM trial writes its poses / positions into the `trial` buffers, an
    // accepted trial swaps the roles, a rejected one costs nothing — no backup copy per trial (round 2: 41 KB of
    // positions copied, and copied back on rejection).  MODE 1 spans launches, so it updates in place and keeps the
    // backup copies (cur == trial).
    // (MODE 2: the shards of a problem share `poses`; each keeps both of its pose buffers in its own workspace)
    double *cur = pts, *trial = MODE != 1 ? pts_b : pts;
    double *pcur = MODE == 2 ? poses_a : poses, *ptrial = MODE != 1 ? poses_b : poses;
    if (MODE == 2 && ll_res) {
        for (int i = tid; i < 2 * nobs; i += BA_THREADS) ll_rec[i] = recL[i];
        for (int j = tid; j < nlm; j += BA_THREADS) {
            const double *s3 = pts_io + 3 * (size_t)lm_orig[j];
#pragma unroll
            for (int c = 0; c < 3; ++c) { ll_pts[3 * j + c] = s3[c]; ll_pts[3 * LL_LCAP + 3 * j + c] = s3[c]; }
        }
        recL = ll_rec; recP = ll_rec + nobs;
        cur = ll_pts; trial = ll_pts + 3 * LL_LCAP;
    } else if (MODE != 1 || sba.first)
        for (int j = tid; j < nlm; j += BA_THREADS) {
            const double *s3 = pts_io + 3 * (size_t)lm_orig[j];
            pts[3 * (size_t)j] = s3[0]; pts[3 * (size_t)j + 1] = s3[1]; pts[3 * (size_t)j + 2] = s3[2];
            if (MODE != 1) { pts_b[3 * (size_t)j] = s3[0]; pts_b[3 * (size_t)j + 1] = s3[1]; pts_b[3 * (size_t)j + 2] = s3[2]; }   // edge-less landmarks never move
        }
    if (MODE != 1) for (int i = tid; i < 7 * nkf; i += BA_THREADS) { const double v = poses[i]; poses_b[i] = v; if (MODE == 2) poses_a[i] = v; }   // keyframes without edges never move
    // ---- MODE 2: this shard's place in its problem, the exchange area, the arrival counter
    const int ll_prob = MODE == 2 ? job / LLW : 0, ll_w = MODE == 2 ? job % LLW : 0;
    const unsigned ll_mask = MODE == 2 ? (unsigned)jd.shmask : 0u;
    const bool ll_leader = MODE == 2 && (ll_mask & ((1u << ll_w) - 1u)) == 0u;       // lowest shard with edges: writes the poses back
    double *ll_xs = MODE == 2 ? sba.xch + (size_t)ll_prob * sba.xch_stride : nullptr;  // [LLW][LL_SLAB]
    double *ll_x0 = MODE == 2 ? ll_xs + (size_t)LLW * LL_SLAB(np) : nullptr;           // [LLW][LL_X0]
    double *ll_xb = MODE == 2 ? ll_x0 + (size_t)LLW * LL_X0(np) : nullptr;             // [LLW][LL_XB]
    unsigned int *ll_cnt = MODE == 2 ? sba.cnt + (size_t)LL_CNT_WORDS * ll_prob : nullptr;
    unsigned ll_ep = 0, ll_epb = 0;
    const unsigned ll_n = __popc(ll_mask);
    auto ll_sync = [&]() -> bool {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave: its write-through stores have left
        __syncthreads();
        ++ll_ep;
        if (tid0 == 0) {
            __hip_atomic_fetch_add(ll_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = ll_n * ll_ep;
            unsigned spins = 0;
            int good = 1;
            const long long t_wait0 = wall_clock64();
            while (__hip_atomic_load(ll_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 63u) == 0 &&
                    (wall_clock64() - t_wait0 > sba.ll_timeout || __hip_atomic_load(ll_cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    __hip_atomic_store(ll_cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    good = 0;
                    break;
                }
            }
            iflag[1] = good;
        }
        __syncthreads();
        return iflag[1] != 0;
    };
    double *sio = MODE == 1 ? sba.io + (size_t)job * SBA_IO_DOUBLES(np) : nullptr;
    double *sio_S = sio, *sio_bs = sio + (size_t)np * np, *sio_bp = sio_bs + np, *sio_hd = sio_bp + np, *sio_sc = sio_hd + np;

    // camera table (constant) and pose table (rebuilt whenever the poses change)
    if (tid < 2) {
        double *CT = CTab + BA_CT * tid;
        d_quat_to_R(cams.ext[tid], CT);
        CT[9] = cams.ext[tid][4]; CT[10] = cams.ext[tid][5]; CT[11] = cams.ext[tid][6];
        CT[12] = cams.cam[tid][0]; CT[13] = cams.cam[tid][1]; CT[14] = cams.cam[tid][2]; CT[15] = cams.cam[tid][3];
    }
    auto pose_table_into = [&](double *tab, const double *src) {
        BA_PHASE_TID;
        __syncthreads();
        if (tid < na) {
            const double *T = src + 7 * act_kf[tid];
            double *PT = tab + BA_PT * tid;
            d_quat_to_R(T, PT);
            PT[9] = T[4]; PT[10] = T[5]; PT[11] = T[6];
        }
        __syncthreads();
    };
    auto pose_table = [&]() { pose_table_into(PTab, pcur); };

    // errors at the current state: thread per edge in landmark-major order (coalesced records,
    // near-coalesced landmark reads, poses from the LDS table)
    // (evaluated through the SECOND pose table: the first keeps the linearisation point of the iteration, which is
    // also where the successor of a rejected trial has to linearise)
    auto error_pass = [&](const double *epts, const double *eposes) -> double {
        pose_table_into(PTab2, eposes);
        BA_PHASE_TID;
        double chi = 0;
        // two dependent global loads per edge (record -> landmark position) and eight edges per thread: the loop is
        // software-pipelined — the record two edges ahead and the position one edge ahead are in flight while an
        // edge is evaluated (with two waves per SIMD there is no other latency hiding; this phase and the pose pass
        // were bound by 16 serialised memory latencies per thread)
        // (prefetch indices are clamped, not predicated: a predicated 16-byte load compiles to a branch per dword)
        BaRec rc = recL[min(tid, nobs - 1)];
        BaRec rn = recL[min(tid + BA_THREADS, nobs - 1)];
        double X[3];
        { const double *Xp = epts + 3 * (size_t)(rc.lmkc & BA_LM_MASK); X[0] = Xp[0]; X[1] = Xp[1]; X[2] = Xp[2]; }
        for (int i = tid; i < nobs; i += BA_THREADS) {
            const BaRec rnn = recL[min(i + 2 * BA_THREADS, nobs - 1)];
            const double *Xq = epts + 3 * (size_t)(rn.lmkc & BA_LM_MASK);
            const double Xn[3] = { Xq[0], Xq[1], Xq[2] };
            const int kc = (unsigned)rc.lmkc >> 24;
            BaProj o;
            ba_project<EID>(PTab2 + BA_PT * (kc >> 1), CTab + BA_CT * (kc & 1), X, rc.u, rc.v, o);
            err[2 * i] = o.ex; err[2 * i + 1] = o.ey;
            double r0, r1;
            d_huber(o.ex * o.ex + o.ey * o.ey, delta, r0, r1);
            chi += r0;
            rc = rn; rn = rnn; X[0] = Xn[0]; X[1] = Xn[1]; X[2] = Xn[2];
        }
        return block_sum(chi, red, tid);
    };

    // ---- pose pass at a state (positions `psrc`, pose table `tab`): Hpp (block diagonal), bp -> `Hout`, `bout`, and the
    // robust chi2 of the state (identical in every thread).  16-lane rows; pose a is shared by the rows a, a + na,
    // a + 2 na ... (< BA_ROWS); 27 normal-equation sums + chi2 ride one recursive-halving butterfly per row, the rows of a
    // pose are added in row order.  Round 3: the SAME pass evaluates an LM trial (it needs the projections anyway) — an
    // accepted trial hands its Hpp / bp / pose table to the next iteration, which then starts at the single-view pass:
    // one sweep over the edges per trial less (the separate error pass) and no per-edge residual array.
    auto pose_pass = [&](const double *psrc, const double *tab, double *Hout, double *bout) -> double {
        BA_PHASE_TID;
        const int row = tid >> 4, rl = tid & 15;
        const int rpp = BA_ROWS / na;                 // rows per pose (>= 1: na <= 32)
        double acc[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) acc[t] = 0;
        const int a = row % na, sub = row / na;
        if (sub < rpp) {
            const int k = act_kf[a];
            const int i0 = kf_estart[k] + sub * 16 + rl, i1 = kf_estart[k + 1], ist = 16 * rpp;
            BaRec rc = recP[min(i0, nobs - 1)];            // software pipeline: record two edges ahead, position one ahead
            BaRec rn = recP[min(i0 + ist, nobs - 1)];
            double X[3];
            { const double *Xp = psrc + 3 * (size_t)(rc.lmkc & BA_LM_MASK); X[0] = Xp[0]; X[1] = Xp[1]; X[2] = Xp[2]; }
            for (int i = i0; i < i1; i += ist) {
                const BaRec rnn = recP[min(i + 2 * ist, nobs - 1)];
                const double *Xq = psrc + 3 * (size_t)(rn.lmkc & BA_LM_MASK);
                const double Xn[3] = { Xq[0], Xq[1], Xq[2] };
                const int kc = (unsigned)rc.lmkc >> 24;
                const double *CT = CTab + BA_CT * (kc & 1);
                BaProj o;
                ba_project<EID>(tab + BA_PT * a, CT, X, rc.u, rc.v, o);
                double r0, w;
                d_huber(o.ex * o.ex + o.ey * o.ey, delta, r0, w);
                acc[27] += r0;
                double M[6], jp[12];
                ba_jac_pose<EID>(CT, o, M, jp);
                ba_acc_pose<EID>(acc, jp, w, o.ex, o.ey);
                rc = rn; rn = rnn; X[0] = Xn[0]; X[1] = Xn[1]; X[2] = Xn[2];
            }
        }
        const int code = ba_row_sum32(acc, lane);
        __syncthreads();                              // `part` may still be read by the previous phase
        reinterpret_cast<double2 *>(part + 32 * row)[code] = make_double2(acc[0], acc[1]);
        __syncthreads();
        for (int z = tid; z < 27 * na; z += BA_THREADS) {
            const int a2 = z / 27, t = z - a2 * 27;
            double v = 0;
            for (int sb = 0; sb < rpp; ++sb) v += part[(sb * na + a2) * 32 + t];
            if (t < 21) {
                int r = 0, rem = t;
                while (rem >= 6 - r) { rem -= 6 - r; ++r; }
                const int c = r + rem;
                Hout[36 * a2 + r * 6 + c] = v; Hout[36 * a2 + c * 6 + r] = v;
            } else bout[6 * a2 + (t - 21)] = v;
        }
        double chi = 0;
#pragma unroll
        for (int r = 0; r < BA_ROWS; ++r) chi += part[r * 32 + 27];
        __syncthreads();
        return chi;
    };

    double lambda = MODE == 1 ? sba.lambda : 0, ni = 2;
    int it_done = 0, trials_done = 0;
    const int npairs = na * (na + 1) / 2;
    double currentChi = 0;
    if (MODE == 1 && sba.phase == 4) {                     // reject the trial
        __syncthreads();
        for (int i = tid; i < 7 * nkf; i += BA_THREADS) poses[i] = poses_b[i];
        for (int i = tid; i < 3 * nlm; i += BA_THREADS) pts[i] = pts_b[i];
        return;
    }
    if (MODE == 1 && sba.phase == 5) iters = 0;            // finalise: straight to the write-back
    const bool lin = MODE != 1 || sba.phase <= 2;          // this launch linearises (pose pass, Schur sweep)
    bool ll_failed = false;                                // MODE 2: an exchange timed out
    // last state whose residuals were evaluated (g2o reports the edge chi2 of its last computeActiveErrors, the state of
    // the last trial even when that trial was rejected): the per-edge chi2 are computed from it at the end (MODE 0)
    const double *last_pts = cur, *last_poses = pcur;
    bool have_lin = false;                             // MODE 0: Hpp / bp / PTab of `cur` are in place (left by the accepted trial)
    for (int it = 0; it < iters; ++it) {
        double chi_lin = 0;
        if (!have_lin) {
            pose_table();
            if (lin) chi_lin = pose_pass(cur, PTab, Hpp, bp);
        }
        __syncthreads();
        BA_PROF(1);
        if (it == 0 && (MODE != 1 || sba.phase == 1)) {
            BA_PHASE_TID;
            // lambda_0 = 1e-5 * max diagonal of the Hessian: the landmark diagonals need one cheap sweep
            double md = 0;
            if (MODE == 0) for (int i = tid; i < np; i += BA_THREADS) md = fmax(md, fabs(Hpp[36 * (i / 6) + (i % 6) * 7]));
            else if (MODE == 1) for (int i = tid; i < np; i += BA_THREADS) sio_hd[i] = Hpp[36 * (i / 6) + (i % 6) * 7];   // summed over the ranks first
            else for (int i = tid; i < np; i += BA_THREADS) ll_st(ll_x0 + (size_t)ll_w * LL_X0(np) + i, Hpp[36 * (i / 6) + (i % 6) * 7]);   // ... over the shards
            for (int j = tid; j < nlm; j += BA_THREADS) {
                const double X[3] = { cur[3 * (size_t)j], cur[3 * (size_t)j + 1], cur[3 * (size_t)j + 2] };
                double h0 = 0, h3 = 0, h5 = 0;
                for (int i = lm_estart[j]; i < lm_estart[j + 1]; ++i) {
                    const BaRec rc = recL[i];
                    const int kc = (unsigned)rc.lmkc >> 24;
                    BaLin L;
                    ba_linearize<EID>(PTab + BA_PT * (kc >> 1), CTab + BA_CT * (kc & 1), X, rc.u, rc.v, delta, L);
                    h0 += L.w * (L.jl[0] * L.jl[0] + L.jl[3] * L.jl[3]);
                    h3 += L.w * (L.jl[1] * L.jl[1] + L.jl[4] * L.jl[4]);
                    h5 += L.w * (L.jl[2] * L.jl[2] + L.jl[5] * L.jl[5]);
                }
                if (lm_estart[j + 1] > lm_estart[j]) md = fmax(md, fmax(fabs(h0), fmax(fabs(h3), fabs(h5))));
            }
            md = block_max(md, red, tid);
            if (MODE == 1) { if (tid == 0) sio_sc[1] = md; return; }
            if (MODE == 2) {
                // sync 0: the pose diagonals are sums over the shards, the landmark maximum a maximum, chi2 a sum
                if (tid == 0) { ll_st(ll_x0 + (size_t)ll_w * LL_X0(np) + np, md); ll_st(ll_x0 + (size_t)ll_w * LL_X0(np) + np + 1, chi_lin); }
                if (!ll_sync()) { ll_failed = true; break; }
                double m2 = 0, chs = 0;
#pragma unroll
                for (int v = 0; v < LLW; ++v) { m2 = fmax(m2, ll_ld(ll_x0 + (size_t)v * LL_X0(np) + np)); chs += ll_ld(ll_x0 + (size_t)v * LL_X0(np) + np + 1); }
                for (int i = tid; i < np; i += BA_THREADS) {
                    double hd = 0;
#pragma unroll
                    for (int v = 0; v < LLW; ++v) hd += ll_ld(ll_x0 + (size_t)v * LL_X0(np) + i);
                    m2 = fmax(m2, fabs(hd));
                }
                md = block_max(m2, red, tid);
                chi_lin = chs;
            }
            lambda = 1e-5 * md; ni = 2;
        }
        double tempChi = 0;
        if (!have_lin && lin) currentChi = chi_lin;      // (afterwards currentChi carries over from the accepted trial)
        double rho = 0; int qmax = 0;
        do {
            // backup, S = blockdiag(Hpp) + lambda I, bs = bp
            if (lin) {
            BA_PHASE_TID;
            if (MODE == 1) {
                for (int i = tid; i < 7 * nkf; i += BA_THREADS) poses_b[i] = poses[i];
                for (int i = tid; i < 3 * nlm; i += BA_THREADS) pts_b[i] = pts[i];
            }
            for (int r = wv; r < np; r += BA_WAVES)
                for (int c = lane; c < np; c += 64) {
                    double v = 0;
                    if (r / 6 == c / 6) v = Hpp[36 * (r / 6) + (r % 6) * 6 + (c % 6)];
                    if (r == c && MODE == 0) v += lambda;          // shared map: added once, to the reduced system
                    S[(size_t)r * ld + c] = v;
                }
            for (int i = tid; i < np; i += BA_THREADS) bs[i] = bp[i];
            }
            __syncthreads();
            BA_PROF(2);
            // ---- single-view landmarks (most of a local window): one block each, so all they change is S(a, a)
            // and bs(a) of their pose.  No LDS tile: a lane linearises its landmark (one or two edges), forms W,
            // (Hll + lambda I)^-1 and Y = W Dinv in registers and adds Y W^T (21 sums) and Y bl (6) to its own
            // accumulators; the rows are dealt over the poses like the pose pass, row totals by the butterfly,
            // the rows of a pose added in row order.
            if (lin && jd.nmv < nlm) {
                BA_PHASE_TID;
                const int row = tid >> 4, rl = tid & 15;
                const int rpp = BA_ROWS / na, a = row % na, sub = row / na;
                double acc[32];
#pragma unroll
                for (int t = 0; t < 32; ++t) acc[t] = 0;
                if (sub < rpp) {
                    const int k = act_kf[a];
                    const double *PT = PTab + BA_PT * a;
                    const int j0 = sv_start[k] + sub * 16 + rl, j1 = sv_start[k + 1], jst = 16 * rpp;
                    int e0, e1;
                    double X[3];
                    { const int jc = min(j0, nlm - 1); e0 = lm_estart[jc]; e1 = lm_estart[jc + 1]; X[0] = cur[3 * (size_t)jc]; X[1] = cur[3 * (size_t)jc + 1]; X[2] = cur[3 * (size_t)jc + 2]; }
                    for (int j = j0; j < j1; j += jst) {
                        const int jn = min(j + jst, nlm - 1);
                        const int ne0 = lm_estart[jn], ne1 = lm_estart[jn + 1];
                        const double Xn[3] = { cur[3 * (size_t)jn], cur[3 * (size_t)jn + 1], cur[3 * (size_t)jn + 2] };
                        double h[6] = { 0, 0, 0, 0, 0, 0 }, b3[3] = { 0, 0, 0 }, ww[18];
#pragma unroll
                        for (int t = 0; t < 18; ++t) ww[t] = 0;
                        BaRec rc = recL[min(e0, nobs - 1)];
                        for (int i = e0; i < e1; ++i) {
                            const BaRec rn = recL[min(i + 1, nobs - 1)];
                            const int kc = (unsigned)rc.lmkc >> 24;
                            BaLin L;
                            ba_linearize<EID>(PT, CTab + BA_CT * (kc & 1), X, rc.u, rc.v, delta, L);
                            const double wl0 = L.w * L.jl[0], wl1 = L.w * L.jl[1], wl2 = L.w * L.jl[2],
                                         wl3 = L.w * L.jl[3], wl4 = L.w * L.jl[4], wl5 = L.w * L.jl[5];
                            ba_acc_w<EID>(ww, L.jp, wl0, wl1, wl2, wl3, wl4, wl5);
                            BA_SUB2(b3[0], wl0, L.ex, wl3, L.ey); BA_SUB2(b3[1], wl1, L.ex, wl4, L.ey); BA_SUB2(b3[2], wl2, L.ex, wl5, L.ey);
                            BA_ACC2(h[0], wl0, L.jl[0], wl3, L.jl[3]); BA_ACC2(h[1], wl0, L.jl[1], wl3, L.jl[4]); BA_ACC2(h[2], wl0, L.jl[2], wl3, L.jl[5]);
                            BA_ACC2(h[3], wl1, L.jl[1], wl4, L.jl[4]); BA_ACC2(h[4], wl1, L.jl[2], wl4, L.jl[5]); BA_ACC2(h[5], wl2, L.jl[2], wl5, L.jl[5]);
                            rc = rn;
                        }
                        double D[9] = { h[0] + lambda, h[1], h[2], h[1], h[3] + lambda, h[4], h[2], h[4], h[5] + lambda }, Di[9];
                        d_inv3(D, Di);
                        int t = 0;
#pragma unroll
                        for (int r = 0; r < 6; ++r) {
                            const double x0 = ww[r * 3], x1 = ww[r * 3 + 1], x2 = ww[r * 3 + 2];
                            const double y0 = x0 * Di[0] + x1 * Di[1] + x2 * Di[2], y1 = x0 * Di[1] + x1 * Di[4] + x2 * Di[5],
                                         y2 = x0 * Di[2] + x1 * Di[5] + x2 * Di[8];
#pragma unroll
                            for (int cc = r; cc < 6; ++cc) { BA_ACC3(acc[t], y0, ww[cc * 3], y1, ww[cc * 3 + 1], y2, ww[cc * 3 + 2]); ++t; }
                            BA_ACC3(acc[21 + r], y0, b3[0], y1, b3[1], y2, b3[2]);
                        }
                        e0 = ne0; e1 = ne1; X[0] = Xn[0]; X[1] = Xn[1]; X[2] = Xn[2];
                    }
                }
                const int code = ba_row_sum32(acc, lane);
                reinterpret_cast<double2 *>(part + 32 * row)[code] = make_double2(acc[0], acc[1]);
                __syncthreads();
                for (int z = tid; z < 27 * na; z += BA_THREADS) {
                    const int a2 = z / 27, t = z - a2 * 27;
                    double v = 0;
                    for (int sb = 0; sb < rpp; ++sb) v += part[(sb * na + a2) * 32 + t];
                    if (t < 21) {
                        int r = 0, rem = t;
                        while (rem >= 6 - r) { rem -= 6 - r; ++r; }
                        const int cc = r + rem;
                        S[(size_t)(6 * a2 + r) * ld + 6 * a2 + cc] -= v;
                        if (cc != r) S[(size_t)(6 * a2 + cc) * ld + 6 * a2 + r] -= v;
                    } else bs[6 * a2 + (t - 21)] -= v;
                }
                __syncthreads();
            }
            BA_PROF(0);
            // ---- tile sweep (landmarks seen from two or more poses): linearise the tile's landmarks into LDS,
            // then fold the tile into S / bs
            for (int tl = 0; lin && tl < ntile; ++tl) {
                BA_PHASE_TID;
                const int l0 = tile_lm[tl], l1 = tile_lm[tl + 1], bt0 = lm_bstart[l0];
                // the tile's pair ranges and items go to LDS too: issued here, stored after the
                // landmark work, so their latency hides behind it (the Schur pass then never
                // waits for a dependent global load)
                const int it0 = pcs[tl * npairs], it1 = pcs[(tl + 1) * npairs];
                int pre[BA_PIT_CAP / BA_THREADS];
#pragma unroll
                for (int q = 0; q < BA_PIT_CAP / BA_THREADS; ++q) {
                    const int c = it0 + tid + q * BA_THREADS;
                    pre[q] = c < it1 ? pitem[c] : 0;
                }
                const int pcs_mine = tid <= npairs ? pcs[tl * npairs + tid] : 0;   // npairs + 1 <= BA_THREADS
                // thread per BLOCK: its one or two edge records are independent loads, every thread of the workgroup
                // has work (a tile of multi-view landmarks is ~200 landmarks of ~2.5 blocks; one thread per landmark
                // walked five dependent record loads with most of the workgroup idle).  W_block -> LDS; the block's
                // share of Hll and bl goes through LDS too (the Dl / Bl area, indexed by block for the moment) ...
                const int nbt = lm_bstart[l1] - bt0;
                for (int bq = tid; bq < nbt; bq += BA_THREADS) {
                    const int b = bt0 + bq, j = blk_lm[b];
                    const int e0 = blk_es[b], e1 = blk_es[b + 1];
                    const double X[3] = { cur[3 * (size_t)j], cur[3 * (size_t)j + 1], cur[3 * (size_t)j + 2] };
                    double h[6] = { 0, 0, 0, 0, 0, 0 }, b3[3] = { 0, 0, 0 }, wacc[18];
#pragma unroll
                    for (int t = 0; t < 18; ++t) wacc[t] = 0;
                    for (int i = e0; i < e1; ++i) {
                        const BaRec rc = recL[i];
                        const int kc = (unsigned)rc.lmkc >> 24;
                        BaLin L;
                        ba_linearize<EID>(PTab + BA_PT * (kc >> 1), CTab + BA_CT * (kc & 1), X, rc.u, rc.v, delta, L);
                        const double wl0 = L.w * L.jl[0], wl1 = L.w * L.jl[1], wl2 = L.w * L.jl[2],
                                     wl3 = L.w * L.jl[3], wl4 = L.w * L.jl[4], wl5 = L.w * L.jl[5];
                        ba_acc_w<EID>(wacc, L.jp, wl0, wl1, wl2, wl3, wl4, wl5);
                        BA_SUB2(b3[0], wl0, L.ex, wl3, L.ey); BA_SUB2(b3[1], wl1, L.ex, wl4, L.ey); BA_SUB2(b3[2], wl2, L.ex, wl5, L.ey);
                        BA_ACC2(h[0], wl0, L.jl[0], wl3, L.jl[3]); BA_ACC2(h[1], wl0, L.jl[1], wl3, L.jl[4]); BA_ACC2(h[2], wl0, L.jl[2], wl3, L.jl[5]);
                        BA_ACC2(h[3], wl1, L.jl[1], wl4, L.jl[4]); BA_ACC2(h[4], wl1, L.jl[2], wl4, L.jl[5]); BA_ACC2(h[5], wl2, L.jl[2], wl5, L.jl[5]);
                    }
                    double *wd = Wt + 18 * bq;
                    if (MODE == 2) ll_ba[bq] = (int)((unsigned)recL[e0].lmkc >> 25);      // the block's pose (back-substitution from the stored blocks)
#pragma unroll
                    for (int t = 0; t < 18; ++t) wd[t] = wacc[t];
                    double *dd = Dl + 6 * bq;
#pragma unroll
                    for (int t = 0; t < 6; ++t) dd[t] = h[t];
                    Bl[3 * bq] = b3[0]; Bl[3 * bq + 1] = b3[1]; Bl[3 * bq + 2] = b3[2];
                }
                __syncthreads();
                // ... where the landmark's thread adds its blocks' shares in block order (read, barrier, then
                // overwrite: a landmark's own slot is one of the block slots), inverts Hll + lambda I and leaves
                // Dinv and bl per landmark
                for (int lj0 = 0; lj0 < l1 - l0; lj0 += BA_THREADS) {
                    const int lj = lj0 + tid;
                    double h[6] = { 0, 0, 0, 0, 0, 0 }, b3[3] = { 0, 0, 0 };
                    if (lj < l1 - l0) {
                        const int j = l0 + lj;
                        for (int b = lm_bstart[j] - bt0; b < lm_bstart[j + 1] - bt0; ++b) {
#pragma unroll
                            for (int t = 0; t < 6; ++t) h[t] += Dl[6 * b + t];
                            b3[0] += Bl[3 * b]; b3[1] += Bl[3 * b + 1]; b3[2] += Bl[3 * b + 2];
                        }
                    }
                    __syncthreads();
                    if (lj < l1 - l0) {
                        double D[9] = { h[0] + lambda, h[1], h[2], h[1], h[3] + lambda, h[4], h[2], h[4], h[5] + lambda }, Di[9];
                        d_inv3(D, Di);
                        double *dd = Dl + 6 * lj;
                        dd[0] = Di[0]; dd[1] = Di[1]; dd[2] = Di[2]; dd[3] = Di[4]; dd[4] = Di[5]; dd[5] = Di[8];
                        Bl[3 * lj] = b3[0]; Bl[3 * lj + 1] = b3[1]; Bl[3 * lj + 2] = b3[2];
                    }
                }
#pragma unroll
                for (int q = 0; q < BA_PIT_CAP / BA_THREADS; ++q) Pit[tid + q * BA_THREADS] = pre[q];
                if (tid <= npairs) Pcs[tid] = pcs_mine;
                for (int i2 = BA_THREADS + tid; i2 <= npairs; i2 += BA_THREADS) Pcs[i2] = pcs[tl * npairs + i2];
                __syncthreads();
                BA_PROF(8);
                {
                    // task = (pose pair, two of the six output rows).  Splitting a pair by output rows
                    // (not by items) keeps every S entry owned by one lane group, so no partial sums have
                    // to be combined.  Diagonal pairs get every block of their keyframe in the tile (a tile
                    // of single-view landmarks is nothing but two to five diagonal pairs), so they run on
                    // 16-lane rows, 32 at a time; the sparse off-diagonal pairs on 8-lane groups.
                    for (int tk = tid >> 4; tk < 3 * na; tk += BA_THREADS / 16) {
                        const int a = tk / 3, rg = tk - 3 * a;
                        const int pr = a * na - a * (a - 1) / 2;
                        const int c0 = Pcs[pr], c1 = Pcs[pr + 1];
                        if (c0 == c1) continue;
                        ba_schur_task<16>(a, a, rg, c0, c1, tid & 15, it0, Pit, pitem, Wt, Dl, Bl, S, bs, ld);
                    }
                    if (prof) { __syncthreads(); BA_PROF(10); }         // development: diagonal / off-diagonal split
                    if (prof) { __syncthreads(); BA_PROF(10); }         // development: diagonal / off-diagonal split
                    for (int tk = tid >> 3; tk < 3 * npairs; tk += BA_THREADS / 8) {
                        const int pr = tk / 3, rg = tk - 3 * pr;
                        const int c0 = Pcs[pr], c1 = Pcs[pr + 1];
                        if (c0 == c1) continue;
                        int a = 0, rem = pr;
                        while (rem >= na - a) { rem -= na - a; ++a; }
                        if (rem == 0) continue;                 // diagonal: done above
                        ba_schur_task<8>(a, a + rem, rg, c0, c1, tid & 7, it0, Pit, pitem, Wt, Dl, Bl, S, bs, ld);
                    }
                }
                __syncthreads();
                BA_PROF(9);
            }
            tempChi = currentChi;
            BA_PROF(3);
            if (MODE == 1 && sba.phase == 2) {                 // hand the partial sums to the host
                for (int i = tid; i < np * np; i += BA_THREADS) sio_S[i] = S[(size_t)(i / np) * ld + (i % np)];
                for (int i = tid; i < np; i += BA_THREADS) { sio_bs[i] = bs[i]; sio_bp[i] = bp[i]; }
                if (tid == 0) sio_sc[0] = currentChi;
                return;
            }
            if (MODE == 1) {                                   // phase 3: the reduced system comes back
                for (int i = tid; i < np * np; i += BA_THREADS)
                    S[(size_t)(i / np) * ld + (i % np)] = sio_S[i] + ((sba.add_lambda && i / np == i % np) ? sba.lambda : 0.0);
                for (int i = tid; i < np; i += BA_THREADS) { bs[i] = sio_bs[i]; bp[i] = sio_bp[i]; }
                __syncthreads();
            }
            if (MODE == 2) {
                // sync A: publish this shard's partial system, then every shard adds all partials in shard order
                // (shards without edges were zeroed by k_ba_split), lambda I goes on once
                // (16-byte write-through stores / past-L1 loads through a buffer descriptor: pairs of columns 2p, 2p + 1 <= r + 1
                // of row r — S is kept symmetric in LDS, so the pair that straddles the diagonal is valid too)
                BA_PHASE_TID;
                const __amdgpu_buffer_rsrc_t rs_all = ll_rsrc(ll_xs, (size_t)LLW * LL_SLAB(np));
                const size_t mine = (size_t)ll_w * LL_SLAB(np);
                const int hp_ = np >> 1;
                for (int it2 = tid; it2 < np * hp_; it2 += BA_THREADS) {
                    const int r = it2 / hp_, pc = 2 * (it2 - r * hp_);
                    if (pc <= r) ll_st2(rs_all, mine + (size_t)r * np + pc, S[(size_t)r * ld + pc], S[(size_t)r * ld + pc + 1]);
                }
                for (int it2 = tid; it2 < np; it2 += BA_THREADS) {          // bs | bp behind the matrix, in pairs
                    const double *src = it2 < hp_ ? bs + 2 * it2 : bp + 2 * (it2 - hp_);
                    ll_st2(rs_all, mine + (size_t)np * np + 2 * it2, src[0], src[1]);
                }
                if (!ll_sync()) { ll_failed = true; break; }
                for (int it2 = tid; it2 < np * hp_; it2 += BA_THREADS) {
                    const int r = it2 / hp_, pc = 2 * (it2 - r * hp_);
                    if (pc > r) continue;
                    ll_u4 part[LLW];
#pragma unroll
                    for (int v = 0; v < LLW; ++v) part[v] = ll_ld2(rs_all, (size_t)v * LL_SLAB(np) + (size_t)r * np + pc);
                    double s0 = ll_lo(part[0]), s1 = ll_hi(part[0]);
#pragma unroll
                    for (int v = 1; v < LLW; ++v) { s0 += ll_lo(part[v]); s1 += ll_hi(part[v]); }
                    if (pc == r) s0 += lambda;
                    if (pc + 1 == r) s1 += lambda;
                    S[(size_t)r * ld + pc] = s0; S[(size_t)pc * ld + r] = s0;
                    if (pc + 1 <= r) { S[(size_t)r * ld + pc + 1] = s1; S[(size_t)(pc + 1) * ld + r] = s1; }
                }
                for (int it2 = tid; it2 < np; it2 += BA_THREADS) {
                    ll_u4 part[LLW];
#pragma unroll
                    for (int v = 0; v < LLW; ++v) part[v] = ll_ld2(rs_all, (size_t)v * LL_SLAB(np) + (size_t)np * np + 2 * it2);
                    double s0 = ll_lo(part[0]), s1 = ll_hi(part[0]);
#pragma unroll
                    for (int v = 1; v < LLW; ++v) { s0 += ll_lo(part[v]); s1 += ll_hi(part[v]); }
                    double *dst = it2 < hp_ ? bs + 2 * it2 : bpt + 2 * (it2 - hp_);
                    dst[0] = s0; dst[1] = s1;
                }
                __syncthreads();
                BA_PROF(3);
            }
            const int ok2 = ba_chol_solve(S, ld, np, na, bs, xp, iflag, tid0, prof, tprev);
            BA_PROF(4);
            double scale_part = 0, scale_pose_part = 0;
            // MODE 2, the whole shard in ONE tile: W, (Hll + lambda I)^-1 and bl of every landmark are still in LDS
            const bool ll_stored = MODE == 2 && ntile == 1 && jd.nmv == nlm;
            if (ok2 && ll_stored) {
                BA_PHASE_TID;
                for (int j = tid; j < nlm; j += BA_THREADS) {
                    const int b0 = lm_bstart[j], b1 = lm_bstart[j + 1];
                    if (b1 <= b0) continue;
                    double g0 = 0, g1 = 0, g2 = 0;
                    for (int b = b0; b < b1; ++b) {
                        const double *w18 = Wt + 18 * b, *x6 = xp + 6 * ll_ba[b];
#pragma unroll
                        for (int r = 0; r < 6; ++r) { g0 += w18[3 * r] * x6[r]; g1 += w18[3 * r + 1] * x6[r]; g2 += w18[3 * r + 2] * x6[r]; }
                    }
                    const double *Di = Dl + 6 * j;
                    const double bl0 = Bl[3 * j], bl1 = Bl[3 * j + 1], bl2 = Bl[3 * j + 2];
                    const double c0 = bl0 - g0, c1 = bl1 - g1, c2 = bl2 - g2;
                    const double x0 = Di[0] * c0 + Di[1] * c1 + Di[2] * c2, x1 = Di[1] * c0 + Di[3] * c1 + Di[4] * c2,
                                 x2 = Di[2] * c0 + Di[4] * c1 + Di[5] * c2;
                    trial[3 * (size_t)j] = cur[3 * (size_t)j] + x0; trial[3 * (size_t)j + 1] = cur[3 * (size_t)j + 1] + x1;
                    trial[3 * (size_t)j + 2] = cur[3 * (size_t)j + 2] + x2;
                    scale_part += x0 * (lambda * x0 + bl0) + x1 * (lambda * x1 + bl1) + x2 * (lambda * x2 + bl2);
                }
            }
            if (ok2 && !ll_stored) {
                BA_PHASE_TID;
                // back-substitution with the Jacobians recomputed (the pose table still holds the
                // linearisation point): dl = Dinv (bl - sum W^T dp) = -Dinv sum Jl^T w (r + Jp dp)
                // (the next landmark's edge range and position, and the next edge record, are requested while the
                // current ones are worked on)
                int e0, e1;
                double X[3];
                { const int jc = min(tid, nlm - 1); e0 = lm_estart[jc]; e1 = lm_estart[jc + 1]; X[0] = cur[3 * (size_t)jc]; X[1] = cur[3 * (size_t)jc + 1]; X[2] = cur[3 * (size_t)jc + 2]; }
                for (int j = tid; j < nlm; j += BA_THREADS) {
                    const int jn = min(j + BA_THREADS, nlm - 1);
                    const int ne0 = lm_estart[jn], ne1 = lm_estart[jn + 1];
                    const double Xn[3] = { cur[3 * (size_t)jn], cur[3 * (size_t)jn + 1], cur[3 * (size_t)jn + 2] };
                    double h[6] = { 0, 0, 0, 0, 0, 0 }, g3[3] = { 0, 0, 0 }, b3[3] = { 0, 0, 0 };
                    BaRec rc = recL[min(e0, nobs - 1)];
                    for (int i = e0; i < e1; ++i) {
                        const BaRec rn = recL[min(i + 1, nobs - 1)];
                        const int kc = (unsigned)rc.lmkc >> 24;
                        BaLin L;
                        ba_linearize<EID>(PTab + BA_PT * (kc >> 1), CTab + BA_CT * (kc & 1), X, rc.u, rc.v, delta, L);
                        const double *x6 = xp + 6 * (kc >> 1);
                        double t0 = 0, t1 = 0;
#pragma unroll
                        for (int a = 0; a < 6; ++a) { t0 += L.jp[a] * x6[a]; t1 += L.jp[6 + a] * x6[a]; }
                        const double wl0 = L.w * L.jl[0], wl1 = L.w * L.jl[1], wl2 = L.w * L.jl[2],
                                     wl3 = L.w * L.jl[3], wl4 = L.w * L.jl[4], wl5 = L.w * L.jl[5];
                        BA_SUB2(b3[0], wl0, L.ex, wl3, L.ey); BA_SUB2(b3[1], wl1, L.ex, wl4, L.ey); BA_SUB2(b3[2], wl2, L.ex, wl5, L.ey);
                        BA_ACC2(g3[0], wl0, t0, wl3, t1); BA_ACC2(g3[1], wl1, t0, wl4, t1); BA_ACC2(g3[2], wl2, t0, wl5, t1);
                        BA_ACC2(h[0], wl0, L.jl[0], wl3, L.jl[3]); BA_ACC2(h[1], wl0, L.jl[1], wl3, L.jl[4]); BA_ACC2(h[2], wl0, L.jl[2], wl3, L.jl[5]);
                        BA_ACC2(h[3], wl1, L.jl[1], wl4, L.jl[4]); BA_ACC2(h[4], wl1, L.jl[2], wl4, L.jl[5]); BA_ACC2(h[5], wl2, L.jl[2], wl5, L.jl[5]);
                        rc = rn;
                    }
                    if (e1 > e0) {
                        double D[9] = { h[0] + lambda, h[1], h[2], h[1], h[3] + lambda, h[4], h[2], h[4], h[5] + lambda }, Di[9];
                        d_inv3(D, Di);
                        const double c0 = b3[0] - g3[0], c1 = b3[1] - g3[1], c2 = b3[2] - g3[2];
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            const double x = Di[a * 3] * c0 + Di[a * 3 + 1] * c1 + Di[a * 3 + 2] * c2;
                            trial[3 * (size_t)j + a] = X[a] + x;
                            scale_part += x * (lambda * x + b3[a]);
                        }
                    }
                    e0 = ne0; e1 = ne1; X[0] = Xn[0]; X[1] = Xn[1]; X[2] = Xn[2];
                }
            }
            if (ok2) {
                BA_PHASE_TID;
                for (int a = tid; a < na; a += BA_THREADS) {
                    const int k = act_kf[a];
                    double dT[7], Tn[7], x6[6];
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        x6[t] = xp[6 * a + t];
                        // shared map: every rank computes the same pose part; the host counts it once
                        if (MODE == 0) scale_part += x6[t] * (lambda * x6[t] + bp[6 * a + t]);
                        else scale_pose_part += x6[t] * (lambda * x6[t] + (MODE == 2 ? bpt : bp)[6 * a + t]);   // MODE 2: every shard computes the same
                    }
                    d_se3_exp(x6, dT);
                    d_se3_mul(dT, pcur + 7 * k, Tn);
#pragma unroll
                    for (int t = 0; t < 7; ++t) ptrial[7 * k + t] = Tn[t];
                }
            }
            double scale = block_sum(scale_part, red, tid);
            const double scale_pose = MODE != 0 ? block_sum(scale_pose_part, red, tid) : 0.0;
            __syncthreads();
            BA_PROF(5);
            // (a failed factorisation updates nothing: its errors are those of the unchanged state, like the oracle's)
            if (MODE == 1) tempChi = ok2 ? error_pass(trial, ptrial) : error_pass(cur, pcur);
            else if (ok2) {
                // the trial state's chi2 from the pose pass of the trial state: its Hpp / bp / pose table are the next
                // iteration's if the trial is accepted
                pose_table_into(PTab2, ptrial);
                tempChi = pose_pass(trial, PTab2, Hpp2, bp2);
                last_pts = trial; last_poses = ptrial;
            } else { last_pts = cur; last_poses = pcur; }
            BA_PROF(6);
            if (MODE == 1) {                                   // the host sums the partials and runs the rho test
                if (tid == 0) { sio_sc[2] = (double)ok2; sio_sc[3] = scale; sio_sc[4] = scale_pose; sio_sc[5] = tempChi; }
                return;
            }
            if (MODE == 2) {
                // sync B: chi2 of the trial state and the landmark part of the rho denominator, summed in shard order
                // (a failed factorisation still meets here: the exchange also keeps the slabs safe from the next trial)
                // The data is the flag (Guideline 16, R2): each double travels as two 8-byte granules { epoch tag, 32 bits },
                // one aligned store each; wave 0 sweeps the 4 LLW granules until every tag is this exchange's epoch (the
                // granules were zeroed by k_ba_split, epochs count from 1).  No counter, no drain, one hop.
                BA_PHASE_TID;
                ++ll_epb;
                unsigned long long *gran = reinterpret_cast<unsigned long long *>(ll_xb);
                if (tid < 4) {
                    const double val = tid < 2 ? (ok2 ? tempChi : 0.0) : scale;
                    const unsigned half = (tid & 1) ? (unsigned)__double2hiint(val) : (unsigned)__double2loint(val);
                    __hip_atomic_store(gran + 4 * ll_w + tid, ((unsigned long long)ll_epb << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (wv == 0) {
                    const bool mineg = lane < 4 * LLW && ((ll_mask >> (lane >> 2)) & 1u);
                    unsigned long long x = 0;
                    unsigned spins = 0;
                    int good = 1;
                    const long long t_wait0 = wall_clock64();
                    for (;;) {
                        if (mineg) x = __hip_atomic_load(gran + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__all(!mineg || (unsigned)(x >> 32) == ll_epb)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if ((++spins & 63u) == 0 &&
                            (wall_clock64() - t_wait0 > sba.ll_timeout || __hip_atomic_load(ll_cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                            __hip_atomic_store(ll_cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            good = 0;
                            break;
                        }
                    }
                    const int half = mineg ? (int)(unsigned)x : 0;          // shards without edges count as 0.0
                    double cs = 0, ss = 0;
#pragma unroll
                    for (int v = 0; v < LLW; ++v) {
                        cs += __hiloint2double(__builtin_amdgcn_readlane(half, 4 * v + 1), __builtin_amdgcn_readlane(half, 4 * v));
                        ss += __hiloint2double(__builtin_amdgcn_readlane(half, 4 * v + 3), __builtin_amdgcn_readlane(half, 4 * v + 2));
                    }
                    if (lane == 0) { red[0] = cs; red[1] = ss; iflag[1] = good; }
                }
                __syncthreads();
                if (!iflag[1]) { ll_failed = true; break; }
                tempChi = red[0]; scale = red[1] + scale_pose;
            }
            if (!ok2) tempChi = 1.7976931348623157e308;
            rho = currentChi - tempChi;
            scale += 1e-3;
            rho /= scale;
            if (sba.trace && tid == 0 && (MODE != 2 || ll_leader)) lm_trace_put(sba.trace, MODE == 2 ? ll_prob : job, it, lambda, currentChi, tempChi, rho, rho > 0 && isfinite(tempChi));
            if (rho > 0 && isfinite(tempChi)) {
                double t = 2 * rho - 1;
                double alpha = 1. - t * t * t;
                alpha = fmin(alpha, 2. / 3.);
                double sf = fmax(1. / 3., alpha);
                lambda *= sf; ni = 2; currentChi = tempChi;
                if (ok2) {                                 // MODE 0 only gets here: the trial state becomes the current one
                    double *t_ = cur; cur = trial; trial = t_; t_ = pcur; pcur = ptrial; ptrial = t_;
                    t_ = PTab; PTab = PTab2; PTab2 = t_;
                    for (int i = tid; i < 36 * na + np; i += BA_THREADS) Hpp[i] = Hpp2[i];   // bp follows Hpp, bp2 follows Hpp2
                    have_lin = true;
                }
            } else {
                lambda *= ni; ni *= 2;                     // rejected: `cur` was never touched
                if (!isfinite(lambda)) break;
            }
            ++qmax; ++trials_done;
            if (prof && tid == 0) prof[BA_PROF_N - 1] += 1;
        } while (rho < 0 && qmax < 10);
        if (ll_failed) break;
        ++it_done;
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) break;
    }
    __syncthreads();
    if (MODE == 2 && ll_failed) {                          // a shard never arrived: nothing is written back
        // (every shard says so, not only the leader: the shard that never arrived may be the leader)
        if (tid == 0) { jd.iters_done = -1; sba.parents[ll_prob].iters_done = -1; }
        return;
    }
    if (MODE == 1) {
        for (int i = tid; i < nobs; i += BA_THREADS) edge_chi2[lm_edges[i]] = err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1];
    } else {
        // per-edge chi2 of the last evaluated state (see last_pts): one sweep at the end instead of a residual array
        // rewritten by every trial
        pose_table_into(PTab2, last_poses);
        BA_PHASE_TID;
        BaRec rc = recL[min(tid, nobs - 1)];
        BaRec rn = recL[min(tid + BA_THREADS, nobs - 1)];
        double X[3];
        { const double *Xp = last_pts + 3 * (size_t)(rc.lmkc & BA_LM_MASK); X[0] = Xp[0]; X[1] = Xp[1]; X[2] = Xp[2]; }
        for (int i = tid; i < nobs; i += BA_THREADS) {
            const BaRec rnn = recL[min(i + 2 * BA_THREADS, nobs - 1)];
            const double *Xq = last_pts + 3 * (size_t)(rn.lmkc & BA_LM_MASK);
            const double Xn[3] = { Xq[0], Xq[1], Xq[2] };
            const int kc = (unsigned)rc.lmkc >> 24;
            BaProj o;
            ba_project<EID>(PTab2 + BA_PT * (kc >> 1), CTab + BA_CT * (kc & 1), X, rc.u, rc.v, o);
            edge_chi2[lm_edges[i]] = o.ex * o.ex + o.ey * o.ey;
            rc = rn; rn = rnn; X[0] = Xn[0]; X[1] = Xn[1]; X[2] = Xn[2];
        }
    }
    for (int j = tid; j < nlm; j += BA_THREADS) {
        double *d3 = pts_io + 3 * (size_t)lm_orig[j];
        d3[0] = cur[3 * (size_t)j]; d3[1] = cur[3 * (size_t)j + 1]; d3[2] = cur[3 * (size_t)j + 2];
    }
    if (pcur != poses && (MODE != 2 || ll_leader)) for (int i = tid; i < 7 * nkf; i += BA_THREADS) poses[i] = pcur[i];
    if (tid == 0) {
        jd.iters_done = it_done; jd.ntrial = trials_done;
        if (MODE == 2) {        // the problem's totals for its caller: iterations / trials once, block pairs summed over the shards
            if (ll_leader) { sba.parents[ll_prob].iters_done = it_done; sba.parents[ll_prob].ntrial = trials_done; }
            atomicAdd(&sba.parents[ll_prob].ncontrib, jd.ncontrib);
        }
    }
}

static inline size_t ba_lds_fixed_bytes(int max_kf)
{
    size_t np = 6 * (size_t)max_kf;
    return ((np + 1) * (np + 1) + 4 * np + 36 * (size_t)max_kf + BA_WAVES + 2 * BA_PT * (size_t)max_kf + 2 * BA_CT +
            32 * BA_ROWS) * sizeof(double) +
           ((BA_MAX_NP / 6) * (BA_MAX_NP / 6 + 1) / 2 + 1 + BA_PIT_CAP) * sizeof(int) + 64;
}
// landmarks / blocks per LDS tile: what the 160 KB leave after the reduced system, at most BA_TILE_MAX
static inline int ba_tile_cap(int max_kf)
{
    const size_t lim = BA_LDS_LIMIT, fixed = ba_lds_fixed_bytes(max_kf);
    if (fixed + 27 * sizeof(double) * 64 > lim) return 0;
    size_t t = (lim - fixed) / (27 * sizeof(double));
    t = t / 16 * 16;
    return (int)(t > BA_TILE_MAX ? BA_TILE_MAX : t);
}
static inline size_t ba_lds_bytes(int max_kf) { return ba_lds_fixed_bytes(max_kf) + 27 * sizeof(double) * (size_t)ba_tile_cap(max_kf); }
// low-latency shards (MODE 2): the resident records / positions take their share, a block also notes its pose
static inline size_t ba_lds_fixed_bytes_ll(int max_kf) { return ba_lds_fixed_bytes(max_kf) + 2 * LL_ECAP * sizeof(BaRec) + 6 * LL_LCAP * sizeof(double) + 64; }
static inline int ba_tile_cap_ll(int max_kf)
{
    const size_t lim = BA_LDS_LIMIT, fixed = ba_lds_fixed_bytes_ll(max_kf), per = 27 * sizeof(double) + sizeof(int);
    if (fixed + per * 64 > lim) return 0;
    size_t t = (lim - fixed) / per;
    t = t / 16 * 16;
    return (int)(t > BA_TILE_MAX ? BA_TILE_MAX : t);
}
static inline size_t ba_lds_bytes_ll(int max_kf) { return ba_lds_fixed_bytes_ll(max_kf) + (27 * sizeof(double) + sizeof(int)) * (size_t)ba_tile_cap_ll(max_kf); }
#pragma clang fp contract(off)
