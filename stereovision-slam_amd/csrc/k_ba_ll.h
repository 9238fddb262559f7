// k_ba_ll.h — low-latency local BA, the resident shape: ONE problem over LLW workgroups, every shard entirely in LDS.
// Backend::Optimize (reference src/backend.cpp:22-164) for a caller that waits for one camera's keyframe: the batch
// kernel (k_ba.h, one workgroup per problem) leaves 255 CUs idle for 1.2 ms.  Here k_ba_split deals the landmarks of
// a problem to LLW shards, k_ba_build builds every shard's structure (all keyframes active, every landmark in ONE LDS
// tile), and this kernel runs the whole Levenberg-Marquardt loop of g2o (the control flow of k_local_ba_t<0> /
// oracle/orc_geom.c:orc_local_ba) with the shards meeting twice per trial (k_ba.h: "Low-latency BA", syncs 0 / A / B).
//
// What differs from k_local_ba_t<2> (the general fallback, which streams a shard of any size through LDS tiles):
//   * the shard's edge records, landmark positions (current and trial), poses and index arrays are loaded into LDS once;
//     inside the LM loop only the exchanges leave the CU;
//   * ONE linearisation pass per evaluated state: a lane takes a (pose, landmark) block — blocks in pose-major order, so
//     the 27 normal-equation sums of a pose ride the 16-lane butterfly like the batch kernel's pose pass — and leaves the
//     block's W (6x3), its share of Hll and bl in LDS.  The pass over the TRIAL state gives the trial's chi2 and, if the
//     trial is accepted, is the next trial's linearisation (a rejected trial re-linearises the restored state: rare);
//   * the back-substitution uses the stored blocks: dl = (Hll + lambda I)^-1 (bl - sum_b W_b^T dp_b), no Jacobian again;
//   * the landmark part of the rho denominator rides the next pass' butterfly instead of two block reductions.
// A problem whose shards do not all fit the resident layout (LlCaps) is marked by k_ba_split and left to k_local_ba_t<2>.
#pragma once
#include "k_ba.h"

SVS_CONTRACT_FAST

struct LlCaps { int B, L, E; };      // blocks, landmarks, edges of a shard that the resident layout holds

// LDS bytes of the resident layout for `max_kf` keyframes and the capacities
static inline size_t ba_ll_lds_bytes(int max_kf, const LlCaps &c)
{
    const size_t np = 6 * (size_t)max_kf;
    const size_t dbl = (np + 1) * (np + 1) + 5 * np + 72 * (size_t)max_kf + BA_WAVES + 2 * BA_PT * (size_t)max_kf + 2 * BA_CT + 14 * (size_t)max_kf +
                       32 * BA_ROWS + 15 * (size_t)c.L + 27 * (size_t)c.B;
    const size_t ints = (size_t)(c.L + 1) + 3 * (size_t)(c.B + 1) + (size_t)c.E + 40 + (BA_MAX_NP / 6) * (BA_MAX_NP / 6 + 1) / 2 + 1 + BA_PIT_CAP + 16;
    return dbl * sizeof(double) + sizeof(BaRec) * (size_t)c.E + ints * sizeof(int) + 64;
}
// capacities for a K = 10 window over 16 workgroups (~250 edges, ~150 blocks, <= 140 landmarks a shard) with room to spare
static inline LlCaps ba_ll_caps(int max_kf)
{
    LlCaps c{ 0, 208, 448 };
    const size_t lim = BA_LDS_LIMIT;
    const LlCaps zero{ 0, c.L, c.E };
    const size_t fixed = ba_ll_lds_bytes(max_kf, zero);
    if (fixed + 64 * (27 * sizeof(double) + 12) > lim) return LlCaps{ 0, 0, 0 };
    size_t b = (lim - fixed) / (27 * sizeof(double) + 12);
    b = b / 16 * 16;
    c.B = (int)(b > 320 ? 320 : b);
    return c;
}

template <int LLW, bool EID = false>
__global__ void __launch_bounds__(BA_THREADS, BA_MIN_WAVES_PER_SIMD)
k_ba_ll(BaDev *shards, const BaCams *camsp, double *poses_all, double *pts_all, const BaRec *recs_all, const int *aux_all,
        double delta, int iters, double *edge_chi2_all, long long *prof_all, LlCaps cap, SbaArgs sba)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int job = blockIdx.x;
    BaDev &jd = shards[job];
    const int tid0 = threadIdx.x;
    const int tid = tid0, lane = tid & 63, wv = tid >> 6;
    if (jd.reserved != 2) return;                                     // not resident: k_local_ba_t<2> takes the problem
    const int nkf = jd.nkf, nlm = jd.nlm, nobs = jd.nobs, na = jd.na, np = 6 * jd.na, nblk = jd.nblk;
    if (nobs <= 0 || na <= 0) { if (tid == 0) jd.iters_done = 0; return; }
    long long *prof = (prof_all && job == 0) ? prof_all : nullptr;
    long long tprev = prof ? wall_clock64() : 0;
    const int ld = np + 1;
    // ---- LDS carve
    double *S = reinterpret_cast<double *>(smem);
    double *bs = S + (size_t)(np + 1) * ld;
    double *xp = bs + np;
    double *bpA = xp + np, *bpB = bpA + np, *bpt = bpB + np;
    double *HppA = bpt + np, *HppB = HppA + 36 * na;
    double *red = HppB + 36 * na;
    double *PTabA = red + BA_WAVES, *PTabB = PTabA + BA_PT * na;
    double *CTab = PTabB + BA_PT * na;
    double *PcA = CTab + 2 * BA_CT, *PcB = PcA + 7 * nkf;
    // (16-byte alignment by OFFSET from the LDS base: rounding a pointer through an integer loses its address space and every
    // access behind it becomes a FLAT instruction)
    auto al16 = [&](double *p) { return reinterpret_cast<double *>(smem + (((size_t)(reinterpret_cast<unsigned char *>(p) - smem) + 15) & ~(size_t)15)); };
    double *part = al16(PcB + 7 * nkf);                               // (double2 stores of the row butterfly)
    double *XA = part + 32 * BA_ROWS, *XB = XA + 3 * cap.L;
    double *Dl = XB + 3 * cap.L, *Bl = Dl + 6 * cap.L;
    double *Wt = Bl + 3 * cap.L, *Hb = Wt + 18 * cap.B, *Bb = Hb + 6 * cap.B;
    BaRec *rec = reinterpret_cast<BaRec *>(al16(Bb + 3 * cap.B));
    int *lm_bs = reinterpret_cast<int *>(rec + cap.E);
    int *blk_es = lm_bs + cap.L + 1, *blk_lm = blk_es + cap.B + 1, *blk_a = blk_lm + cap.B + 1;
    int *permP = blk_a + cap.B + 1;                                   // [cap.E] pose-major position -> landmark-major position of an edge
    int *estartP = permP + cap.E;                                     // [na + 1] (<= 33) pose-major edge ranges
    int *Pcs = estartP + 40;
    int *Pit = Pcs + (BA_MAX_NP / 6) * (BA_MAX_NP / 6 + 1) / 2 + 1;
    int *iflag = Pit + BA_PIT_CAP;

    const BaCams &cams = *camsp;
    double *poses = poses_all + (size_t)jd.kf_ofs * 7;
    double *pts_io = pts_all + (size_t)jd.lm_ofs * 3;
    double *edge_chi2 = edge_chi2_all + jd.obs_ofs;
    const BaRec *recL = recs_all + jd.rec_ofs, *recP = recL + nobs;
    const int *aux = aux_all + jd.aux_ofs;
    const BaAuxLayout AL = ba_aux_layout(nkf, nlm, nobs, jd.lay_nblk, jd.lay_na, jd.ncontrib, jd.lay_ntile);
    const int *g_lm_edges = aux + AL.lm_edges, *g_kf_estart = aux + AL.kf_estart, *g_lm_orig = aux + AL.lm_orig;
    const int *g_lm_bstart = aux + AL.lm_bstart, *g_blk_lm = aux + AL.blk_lm, *g_blk_es = aux + AL.blk_es;
    const int *g_act_kf = aux + AL.act_kf, *g_pcs = aux + AL.pcs, *g_pitem = aux + AL.pitem;
    const int npairs = na * (na + 1) / 2;

    // ---- this shard's place in its problem
    const int ll_prob = job / LLW, ll_w = job % LLW;
    const unsigned ll_mask = (unsigned)jd.shmask;
    const bool ll_leader = (ll_mask & ((1u << ll_w) - 1u)) == 0u;
    double *ll_xs = sba.xch + (size_t)ll_prob * sba.xch_stride;
    double *ll_x0 = ll_xs + (size_t)LLW * LL_SLAB(np);
    double *ll_xb = ll_x0 + (size_t)LLW * LL_X0(np);
    unsigned int *ll_cnt = sba.cnt + (size_t)LL_CNT_WORDS * ll_prob;
    unsigned ll_ep = 0, ll_epb = 0;
    const unsigned ll_n = __popc(ll_mask);
    auto ll_sync = [&]() -> bool {                                    // counter form (k_ba.h): drain, arrive, poll relaxed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ++ll_ep;
        if (tid0 == 0) {
            __hip_atomic_fetch_add(ll_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = ll_n * ll_ep;
            unsigned spins = 0;
            int good = 1;
            const long long t_wait0 = wall_clock64();
            while (__hip_atomic_load(ll_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
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

    // ---- load the shard
    for (int i = tid; i < nobs; i += BA_THREADS) rec[i] = recL[i];
    for (int j = tid; j <= nlm; j += BA_THREADS) lm_bs[j] = g_lm_bstart[j];
    for (int b = tid; b <= nblk; b += BA_THREADS) { blk_es[b] = b < nblk ? g_blk_es[b] : nobs; blk_lm[b] = b < nblk ? g_blk_lm[b] : 0; }
    for (int j = tid; j < nlm; j += BA_THREADS) {
        const double *s3 = pts_io + 3 * (size_t)g_lm_orig[j];
#pragma unroll
        for (int c = 0; c < 3; ++c) { XA[3 * j + c] = s3[c]; XB[3 * j + c] = s3[c]; }     // edge-less landmarks never move
    }
    for (int i = tid; i < 7 * nkf; i += BA_THREADS) { const double v = poses[i]; PcA[i] = v; PcB[i] = v; }
    for (int i = tid; i <= npairs; i += BA_THREADS) Pcs[i] = g_pcs[i];
    for (int i = tid; i < min(jd.ncontrib, BA_PIT_CAP); i += BA_THREADS) Pit[i] = g_pitem[i];
    if (tid < 2) {
        double *CT = CTab + BA_CT * tid;
        d_quat_to_R(cams.ext[tid], CT);
        CT[9] = cams.ext[tid][4]; CT[10] = cams.ext[tid][5]; CT[11] = cams.ext[tid][6];
        CT[12] = cams.cam[tid][0]; CT[13] = cams.cam[tid][1]; CT[14] = cams.cam[tid][2]; CT[15] = cams.cam[tid][3];
    }
    // the edges in pose-major order (the batch kernel's second record list) as positions in the resident landmark-major
    // list: record i of that list is the k-th edge of its block, the block starts at blk_es[blk]; every keyframe is an
    // active pose here (k_ba_build all_active), so pose a = keyframe act_kf[a]
    for (int i = tid; i < nobs; i += BA_THREADS) {
        const int b = recP[i].blk;
        int k = 0;                                                    // (the pose-major list keeps the order inside a block)
        while (i - k - 1 >= 0 && recP[i - k - 1].blk == b) ++k;
        permP[i] = g_blk_es[b] + k;
    }
    if (tid <= na) estartP[tid] = tid == na ? nobs : g_kf_estart[g_act_kf[tid]];
    __syncthreads();
    for (int b = tid; b < nblk; b += BA_THREADS) blk_a[b] = (int)((unsigned)rec[blk_es[b]].lmkc >> 25);
    __syncthreads();
    BA_PROF(1);

    // Current and trial state live in buffer pairs (positions, poses, pose table, Hpp, bp); `sx` says which half is the
    // current one.  (An index, not swapped pointers: pointers that may be either half lose their LDS address space and every
    // access through them becomes a FLAT instruction — measured 3x on the whole kernel.)
    int sx = 0, slast = 0;
#define LLX(s_)  (XA + (size_t)(s_) * 3 * cap.L)
#define LLP(s_)  (PcA + (size_t)(s_) * 7 * nkf)
#define LLT(s_)  (PTabA + (size_t)(s_) * BA_PT * na)
#define LLH(s_)  (HppA + (size_t)(s_) * 36 * na)
#define LLB(s_)  (bpA + (size_t)(s_) * np)

    auto pose_table_into = [&](double *tab, const double *src) {
        BA_PHASE_TID;
        __syncthreads();
        if (tid < na) {
            const double *T = src + 7 * g_act_kf[tid];
            double *PT = tab + BA_PT * tid;
            d_quat_to_R(T, PT);
            PT[9] = T[4]; PT[10] = T[5]; PT[11] = T[6];
        }
        __syncthreads();
    };

    // ---- the linearisation pass at a state, the workgroup split in two (the register sets of the two jobs do not add up):
    //   waves 4..7  a lane takes a (pose, landmark) BLOCK: W (6x3), its share of Hll and bl -> LDS (Wt / Hb / Bb)
    //   waves 0..3  a lane takes an EDGE, pose-major like the batch kernel's pose pass: the 27 normal-equation sums of a pose
    //               and chi2 ride the 16-lane butterfly -> Hout / bout; c28 / c29 are two more per-thread values of these
    //               waves to be summed (the rho denominator: the back-substitution ran on threads < nlm <= 256)
    auto lin_pass = [&](const double *Xs, const double *tab, double *Hout, double *bout, double c28, double c29, double &o28, double &o29) -> double {
        BA_PHASE_TID;
        constexpr int HROWS = BA_ROWS / 2;                          // 16-lane rows of the edge half
        const int rpp = HROWS / na;                                   // rows per pose (>= 1: na <= 16)
        double acc[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) acc[t] = 0;
        const int row = tid >> 4, rl = tid & 15;
        if (wv >= BA_WAVES / 2) {
            for (int b = tid - BA_THREADS / 2; b < nblk; b += BA_THREADS / 2) {
                const int j = blk_lm[b];
                const double *PT = tab + BA_PT * blk_a[b];
                const double X[3] = { Xs[3 * j], Xs[3 * j + 1], Xs[3 * j + 2] };
                double h[6] = { 0, 0, 0, 0, 0, 0 }, b3[3] = { 0, 0, 0 }, wacc[18];
#pragma unroll
                for (int t = 0; t < 18; ++t) wacc[t] = 0;
                for (int e = blk_es[b]; e < blk_es[b + 1]; ++e) {
                    const BaRec rc = rec[e];
                    const int kc = (unsigned)rc.lmkc >> 24;
                    BaLin L;
                    ba_linearize<EID>(PT, CTab + BA_CT * (kc & 1), X, rc.u, rc.v, delta, L);
                    const double wl0 = L.w * L.jl[0], wl1 = L.w * L.jl[1], wl2 = L.w * L.jl[2],
                                 wl3 = L.w * L.jl[3], wl4 = L.w * L.jl[4], wl5 = L.w * L.jl[5];
                    ba_acc_w<EID>(wacc, L.jp, wl0, wl1, wl2, wl3, wl4, wl5);
                    BA_SUB2(b3[0], wl0, L.ex, wl3, L.ey); BA_SUB2(b3[1], wl1, L.ex, wl4, L.ey); BA_SUB2(b3[2], wl2, L.ex, wl5, L.ey);
                    BA_ACC2(h[0], wl0, L.jl[0], wl3, L.jl[3]); BA_ACC2(h[1], wl0, L.jl[1], wl3, L.jl[4]); BA_ACC2(h[2], wl0, L.jl[2], wl3, L.jl[5]);
                    BA_ACC2(h[3], wl1, L.jl[1], wl4, L.jl[4]); BA_ACC2(h[4], wl1, L.jl[2], wl4, L.jl[5]); BA_ACC2(h[5], wl2, L.jl[2], wl5, L.jl[5]);
                }
                double *wd = Wt + 18 * b;
#pragma unroll
                for (int t = 0; t < 18; ++t) wd[t] = wacc[t];
                double *hd = Hb + 6 * b;
#pragma unroll
                for (int t = 0; t < 6; ++t) hd[t] = h[t];
                Bb[3 * b] = b3[0]; Bb[3 * b + 1] = b3[1]; Bb[3 * b + 2] = b3[2];
            }
        } else {
            const int a = row % na, sub = row / na;
            acc[28] = c28; acc[29] = c29;
            if (sub < rpp) {
                const double *PT = tab + BA_PT * a;
                for (int i = estartP[a] + sub * 16 + rl; i < estartP[a + 1]; i += 16 * rpp) {
                    const BaRec rc = rec[permP[i]];
                    const int kc = (unsigned)rc.lmkc >> 24, j = rc.lmkc & BA_LM_MASK;
                    const double *CT = CTab + BA_CT * (kc & 1);
                    const double X[3] = { Xs[3 * j], Xs[3 * j + 1], Xs[3 * j + 2] };
                    BaProj o;
                    ba_project<EID>(PT, CT, X, rc.u, rc.v, o);
                    double r0, w;
                    d_huber(o.ex * o.ex + o.ey * o.ey, delta, r0, w);
                    acc[27] += r0;
                    double M[6], jp[12];
                    ba_jac_pose<EID>(CT, o, M, jp);
                    ba_acc_pose<EID>(acc, jp, w, o.ex, o.ey);
                }
            }
        }
        int code = 0;
        if (wv < BA_WAVES / 2) code = ba_row_sum32(acc, lane);
        __syncthreads();                              // `part` may still be read by the previous phase
        if (wv < BA_WAVES / 2) reinterpret_cast<double2 *>(part + 32 * row)[code] = make_double2(acc[0], acc[1]);
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
        double chi = 0, s28 = 0, s29 = 0;
#pragma unroll
        for (int r = 0; r < HROWS; ++r) { chi += part[r * 32 + 27]; s28 += part[r * 32 + 28]; s29 += part[r * 32 + 29]; }
        o28 = s28; o29 = s29;
        __syncthreads();
        return chi;
    };

    double lambda = 0, ni = 2;
    int it_done = 0, trials_done = 0;
    double currentChi = 0;
    bool ll_failed = false;
    const __amdgpu_buffer_rsrc_t rs_all = ll_rsrc(ll_xs, (size_t)LLW * LL_SLAB(np));
    const size_t mine = (size_t)ll_w * LL_SLAB(np);
    const int hp_ = np >> 1;
    for (int it = 0; it < iters; ++it) {
        if (it == 0) {
            BA_PHASE_TID;
            double d0, d1;
            pose_table_into(LLT(sx), LLP(sx));
            const double chi_lin = lin_pass(LLX(sx), LLT(sx), LLH(sx), LLB(sx), 0.0, 0.0, d0, d1);
            // lambda_0 = 1e-5 * max diagonal of the Hessian (sync 0: pose diagonals summed over the shards)
            double md = 0;
            for (int j = tid; j < nlm; j += BA_THREADS) {
                double h0 = 0, h3 = 0, h5 = 0;
                for (int b = lm_bs[j]; b < lm_bs[j + 1]; ++b) { h0 += Hb[6 * b]; h3 += Hb[6 * b + 3]; h5 += Hb[6 * b + 5]; }
                if (lm_bs[j + 1] > lm_bs[j]) md = fmax(md, fmax(fabs(h0), fmax(fabs(h3), fabs(h5))));
            }
            md = block_max(md, red, tid);
            for (int i = tid; i < np; i += BA_THREADS) ll_st(ll_x0 + (size_t)ll_w * LL_X0(np) + i, LLH(sx)[36 * (i / 6) + (i % 6) * 7]);
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
            currentChi = chs;
            lambda = 1e-5 * md; ni = 2;
            BA_PROF(0);
        }
        double rho = 0;
        int qmax = 0;
        do {
            double tempChi = currentChi;
            // ---- (Hll + lambda I)^-1 and bl per landmark from the blocks' shares; S = blockdiag(Hpp) (this shard's), bs = bp
            {
                BA_PHASE_TID;
                const double *Hpp = LLH(sx), *bp = LLB(sx);
                for (int j = tid; j < nlm; j += BA_THREADS) {
                    double h[6] = { 0, 0, 0, 0, 0, 0 }, b3[3] = { 0, 0, 0 };
                    for (int b = lm_bs[j]; b < lm_bs[j + 1]; ++b) {
#pragma unroll
                        for (int t = 0; t < 6; ++t) h[t] += Hb[6 * b + t];
                        b3[0] += Bb[3 * b]; b3[1] += Bb[3 * b + 1]; b3[2] += Bb[3 * b + 2];
                    }
                    double D[9] = { h[0] + lambda, h[1], h[2], h[1], h[3] + lambda, h[4], h[2], h[4], h[5] + lambda }, Di[9];
                    d_inv3(D, Di);
                    double *dd = Dl + 6 * j;
                    dd[0] = Di[0]; dd[1] = Di[1]; dd[2] = Di[2]; dd[3] = Di[4]; dd[4] = Di[5]; dd[5] = Di[8];
                    Bl[3 * j] = b3[0]; Bl[3 * j + 1] = b3[1]; Bl[3 * j + 2] = b3[2];
                }
                for (int r = wv; r < np; r += BA_WAVES)
                    for (int c = lane; c < np; c += 64) {
                        double v = 0;
                        if (r / 6 == c / 6) v = Hpp[36 * (r / 6) + (r % 6) * 6 + (c % 6)];
                        S[(size_t)r * ld + c] = v;
                    }
                for (int i = tid; i < np; i += BA_THREADS) bs[i] = bp[i];
            }
            __syncthreads();
            BA_PROF(2);
            // ---- Schur complement of the shard's landmarks: the tasks of the batch kernel's tile sweep on the one resident tile
            {
                BA_PHASE_TID;
                for (int tk = tid >> 4; tk < 3 * na; tk += BA_THREADS / 16) {
                    const int a = tk / 3, rg = tk - 3 * a;
                    const int pr = a * na - a * (a - 1) / 2;
                    const int c0 = Pcs[pr], c1 = Pcs[pr + 1];
                    if (c0 == c1) continue;
                    ba_schur_task<16>(a, a, rg, c0, c1, tid & 15, 0, Pit, g_pitem, Wt, Dl, Bl, S, bs, ld);
                }
                for (int tk = tid >> 3; tk < 3 * npairs; tk += BA_THREADS / 8) {
                    const int pr = tk / 3, rg = tk - 3 * pr;
                    const int c0 = Pcs[pr], c1 = Pcs[pr + 1];
                    if (c0 == c1) continue;
                    int a = 0, rem = pr;
                    while (rem >= na - a) { rem -= na - a; ++a; }
                    if (rem == 0) continue;
                    ba_schur_task<8>(a, a + rem, rg, c0, c1, tid & 7, 0, Pit, g_pitem, Wt, Dl, Bl, S, bs, ld);
                }
            }
            __syncthreads();
            BA_PROF(9);
            // ---- sync A: publish the partial system, add all partials in shard order, lambda I once
            {
                BA_PHASE_TID;
                for (int it2 = tid; it2 < np * hp_; it2 += BA_THREADS) {
                    const int r = it2 / hp_, pc = 2 * (it2 - r * hp_);
                    if (pc <= r) ll_st2(rs_all, mine + (size_t)r * np + pc, S[(size_t)r * ld + pc], S[(size_t)r * ld + pc + 1]);
                }
                for (int it2 = tid; it2 < np; it2 += BA_THREADS) {
                    const double *src = it2 < hp_ ? bs + 2 * it2 : LLB(sx) + 2 * (it2 - hp_);
                    ll_st2(rs_all, mine + (size_t)np * np + 2 * it2, src[0], src[1]);
                }
                if (!ll_sync()) { ll_failed = true; break; }
                for (int it2 = tid; it2 < np * hp_; it2 += BA_THREADS) {
                    const int r = it2 / hp_, pc = 2 * (it2 - r * hp_);
                    if (pc > r) continue;
                    ll_u4 pv[LLW];
#pragma unroll
                    for (int v = 0; v < LLW; ++v) pv[v] = ll_ld2(rs_all, (size_t)v * LL_SLAB(np) + (size_t)r * np + pc);
                    double s0 = ll_lo(pv[0]), s1 = ll_hi(pv[0]);
#pragma unroll
                    for (int v = 1; v < LLW; ++v) { s0 += ll_lo(pv[v]); s1 += ll_hi(pv[v]); }
                    if (pc == r) s0 += lambda;
                    if (pc + 1 == r) s1 += lambda;
                    S[(size_t)r * ld + pc] = s0; S[(size_t)pc * ld + r] = s0;
                    if (pc + 1 <= r) { S[(size_t)r * ld + pc + 1] = s1; S[(size_t)(pc + 1) * ld + r] = s1; }
                }
                for (int it2 = tid; it2 < np; it2 += BA_THREADS) {
                    ll_u4 pv[LLW];
#pragma unroll
                    for (int v = 0; v < LLW; ++v) pv[v] = ll_ld2(rs_all, (size_t)v * LL_SLAB(np) + (size_t)np * np + 2 * it2);
                    double s0 = ll_lo(pv[0]), s1 = ll_hi(pv[0]);
#pragma unroll
                    for (int v = 1; v < LLW; ++v) { s0 += ll_lo(pv[v]); s1 += ll_hi(pv[v]); }
                    double *dst = it2 < hp_ ? bs + 2 * it2 : bpt + 2 * (it2 - hp_);
                    dst[0] = s0; dst[1] = s1;
                }
                __syncthreads();
            }
            BA_PROF(3);
            const int ok2 = ba_chol_solve(S, ld, np, na, bs, xp, iflag, tid0, prof, tprev);
            BA_PROF(4);
            double scale_part = 0, scale_pose_part = 0, scale = 0, scale_pose = 0;
            if (ok2) {
                BA_PHASE_TID;
                const double *Xc = LLX(sx), *Pc = LLP(sx);
                double *Xt = LLX(sx ^ 1), *Pt = LLP(sx ^ 1);
                // back-substitution from the stored blocks: dl = Dinv (bl - sum_b W_b^T dp_b)
                for (int j = tid; j < nlm; j += BA_THREADS) {
                    const int b0 = lm_bs[j], b1 = lm_bs[j + 1];
                    if (b1 <= b0) continue;
                    double g0 = 0, g1 = 0, g2 = 0;
                    for (int b = b0; b < b1; ++b) {
                        const double *w18 = Wt + 18 * b, *x6 = xp + 6 * blk_a[b];
#pragma unroll
                        for (int r = 0; r < 6; ++r) { g0 += w18[3 * r] * x6[r]; g1 += w18[3 * r + 1] * x6[r]; g2 += w18[3 * r + 2] * x6[r]; }
                    }
                    const double *Di = Dl + 6 * j;
                    const double bl0 = Bl[3 * j], bl1 = Bl[3 * j + 1], bl2 = Bl[3 * j + 2];
                    const double c0 = bl0 - g0, c1 = bl1 - g1, c2 = bl2 - g2;
                    const double x0 = Di[0] * c0 + Di[1] * c1 + Di[2] * c2, x1 = Di[1] * c0 + Di[3] * c1 + Di[4] * c2,
                                 x2 = Di[2] * c0 + Di[4] * c1 + Di[5] * c2;
                    Xt[3 * j] = Xc[3 * j] + x0; Xt[3 * j + 1] = Xc[3 * j + 1] + x1; Xt[3 * j + 2] = Xc[3 * j + 2] + x2;
                    scale_part += x0 * (lambda * x0 + bl0) + x1 * (lambda * x1 + bl1) + x2 * (lambda * x2 + bl2);
                }
                for (int a = tid; a < na; a += BA_THREADS) {
                    const int k = g_act_kf[a];
                    double dT[7], Tn[7], x6[6];
#pragma unroll
                    for (int t = 0; t < 6; ++t) { x6[t] = xp[6 * a + t]; scale_pose_part += x6[t] * (lambda * x6[t] + bpt[6 * a + t]); }   // every shard computes the same
                    d_se3_exp(x6, dT);
                    d_se3_mul(dT, Pc + 7 * k, Tn);
#pragma unroll
                    for (int t = 0; t < 7; ++t) Pt[7 * k + t] = Tn[t];
                }
                BA_PROF(5);
                // the trial state: its chi2 — and, if it is accepted, the next trial's linearisation
                pose_table_into(LLT(sx ^ 1), LLP(sx ^ 1));
                tempChi = lin_pass(LLX(sx ^ 1), LLT(sx ^ 1), LLH(sx ^ 1), LLB(sx ^ 1), scale_part, scale_pose_part, scale, scale_pose);
                slast = sx ^ 1;
            } else slast = sx;
            BA_PROF(6);
            // ---- sync B (granules, k_ba.h): chi2 of the trial state and the landmark part of the rho denominator
            {
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
                    const int half = mineg ? (int)(unsigned)x : 0;
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
                __syncthreads();                       // red is reused by the next block reduction
            }
            BA_PROF(8);
            if (!ok2) tempChi = 1.7976931348623157e308;
            rho = currentChi - tempChi;
            scale += 1e-3;
            rho /= scale;
            if (sba.trace && tid == 0 && ll_leader) lm_trace_put(sba.trace, ll_prob, it, lambda, currentChi, tempChi, rho, rho > 0 && isfinite(tempChi));
            ++qmax; ++trials_done;
            if (rho > 0 && isfinite(tempChi)) {
                double t = 2 * rho - 1;
                double alpha = 1. - t * t * t;
                alpha = fmin(alpha, 2. / 3.);
                double sf = fmax(1. / 3., alpha);
                lambda *= sf; ni = 2; currentChi = tempChi;
                sx ^= 1;                                 // the trial state (and its linearisation, already in LDS) becomes the current one
            } else {
                lambda *= ni; ni *= 2;
                if (!isfinite(lambda)) break;
                // the blocks in LDS are the rejected trial's: linearise the restored state again if another trial follows
                if (ok2 && rho < 0 && qmax < 10) { double d0, d1; (void)lin_pass(LLX(sx), LLT(sx), LLH(sx), LLB(sx), 0.0, 0.0, d0, d1); }
            }
            if (prof && tid == 0) prof[BA_PROF_N - 1] += 1;
        } while (rho < 0 && qmax < 10);
        if (ll_failed) break;
        ++it_done;
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) break;
    }
    __syncthreads();
    if (ll_failed) {
        // (every shard says so, not only the leader: the shard that never arrived may be the leader)
        if (tid == 0) { jd.iters_done = -1; sba.parents[ll_prob].iters_done = -1; }
        return;
    }
    // per-edge chi2 of the last evaluated state (g2o reports the errors of its last computeActiveErrors)
    // (the table of the other half is free: the state evaluated last is the current one or the rejected trial)
    pose_table_into(LLT(slast), LLP(slast));
    for (int i = tid; i < nobs; i += BA_THREADS) {
        const BaRec rc = rec[i];
        const int j = rc.lmkc & BA_LM_MASK, kc = (unsigned)rc.lmkc >> 24;
        const double *last_pts = LLX(slast);
        const double X[3] = { last_pts[3 * j], last_pts[3 * j + 1], last_pts[3 * j + 2] };
        BaProj o;
        ba_project<EID>(LLT(slast) + BA_PT * (kc >> 1), CTab + BA_CT * (kc & 1), X, rc.u, rc.v, o);
        edge_chi2[g_lm_edges[i]] = o.ex * o.ex + o.ey * o.ey;
    }
    for (int j = tid; j < nlm; j += BA_THREADS) {
        double *d3 = pts_io + 3 * (size_t)g_lm_orig[j];
        d3[0] = LLX(sx)[3 * j]; d3[1] = LLX(sx)[3 * j + 1]; d3[2] = LLX(sx)[3 * j + 2];
    }
    if (ll_leader) for (int i = tid; i < 7 * nkf; i += BA_THREADS) poses[i] = LLP(sx)[i];
    if (tid == 0) {
        jd.iters_done = it_done; jd.ntrial = trials_done;
        if (ll_leader) { sba.parents[ll_prob].iters_done = it_done; sba.parents[ll_prob].ntrial = trials_done; }
        atomicAdd(&sba.parents[ll_prob].ncontrib, jd.ncontrib);     // block pairs of the problem = sum over its shards
    }
}
#undef LLX
#undef LLP
#undef LLT
#undef LLH
#undef LLB
#pragma clang fp contract(off)
