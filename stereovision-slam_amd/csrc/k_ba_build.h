// k_ba_build.h — the structure of a local-BA problem built on the device.
// k_local_ba (k_ba.h) works on edge records in two orders, landmark ranges, (pose, landmark) blocks,
// LDS tiles and per-tile lists of block pairs.  BaHostStruct::build derives all of that from the raw
// observation list the backend gathers (src/backend.cpp:83-160) — on the host it was the largest
// single consumer of CPU time of the whole path (6 ms of wall time per 100 problems on 4 threads).
// This kernel produces exactly the same arrays (same numbering, same orders, hence bit-identical
// optimisation results) with one workgroup per problem: counting sorts become flattened
// [key][thread] exclusive scans, the per-pair item lists are emitted by one thread per list walking
// the landmarks' pose bit masks.  The host only validates indices and supplies the edges'
// (landmark, keyframe) order (the identity for the order the pipeline gathers them in).
//
// aux layout: ba_aux_layout() with CAPACITIES for the block / tile counts (BaDev::lay_*), so the
// space of a problem can be reserved before its counts are known.
#pragma once
#include "k_ba.h"

#pragma clang fp contract(off)

#define BB_THREADS 256
// BB_MAXKEYS (k_ba.h): block counts 0..32 / keyframes 0..31; the grouped landmark sort needs max count + nkf keys

// upper bound of the number of LDS tiles of a problem (greedy packing with two capacities)
__host__ __device__ inline int ba_tile_bound(int nlm, int nobs, int nkf, int tile_cap)
{
    const int per = tile_cap - nkf > 1 ? tile_cap - nkf : 1;
    return nobs / per + nlm / tile_cap + 2;
}
// pitem capacity: a landmark seen from k keyframes lists k (k + 1) / 2 pairs
__host__ __device__ inline size_t ba_pitem_bound(int nobs, int nkf) { return (size_t)nobs * (size_t)(nkf + 1) / 2 + 1; }

// exclusive scan of a[0..n) in LDS by the whole workgroup; returns the total.  tmp: BB_THREADS ints.
__device__ __forceinline__ int bb_exscan(int *a, int n, int *tmp, int tid)
{
    const int per = (n + BB_THREADS - 1) / BB_THREADS;
    const int i0 = min(tid * per, n), i1 = min(i0 + per, n);
    int s = 0;
    for (int i = i0; i < i1; ++i) s += a[i];
    __syncthreads();
    tmp[tid] = s;
    __syncthreads();
    for (int d = 1; d < BB_THREADS; d <<= 1) {
        const int add = tid >= d ? tmp[tid - d] : 0;
        __syncthreads();
        tmp[tid] += add;
        __syncthreads();
    }
    const int total = tmp[BB_THREADS - 1];
    int run = tmp[tid] - s;
    for (int i = i0; i < i1; ++i) { const int v = a[i]; a[i] = run; run += v; }
    __syncthreads();
    return total;
}

// LDS ints needed for a batch whose largest problem has max_nlm landmarks (and, when it still fits, one packed
// word per edge: the edges' landmark / keyframe / camera are then read from global memory once)
static inline size_t bb_lds_ints(int max_nlm)
{
    return (size_t)6 * ((size_t)max_nlm + 2) + (size_t)BB_MAXKEYS * BB_THREADS + BB_THREADS + 5 * BB_MAXKEYS + 64;
}
static inline bool bb_edge_cache_fits(int max_nlm, int max_nobs) { return (bb_lds_ints(max_nlm) + (size_t)max_nobs) * sizeof(int) <= 150 * 1024; }
static inline size_t bb_lds_bytes(int max_nlm, int max_nobs = 0)
{
    return (bb_lds_ints(max_nlm) + (bb_edge_cache_fits(max_nlm, max_nobs) ? (size_t)max_nobs : 0)) * sizeof(int);
}

__global__ void __launch_bounds__(BB_THREADS)
k_ba_build(BaDev *jobs, const unsigned int *obs_packed, const float2 *obs_uv,
           const int *srt_all, BaRec *recs_all, int *aux_all, int tile_cap, int max_nlm, int *err_flag, int all_active,
           int edge_cache, int no_group)
{
    extern __shared__ __attribute__((aligned(16))) int bb_lds[];
    BaDev &jd = jobs[blockIdx.x];
    const int tid = threadIdx.x;
    const int nkf = jd.nkf, nlm = jd.nlm, nobs = jd.nobs;
    // edges in caller order, one packed word each (landmark | keyframe << 16 | camera << 24, written by the host
    // while it validates them); srt = their (landmark, keyframe) order, not even read when that is the identity
    const unsigned int *opk = obs_packed + jd.obs_ofs;
    const int *srt = srt_all + jd.obs_ofs;
    const bool ident = jd.reserved != 0;
    const unsigned int lm_base = (unsigned int)jd.lm_base;      // a low-latency shard: landmarks [lm_base, lm_base + nlm) of its parent
    const float2 *ouv = obs_uv + jd.obs_ofs;
    BaRec *recL = recs_all + jd.rec_ofs, *recP = recL + nobs;
    int *aux = aux_all + jd.aux_ofs;
    if (nobs <= 0 || nlm <= 0 || nkf <= 0) {
        if (tid == 0) { jd.nblk = 0; jd.na = 0; jd.ncontrib = 0; jd.ntile = 0; jd.nmv = 0; jd.ntrial = 0; }
        return;
    }
    const BaAuxLayout AL = ba_aux_layout(nkf, nlm, nobs, jd.lay_nblk, jd.lay_na, 0, jd.lay_ntile);
    int *g_lm_estart = aux + AL.lm_estart, *g_lm_edges = aux + AL.lm_edges, *g_kf_estart = aux + AL.kf_estart;
    int *g_lm_orig = aux + AL.lm_orig, *g_lm_bstart = aux + AL.lm_bstart, *g_blk_kf = aux + AL.blk_kf, *g_blk_lm = aux + AL.blk_lm;
    int *g_kf_pidx = aux + AL.kf_pidx, *g_act_kf = aux + AL.act_kf, *g_tile_lm = aux + AL.tile_lm, *g_pcs = aux + AL.pcs;
    int *g_sv_start = aux + AL.sv_start, *g_blk_es = aux + AL.blk_es;
    int *g_pitem = aux + AL.pitem;

    // LDS carve
    const int N1 = max_nlm + 2;
    int *ostart = bb_lds;                 // [nlm + 1] edge range of landmark l (caller numbering) in sorted order
    int *cnt = ostart + N1;               // [nlm]     blocks of landmark l
    int *lm_orig = cnt + N1;              // [nlm]     new -> caller
    int *lm_bs = lm_orig + N1;            // [nlm + 1] block range (new numbering)
    int *lm_es = lm_bs + N1;              // [nlm + 1] edge range (new numbering)
    int *pmask = lm_es + N1;              // [nlm]     active-pose bits of a landmark (new numbering)
    int *flat = pmask + N1;               // [BB_MAXKEYS * BB_THREADS]
    int *tmp = flat + BB_MAXKEYS * BB_THREADS;   // [BB_THREADS]
    int *kcnt = tmp + BB_THREADS;         // [BB_MAXKEYS] edges per keyframe, then kf_estart
    int *pidx = kcnt + BB_MAXKEYS;        // [BB_MAXKEYS]
    int *small = pidx + BB_MAXKEYS;       // scalars: 0 maxc, 1 na, 2 ntile, 3 ncontrib; + 8: active pose -> keyframe
    unsigned int *ed = reinterpret_cast<unsigned int *>(small + 8 + BB_MAXKEYS + 8);   // [nobs] landmark | keyframe << 16 | camera << 24, sorted order
    // the edges' indices, read from global memory ONCE (independent, coalesced loads, four in flight per thread)
    if (edge_cache) {
        for (int i0 = tid; i0 < nobs; i0 += 4 * BB_THREADS) {
            int e4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * BB_THREADS; e4[u] = i < nobs ? (ident ? i : srt[i]) : 0; }
            unsigned int v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v4[u] = opk[e4[u]] - lm_base;
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * BB_THREADS; if (i < nobs) ed[i] = v4[u]; }
        }
    }
    auto edge = [&](int i) -> unsigned int {
        if (edge_cache) return ed[i];
        return opk[ident ? i : srt[i]] - lm_base;
    };

    for (int l = tid; l <= nlm; l += BB_THREADS) { cnt[l] = 0; }
    if (tid < BB_MAXKEYS) { kcnt[tid] = 0; pidx[tid] = -1; }
    if (tid < 8) small[tid] = 0;
    __syncthreads();
    // ---- 1. edge ranges per landmark (the edges arrive landmark-major through srt), blocks per landmark,
    //         edges per keyframe
    for (int i = tid; i < nobs; i += BB_THREADS) {
        const unsigned int ev = edge(i);
        const int l = (int)(ev & 0xffffu), k = (int)((ev >> 16) & 0xffu);
        int lp = -1, kp = -1;
        if (i > 0) { const unsigned int pv = edge(i - 1); lp = (int)(pv & 0xffffu); kp = (int)((pv >> 16) & 0xffu); }
        if (l != lp) for (int q = lp + 1; q <= l; ++q) ostart[q] = i;
        if (i == nobs - 1) for (int q = l + 1; q <= nlm; ++q) ostart[q] = nobs;
        if (l != lp || k != kp) atomicAdd(&cnt[l], 1);
        atomicAdd(&kcnt[k], 1);
    }
    __syncthreads();
    // ---- 2. keyframes: edge ranges, active poses (tiny, one thread)
    if (tid == 0) {
        int run = 0, na = 0;
        for (int k = 0; k < nkf; ++k) {
            const int c = kcnt[k];
            g_kf_estart[k] = run;
            // shared-map BA: every keyframe takes part on every rank (a rank's shard may not see all of them)
            if (c > 0 || all_active) { pidx[k] = na; g_act_kf[na] = k; ++na; }
            g_kf_pidx[k] = pidx[k];
            kcnt[k] = run;                // now: kf_estart
            run += c;
        }
        g_kf_estart[nkf] = run;
        for (int a = na; a < nkf; ++a) g_act_kf[a] = -1;
        small[1] = na;
    }
    // max block count
    {
        int m = 0;
        for (int l = tid; l < nlm; l += BB_THREADS) m = max(m, cnt[l]);
        atomicMax(&small[0], m);
    }
    __syncthreads();
    const int maxc = small[0], na = small[1];
    const int npairs = na * (na + 1) / 2;
    // ---- 3. landmarks renumbered by descending block count, the single-view ones grouped by their keyframe, the
    //         ones without edges last (BaHostStruct::build has the same key), stable: flattened [key][thread]
    //         counting sort.  sv_start[k] = first landmark of keyframe k's single-view group; nmv = sv_start[0].
    // (no_group: a low-latency shard keeps every landmark in its LDS tile, whose blocks the back-substitution reuses)
    const bool grouped = !no_group && maxc >= 1 && maxc + nkf <= BB_MAXKEYS;
    const int nkeys = grouped ? maxc + nkf : maxc + 1;
    const int lper = (nlm + BB_THREADS - 1) / BB_THREADS;
    {
        const int l0 = min(tid * lper, nlm), l1 = min(l0 + lper, nlm);
        auto key = [&](int l) -> int {
            const int c = cnt[l];
            if (!grouped) return maxc - c;
            if (c >= 2) return maxc - c;
            if (c == 0) return maxc - 1 + nkf;
            return maxc - 1 + (int)((edge(ostart[l]) >> 16) & 0xffu);
        };
        for (int k2 = 0; k2 < nkeys; ++k2) flat[k2 * BB_THREADS + tid] = 0;
        for (int l = l0; l < l1; ++l) flat[key(l) * BB_THREADS + tid]++;
        __syncthreads();
        bb_exscan(flat, nkeys * BB_THREADS, tmp, tid);
        if (tid <= nkf) {
            const int v = grouped ? flat[(maxc - 1 + tid) * BB_THREADS] : nlm;     // start of the bucket = its first thread's offset
            g_sv_start[tid] = v;
            if (tid == 0) small[4] = v;
        }
        __syncthreads();                      // thread 0's column is about to be advanced
        for (int l = l0; l < l1; ++l) {
            const int pos = flat[key(l) * BB_THREADS + tid]++;
            lm_orig[pos] = l;
        }
    }
    __syncthreads();
    const int nmv = small[4];
    // ---- 4. edge and block ranges in the new numbering
    for (int jn = tid; jn < nlm; jn += BB_THREADS) {
        const int l = lm_orig[jn];
        lm_es[jn] = ostart[l + 1] - ostart[l];
        lm_bs[jn] = cnt[l];
        g_lm_orig[jn] = l;
    }
    if (tid == 0) { lm_es[nlm] = 0; lm_bs[nlm] = 0; }
    __syncthreads();
    bb_exscan(lm_es, nlm + 1, tmp, tid);
    const int nblk = bb_exscan(lm_bs, nlm + 1, tmp, tid);
    for (int jn = tid; jn <= nlm; jn += BB_THREADS) { g_lm_estart[jn] = lm_es[jn]; g_lm_bstart[jn] = lm_bs[jn]; }
    if (tid == 0) g_blk_es[nblk] = nobs;
    // ---- 5. landmark-major records, blocks, pose masks (thread per landmark; lanes of a wave walk
    //         similar edge counts because of the renumbering)
    for (int jn = tid; jn < nlm; jn += BB_THREADS) {
        const int l = lm_orig[jn];
        int i = lm_es[jn], b = lm_bs[jn] - 1, prev_kf = -1;
        unsigned int mask = 0;
        for (int q = ostart[l]; q < ostart[l + 1]; ++q, ++i) {
            const unsigned int ev = edge(q);
            const int e = ident ? q : srt[q], k = (int)((ev >> 16) & 0xffu);
            if (k != prev_kf) { ++b; g_blk_kf[b] = k; g_blk_lm[b] = jn; g_blk_es[b] = i; prev_kf = k; mask |= 1u << pidx[k]; }
            g_lm_edges[i] = e;
            const float2 uv = ouv[e];
            BaRec r;
            r.u = uv.x; r.v = uv.y;
            r.lmkc = jn | (((pidx[k] << 1) | (int)(ev >> 24)) << 24);
            r.blk = b;
            recL[i] = r;
        }
        pmask[jn] = (int)mask;
    }
    __syncthreads();
    // ---- 6. pose-major copy of the records, landmark-ascending inside a pose: stable partition of the
    //         landmark-major order by keyframe, again as a flattened [keyframe][thread] scan
    {
        // (the keyframe of record i is read back from the record itself — active-pose index -> keyframe through
        // LDS — instead of through two dependent global loads per edge)
        const int eper = (nobs + BB_THREADS - 1) / BB_THREADS;
        const int i0 = min(tid * eper, nobs), i1 = min(i0 + eper, nobs);
        int *actk = small + 8;                    // [BB_MAXKEYS] active pose -> keyframe
        if (tid < BB_MAXKEYS) actk[tid] = tid < na ? g_act_kf[tid] : 0;
        for (int k = 0; k < nkf; ++k) flat[k * BB_THREADS + tid] = 0;
        __syncthreads();
        for (int i = i0; i < i1; ++i) flat[actk[(unsigned)recL[i].lmkc >> 25] * BB_THREADS + tid]++;
        __syncthreads();
        bb_exscan(flat, nkf * BB_THREADS, tmp, tid);
        for (int i = i0; i < i1; ++i) {
            const BaRec r = recL[i];
            recP[flat[actk[(unsigned)r.lmkc >> 25] * BB_THREADS + tid]++] = r;
        }
    }
    // ---- 7. LDS tiles: greedy packing of consecutive landmarks under two capacities (landmarks and blocks).
    //         Both running counts are monotone in the landmark index, so the end of a tile that starts at s is
    //         min(s + cap, first e with lm_bs[e] - lm_bs[s] > cap): a binary search per tile instead of a
    //         walk over every landmark (the walk was a third of this kernel's time)
    const int lay_ntile = jd.lay_ntile;
    if (tid == 0) {
        int nt = 0, st = 0;
        g_tile_lm[0] = 0;
        while (st < nmv) {
            const int lim = lm_bs[st] + tile_cap;
            int lo = st + 1, hi = min(st + tile_cap, nmv);        // the tile ends in (st, hi]; lm_bs[lo] - lm_bs[st] <= cap holds for lo = st + 1
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (lm_bs[mid] <= lim) lo = mid; else hi = mid - 1;
            }
            st = lo;
            ++nt;
            if (nt <= lay_ntile) g_tile_lm[nt] = st;
        }
        if (nt > lay_ntile) atomicExch(err_flag, 1);
        small[2] = nt;
    }
    __syncthreads();
    const int ntile = small[2];
    if (ntile > lay_ntile) return;
    // ---- 8. block pairs of every (tile, pose pair): one thread per list walks the tile's landmarks;
    //         a landmark that sees poses a and b contributes the pair of its blocks popcount(mask below a / b)
    const int nlist = ntile * npairs;
    for (int q0 = 0; q0 < nlist; q0 += BB_MAXKEYS * BB_THREADS) {
        // counts (flat doubles as the count / offset table of up to BB_MAXKEYS * BB_THREADS lists at a time)
        const int nq = min(nlist - q0, BB_MAXKEYS * BB_THREADS);
        for (int q = tid; q < nq; q += BB_THREADS) {
            const int t = (q0 + q) / npairs, pr = (q0 + q) - t * npairs;
            int a = 0, rem = pr;
            while (rem >= na - a) { rem -= na - a; ++a; }
            const unsigned int need = (1u << a) | (1u << (a + rem));
            int c = 0;
            const int lt0 = g_tile_lm[t], lt1 = g_tile_lm[t + 1];
            for (int l = lt0; l < lt1; ++l) c += ((unsigned int)pmask[l] & need) == need;
            flat[q] = c;
        }
        __syncthreads();
        const int base = small[3];
        const int tot = bb_exscan(flat, nq, tmp, tid);
        for (int q = tid; q < nq; q += BB_THREADS) g_pcs[q0 + q] = base + flat[q];
        if ((size_t)(base + tot) > ba_pitem_bound(nobs, nkf)) { if (tid == 0) atomicExch(err_flag, 2); return; }
        for (int q = tid; q < nq; q += BB_THREADS) {
            const int t = (q0 + q) / npairs, pr = (q0 + q) - t * npairs;
            int a = 0, rem = pr;
            while (rem >= na - a) { rem -= na - a; ++a; }
            const int b2 = a + rem;
            const unsigned int need = (1u << a) | (1u << b2);
            const int l0 = g_tile_lm[t], l1 = g_tile_lm[t + 1], bt0 = lm_bs[l0];
            int w = base + flat[q];
            for (int l = l0; l < l1; ++l) {
                const unsigned int m = (unsigned int)pmask[l];
                if ((m & need) != need) continue;
                const int u = lm_bs[l] - bt0 + __popc(m & ((1u << a) - 1u));
                const int v = lm_bs[l] - bt0 + __popc(m & ((1u << b2) - 1u));
                g_pitem[w++] = u | (v << 10) | ((l - l0) << 20);
            }
        }
        __syncthreads();
        if (tid == 0) small[3] = base + tot;
        __syncthreads();
    }
    if (tid == 0) {
        g_pcs[nlist] = small[3];
        jd.nblk = nblk; jd.na = na; jd.ncontrib = small[3]; jd.ntile = ntile; jd.nmv = nmv;
    }
}

// ---------------------------------------------------------------- low-latency BA: one problem -> LLW shards
// Deals the landmarks of a problem (edges landmark-major, the order the backend gathers them in) to `llw` shards of
// CONTIGUOUS landmark ranges with about equal cost (a landmark with e edges costs e (4 + e): linearisation + block pairs),
// so a shard is a sub-range of the parent's landmark, position and edge arrays and k_ba_build / k_local_ba_t<2> work on it
// in place: results land where the parent's caller expects them.  Every shard gets all keyframes (k_ba_build all_active).
// One workgroup per problem.  Shards without edges are masked out (BaDev::shmask) and their exchange slots zeroed.
#define SP_THREADS 1024     // (the pass over the edges is a chain of dependent global loads per thread: more threads, fewer rounds)
__global__ void __launch_bounds__(SP_THREADS)
k_ba_split(const BaDev *parents, BaDev *shards, const unsigned int *obs_packed, int llw, int tile_cap, int max_nlm,
           double *xch, size_t xch_stride, unsigned int *cnt, int res_blocks, int res_landmarks, int res_edges)
{
    extern __shared__ __attribute__((aligned(16))) int sp_lds[];
    const BaDev &pd = parents[blockIdx.x];
    const int tid = threadIdx.x;
    const int nkf = pd.nkf, nlm = pd.nlm, nobs = pd.nobs, np = 6 * nkf;
    BaDev *sh = shards + (size_t)blockIdx.x * llw;
    int *estart = sp_lds;                     // [nlm + 1] first edge of landmark l
    long long *cost = reinterpret_cast<long long *>(sp_lds + ((max_nlm + 2 + 1) & ~1));   // [nlm + 1] exclusive prefix of the cost
    long long *tmpl = cost + max_nlm + 2;     // [SP_THREADS]
    int *cut = reinterpret_cast<int *>(tmpl + SP_THREADS);    // [llw + 1] first landmark of shard w
    if (tid < LL_CNT_WORDS) cnt[(size_t)LL_CNT_WORDS * blockIdx.x + tid] = 0u;    // arrival counter + abort word of this call
    const bool empty = nobs <= 0 || nlm <= 0 || nkf <= 0 || pd.reserved == 0;        // (unsorted edges: the host does not come here)
    if (empty) {
        for (int w = tid; w < llw; w += SP_THREADS) {
            BaDev d = pd;
            d.nlm = 0; d.nobs = 0; d.nblk = d.na = d.ncontrib = d.ntile = d.nmv = 0; d.shmask = 0; d.lm_base = 0; d.iters_done = 0; d.ntrial = 0;
            sh[w] = d;
        }
        return;
    }
    const unsigned int *opk = obs_packed + pd.obs_ofs;
    for (int i = tid; i < nobs; i += SP_THREADS) {
        const int l = (int)(opk[i] & 0xffffu);
        const int lp = i > 0 ? (int)(opk[i - 1] & 0xffffu) : -1;
        if (l != lp) for (int q = lp + 1; q <= l; ++q) estart[q] = i;
        if (i == nobs - 1) for (int q = l + 1; q <= nlm; ++q) estart[q] = nobs;
    }
    __syncthreads();
    // exclusive prefix of the landmark costs
    const int per = (nlm + SP_THREADS - 1) / SP_THREADS;
    const int l0 = min(tid * per, nlm), l1 = min(l0 + per, nlm);
    long long sum = 0;
    for (int l = l0; l < l1; ++l) { const long long e = estart[l + 1] - estart[l]; sum += e * (4 + e); }
    tmpl[tid] = sum;
    __syncthreads();
    for (int d = 1; d < SP_THREADS; d <<= 1) {
        const long long add = tid >= d ? tmpl[tid - d] : 0;
        __syncthreads();
        tmpl[tid] += add;
        __syncthreads();
    }
    const long long total = tmpl[SP_THREADS - 1];
    {
        long long run = tmpl[tid] - sum;
        for (int l = l0; l < l1; ++l) { const long long e = estart[l + 1] - estart[l]; cost[l] = run; run += e * (4 + e); }
        if (tid == 0) cost[nlm] = total;
    }
    if (tid <= llw) cut[tid] = tid == llw ? nlm : -1;
    __syncthreads();
    // shard of landmark l = floor(cost_before(l) * llw / total); cut[w] = first landmark of shard w (monotone)
    auto shard_of = [&](int l) -> int { const int w = (int)((cost[l] * llw) / (total > 0 ? total : 1)); return w < llw ? w : llw - 1; };
    for (int l = tid; l < nlm; l += SP_THREADS) {
        const int w = shard_of(l), wp = l > 0 ? shard_of(l - 1) : -1;
        for (int q = wp + 1; q <= w; ++q) cut[q] = l;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = llw - 1; w >= 0; --w) if (cut[w] < 0) cut[w] = cut[w + 1];      // shards nothing fell into: empty ranges
        unsigned mask = 0;
        for (int w = 0; w < llw; ++w) if (estart[cut[w + 1]] > estart[cut[w]]) mask |= 1u << w;
        size_t aux = (size_t)pd.aux_ofs;
        for (int w = 0; w < llw; ++w) {
            const int a = cut[w], b = cut[w + 1], e0 = estart[a], e1 = estart[b];
            BaDev d = pd;
            d.lm_ofs = pd.lm_ofs + a; d.nlm = b - a; d.obs_ofs = pd.obs_ofs + e0; d.nobs = e1 - e0; d.lm_base = a; d.shmask = (int)mask;
            d.nblk = d.na = d.ncontrib = d.ntile = d.nmv = 0; d.iters_done = 0; d.reserved = 1; d.ntrial = 0;
            d.rec_ofs = pd.rec_ofs + 2 * e0;
            d.lay_nblk = d.nobs; d.lay_na = nkf; d.lay_ntile = ba_tile_bound(d.nlm, d.nobs, nkf, tile_cap);
            d.aux_ofs = (int)aux;
            aux += ba_aux_layout(nkf, d.nlm, d.nobs, d.lay_nblk, d.lay_na, 0, d.lay_ntile).total + ba_pitem_bound(d.nobs, nkf);
            sh[w] = d;
        }
        tmpl[0] = (long long)mask;
        for (int w = 0; w <= llw; ++w) cut[w] = estart[cut[w]];      // from here: first EDGE of shard w
        for (int w = 0; w < llw; ++w) tmpl[1 + w] = 0;               // blocks per shard
    }
    __syncthreads();
    // Does every shard fit the resident layout of k_ba_ll (blocks / landmarks / edges in LDS, ONE tile)?  Blocks are counted
    // exactly: an edge opens a block when its (landmark, keyframe) differs from its predecessor's.
    {
        int *nb = reinterpret_cast<int *>(tmpl + 1);
        for (int i = tid; i < nobs; i += SP_THREADS) {
            const unsigned int ev = opk[i] & 0x00ffffffu, pv = i > 0 ? (opk[i - 1] & 0x00ffffffu) : 0xffffffffu;
            if (ev == pv) continue;
            int w = 0;
            while (w + 1 < llw && i >= cut[w + 1]) ++w;
            atomicAdd(&nb[w], 1);
        }
        __syncthreads();
        if (tid == 0) {
            bool fits = res_blocks > 0;
            const int bcap = min(res_blocks, tile_cap), lcap = min(res_landmarks, tile_cap);
            for (int w = 0; w < llw; ++w) fits = fits && nb[w] <= bcap && sh[w].nlm <= lcap && sh[w].nobs <= res_edges;
            for (int w = 0; w < llw; ++w) sh[w].reserved = fits ? 2 : 1;
        }
    }
    // shards without edges never publish: their slots of the exchange area read as zero
    const unsigned mask = (unsigned)tmpl[0];
    double *xs = xch + (size_t)blockIdx.x * xch_stride;
    for (int w = 0; w < llw; ++w) {
        if ((mask >> w) & 1u) continue;
        double *slab = xs + (size_t)w * LL_SLAB(np), *x0 = xs + (size_t)llw * LL_SLAB(np) + (size_t)w * LL_X0(np);
        for (size_t i = tid; i < LL_SLAB(np); i += SP_THREADS) slab[i] = 0.0;
        for (size_t i = tid; i < LL_X0(np); i += SP_THREADS) x0[i] = 0.0;
    }
    // the granules of the rho exchange are tagged with epochs that count from 1 in every launch: all of them start at 0
    {
        double *xb = xs + (size_t)llw * (LL_SLAB(np) + LL_X0(np));
        for (int i = tid; i < llw * LL_XB; i += SP_THREADS) xb[i] = 0.0;
    }
}
// extra aux ints a parent's reservation needs so that its llw shards fit (per-shard constants of ba_aux_layout, two tiles of
// slack per shard in ba_tile_bound, one pair-item of slack)
__host__ __device__ inline size_t ba_split_aux_extra(int nkf, int llw)
{
    return (size_t)llw * (16 + 4 * (size_t)nkf + 3 * ((size_t)nkf * (nkf + 1) / 2 + 1) + 4);
}
static inline size_t ba_split_lds_bytes(int max_nlm, int llw)
{
    return sizeof(int) * (size_t)((max_nlm + 2 + 1) & ~1) + sizeof(long long) * ((size_t)max_nlm + 2 + SP_THREADS) + sizeof(int) * (size_t)(llw + 2) + 64;      // (SP_THREADS >= 1 + llw ints for the per-shard block counts)
}
