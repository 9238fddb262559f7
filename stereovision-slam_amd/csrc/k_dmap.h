// k_dmap.h — the map of every stream resident in HBM: keyframe window, the features of its
// keyframes, the landmarks and their observation counts live in per-stream arenas on the device, and the keyframe
// path of Frontend::InsertKeyframe / StereoInit + Backend::Optimize runs as one chain of launches with no
// per-feature, per-landmark or per-edge work on the host:
//
//   k_dmap_begin          Map::InsertKeyFrame / RemoveOldKeyframe / CleanMap + SetObservationsForKeyFrame
//                         (src/map.cpp:53-181, src/frontend.cpp:560-574); unreachable landmarks are freed
//   k_gftt_*              DetectFeatures (:36-70) — mask squares = the keyframe's features, read in place
//   k_dmap_stereo_prep    corners appended (:52-66); start guesses of FindFeaturesInRight (:79-103)
//   k_lk                  left -> right (:105-109)
//   k_dmap_stereo_finish  right features (:111-135); StereoInit's count (:227); pairs to triangulate (:265-275 / :156-163)
//   k_triangulate         slam::triangulation
//   k_dmap_commit         new landmarks + their two observations (:289-305 / :176-191), the keyframe of StereoInit
//   k_dmap_ba_gather      Backend::Optimize's problem (src/backend.cpp:39-160) in the layout k_ba_build reads
//   k_ba_build, k_local_ba_t<0>
//   k_dmap_ba_scatter     outlier threshold, outlier observations removed, poses / positions written back (:167-246)
//   k_dmap_refresh        the keyframe's features become the resident list the next frame tracks from
//
// What stays on the host per keyframe is O(window): the choice of the keyframe to retire (se3 log of <= 10 poses —
// kept on the host so that the decision is bit-identical to the host-map pipeline whatever the device's libm
// rounds like) and the ids / slots of the window.  Observations are implicit: feature (slot, i) of an active
// keyframe observes landmark f_lm / f_lmr, a landmark counts them in lm_obs — RemoveObservation is a decrement.
// The order of a landmark's observations (chronological = keyframe id, left before right) is all the reference's
// list carries, and the BA gather reproduces it by sorting, so the problems are identical to the host gather's.
#pragma once
#include "dev_common.h"
#include "k_ba.h"
#include "k_ba_build.h"
#include "k_geom.h"
#include "k_gftt.h"
#include "k_lk.h"

#pragma clang fp contract(off)

#define DM_THREADS 256
#define DM_FL_RIGHT_OK 1

struct DMap {                 // per-stream arenas, SoA over streams
    int KW, NF, NL;           // keyframe slots (num_active + 1), features per keyframe, landmark slots
    long long *kf_frame;      // [S][KW] frame id, -1 = empty slot
    int *kf_id;               // [S][KW]
    double *kf_pose;          // [S][KW][7]
    int *kf_n;                // [S][KW]
    float2 *f_xy, *f_xyr;     // [S][KW][NF] left / right pixel
    int *f_lm, *f_lmr;        // [S][KW][NF] landmark slot observed by the left / right feature, -1 = none
    uint8_t *f_fl;            // [S][KW][NF] DM_FL_*
    double *lm_pos;           // [S][NL][3]
    int *lm_id;               // [S][NL] id (creation order), -1 = free slot
    int *lm_obs;              // [S][NL] observations (MapPoint::observed_times_)
    uint8_t *lm_st;           // [S][NL] 0 free, 1 active (Map::active_landmarks_), 2 outside the window
    int *lm_stamp;            // [S][NL] last keyframe step whose tracked list named the landmark
    int *next_lm_id;          // [S]
};

struct DmJob {                // device copy of svslam_dmap_job + placement (host-filled unless noted)
    int stream, slot_cur, slot_right, is_init;
    int kf_slot, remove_slot, kf_id, npts;        // npts: features the frame already has (tracked survivors; 0 at init)
    long long frame_id;
    double pose[7];           // T_cw of the frame; out: after local BA
    double T_camr_w[7];       // cam_right.pose * T_cw
    double T_wc[7];           // inverse (identity at init)
    int src_buf, dst_buf;     // resident list: read (survivors) / write (the keyframe's features)
    int stamp, pad0;
    // outputs (device-written)
    int ok, n_features, n_corners, n_right_ok, n_tri_in, n_tri_ok, ba_nkf, ba_nlm, ba_nobs, ba_iters, flags, dead;
    int ba_npair, ba_ntrial;  // block pairs of the Schur complement, LM trials: the flop accounting of bench.py
    int ev_ofs, ev_n;         // the landmarks this call freed: records [ev_ofs, ev_ofs + ev_n) of the batch's evicted list
    double win_pose[12][7];   // poses of the BA problem's keyframes after the solve, by local index
    int win_slot[12];
};
#define DM_FLAG_CORNERS_DROPPED 1
#define DM_FLAG_LM_FULL 2
#define DM_FLAG_BA_SKIPPED 4

struct DmEvicted { int id; float pos[3]; };      // == svslam_dmap_evicted_rec: a landmark leaving the device map (id, last position)

struct DmParams {
    int num_features, num_features_init, num_active;
    double zmax;
    double chi2_th;
    double cam_l[4], cam_r[4];
    int max_obs, max_lm;
    int w, h;
};

__device__ __forceinline__ size_t dm_kf(const DMap &m, int s, int k) { return (size_t)s * m.KW + k; }
__device__ __forceinline__ size_t dm_f(const DMap &m, int s, int k) { return ((size_t)s * m.KW + k) * m.NF; }
__device__ __forceinline__ size_t dm_l(const DMap &m, int s) { return (size_t)s * m.NL; }

// block-wide exclusive scan of one int per thread (DM_THREADS threads); returns the exclusive prefix, total in *tot
__device__ __forceinline__ int dm_exscan(int v, int *tmp, int tid, int *tot)
{
    __syncthreads();
    tmp[tid] = v;
    __syncthreads();
    for (int d = 1; d < DM_THREADS; d <<= 1) {
        const int add = tid >= d ? tmp[tid - d] : 0;
        __syncthreads();
        tmp[tid] += add;
        __syncthreads();
    }
    *tot = tmp[DM_THREADS - 1];
    return tmp[tid] - v;
}

// ---------------------------------------------------------------- keyframe insertion (tracked frames)
__global__ void __launch_bounds__(DM_THREADS)
k_dmap_begin(DmJob *jobs, DMap m, RtStore rs, DmEvicted *ev, int *ev_cursor, int ev_quota)
{
    __shared__ int tmp[DM_THREADS];
    __shared__ int s_base, s_room;
    DmJob &jb = jobs[blockIdx.x];
    const int tid = threadIdx.x, s = jb.stream;
    if (tid == 0) {
        jb.ok = jb.is_init ? 0 : 1; jb.dead = 0; jb.flags = 0; jb.n_corners = jb.n_right_ok = jb.n_tri_in = jb.n_tri_ok = 0;
        jb.ba_nkf = jb.ba_nlm = jb.ba_nobs = jb.ba_iters = 0; jb.n_features = jb.npts; jb.pad0 = 0; jb.ev_ofs = jb.ev_n = 0;
        jb.ba_npair = jb.ba_ntrial = 0;
    }
    if (jb.is_init) return;
    const size_t L = dm_l(m, s);
    // Map::RemoveOldKeyframe (the host chose which): its observations go, then CleanMap
    if (jb.remove_slot >= 0) {
        const size_t F = dm_f(m, s, jb.remove_slot);
        const int n = m.kf_n[dm_kf(m, s, jb.remove_slot)];
        for (int i = tid; i < n; i += DM_THREADS) {
            const int a = m.f_lm[F + i], b = m.f_lmr[F + i];
            if (a >= 0) atomicSub(&m.lm_obs[L + a], 1);
            if (b >= 0) atomicSub(&m.lm_obs[L + b], 1);
        }
        if (tid == 0) m.kf_frame[dm_kf(m, s, jb.remove_slot)] = -1;
        __syncthreads();
        __threadfence_block();
        for (int l = tid; l < m.NL; l += DM_THREADS)
            if (m.lm_st[L + l] == 1 && m.lm_obs[L + l] == 0) m.lm_st[L + l] = 2;
    }
    __syncthreads();
    // the new keyframe: header, the tracked survivors become its features, SetObservationsForKeyFrame
    const size_t K = dm_kf(m, s, jb.kf_slot), F = dm_f(m, s, jb.kf_slot);
    if (tid == 0) { m.kf_frame[K] = jb.frame_id; m.kf_id[K] = jb.kf_id; m.kf_n[K] = jb.npts; }
    if (tid < 7) m.kf_pose[K * 7 + tid] = jb.pose[tid];
    const size_t R = (size_t)s * rs.max_pts;
    for (int i = tid; i < jb.npts; i += DM_THREADS) {
        const int mp = rs.mp[jb.src_buf][R + i];
        m.f_xy[F + i] = rs.xy[jb.src_buf][R + i];
        m.f_lm[F + i] = mp; m.f_lmr[F + i] = -1; m.f_fl[F + i] = 0;
        if (mp >= 0) { m.lm_obs[L + mp] += 1; m.lm_stamp[L + mp] = jb.stamp; }      // a landmark is named by one feature of a frame at most
    }
    __syncthreads();
    __threadfence_block();
    // landmarks nothing can reach any more (no observation, outside the window, not carried by tracking) free their slot.
    // The reference keeps every landmark for saveSLAMOutputInFile (src/visual_odometry.cpp:226-304): id and last position
    // of each go to the batch's evicted list first.  A job hands over at most ev_quota landmarks per call (the list holds
    // quota x jobs records, so the reservation below never overflows and WHICH landmarks a job frees depends on nothing but its
    // own map: the first ev_quota candidates in (thread, slot) order); the rest stay where they are until the stream's next
    // keyframe — nothing is lost, the slots just stay taken that long.
    int mine = 0;
    for (int l = tid; l < m.NL; l += DM_THREADS)
        mine += (m.lm_st[L + l] == 2 && m.lm_obs[L + l] == 0 && m.lm_stamp[L + l] != jb.stamp) ? 1 : 0;
    int tot;
    const int pos = dm_exscan(mine, tmp, tid, &tot);
    if (tid == 0) {
        const int room = min(tot, ev_quota);
        s_base = room > 0 ? atomicAdd(ev_cursor, room) : 0; s_room = room;
        jb.ev_ofs = s_base; jb.ev_n = room;
    }
    __syncthreads();
    if (s_room == 0) return;
    int k = pos;
    for (int l = tid; l < m.NL; l += DM_THREADS)
        if (m.lm_st[L + l] == 2 && m.lm_obs[L + l] == 0 && m.lm_stamp[L + l] != jb.stamp) {
            if (k >= s_room) break;                        // over the quota: this one waits for the next keyframe
            DmEvicted &e = ev[s_base + k];
            ++k;
            e.id = m.lm_id[L + l];
            e.pos[0] = (float)m.lm_pos[3 * (L + l)]; e.pos[1] = (float)m.lm_pos[3 * (L + l) + 1]; e.pos[2] = (float)m.lm_pos[3 * (L + l) + 2];
            m.lm_st[L + l] = 0; m.lm_id[L + l] = -1;
        }
}

// ---------------------------------------------------------------- corners -> features, stereo LK inputs
// LK points of job j live at [j * NF, (j + 1) * NF) of the call's point arrays
__global__ void __launch_bounds__(DM_THREADS)
k_dmap_stereo_prep(DmJob *jobs, DMap m, DmParams prm, const float2 *corners, const int *ncorners, int max_corners,
                   LkJob *lkjobs, float2 *prev_xy, float2 *next_xy)
{
    DmJob &jb = jobs[blockIdx.x];
    const int tid = threadIdx.x, s = jb.stream, j = blockIdx.x;
    const size_t F = dm_f(m, s, jb.kf_slot), L = dm_l(m, s);
    int take = ncorners[j];
    const int room = m.NF - jb.npts;
    if (take > room) {                                   // capacity of the keyframe's feature list: surplus corners (the weakest) are dropped
        const int kept = room > 0 ? room : 0;
        if (tid == 0) { jb.flags |= DM_FLAG_CORNERS_DROPPED; jb.pad0 = take - kept; }
        take = kept;
    }
    for (int c = tid; c < take; c += DM_THREADS) {
        m.f_xy[F + jb.npts + c] = corners[(size_t)j * max_corners + c];
        m.f_lm[F + jb.npts + c] = -1; m.f_lmr[F + jb.npts + c] = -1; m.f_fl[F + jb.npts + c] = 0;
    }
    const int n = jb.npts + take;
    __syncthreads();
    if (tid == 0) {
        jb.n_corners = take; jb.n_features = n;
        m.kf_n[dm_kf(m, s, jb.kf_slot)] = n;
        lkjobs[j].prev_slot = jb.slot_cur; lkjobs[j].next_slot = jb.slot_right; lkjobs[j].pt_ofs = j * m.NF; lkjobs[j].npts = n;
    }
    for (int p = tid; p < n; p += DM_THREADS) {
        const float2 xy = m.f_xy[F + p];
        const int mp = m.f_lm[F + p];
        float2 g = xy;
        if (mp >= 0) {
            const double X[3] = { m.lm_pos[(L + mp) * 3], m.lm_pos[(L + mp) * 3 + 1], m.lm_pos[(L + mp) * 3 + 2] };
            double uv[2];
            d_project_exact(jb.T_camr_w, prm.cam_r, X, uv);
            g = make_float2((float)uv[0], (float)uv[1]);
        }
        prev_xy[(size_t)j * m.NF + p] = xy;
        next_xy[(size_t)j * m.NF + p] = g;
    }
}

// ---------------------------------------------------------------- right features, StereoInit's test, triangulation list
__global__ void __launch_bounds__(DM_THREADS)
k_dmap_stereo_finish(DmJob *jobs, DMap m, DmParams prm, const float2 *next_xy, const uint8_t *status,
                     TriJob *trijobs, float2 *uv_l, float2 *uv_r, int *tri_idx)
{
    __shared__ int tmp[DM_THREADS];
    DmJob &jb = jobs[blockIdx.x];
    const int tid = threadIdx.x, s = jb.stream, j = blockIdx.x;
    const size_t F = dm_f(m, s, jb.kf_slot);
    const int n = jb.n_features;
    int good = 0;
    for (int p = tid; p < n; p += DM_THREADS) {
        const float2 q = next_xy[(size_t)j * m.NF + p];
        const bool ok = status[(size_t)j * m.NF + p] && q.y >= 0.f && q.y < (float)prm.h && q.x >= 0.f && q.x < (float)prm.w;
        m.f_fl[F + p] = ok ? DM_FL_RIGHT_OK : 0;
        m.f_xyr[F + p] = ok ? q : make_float2(0.f, 0.f);
        good += ok ? 1 : 0;
    }
    int tot;
    (void)dm_exscan(good, tmp, tid, &tot);
    const bool dead = jb.is_init && tot < prm.num_features_init;          // src/frontend.cpp:227
    // pairs to triangulate, in feature order: right found and (init or no map point yet)
    int base = 0, ntri = 0;
    for (int p0 = 0; p0 < n && !dead; p0 += DM_THREADS) {
        const int p = p0 + tid;
        const bool sel = p < n && (m.f_fl[F + p] & DM_FL_RIGHT_OK) && (jb.is_init || m.f_lm[F + p] < 0);
        int cnt;
        const int pos = dm_exscan(sel ? 1 : 0, tmp, tid, &cnt);
        if (sel) {
            const size_t o = (size_t)j * m.NF + base + pos;
            uv_l[o] = m.f_xy[F + p]; uv_r[o] = m.f_xyr[F + p]; tri_idx[o] = p;
        }
        base += cnt;
    }
    ntri = base;
    if (tid == 0) {
        jb.n_right_ok = tot; jb.dead = dead ? 1 : 0; jb.n_tri_in = ntri;
        if (jb.is_init) jb.ok = dead ? 0 : 1;
        TriJob &t = trijobs[j];
        t.pt_ofs = j * m.NF; t.npts = dead ? 0 : ntri; t.zmax = jb.is_init ? 0.0 : prm.zmax;
        for (int i = 0; i < 7; ++i) t.T_wc[i] = jb.T_wc[i];
    }
}

// ---------------------------------------------------------------- new landmarks (and StereoInit's keyframe)
__global__ void __launch_bounds__(DM_THREADS)
k_dmap_commit(DmJob *jobs, DMap m, const double *tri_xyz, const uint8_t *tri_ok, const int *tri_idx, int *slot_tmp)
{
    __shared__ int tmp[DM_THREADS];
    DmJob &jb = jobs[blockIdx.x];
    if (jb.dead) return;
    const int tid = threadIdx.x, s = jb.stream, j = blockIdx.x;
    const size_t F = dm_f(m, s, jb.kf_slot), L = dm_l(m, s), K = dm_kf(m, s, jb.kf_slot);
    const int ntri = jb.n_tri_in;
    int *slots = slot_tmp + (size_t)j * m.NF;          // rank among the new landmarks -> landmark slot
    // rank of every accepted point (feature order = the order the reference creates them in)
    // 1. the free slots, ascending, are dealt to the ranks
    int nfree_before = 0;
    for (int l0 = 0; l0 < m.NL; l0 += DM_THREADS) {
        const int l = l0 + tid;
        const bool fr = l < m.NL && m.lm_st[L + l] == 0;
        int cnt;
        const int pos = dm_exscan(fr ? 1 : 0, tmp, tid, &cnt);
        if (fr && nfree_before + pos < m.NF) slots[nfree_before + pos] = l;
        nfree_before += cnt;
        if (nfree_before >= m.NF) break;                 // uniform
    }
    __syncthreads();
    const int nfree = nfree_before;
    const int id0 = m.next_lm_id[s];
    int base = 0;
    for (int q0 = 0; q0 < ntri; q0 += DM_THREADS) {
        const int q = q0 + tid;
        const size_t o = (size_t)j * m.NF + q;
        const bool ok = q < ntri && tri_ok[o];
        int cnt;
        const int pos = dm_exscan(ok ? 1 : 0, tmp, tid, &cnt);
        if (ok) {
            const int r = base + pos;
            if (r < nfree) {
                const int l = slots[r], p = tri_idx[o];
                m.lm_pos[(L + l) * 3] = tri_xyz[o * 3]; m.lm_pos[(L + l) * 3 + 1] = tri_xyz[o * 3 + 1]; m.lm_pos[(L + l) * 3 + 2] = tri_xyz[o * 3 + 2];
                m.lm_id[L + l] = id0 + r; m.lm_obs[L + l] = 2; m.lm_st[L + l] = 1; m.lm_stamp[L + l] = jb.stamp;
                m.f_lm[F + p] = l; m.f_lmr[F + p] = l;
            }
        }
        base += cnt;
    }
    if (tid == 0) {
        const int made = base < nfree ? base : nfree;
        if (base > nfree) jb.flags |= DM_FLAG_LM_FULL;
        jb.n_tri_ok = made;
        m.next_lm_id[s] = id0 + made;
        if (jb.is_init) { m.kf_frame[K] = jb.frame_id; m.kf_id[K] = jb.kf_id; }      // StereoInit's keyframe (:232-246)
    }
    if (jb.is_init && tid < 7) m.kf_pose[K * 7 + tid] = jb.pose[tid];
}

// ---------------------------------------------------------------- Backend::Optimize: the problem, in k_ba_build's input layout
// Per job the BA arrays have fixed strides (max_kf poses, NL points, max_obs edges).  LDS: sort keys [NL] u64,
// local index by slot [NL] i32, edge counts / starts [NL + 1] i32.
// DMG_T threads per problem: 1024 for a few problems per call (the latency of the sort and the scans is the call's latency),
// 512 for a batch (more problems per CU at once: the 1024-thread form took 836 us per ~300 problems beside the other groups'
// kernels where this one takes 448)
template <int DMG_T> static inline size_t dmg_lds_bytes_t(int NL) { return (size_t)NL * 8 + (size_t)NL * 4 + ((size_t)NL + 2) * 4 + DMG_T * 4 + 256; }
static inline size_t dmg_lds_bytes(int NL) { return dmg_lds_bytes_t<1024>(NL); }      // the larger of the two

template <int DMG_THREADS>
__global__ void __launch_bounds__(DMG_THREADS)
k_dmap_ba_gather(DmJob *jobs, DMap m, DmParams prm, BaDev *badev, double *poses, double *pts, unsigned int *packed, float2 *uv,
                 int *edge_ref, int *lm_slot_of, int max_kf, int tile_cap, size_t aux_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dmg_lds[];
    DmJob &jb = jobs[blockIdx.x];
    const int tid = threadIdx.x, s = jb.stream, j = blockIdx.x;
    BaDev &bd = badev[j];
    const int NL = m.NL;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(dmg_lds);
    int *local_of = reinterpret_cast<int *>(keys + NL);
    int *estart = local_of + NL;                      // [NL + 2]
    int *tmp = estart + NL + 2;                       // [DMG_THREADS]
    int *small = tmp + DMG_THREADS;                   // kf order etc.
    const size_t L = dm_l(m, s);
    if (tid == 0) {
        bd.kf_ofs = j * max_kf; bd.lm_ofs = j * NL; bd.obs_ofs = j * prm.max_obs; bd.rec_ofs = 2 * j * prm.max_obs;
        bd.nkf = bd.nlm = bd.nobs = 0; bd.nblk = bd.na = bd.ncontrib = bd.ntile = 0; bd.iters_done = 0; bd.nmv = 0; bd.reserved = 1;
        bd.aux_ofs = (int)(aux_stride * j); bd.lay_nblk = bd.lay_na = bd.lay_ntile = 0; bd.lm_base = 0; bd.shmask = 0; bd.ntrial = 0;
    }
    if (jb.dead) return;
    // active keyframes in id order (Map::active_keyframes_ is id-ordered): <= KW of them
    if (tid == 0) {
        int n = 0;
        for (int k = 0; k < m.KW; ++k) if (m.kf_frame[dm_kf(m, s, k)] >= 0) small[n++] = k;
        for (int a = 1; a < n; ++a) {                  // insertion sort by keyframe id
            const int k = small[a]; const int id = m.kf_id[dm_kf(m, s, k)];
            int b = a - 1;
            while (b >= 0 && m.kf_id[dm_kf(m, s, small[b])] > id) { small[b + 1] = small[b]; --b; }
            small[b + 1] = k;
        }
        small[16] = n;
    }
    // vertices: active landmarks with an observation, in id order (Map::active_landmarks_ is id-ordered; a landmark
    // becomes a vertex with its first observation, src/backend.cpp:118-130)
    // The vertices' keys are compacted first (thread t owns slots [t per, (t + 1) per): stable), so that the bitonic sort
    // runs over the next power of two above their number — ~1 700 of 4 096 slots hold a vertex in a full window — instead of
    // over all NL slots: 99 -> ~30 us for a lone camera's keyframe (round 4).  The compacted copy borrows the local_of /
    // estart area, which is initialised afterwards.
    unsigned long long *ck = reinterpret_cast<unsigned long long *>(local_of);      // [NL] (local_of + estart: 8 NL + 8 bytes)
    int nvalid = 0, n2 = 2;
    {
        const int per = (NL + DMG_THREADS - 1) / DMG_THREADS;
        const int l0 = min(tid * per, NL), l1 = min(l0 + per, NL);
        int cnt = 0;
        for (int l = l0; l < l1; ++l) {
            const bool v = m.lm_st[L + l] == 1 && m.lm_obs[L + l] > 0;
            keys[l] = v ? (((unsigned long long)(unsigned)m.lm_id[L + l] << 32) | (unsigned)l) : ~0ull;
            cnt += v ? 1 : 0;
        }
        tmp[tid] = cnt;
        __syncthreads();
        for (int d = 1; d < DMG_THREADS; d <<= 1) {
            const int add = tid >= d ? tmp[tid - d] : 0;
            __syncthreads();
            tmp[tid] += add;
            __syncthreads();
        }
        nvalid = tmp[DMG_THREADS - 1];
        int pos = tmp[tid] - cnt;
        for (int l = l0; l < l1; ++l) if (keys[l] != ~0ull) ck[pos++] = keys[l];
        while (n2 < nvalid) n2 <<= 1;
        __syncthreads();
        for (int i = nvalid + tid; i < n2; i += DMG_THREADS) ck[i] = ~0ull;
        __syncthreads();
    }
    for (int k2 = 2; k2 <= n2; k2 <<= 1)               // bitonic sort of the compacted keys
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (int i = tid; i < n2; i += DMG_THREADS) {
                const int ixj = i ^ j2;
                if (ixj > i) {
                    const unsigned long long a = ck[i], b = ck[ixj];
                    const bool up = (i & k2) == 0;
                    if ((a > b) == up) { ck[i] = b; ck[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < NL; i += DMG_THREADS) keys[i] = i < n2 ? ck[i] : ~0ull;     // sorted vertices first, the rest empty
    __syncthreads();
    for (int l = tid; l < NL; l += DMG_THREADS) local_of[l] = -1;
    __syncthreads();
    for (int i = tid; i < nvalid; i += DMG_THREADS) local_of[(int)(keys[i] & 0xffffffffu)] = i;      // (the vertices are keys [0, nvalid))
    __syncthreads();
    const int nkf = small[16], nlm = nvalid;
    // edges per vertex
    for (int i = tid; i <= nlm + 1; i += DMG_THREADS) estart[i] = 0;
    __syncthreads();
    for (int a = 0; a < nkf; ++a) {
        const int k = small[a];
        const size_t F = dm_f(m, s, k);
        const int n = m.kf_n[dm_kf(m, s, k)];
        for (int p = tid; p < n; p += DMG_THREADS) {
            const int la = m.f_lm[F + p], lb = m.f_lmr[F + p];
            const int va = la >= 0 ? local_of[la] : -1, vb = lb >= 0 ? local_of[lb] : -1;
            if (va >= 0) atomicAdd(&estart[va], 1);
            if (vb >= 0) atomicAdd(&estart[vb], 1);
        }
    }
    __syncthreads();
    // exclusive scan of the counts (nlm <= NL): per-thread chunk + scan of the partials
    {
        const int per = (nlm + DMG_THREADS - 1) / DMG_THREADS;
        const int i0 = min(tid * per, nlm), i1 = min(i0 + per, nlm);
        int sum = 0;
        for (int i = i0; i < i1; ++i) sum += estart[i];
        tmp[tid] = sum;
        __syncthreads();
        for (int d = 1; d < DMG_THREADS; d <<= 1) {
            const int add = tid >= d ? tmp[tid - d] : 0;
            __syncthreads();
            tmp[tid] += add;
            __syncthreads();
        }
        int run = tmp[tid] - sum;
        for (int i = i0; i < i1; ++i) { const int v = estart[i]; estart[i] = run; run += v; }
        __syncthreads();
        if (tid == 0) estart[nlm] = tmp[DMG_THREADS - 1];
        __syncthreads();
    }
    const int nobs = estart[nlm];
    if (nobs > prm.max_obs || nkf > max_kf) {          // over the solver's capacity: this keyframe goes without a BA
        if (tid == 0) jb.flags |= DM_FLAG_BA_SKIPPED;
        return;
    }
    // fill, keyframe by keyframe in id order: a landmark's edges come out keyframe-ascending, left before right.
    // `estart` doubles as the running fill position (a landmark is touched by one feature of a keyframe at most).
    const size_t EO = (size_t)j * prm.max_obs;
    for (int a = 0; a < nkf; ++a) {
        const int k = small[a];
        const size_t F = dm_f(m, s, k);
        const int n = m.kf_n[dm_kf(m, s, k)];
        for (int p = tid; p < n; p += DMG_THREADS) {
            const int la = m.f_lm[F + p], lb = m.f_lmr[F + p];
            const int va = la >= 0 ? local_of[la] : -1, vb = lb >= 0 ? local_of[lb] : -1;
            if (va >= 0) {
                const int e = estart[va]++;
                packed[EO + e] = (unsigned)va | ((unsigned)a << 16);
                uv[EO + e] = m.f_xy[F + p];
                edge_ref[EO + e] = (k << 16) | (p << 1);
            }
            if (vb >= 0) {
                const int e = estart[vb]++;
                packed[EO + e] = (unsigned)vb | ((unsigned)a << 16) | (1u << 24);
                uv[EO + e] = m.f_xyr[F + p];
                edge_ref[EO + e] = (k << 16) | (p << 1) | 1;
            }
        }
        __syncthreads();
    }
    for (int a = tid; a < nkf * 7; a += DMG_THREADS) poses[(size_t)j * max_kf * 7 + a] = m.kf_pose[dm_kf(m, s, small[a / 7]) * 7 + a % 7];
    for (int i = tid; i < nlm; i += DMG_THREADS) {
        const int l = (int)(keys[i] & 0xffffffffu);
        lm_slot_of[(size_t)j * NL + i] = l;
        pts[((size_t)j * NL + i) * 3] = m.lm_pos[(L + l) * 3]; pts[((size_t)j * NL + i) * 3 + 1] = m.lm_pos[(L + l) * 3 + 1];
        pts[((size_t)j * NL + i) * 3 + 2] = m.lm_pos[(L + l) * 3 + 2];
    }
    if (tid == 0) {
        bd.nkf = nkf; bd.nlm = nlm; bd.nobs = nobs;
        bd.lay_nblk = nobs; bd.lay_na = nkf; bd.lay_ntile = ba_tile_bound(nlm, nobs, nkf, tile_cap);
        jb.ba_nkf = nkf; jb.ba_nlm = nlm; jb.ba_nobs = nobs;
        for (int a = 0; a < nkf; ++a) jb.win_slot[a] = small[a];
    }
}

// ---------------------------------------------------------------- BA results back into the map (src/backend.cpp:167-246)
__global__ void __launch_bounds__(DM_THREADS)
k_dmap_ba_scatter(DmJob *jobs, DMap m, DmParams prm, const BaDev *badev, const double *poses, const double *pts,
                  const double *chi2, const int *edge_ref, const int *lm_slot_of, int max_kf)
{
    __shared__ int tmp[DM_THREADS];
    DmJob &jb = jobs[blockIdx.x];
    const BaDev &bd = badev[blockIdx.x];
    const int tid = threadIdx.x, s = jb.stream, j = blockIdx.x;
    if (jb.dead || bd.nobs <= 0) return;
    if (bd.iters_done < 0) { if (tid == 0) jb.ba_iters = -1; return; }      // the low-latency solver gave up: nothing to apply (the host fails the call)
    const size_t EO = (size_t)j * prm.max_obs, L = dm_l(m, s);
    const int nobs = bd.nobs;
    // threshold doubling until more than half of the edges are inliers (:167-193)
    double th = prm.chi2_th;
    for (int it = 0; it < 5; ++it) {
        int out = 0;
        for (int e = tid; e < nobs; e += DM_THREADS) out += chi2[EO + e] > th ? 1 : 0;
        int tot;
        (void)dm_exscan(out, tmp, tid, &tot);
        const int cnt_outlier = tot, cnt_inlier = nobs - tot;
        const double inlier_ratio = cnt_inlier / double(cnt_inlier + cnt_outlier);
        if (inlier_ratio > 0.5) break;
        th *= 2;
    }
    for (int e = tid; e < nobs; e += DM_THREADS) {
        if (chi2[EO + e] > th) {                       // feature becomes an outlier: observation removed, map point dropped
            const int r = edge_ref[EO + e], k = r >> 16, p = (r >> 1) & 0x7fff, right = r & 1;
            const size_t F = dm_f(m, s, k);
            const int l = right ? m.f_lmr[F + p] : m.f_lm[F + p];
            if (right) m.f_lmr[F + p] = -1; else m.f_lm[F + p] = -1;
            if (l >= 0) atomicSub(&m.lm_obs[L + l], 1);
        }
    }
    for (int a = tid; a < bd.nkf * 7; a += DM_THREADS) {
        const double v = poses[(size_t)j * max_kf * 7 + a];
        m.kf_pose[dm_kf(m, s, jb.win_slot[a / 7]) * 7 + a % 7] = v;
        jb.win_pose[a / 7][a % 7] = v;
        if (jb.win_slot[a / 7] == jb.kf_slot) jb.pose[a % 7] = v;
    }
    for (int i = tid; i < bd.nlm; i += DM_THREADS) {
        const int l = lm_slot_of[(size_t)j * m.NL + i];
        m.lm_pos[(L + l) * 3] = pts[((size_t)j * m.NL + i) * 3]; m.lm_pos[(L + l) * 3 + 1] = pts[((size_t)j * m.NL + i) * 3 + 1];
        m.lm_pos[(L + l) * 3 + 2] = pts[((size_t)j * m.NL + i) * 3 + 2];
    }
    if (tid == 0) { jb.ba_iters = bd.iters_done; jb.ba_npair = bd.ncontrib; jb.ba_ntrial = bd.ntrial; }
}

// ---------------------------------------------------------------- after a local BA that ran outside a keyframe step
// (Backend::UpdateMap called from outside the frontend, include/StereoVisionSLAM/backend.h:30): the resident list of the
// frame being tracked keeps its features, only the positions of the landmarks they name are the optimised ones
// list3: per job { stream, buffer of the resident list, features in it } as they are NOW (a deferred BA lands frames later)
__global__ void __launch_bounds__(DM_THREADS)
k_dmap_refresh_xyz(const int *list3, DMap m, RtStore rs)
{
    const int tid = threadIdx.x, s = list3[3 * blockIdx.x], buf = list3[3 * blockIdx.x + 1], npts = list3[3 * blockIdx.x + 2];
    const size_t L = dm_l(m, s), R = (size_t)s * rs.max_pts;
    for (int p = tid; p < npts; p += DM_THREADS) {
        const int mp = rs.mp[buf][R + p];
        if (mp < 0) continue;
        double *X = rs.xyz[buf] + 3 * (R + p);
        X[0] = m.lm_pos[(L + mp) * 3]; X[1] = m.lm_pos[(L + mp) * 3 + 1]; X[2] = m.lm_pos[(L + mp) * 3 + 2];
    }
}

// ---------------------------------------------------------------- the keyframe's features -> the list the next frame tracks from
__global__ void __launch_bounds__(DM_THREADS)
k_dmap_refresh(DmJob *jobs, DMap m, RtStore rs)
{
    DmJob &jb = jobs[blockIdx.x];
    const int tid = threadIdx.x, s = jb.stream;
    // (a failed StereoInit uploads its corner list too, like the host pipeline; the next frame initialises again)
    const size_t F = dm_f(m, s, jb.kf_slot), L = dm_l(m, s), R = (size_t)s * rs.max_pts;
    const int n = jb.n_features;
    for (int p = tid; p < n; p += DM_THREADS) {
        const int mp = m.f_lm[F + p];
        rs.xy[jb.dst_buf][R + p] = m.f_xy[F + p];
        rs.mp[jb.dst_buf][R + p] = mp;
        double *X = rs.xyz[jb.dst_buf] + 3 * (R + p);
        if (mp >= 0) { X[0] = m.lm_pos[(L + mp) * 3]; X[1] = m.lm_pos[(L + mp) * 3 + 1]; X[2] = m.lm_pos[(L + mp) * 3 + 2]; }
        else { X[0] = 0; X[1] = 0; X[2] = 1; }
    }
}
