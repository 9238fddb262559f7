// svslam_hip.hip — C-ABI entry points of libsvslam_hip.so (see include/svslam.h).
// Host side: context, HBM-resident pyramid slots, one pinned staging arena
// (one H2D + one D2H per batched call), launches on the context's own stream.
// No CPU fallback exists: if no HIP device is usable svslam_create fails.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <ctime>
#include <sys/prctl.h>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include <dlfcn.h>
// RCCL is bound with dlopen when a shared-map communicator is asked for; the few types and constants of its C API used here
// are declared below (values of rccl.h, the NCCL ABI), so that building this library needs no RCCL headers
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0, ncclMax = 2 } ncclRedOp_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;
}
#include "../../include/svslam.h"
#include "../host/thread_pool.h"
#include <atomic>
#include <memory>
#include <mutex>
#include "dev_common.h"
#include "k_pyramid.h"
#include "k_lk.h"
#include "k_gftt.h"
#include "k_geom.h"
#include "k_ba.h"
#include "k_ba_build.h"
#include "k_ba_ll.h"
#include "k_dmap.h"

#define SVSLAM_DMAP_CHUNK 512     /* keyframe jobs per svslam_dmap_keyframe_batch call the staging arena is sized for */
#define SVSLAM_DMAP_EVICT_PER_JOB 512   /* evicted-landmark records per job of a call (shared by the call's jobs; the surplus waits) */
#define SVSLAM_LL_MAX_PROBLEMS 8        /* local-BA problems per call that the low-latency path (one problem over several workgroups) takes */

namespace {

enum { FAM_PYR = 0, FAM_LK, FAM_GFTT, FAM_TRI, FAM_POSE, FAM_BA, FAM_DBG0, FAM_DBG1, FAM_DBG2, FAM_DBG3,
       FAM_BA_SOLVE,   // the local-BA solver kernel(s) alone, nested inside FAM_BA (gather + build + solve + scatter)
       FAM_COUNT };

struct Timing {
    double ms[FAM_COUNT] = { 0 };
    long long launches[FAM_COUNT] = { 0 };
    long long units[FAM_COUNT] = { 0 };
};

struct Arena {
    unsigned char *h = nullptr; // pinned host
    unsigned char *d = nullptr; // device mirror
    size_t cap = 0, off = 0;
    void reset() { off = 0; }
    size_t take(size_t bytes)
    {
        size_t o = (off + 255) & ~(size_t)255;
        off = o + bytes;
        return o;
    }
};

} // namespace

struct svslam_ctx {
    svslam_limits lim;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    PyrGeom geom;
    std::vector<unsigned> dm_seen; unsigned dm_seen_gen = 0;     // svslam_dmap_keyframe_batch: streams of the current call (duplicate check)
    PyrFusedPlan pyr_plan;        // all levels in one launch (k_pyr_fused), when the geometry allows it
    bool pyr_fused = false;
    uint8_t *d_pyr = nullptr;
    Arena ar;
    // image upload area (host-pointer sources)
    uint8_t *d_img = nullptr;
    size_t d_img_cap = 0;
    uint8_t *h_img = nullptr;
    size_t h_img_cap = 0;
    // GFTT scratch
    GfttWork gw;
    // BA scratch
    BaWork bw;
    int bw_jobs = 0;          // problems per call the BA scratch holds
    bool ba_eid = false;      // the current BA call's cameras have identity extrinsic rotations (the reference's rig): the solver
                              // kernels' EID instantiations leave the products with Re's zeros and ones out (k_ba.h: ba_project)
    std::unique_ptr<svs::ThreadPool> pool;   // host-side per-problem preparation
    long long *d_ba_prof = nullptr;
    // low-latency local BA (svslam_set_low_latency): a problem is dealt over ll.w workgroups (k_ba_split, k_local_ba_t<2>)
    struct { int w = 0; BaDev *shards = nullptr; double *xch = nullptr; unsigned int *cnt = nullptr; size_t xch_stride = 0; BaWork bw;
             LlCaps caps = { 0, 0, 0 };          // caps.B > 0: problems whose shards fit go to the resident kernel (k_ba_ll)
             // The shards of a problem meet at in-launch barriers, so all of them have to be resident at once: max_problems is
             // what the occupancy of both solver kernels x the device's CUs allows (svslam_set_low_latency); calls with more
             // problems take the batch solver.  A shard that still never arrives (the CUs were taken by something else) makes
             // the problem give up after ~1 s; the call then solves it again with the batch solver (fallbacks counts those).
             int max_problems = 0, cus = 0, blocks_per_cu = 0; bool force_batch = false; int test_drop = 0; long long fallbacks = 0;
             // round 6: the give-up is a wall-clock limit per exchange (SVSLAM_LL_TIMEOUT_US, default 2000 us; 100 MHz ticks here);
             // coop: the two solver kernels go through hipLaunchCooperativeKernel (SVSLAM_LL_COOP=1; DESIGN 4.3 has the A/B)
             long long timeout_ticks = 200000; bool coop = false; long long coop_refused = 0;
             std::vector<BaDev> saved; } ll;
    double *d_lm_trace = nullptr;            // svslam_lm_trace test hook: [max_jobs][LM_TRACE_STRIDE]
    // host-side wall time (ns): 0 h2d enqueue, 1 d2h enqueue, 2 stream wait, 3 launches, 4 staging memcpy/prep
    long long host_ns[8] = { 0 };
    bool wait_poll = true;
    bool wait_block = false;    // SVSLAM_WAIT=block: the completion event sleeps in the driver (hipEventBlockingSync)
    bool low_latency = false;   // svslam_set_low_latency: 4-wave pose-only blocks
    double po_xtol = 1e-12;     // svslam_set_pose_only_xtol: parameter tolerance of the pose-only LM (0 = g2o's schedule to the last trial)
    bool zero_copy = false;     // low latency: the small job / result structs of the tracking path are read and written by the
                                // kernels straight in the pinned staging memory (no copy kernel, no boundary before / after it)
    bool timing_split = false;  // SVSLAM_TIMING_SPLIT: per-kernel events of the multi-kernel families (families 6..9)
    bool ba_host_build = false; // SVSLAM_BA_HOST_BUILD: problem structure on the host (the checker of k_ba_build), A/B
    int src_w = 0, src_h = 0;     // > 0: level 0 is the 2:1 decimation of src_w x src_h inputs
    // resident feature lists (svslam_rtrack_*): two alternating buffers per stream
    RtStore rt = {};
    std::vector<int> rt_which, rt_count;
    std::vector<DmEvicted> evicted;      // landmarks the last svslam_dmap_keyframe_batch freed (svslam_dmap_evicted)
    DMap dm = {};                 // device-resident maps (limits.device_map)
    void *dm_all = nullptr;
    int dm_stamp = 0;
    // a deferred local BA of the device map (svslam_dmap_params::ba_defer): its problem, solver scratch and a copy of the jobs
    // live in a buffer of their own (the staging arena is recycled by every call), the solve runs on a second stream
    struct { unsigned char *buf = nullptr; size_t cap = 0; bool inflight = false; int njobs = 0; hipStream_t stream = nullptr;
             hipEvent_t gathered = nullptr, solved = nullptr, t0 = nullptr, t1 = nullptr;
             size_t ojobs = 0, obd = 0, oposes = 0, opts = 0, ochi = 0, oref = 0, olms = 0, oflag = 0; DmParams prm; int MK = 0;
             BaCams cams; int ba_iters = 0; } dmba;
    hipEvent_t done = nullptr;   // recorded after the last enqueue of a call; the stream may be shared
    // a submitted, not yet collected local-BA batch owns the staging arena
    struct { bool active = false; int njobs = 0, total_kf = 0, total_lm = 0, total_obs = 0;
             size_t ojobs = 0, oposes = 0, opts = 0, ochi = 0, oflag = 0;
             // what a repeat with the batch solver needs when the low-latency solver gives up (svslam_local_ba_collect)
             bool ll_used = false; size_t ocams = 0, opk = 0, ouv = 0, osrt = 0, orecs = 0, oaux = 0, out_end = 0; int max_nlm = 0, max_nobs = 0, iters = 0;
             double delta = 0; } ba_pending;
    // an open shared-map BA problem (svslam_sba_*): this rank's shard lives in the arena like a submitted batch
    struct { bool open = false; int nkf = 0, nlm = 0, nobs = 0, np = 0; size_t ojobs = 0, ocams = 0, oposes = 0, opts = 0, orecs = 0,
             oaux = 0, ochi = 0, oio = 0; double delta = 0; int launches = 0; } sba;
    // shared-map BA over RCCL (svslam_sba_comm_* / svslam_sba_solve): communicator of this context's rank
    struct { ncclComm_t comm = nullptr; int nranks = 1, rank = 0; } sbac;
    // timing
    bool timing = false;
    Timing tm;
    hipEvent_t ev[2 * 8];
    int nev = 0;
    int ev_open[4]; int ev_depth = 0;   // event pairs begun and not yet ended (an interval may nest inside another)
    int ev_fam[8];
    long long ev_units[8];
};

static void ll_release(svslam_ctx *c);

namespace {

int fail(svslam_ctx *c, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return -1;
}

#define HIPCHK(c, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) return fail((c), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

void make_geom(PyrGeom &g, int w, int h)
{
    memset(&g, 0, sizeof(g));
    size_t off = 0;
    int lw = w, lh = h, n = 0;
    for (int l = 0; l < SVS_LEVELS; ++l) {
        g.w[l] = lw; g.h[l] = lh;
        g.pitch[l] = (lw + 2 * SVS_BORDER + 15) & ~15;      // 16-byte rows (round 4: 64-byte rows cost 8 % more slot bytes for nothing measurable)
        g.ofs[l] = off;
        off += (size_t)g.pitch[l] * (lh + 2 * SVS_BORDER);
        off = (off + 255) & ~(size_t)255;
        n = l + 1;
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
        // buildOpticalFlowPyramid stops when the next level is not larger than the window
        if (lw <= SVSLAM_LK_WIN || lh <= SVSLAM_LK_WIN) break;
    }
    g.nlevels = n;
    g.slot_bytes = off;
}

// HIP-event intervals of a call, on the stream the kernels are launched on.  Intervals may nest (the solver kernel inside the
// local-BA family): tm_end closes the innermost open one.  At most 8 intervals per call; beyond that they are not recorded.
void tm_begin(svslam_ctx *c, int fam, long long units)
{
    if (!c->timing) return;
    if (c->nev >= 8 || c->ev_depth >= 4) { if (c->ev_depth < 4) c->ev_open[c->ev_depth] = -1; c->ev_depth++; return; }
    const int slot = c->nev++;
    c->ev_open[c->ev_depth++] = slot;
    c->ev_fam[slot] = fam;
    c->ev_units[slot] = units;
    (void)hipEventRecord(c->ev[2 * slot], c->stream);
}
void tm_end(svslam_ctx *c)
{
    if (!c->timing || c->ev_depth <= 0) return;
    const int d = --c->ev_depth;
    const int slot = d < 4 ? c->ev_open[d] : -1;
    if (slot >= 0) (void)hipEventRecord(c->ev[2 * slot + 1], c->stream);
}
// pose-only LM: one wave per job by default, four when the context is in low-latency mode
void launch_pose_only(svslam_ctx *c, int njobs, PoseJob *jobs, const double *cam, const double *xyz, const float2 *uv,
                      const uint8_t *valid, uint8_t *outlier, double chi2_th, int rounds, int iters, const PoFuse *fz = nullptr)
{
    if (c->d_lm_trace) (void)hipMemsetAsync(c->d_lm_trace, 0, sizeof(double) * LM_TRACE_STRIDE * (size_t)njobs, c->stream);
    const PoFuse none = {};
    if (c->low_latency) {
        if (fz) hipLaunchKernelGGL((k_pose_only<4, true>), dim3(njobs), dim3(256), 0, c->stream, jobs, cam, xyz, uv, valid, outlier, chi2_th, rounds, iters, c->d_lm_trace, *fz, c->po_xtol);
        else hipLaunchKernelGGL((k_pose_only<4, false>), dim3(njobs), dim3(256), 0, c->stream, jobs, cam, xyz, uv, valid, outlier, chi2_th, rounds, iters, c->d_lm_trace, none, c->po_xtol);
    } else {
        if (fz) hipLaunchKernelGGL((k_pose_only<1, true>), dim3(njobs), dim3(64), 0, c->stream, jobs, cam, xyz, uv, valid, outlier, chi2_th, rounds, iters, c->d_lm_trace, *fz, c->po_xtol);
        else hipLaunchKernelGGL((k_pose_only<1, false>), dim3(njobs), dim3(64), 0, c->stream, jobs, cam, xyz, uv, valid, outlier, chi2_th, rounds, iters, c->d_lm_trace, none, c->po_xtol);
    }
}
void tm_collect(svslam_ctx *c)
{
    // call after the stream has been synchronised
    for (int i = 0; i < c->nev; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev[2 * i], c->ev[2 * i + 1]) == hipSuccess) {
            c->tm.ms[c->ev_fam[i]] += ms;
            c->tm.launches[c->ev_fam[i]] += 1;
            c->tm.units[c->ev_fam[i]] += c->ev_units[i];
        }
    }
    c->nev = 0; c->ev_depth = 0;
}

// both extrinsics are pure translations: quaternion exactly (0, 0, 0, 1)
static bool ba_ext_identity(const double *ext_l, const double *ext_r)
{
    static const bool off = []{ const char *e = std::getenv("SVSLAM_BA_GENERIC_EXT"); return e && atoi(e) != 0; }();   // A/B and parity tests: the general code
    for (const double *e : { ext_l, ext_r }) if (!(e[0] == 0.0 && e[1] == 0.0 && e[2] == 0.0 && e[3] == 1.0)) return false;
    return !off;
}
// The local-BA solver of a batch whose structure the device builds: the batch kernel (one workgroup per problem), or — in
// low-latency mode, for a few problems with landmark-major edges — every problem dealt over ll.w workgroups.
// A solver kernel of the low-latency path: an ordinary launch, or — ll.coop — a cooperative one: the runtime then checks that the
// whole grid can be resident at once (and refuses the launch otherwise: counted, the ordinary launch + the in-kernel give-up
// take over) and runs cooperative launches of different streams one after the other, so two half-resident problems of two
// contexts cannot wait for each other's CUs.
template <class... P, class... A> void ll_launch(svslam_ctx *c, void (*kern)(P...), int nshards, size_t lds, A... a)
{
    if (c->ll.coop) {
        std::tuple<P...> args(a...);
        void *ptrs[sizeof...(P)];
        size_t i = 0;
        std::apply([&](auto &...x) { ((ptrs[i++] = (void *)&x), ...); }, args);
        if (hipLaunchCooperativeKernel(reinterpret_cast<const void *>(kern), dim3(nshards), dim3(BA_THREADS), ptrs, (unsigned)lds, c->stream) == hipSuccess) return;
        (void)hipGetLastError();
        c->ll.coop_refused++;
    }
    hipLaunchKernelGGL(kern, dim3(nshards), dim3(BA_THREADS), lds, c->stream, a...);
}
template <int W> void launch_ba_ll_t(svslam_ctx *c, int nshards, BaDev *shards, const BaCams *cams, double *poses, double *pts, const BaRec *recs,
                                      const int *aux, double delta, int iters, double *chi, int tile_cap, BaDev *parents)
{
    ll_launch(c, k_local_ba_t<2, W>, nshards, ba_lds_bytes_ll(c->lim.max_kf), shards, cams, poses, pts, recs, aux, c->ll.bw, delta, iters, chi,
              c->d_ba_prof, tile_cap,
              SbaArgs{ 0, 0, 0.0, nullptr, c->d_lm_trace, 0, c->ll.xch, c->ll.cnt, parents, c->ll.xch_stride, c->ll.timeout_ticks });
}
template <int W> void launch_ba_ll_resident(svslam_ctx *c, int nshards, BaDev *shards, const BaCams *cams, double *poses, double *pts, const BaRec *recs,
                                            const int *aux, double delta, int iters, double *chi, BaDev *parents)
{
    const SbaArgs sa{ 0, 0, 0.0, nullptr, c->d_lm_trace, 0, c->ll.xch, c->ll.cnt, parents, c->ll.xch_stride, c->ll.timeout_ticks };
    if (c->ba_eid)
        ll_launch(c, k_ba_ll<W, true>, nshards, ba_ll_lds_bytes(c->lim.max_kf, c->ll.caps), shards, cams, poses, pts, recs, aux, delta, iters, chi,
                  c->d_ba_prof, c->ll.caps, sa);
    else
        ll_launch(c, k_ba_ll<W, false>, nshards, ba_ll_lds_bytes(c->lim.max_kf, c->ll.caps), shards, cams, poses, pts, recs, aux, delta, iters, chi,
                  c->d_ba_prof, c->ll.caps, sa);
}
// test hook (SVSLAM_LL_TEST_DROP_SHARD = n: the next n low-latency launches): shard 0 of problem 0 never runs, its peers give up at
// their first exchange and the call falls back to the batch solver — the path a GPU without room for every shard takes
__global__ void k_ll_test_drop(BaDev *shards) { shards[0].reserved = 3; }
bool ba_ll_usable(const svslam_ctx *c, int njobs) { return c->low_latency && c->ll.w > 0 && !c->ll.force_batch && njobs <= c->ll.max_problems; }
// tile capacity the shards of the low-latency solver are built with: what both of its kernels can hold
int ba_ll_tile_cap(const svslam_ctx *c)
{
    const int t = ba_tile_cap_ll(c->lim.max_kf);
    return c->ll.caps.B > 0 ? std::min(t, c->ll.caps.B) : t;
}
void launch_ba_solver(svslam_ctx *c, int njobs, bool ll, BaDev *jobs, const BaCams *cams, double *poses, double *pts, const unsigned int *packed,
                      const float2 *uv, const int *srt, BaRec *recs, int *aux, double *chi, int *flag, int max_nlm, int max_nobs,
                      double delta, int iters, bool split_timing, bool time_solver = true)
{
    // (time_solver: FAM_BA_SOLVE interval around the solver kernel(s); off where the solve runs on a stream the call does not wait for)
    const bool tms = time_solver && !split_timing;
    const int tile_cap = ll ? ba_ll_tile_cap(c) : ba_tile_cap(c->lim.max_kf);
    const int ec = bb_edge_cache_fits(max_nlm, max_nobs) ? 1 : 0;
    if (!ll) {
        hipLaunchKernelGGL(k_ba_build, dim3(njobs), dim3(BB_THREADS), bb_lds_bytes(max_nlm, max_nobs), c->stream, jobs, packed, uv, srt, recs, aux,
                           tile_cap, max_nlm, flag, 0, ec, 0);
        if (split_timing) { tm_end(c); tm_begin(c, FAM_DBG3, njobs); }
        const SbaArgs sa{ 0, 0, 0.0, nullptr, c->d_lm_trace, 0, nullptr, nullptr, nullptr, 0, 0 };
        if (tms) tm_begin(c, FAM_BA_SOLVE, njobs);
        if (c->ba_eid)
            hipLaunchKernelGGL((k_local_ba_t<0, 1, true>), dim3(njobs), dim3(BA_THREADS), ba_lds_bytes(c->lim.max_kf), c->stream, jobs, cams, poses, pts, recs, aux,
                               c->bw, delta, iters, chi, c->d_ba_prof, tile_cap, sa);
        else
            hipLaunchKernelGGL((k_local_ba_t<0, 1, false>), dim3(njobs), dim3(BA_THREADS), ba_lds_bytes(c->lim.max_kf), c->stream, jobs, cams, poses, pts, recs, aux,
                               c->bw, delta, iters, chi, c->d_ba_prof, tile_cap, sa);
        if (tms) tm_end(c);
        return;
    }
    const int W = c->ll.w;
    c->host_ns[6] += njobs;                    // svslam_debug_host_ns slot 6: problems the low-latency solver took
    hipLaunchKernelGGL(k_ba_split, dim3(njobs), dim3(SP_THREADS), ba_split_lds_bytes(max_nlm, W), c->stream, jobs, c->ll.shards, packed, W, tile_cap,
                       max_nlm, c->ll.xch, c->ll.xch_stride, c->ll.cnt, c->ll.caps.B, c->ll.caps.L, c->ll.caps.E);
    hipLaunchKernelGGL(k_ba_build, dim3(njobs * W), dim3(BB_THREADS), bb_lds_bytes(max_nlm, max_nobs), c->stream, c->ll.shards, packed, uv, srt, recs, aux,
                       tile_cap, max_nlm, flag, 1 /* every keyframe active in every shard */, ec, 1 /* every landmark in the tile */);
    if (split_timing) { tm_end(c); tm_begin(c, FAM_DBG3, njobs); }
    if (c->ll.test_drop > 0) { --c->ll.test_drop; hipLaunchKernelGGL(k_ll_test_drop, dim3(1), dim3(1), 0, c->stream, c->ll.shards); }
    // problems whose shards all fit LDS: the resident kernel; the others: the streaming one (each kernel skips the other's)
    if (tms) tm_begin(c, FAM_BA_SOLVE, njobs);
    if (c->ll.caps.B > 0) {
        if (W == 4) launch_ba_ll_resident<4>(c, njobs * W, c->ll.shards, cams, poses, pts, recs, aux, delta, iters, chi, jobs);
        else if (W == 8) launch_ba_ll_resident<8>(c, njobs * W, c->ll.shards, cams, poses, pts, recs, aux, delta, iters, chi, jobs);
        else launch_ba_ll_resident<16>(c, njobs * W, c->ll.shards, cams, poses, pts, recs, aux, delta, iters, chi, jobs);
    }
    if (W == 4) launch_ba_ll_t<4>(c, njobs * W, c->ll.shards, cams, poses, pts, recs, aux, delta, iters, chi, tile_cap, jobs);
    else if (W == 16) launch_ba_ll_t<16>(c, njobs * W, c->ll.shards, cams, poses, pts, recs, aux, delta, iters, chi, tile_cap, jobs);
    else if (W == 8) launch_ba_ll_t<8>(c, njobs * W, c->ll.shards, cams, poses, pts, recs, aux, delta, iters, chi, tile_cap, jobs);
    if (tms) tm_end(c);
}

inline long long now_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// Wait for the context's stream.  The runtime's own wait spins; with thousands of streams
// the host bookkeeping needs every core that would burn, so the default here is a short
// spin followed by sleep-polling (SVSLAM_WAIT=spin restores hipStreamSynchronize).
hipError_t wait_stream(svslam_ctx *c)
{
    if (!c->wait_poll) return hipEventSynchronize(c->done);
    static thread_local bool slack_set = false;
    if (!slack_set) { (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0); slack_set = true; }
    const long long t0 = now_ns();
    for (;;) {
        const hipError_t e = hipEventQuery(c->done);
        if (e != hipErrorNotReady) return e;
        const long long dt = now_ns() - t0;
        // low-latency mode (a few cameras per GPU): the wait IS the frame time, so the thread keeps polling — a nap of 20 us
        // plus its wake-up was a tenth of a lone camera's frame; beyond a few ms (nothing of that mode takes that long) it naps
        if (dt < (c->low_latency ? 3000000LL : 10000LL)) continue;
        // back off with the age of the wait: a 100 us kernel is polled every ~20 us, a
        // multi-millisecond BA batch every ~150 us (overshoot stays below ~1/8 of the wait)
        static const long long nap_min = []{ const char *e = std::getenv("SVSLAM_POLL_MIN_US"); return e ? atoll(e) * 1000 : 20000LL; }();
        static const long long nap_max = []{ const char *e = std::getenv("SVSLAM_POLL_MAX_US"); return e ? atoll(e) * 1000 : 150000LL; }();
        long long nap = dt >> 3;
        nap = nap < nap_min ? nap_min : (nap > nap_max ? nap_max : nap);
        struct timespec ts = { 0, (long)nap };
        nanosleep(&ts, nullptr);
    }
}
// HIP streams are handed out from a small per-device pool: beyond ~16 hardware queues the
// GPU time-slices them and every stream stalls, while a host that runs many single-threaded
// pipelines wants one context each.  Contexts that share a stream stay independent: each
// waits on its own event (done), never on the stream.  Pool streams live until process exit.
hipError_t pool_stream(int device, hipStream_t *out)
{
    static std::mutex m;
    static std::vector<hipStream_t> pool[64];
    static size_t next[64] = { 0 };
    std::lock_guard<std::mutex> lk(m);
    const char *e = std::getenv("SVSLAM_MAX_STREAMS");
    const size_t cap = (size_t)std::max(1, e ? atoi(e) : 16);
    std::vector<hipStream_t> &p = pool[device & 63];
    if (p.size() < cap) {
        hipStream_t s = nullptr;
        hipError_t rc = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (rc != hipSuccess) return rc;
        p.push_back(s);
        *out = s;
        return hipSuccess;
    }
    *out = p[next[device & 63]++ % std::min(p.size(), cap)];
    return hipSuccess;
}
int h2d(svslam_ctx *c, size_t from, size_t to)
{
    const long long t0 = now_ns();
    if (to > from) HIPCHK(c, hipMemcpyAsync(c->ar.d + from, c->ar.h + from, to - from, hipMemcpyHostToDevice, c->stream));
    c->host_ns[0] += now_ns() - t0;
    return 0;
}
int d2h_enqueue(svslam_ctx *c, size_t from, size_t to)
{
    const long long t0 = now_ns();
    if (to > from) HIPCHK(c, hipMemcpyAsync(c->ar.h + from, c->ar.d + from, to - from, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->done, c->stream));
    c->host_ns[1] += now_ns() - t0;
    return 0;
}
int finish(svslam_ctx *c)
{
    const long long t1 = now_ns();
    HIPCHK(c, wait_stream(c));
    const long long t0 = now_ns();
    c->host_ns[2] += t0 - t1;
    tm_collect(c);
    c->host_ns[5] += now_ns() - t0;
    return 0;
}
int d2h_sync(svslam_ctx *c, size_t from, size_t to)
{
    if (d2h_enqueue(c, from, to)) return -1;
    return finish(c);
}

template <typename T> T *hp(svslam_ctx *c, size_t off) { return reinterpret_cast<T *>(c->ar.h + off); }
template <typename T> T *dp(svslam_ctx *c, size_t off) { return reinterpret_cast<T *>(c->ar.d + off); }
// a staged struct array as the kernels see it: the device mirror, or (zero_copy) the pinned host memory itself — hipHostMalloc
// memory is mapped into the device's address space at the same address, coherent, and what a kernel wrote there is visible to
// the host once the event behind the kernel has completed
template <typename T> T *dpz(svslam_ctx *c, size_t off) { return c->zero_copy ? hp<T>(c, off) : dp<T>(c, off); }

// The staging arena (and job 0 of the BA scratch) belongs to a submitted, uncollected local-BA batch or to an
// open shared-map shard: every other batched call on the context is refused until collect / close.
int arena_busy(svslam_ctx *c)
{
    if (c->ba_pending.active) return fail(c, "a submitted local-BA batch owns this context: call svslam_local_ba_collect first");
    if (c->sba.open) return fail(c, "an open shared-map BA shard owns this context: call svslam_sba_close first");
    return 0;
}

int check_slot(svslam_ctx *c, int s)
{
    if (s < 0 || s >= c->lim.max_slots) return fail(c, "slot %d out of range [0,%d)", s, c->lim.max_slots);
    return 0;
}

// enqueue pyramid construction for n jobs whose PyrJob array is at device offset djobs
// ga (optional): the feature gather of resident tracking, riding on the fused launch as extra workgroups (gathered = true
// says it did; with the per-level kernels the caller launches k_rt_gather itself)
int launch_pyramid(svslam_ctx *c, const PyrJob *djobs, int n, bool decimate, int src_w, int src_h, const RtGatherArgs *ga = nullptr,
                   bool *gathered = nullptr)
{
    const PyrGeom &g = c->geom;
    tm_begin(c, FAM_PYR, n);
    if (c->pyr_fused) {
        RtGatherArgs a = {};
        a.chunks = 1;
        if (ga && ga->njobs > 0) { a = *ga; if (gathered) *gathered = true; }
        const dim3 grd(((n + 7) / 8) * 8 * c->pyr_plan.nstrips + a.njobs * a.chunks);      // jobs in groups of 8 (one per XCD), all strips of a job on its XCD
        if (decimate) hipLaunchKernelGGL(k_pyr_fused<true>, grd, dim3(PF_THREADS), c->pyr_plan.lds_bytes, c->stream, djobs, n, c->d_pyr, g, src_w, src_h, c->pyr_plan, a);
        else hipLaunchKernelGGL(k_pyr_fused<false>, grd, dim3(PF_THREADS), c->pyr_plan.lds_bytes, c->stream, djobs, n, c->d_pyr, g, src_w, src_h, c->pyr_plan, a);
        tm_end(c);
        HIPCHK(c, hipGetLastError());
        return 0;
    }
    auto fast_ok = [](int w, int h) { return w >= 18 && h >= 18; };
    if (fast_ok(g.w[0], g.h[0])) {
        dim3 blk(16, 16);
        dim3 grd(cdiv(cdiv(g.w[0], 16), 16), cdiv(g.h[0], 16), n);
        if (decimate) hipLaunchKernelGGL(k_pyr_level0_fast<true>, grd, blk, 0, c->stream, djobs, c->d_pyr, g, src_w, src_h);
        else hipLaunchKernelGGL(k_pyr_level0_fast<false>, grd, blk, 0, c->stream, djobs, c->d_pyr, g, src_w, src_h);
    } else {
        dim3 blk(64, 4);
        dim3 grd(cdiv((g.w[0] + 2 * SVS_BORDER + 3) / 4, 64), cdiv(g.h[0] + 2 * SVS_BORDER, 4), n);
        if (decimate) hipLaunchKernelGGL(k_pyr_level0<true>, grd, blk, 0, c->stream, djobs, c->d_pyr, g, src_w, src_h);
        else hipLaunchKernelGGL(k_pyr_level0<false>, grd, blk, 0, c->stream, djobs, c->d_pyr, g, src_w, src_h);
    }
    for (int l = 1; l < g.nlevels; ++l) {
        if (fast_ok(g.w[l], g.h[l])) {
            dim3 blk(32, 8);
            dim3 grd(cdiv(cdiv(g.w[l], 4), 32), cdiv(g.h[l], 8), n);
            hipLaunchKernelGGL(k_pyr_down_fast, grd, blk, 0, c->stream, djobs, c->d_pyr, g, l);
        } else {
            dim3 blk(64, 4);
            dim3 grd(cdiv(g.w[l] + 2 * SVS_BORDER, 64), cdiv(g.h[l] + 2 * SVS_BORDER, 4), n);
            hipLaunchKernelGGL(k_pyr_down, grd, blk, 0, c->stream, djobs, c->d_pyr, g, l);
        }
    }
    tm_end(c);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// more (optional): called once the pyramid's own jobs are staged and before its launch; stages the caller's further inputs behind
// them (the arena is not reset again) and may fill *ga to let the resident-tracking gather ride on the pyramid launch
struct PyrMore { std::function<int(RtGatherArgs *)> stage; bool gathered = false; };
int pyramid_common(svslam_ctx *c, int n, const int *slots, const void *const *imgs, const int *strides,
                   int src_is_device, bool decimate, int src_w, int src_h, bool sync, PyrMore *more = nullptr)
{
    if (n <= 0) return 0;
    if (n > c->lim.max_jobs) return fail(c, "pyramid: %d jobs > max_jobs %d", n, c->lim.max_jobs);
    for (int i = 0; i < n; ++i) if (check_slot(c, slots[i])) return -1;
    const int iw = decimate ? src_w : c->geom.w[0], ih = decimate ? src_h : c->geom.h[0];
    if (arena_busy(c)) return -1;
    c->ar.reset();
    size_t ojobs = c->ar.take(sizeof(PyrJob) * n);
    PyrJob *hj = hp<PyrJob>(c, ojobs);
    if (!src_is_device) {
        size_t per = ((size_t)iw * ih + 255) & ~(size_t)255;
        size_t need = per * n;
        if (need > c->d_img_cap) {
            if (c->d_img) (void)hipFree(c->d_img);
            if (c->h_img) (void)hipHostFree(c->h_img);
            c->d_img = nullptr; c->h_img = nullptr; c->d_img_cap = 0;
            HIPCHK(c, hipMalloc(&c->d_img, need));
            HIPCHK(c, hipHostMalloc(&c->h_img, need));
            c->d_img_cap = need;
        }
        for (int i = 0; i < n; ++i) {
            const uint8_t *s = static_cast<const uint8_t *>(imgs[i]);
            uint8_t *d = c->h_img + per * i;
            for (int y = 0; y < ih; ++y) memcpy(d + (size_t)y * iw, s + (size_t)y * strides[i], iw);
            hj[i].src = c->d_img + per * i;
            hj[i].src_stride = iw;
            hj[i].slot = slots[i];
        }
        HIPCHK(c, hipMemcpyAsync(c->d_img, c->h_img, need, hipMemcpyHostToDevice, c->stream));
    } else {
        for (int i = 0; i < n; ++i) {
            hj[i].src = static_cast<const uint8_t *>(imgs[i]);
            hj[i].src_stride = strides[i];
            hj[i].slot = slots[i];
        }
    }
    if (!c->zero_copy && h2d(c, ojobs, c->ar.off)) return -1;
    RtGatherArgs ga = {};
    if (more && more->stage(&ga)) return -1;
    if (launch_pyramid(c, dpz<PyrJob>(c, ojobs), n, decimate, src_w, src_h, more ? &ga : nullptr, more ? &more->gathered : nullptr)) return -1;
    if (sync) return d2h_sync(c, 0, 0);
    return 0;
}

} // namespace

extern "C" {

const char *svslam_build_info(void)
{
#ifndef SVS_SRC_HASH
#define SVS_SRC_HASH "unstamped"
#endif
    return "libsvslam_hip gfx950 (hipcc " __VERSION__ "), -ffp-contract=off, src " SVS_SRC_HASH;
}

const char *svslam_last_error(const svslam_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int svslam_create(const svslam_limits *lim, svslam_ctx **out)
{
    if (!lim || !out) return -1;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "svslam_create: no HIP device available (this library has no CPU path)\n");
        return -2;
    }
    if (lim->device < 0 || lim->device >= ndev) return -3;
    if (lim->width < 16 || lim->height < 16 || lim->max_slots < 1 || lim->max_jobs < 1 || lim->max_pts < 1)
        return -4;
    if (lim->max_pts > PO_MAX_EDGES) return -5;
    if (lim->max_corners < 1 || lim->max_corners > GF_MAX_CORNERS) return -6;
    svslam_ctx *c = new (std::nothrow) svslam_ctx();
    if (!c) return -7;
    c->lim = *lim;
    c->device = lim->device;
    *out = c; // returned even on failure so the caller can read the error
    HIPCHK(c, hipSetDevice(c->device));
    {
        const char *wm = std::getenv("SVSLAM_WAIT");        // spin | block | poll (default)
        c->wait_poll = !(wm && (std::strcmp(wm, "spin") == 0 || std::strcmp(wm, "block") == 0));
        c->wait_block = wm && std::strcmp(wm, "block") == 0;
        c->timing_split = std::getenv("SVSLAM_TIMING_SPLIT") != nullptr;
        if (const char *xt = std::getenv("SVSLAM_PO_XTOL")) {      // 0: g2o's schedule; a value the setter would refuse fails the create
            char *end = nullptr;
            const double v = strtod(xt, &end);
            if (end == xt || !(v >= 0) || v > 1e-6) return fail(c, "svslam_create: SVSLAM_PO_XTOL=%s is outside [0, 1e-6]", xt);
            c->po_xtol = v;
        }
        // the device build takes the edge indices packed into one word (landmark < 2^16, keyframe < 2^8)
        c->ba_host_build = std::getenv("SVSLAM_BA_HOST_BUILD") != nullptr || lim->max_lm >= 65536 || lim->max_kf >= 256;
    }
    HIPCHK(c, pool_stream(c->device, &c->stream));
    HIPCHK(c, hipEventCreateWithFlags(&c->done, hipEventDisableTiming | (c->wait_block ? hipEventBlockingSync : 0)));
    for (int i = 0; i < 16; ++i) HIPCHK(c, hipEventCreate(&c->ev[i]));
    make_geom(c->geom, lim->width, lim->height);
    int pyr_lds_kb = 64;            // two strips' workgroups per CU; development: SVSLAM_PYR_LDS_KB = 32 .. 160
    if (const char *e = std::getenv("SVSLAM_PYR_LDS_KB")) { const int v = std::atoi(e); if (v >= 32 && v <= 160) pyr_lds_kb = v; }
    c->pyr_fused = !std::getenv("SVSLAM_PYR_LEGACY") && pyr_fused_plan(c->geom, pyr_lds_kb * 1024, c->pyr_plan);
    if (c->pyr_fused && c->pyr_plan.lds_bytes > 64 * 1024) {
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_pyr_fused<true>), hipFuncAttributeMaxDynamicSharedMemorySize, c->pyr_plan.lds_bytes));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_pyr_fused<false>), hipFuncAttributeMaxDynamicSharedMemorySize, c->pyr_plan.lds_bytes));
    }
    if (c->pyr_fused && std::getenv("SVSLAM_PYR_PROF")) {
        HIPCHK(c, hipMalloc(&c->pyr_plan.prof, sizeof(long long) * 8));
        HIPCHK(c, hipMemset(c->pyr_plan.prof, 0, sizeof(long long) * 8));
    }
    HIPCHK(c, hipMalloc(&c->d_pyr, c->geom.slot_bytes * (size_t)lim->max_slots));
    HIPCHK(c, hipMemsetAsync(c->d_pyr, 0, c->geom.slot_bytes * (size_t)lim->max_slots, c->stream));

    // staging arena: the largest batched call decides
    const size_t J = lim->max_jobs, N = lim->max_pts;
    size_t per_job_pts = N * (8 + 8 + 1 + 4 + 24 + 8 + 1 + 1 + 8) + 1024;
    size_t per_job_ba = (size_t)lim->max_kf * 56 + (size_t)lim->max_lm * 24 +
                        (size_t)lim->max_obs * (21 + 8 + 32 + 16 + 2 * (size_t)(lim->max_kf + 1) + 8) + (size_t)lim->max_lm * 16 + 8192;
    // (per edge: raw <= 21 B, chi2 8 B, two records 32 B, edge + block lists 16 B, pair items <= 2 (K + 1) B)
    per_job_ba += 4 * (size_t)(ba_tile_bound(lim->max_lm, lim->max_obs, lim->max_kf, std::max(ba_tile_cap(lim->max_kf), 64)) + 2) *
                  ((size_t)lim->max_kf * (lim->max_kf + 1) / 2 + 1);          // per-tile pair ranges at their upper bound
    per_job_ba += 4 * ba_split_aux_extra(lim->max_kf, LL_MAX_W);                  // low-latency shards of a problem (k_ba_split)
    size_t per_job = std::max(per_job_pts, per_job_ba) + (size_t)lim->max_corners * 8 + 4096;
    c->ar.cap = per_job * J + (1 << 20);
    if (lim->device_map) {
        // With the map on the device the big consumer — BA staging for max_jobs host-gathered problems — is not used:
        // the keyframe path takes at most SVSLAM_DMAP_CHUNK jobs per call and keeps its scratch device-side (the arena
        // mirrors host and device, and 12 contexts x 4 GB of pinned host memory per process was the price of the old
        // sizing).  Tracking calls still get their per-point staging for every job; host-side BA calls on such a
        // context are limited to 64 problems per call.
        const size_t MO = lim->max_obs, NL = lim->max_lm, NF = lim->max_pts, MK = lim->max_kf;
        const int tc = std::max(ba_tile_cap(lim->max_kf), 64);
        const size_t aux = ba_aux_layout((int)MK, (int)NL, (int)MO, (int)MO, (int)MK, 0, ba_tile_bound((int)NL, (int)MO, (int)MK, tc)).total + ba_pitem_bound((int)MO, (int)MK);
        const size_t per_dm = sizeof(DmJob) + NF * 80 + NL * 32 + MO * 64 + (aux + ba_split_aux_extra((int)MK, LL_MAX_W)) * 4 + MK * 56 + (size_t)lim->max_corners * 8 + 8192 +
                              sizeof(DmEvicted) * SVSLAM_DMAP_EVICT_PER_JOB;
        const size_t chunk = std::min<size_t>(SVSLAM_DMAP_CHUNK, (size_t)std::max(1, lim->max_streams));
        c->ar.cap = std::max(chunk * per_dm, std::max(per_job_pts * J, std::min<size_t>(J, 64) * per_job)) + (4 << 20);
    }
    HIPCHK(c, hipHostMalloc(&c->ar.h, c->ar.cap));
    HIPCHK(c, hipMalloc(&c->ar.d, c->ar.cap));

    // GFTT scratch
    const size_t P = (size_t)lim->width * lim->height;
    int cap = 1;
    while ((size_t)cap < P) cap <<= 1;
    c->gw.cap = cap;
    HIPCHK(c, hipMalloc(&c->gw.keys, sizeof(unsigned long long) * (size_t)cap * J));
    HIPCHK(c, hipMalloc(&c->gw.counters, sizeof(unsigned int) * GF_CNT_STRIDE * J));
    HIPCHK(c, hipMemsetAsync(c->gw.counters, 0, sizeof(unsigned int) * GF_CNT_STRIDE * J, c->stream));   // kept zero by k_gftt_select2
    c->gw.prof = nullptr;
    if (std::getenv("SVSLAM_GFTT_PROF")) {
        HIPCHK(c, hipMalloc(&c->gw.prof, sizeof(long long) * 16));
        HIPCHK(c, hipMemsetAsync(c->gw.prof, 0, sizeof(long long) * 16, c->stream));
    }
    // BA scratch
    if (lim->max_kf > 0) {
        if (6 * lim->max_kf > BA_MAX_NP) return fail(c, "max_kf %d too large (<= %d)", lim->max_kf, BA_MAX_NP / 6);
        if (ba_tile_cap(lim->max_kf) < std::max(lim->max_kf, 64) || ba_lds_bytes(lim->max_kf) > 160 * 1024) return fail(c, "max_kf %d: reduced system does not fit LDS", lim->max_kf);
        // per-problem solver scratch (0.5 MB + 0.2 KB per landmark slot): one per job of the largest BA call — with the map on the
        // device that is a keyframe chunk (or the 64 problems a host-side BA call may bring), not max_jobs
        c->bw_jobs = lim->device_map ? (int)std::max<size_t>(std::min<size_t>(SVSLAM_DMAP_CHUNK, (size_t)std::max(1, lim->max_streams)), std::min<size_t>(J, 64)) : lim->max_jobs;
        if (ba_work_alloc(c->bw, c->bw_jobs, lim->max_kf, lim->max_lm, lim->max_obs) != hipSuccess)
            return fail(c, "BA workspace allocation failed");
        for (const void *f : { reinterpret_cast<const void *>(k_local_ba_t<0, 1, false>), reinterpret_cast<const void *>(k_local_ba_t<0, 1, true>),
                               reinterpret_cast<const void *>(k_local_ba_t<1, 1>) })
            HIPCHK(c, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ba_lds_bytes(lim->max_kf)));
        if (ba_tile_cap_ll(lim->max_kf) >= std::max(lim->max_kf, 64))
            for (const void *f : { reinterpret_cast<const void *>(k_local_ba_t<2, 4>), reinterpret_cast<const void *>(k_local_ba_t<2, 8>),
                                   reinterpret_cast<const void *>(k_local_ba_t<2, 16>) })
                HIPCHK(c, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ba_lds_bytes_ll(lim->max_kf)));
        if (bb_lds_bytes(lim->max_lm) > 160 * 1024 || lim->max_kf > 32)
            return fail(c, "max_lm %d / max_kf %d: problem structure does not fit LDS", lim->max_lm, lim->max_kf);
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_ba_build),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (lim->max_streams > 0) {
        const size_t n = (size_t)lim->max_streams * lim->max_pts;
        for (int b = 0; b < 2; ++b) {
            HIPCHK(c, hipMalloc(&c->rt.xy[b], n * sizeof(float2)));
            HIPCHK(c, hipMalloc(&c->rt.mp[b], n * sizeof(int)));
            HIPCHK(c, hipMalloc(&c->rt.xyz[b], n * 3 * sizeof(double)));
        }
        c->rt.max_pts = lim->max_pts;
        c->rt_which.assign((size_t)lim->max_streams, 0);
        c->rt_count.assign((size_t)lim->max_streams, 0);
    }
    if (lim->device_map) {
        if (lim->max_streams <= 0 || lim->max_kf <= 0) return fail(c, "device_map needs max_streams > 0 and BA limits");
        if (lim->max_lm & (lim->max_lm - 1)) return fail(c, "device_map: max_lm %d must be a power of two", lim->max_lm);
        if (lim->max_kf > 12 || lim->max_pts >= 32768 || lim->max_lm >= 65536) return fail(c, "device_map: max_kf <= 12, max_pts < 32768, max_lm < 65536");
        if (dmg_lds_bytes(lim->max_lm) > 160 * 1024) return fail(c, "device_map: max_lm %d does not fit the gather's LDS", lim->max_lm);
        DMap &m = c->dm;
        m.KW = lim->max_kf; m.NF = lim->max_pts; m.NL = lim->max_lm;
        const size_t S = lim->max_streams, KF = S * m.KW, FT = KF * m.NF, LM = S * m.NL;
        const size_t bytes = KF * (8 + 4 + 56 + 4) + FT * (8 + 8 + 4 + 4 + 1) + LM * (24 + 4 + 4 + 1 + 4) + S * 4 + 4096;
        HIPCHK(c, hipMalloc(&c->dm_all, bytes));
        HIPCHK(c, hipMemsetAsync(c->dm_all, 0, bytes, c->stream));
        unsigned char *q = static_cast<unsigned char *>(c->dm_all);
        auto take = [&](size_t n) { unsigned char *r = q; q += (n + 255) & ~(size_t)255; return r; };
        m.kf_pose = reinterpret_cast<double *>(take(KF * 56)); m.lm_pos = reinterpret_cast<double *>(take(LM * 24));
        m.kf_frame = reinterpret_cast<long long *>(take(KF * 8));
        m.f_xy = reinterpret_cast<float2 *>(take(FT * 8)); m.f_xyr = reinterpret_cast<float2 *>(take(FT * 8));
        m.kf_id = reinterpret_cast<int *>(take(KF * 4)); m.kf_n = reinterpret_cast<int *>(take(KF * 4));
        m.f_lm = reinterpret_cast<int *>(take(FT * 4)); m.f_lmr = reinterpret_cast<int *>(take(FT * 4));
        m.lm_id = reinterpret_cast<int *>(take(LM * 4)); m.lm_obs = reinterpret_cast<int *>(take(LM * 4)); m.lm_stamp = reinterpret_cast<int *>(take(LM * 4));
        m.next_lm_id = reinterpret_cast<int *>(take(S * 4));
        m.f_fl = take(FT); m.lm_st = take(LM);
        // empty window slots / free landmark slots are marked by -1
        HIPCHK(c, hipMemsetAsync(m.kf_frame, 0xff, KF * 8, c->stream));
        HIPCHK(c, hipMemsetAsync(m.lm_id, 0xff, LM * 4, c->stream));
        HIPCHK(c, hipMemsetAsync(m.lm_stamp, 0xff, LM * 4, c->stream));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_dmap_ba_gather<1024>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)dmg_lds_bytes_t<1024>(lim->max_lm)));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_dmap_ba_gather<512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)dmg_lds_bytes_t<512>(lim->max_lm)));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

void svslam_destroy(svslam_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int b = 0; b < 2; ++b) { (void)hipFree(c->rt.xy[b]); (void)hipFree(c->rt.mp[b]); (void)hipFree(c->rt.xyz[b]); }
    (void)hipFree(c->d_pyr);
    if (c->dm_all) (void)hipFree(c->dm_all);
    if (c->dmba.stream) (void)hipStreamSynchronize(c->dmba.stream);
    if (c->dmba.buf) (void)hipFree(c->dmba.buf);
    for (hipEvent_t e : { c->dmba.gathered, c->dmba.solved, c->dmba.t0, c->dmba.t1 }) if (e) (void)hipEventDestroy(e);
    if (c->dmba.stream) (void)hipStreamDestroy(c->dmba.stream);
    (void)hipFree(c->ar.d);
    if (c->ar.h) (void)hipHostFree(c->ar.h);
    (void)hipFree(c->d_img);
    if (c->h_img) (void)(void)hipHostFree(c->h_img);
    if (c->pyr_fused && c->pyr_plan.prof) {
        long long p[8] = { 0 };
        (void)hipMemcpy(p, c->pyr_plan.prof, sizeof(p), hipMemcpyDeviceToHost);
        const double k = p[7] ? 0.01 / (double)p[7] : 0.0;
        fprintf(stderr, "[pyr fused, workgroup (1,0), %lld launches, %d strips, lds %d] us: fill %.1f | border+store+pyrDown per level %.1f %.1f %.1f | last level %.1f\n",
                p[7], c->pyr_plan.nstrips, c->pyr_plan.lds_bytes, k * p[0], k * p[1], k * p[2], k * p[3], k * p[5]);
        (void)hipFree(c->pyr_plan.prof);
    }
    if (c->gw.prof) {
        long long p[16] = { 0 };
        (void)hipMemcpy(p, c->gw.prof, sizeof(p), hipMemcpyDeviceToHost);
        const double k = p[7] ? 0.01 / (double)p[7] : 0.0;      // 100 MHz ticks -> us per call
        fprintf(stderr, "[gftt select2, job 0, %lld calls] us/call: clear %.1f hist %.1f scan %.1f gather %.1f sort %.1f greedy %.1f | candidates %.0f, sorted %.0f\n",
                p[7], k * p[8], k * p[0], k * p[1], k * p[2], k * p[3], k * p[4], p[7] ? (double)p[5] / p[7] : 0.0, p[7] ? (double)p[6] / p[7] : 0.0);
        (void)hipFree(c->gw.prof);
    }
    (void)hipFree(c->gw.keys); (void)hipFree(c->gw.counters);
    ba_work_free(c->bw);
    ll_release(c);
    if (c->d_ba_prof) (void)hipFree(c->d_ba_prof);
    if (c->d_lm_trace) (void)hipFree(c->d_lm_trace);
    (void)svslam_sba_comm_destroy(c);
    for (int i = 0; i < 16; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->done) (void)hipEventDestroy(c->done);   // the stream belongs to the pool
    delete c;
}

int svslam_sync(svslam_ctx *c)
{
    HIPCHK(c, hipStreamSynchronize(c->stream));
    tm_collect(c);
    return 0;
}

int svslam_dev_alloc(svslam_ctx *c, size_t bytes, void **out) { HIPCHK(c, hipMalloc(out, bytes)); return 0; }
int svslam_dev_free(svslam_ctx *c, void *p) { HIPCHK(c, hipFree(p)); return 0; }
int svslam_dev_upload(svslam_ctx *c, void *dst, const void *src, size_t bytes)
{
    HIPCHK(c, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
int svslam_dev_download(svslam_ctx *c, void *dst, const void *src, size_t bytes)
{
    HIPCHK(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int svslam_timing_enable(svslam_ctx *c, int on) { c->timing = on != 0; return 0; }
int svslam_timing_reset(svslam_ctx *c) { c->tm = Timing(); return 0; }
int svslam_timing_get(svslam_ctx *c, int family, double *total_ms, long long *launches, long long *units)
{
    if (family < 0 || family >= FAM_COUNT) return fail(c, "bad timing family %d", family);
    if (total_ms) *total_ms = c->tm.ms[family];
    if (launches) *launches = c->tm.launches[family];
    if (units) *units = c->tm.units[family];
    return 0;
}

// ------------------------------------------------------------------ pyramids
int svslam_pyramid_batch(svslam_ctx *c, int n, const int *slots, const void *const *imgs,
                         const int *strides, int src_is_device)
{
    const bool dec = c->src_w > 0;
    return pyramid_common(c, n, slots, imgs, strides, src_is_device, dec, dec ? c->src_w : c->geom.w[0],
                          dec ? c->src_h : c->geom.h[0], true);
}

// co-resident workgroups of kernel `f` the device can hold: occupancy per CU x CUs
static int ll_resident_blocks(const void *f, size_t lds, int cus)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, BA_THREADS, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return nb * cus;
}
static void ll_release(svslam_ctx *c)
{
    if (c->ll.shards) (void)hipFree(c->ll.shards);
    if (c->ll.xch) (void)hipFree(c->ll.xch);
    if (c->ll.cnt) (void)hipFree(c->ll.cnt);
    ba_work_free(c->ll.bw);
    c->ll.shards = nullptr; c->ll.xch = nullptr; c->ll.cnt = nullptr; c->ll.w = 0; c->ll.max_problems = 0;
}

double svslam_get_pose_only_xtol(const svslam_ctx *c) { return c ? c->po_xtol : -1.0; }

int svslam_set_pose_only_xtol(svslam_ctx *c, double xtol)
{
    if (!(xtol >= 0) || xtol > 1e-6) return fail(c, "svslam_set_pose_only_xtol: 0 <= xtol <= 1e-6 (got %g)", xtol);
    c->po_xtol = xtol;
    return 0;
}

int svslam_set_low_latency(svslam_ctx *c, int on)
{
    c->low_latency = on != 0;
    // a caller that waits for one camera's frame wants the result, not its core back: block in the
    // runtime instead of sleep-polling the event (SVSLAM_WAIT=poll|spin still overrides)
    if (!std::getenv("SVSLAM_WAIT")) c->wait_poll = !c->low_latency;
    { const char *z = std::getenv("SVSLAM_ZERO_COPY"); c->zero_copy = c->low_latency && !(z && atoi(z) == 0); }
    // one local-BA problem over several workgroups (k_ba_ll / k_local_ba_t<2>): shard descriptors, exchange area, arrival
    // counters and the per-shard solver scratch, once.  SVSLAM_LL_SHARDS = 4 | 8 | 16 (default); 0 keeps one workgroup per problem.
    if (c->low_latency && c->ll.w == 0 && c->lim.max_kf > 0 && !c->ba_host_build && ba_tile_cap_ll(c->lim.max_kf) >= std::max(c->lim.max_kf, 64)) {
        const char *e = std::getenv("SVSLAM_LL_SHARDS");
        int w = e ? atoi(e) : 16;
        if (w != 0 && w != 4 && w != 8 && w != 16) return fail(c, "SVSLAM_LL_SHARDS=%d (4, 8, 16 or 0)", w);
        if (w > 0) {
            // the resident kernel (SVSLAM_LL_RESIDENT=0: every problem through the streaming kernel, A/B and its tests)
            const char *er = std::getenv("SVSLAM_LL_RESIDENT");
            const LlCaps caps = (er && atoi(er) == 0) ? LlCaps{ 0, 0, 0 } : ba_ll_caps(c->lim.max_kf);
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_ba_split), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)ba_split_lds_bytes(c->lim.max_lm, LL_MAX_W)));
            if (caps.B > 0)
                for (const void *f : { reinterpret_cast<const void *>(k_ba_ll<4, false>), reinterpret_cast<const void *>(k_ba_ll<8, false>),
                                       reinterpret_cast<const void *>(k_ba_ll<16, false>), reinterpret_cast<const void *>(k_ba_ll<4, true>),
                                       reinterpret_cast<const void *>(k_ba_ll<8, true>), reinterpret_cast<const void *>(k_ba_ll<16, true>) })
                    HIPCHK(c, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ba_ll_lds_bytes(c->lim.max_kf, caps)));
            // Guard of the in-launch barriers: every shard of every problem of a launch must be resident at
            // the same time.  CUs of the device (SVSLAM_LL_CUS overrides: a CU-masked process, a partitioned device whose
            // property still reports the whole chip) x the occupancy of the two solver kernels at their LDS sizes; if not even
            // one problem fits at w shards the shard count is halved, and without a fit the batch solver keeps the problems.
            int cus = 0;
            HIPCHK(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            if (const char *ec = std::getenv("SVSLAM_LL_CUS")) { const int v = atoi(ec); if (v > 0) cus = std::min(cus, v); }
            int maxp = 0, per_cu = 0;
            for (; w >= 4; w /= 2) {
                // every kernel a launch at this shard count may use: both extrinsic instantiations of the resident solver have their
                // own register count (the reference's rig launches <W, true>), the streaming solver has one
                const void *fr = w == 4 ? reinterpret_cast<const void *>(k_ba_ll<4, false>) : w == 8 ? reinterpret_cast<const void *>(k_ba_ll<8, false>) : reinterpret_cast<const void *>(k_ba_ll<16, false>);
                const void *fe = w == 4 ? reinterpret_cast<const void *>(k_ba_ll<4, true>) : w == 8 ? reinterpret_cast<const void *>(k_ba_ll<8, true>) : reinterpret_cast<const void *>(k_ba_ll<16, true>);
                const void *fs = w == 4 ? reinterpret_cast<const void *>(k_local_ba_t<2, 4>) : w == 8 ? reinterpret_cast<const void *>(k_local_ba_t<2, 8>) : reinterpret_cast<const void *>(k_local_ba_t<2, 16>);
                int blocks = ll_resident_blocks(fs, ba_lds_bytes_ll(c->lim.max_kf), cus);
                if (caps.B > 0)
                    for (const void *f : { fr, fe }) blocks = std::min(blocks, ll_resident_blocks(f, ba_ll_lds_bytes(c->lim.max_kf, caps), cus));
                per_cu = cus > 0 ? blocks / cus : 0;
                maxp = std::min(SVSLAM_LL_MAX_PROBLEMS, blocks / w);
                if (maxp >= 1) break;
            }
            c->ll.cus = cus; c->ll.blocks_per_cu = per_cu;
            if (maxp < 1) return 0;               // no shard count fits: one workgroup per problem (the batch solver)
            const size_t nsh = (size_t)SVSLAM_LL_MAX_PROBLEMS * w;
            c->ll.xch_stride = ll_xch_doubles(6 * c->lim.max_kf, w);
            hipError_t er_ = hipMalloc(&c->ll.shards, sizeof(BaDev) * nsh);
            if (er_ == hipSuccess) er_ = hipMalloc(&c->ll.xch, sizeof(double) * c->ll.xch_stride * SVSLAM_LL_MAX_PROBLEMS);
            if (er_ == hipSuccess) er_ = hipMalloc(&c->ll.cnt, sizeof(unsigned int) * LL_CNT_WORDS * SVSLAM_LL_MAX_PROBLEMS);
            if (er_ == hipSuccess) er_ = hipMemset(c->ll.xch, 0, sizeof(double) * c->ll.xch_stride * SVSLAM_LL_MAX_PROBLEMS);
            if (er_ == hipSuccess) er_ = hipMemset(c->ll.cnt, 0, sizeof(unsigned int) * LL_CNT_WORDS * SVSLAM_LL_MAX_PROBLEMS);
            if (er_ == hipSuccess) er_ = ba_work_alloc(c->ll.bw, (int)nsh, c->lim.max_kf, c->lim.max_lm, c->lim.max_obs);
            if (er_ != hipSuccess) {              // nothing half-built stays behind: a later call starts over
                ll_release(c);
                return fail(c, "low-latency BA workspace allocation failed: %s", hipGetErrorString(er_));
            }
            c->ll.caps = caps;
            c->ll.w = w;
            c->ll.max_problems = maxp;
            if (const char *et = std::getenv("SVSLAM_LL_TEST_DROP_SHARD")) c->ll.test_drop = atoi(et);   // test hook, see launch_ba_solver
            if (const char *eu = std::getenv("SVSLAM_LL_TIMEOUT_US")) { const long long us = atoll(eu); if (us >= 50 && us <= 5000000) c->ll.timeout_ticks = us * 100; }
            if (const char *ecp = std::getenv("SVSLAM_LL_COOP")) c->ll.coop = atoi(ecp) != 0;
        }
    }
    return 0;
}

int svslam_set_source_size(svslam_ctx *c, int src_w, int src_h)
{
    if (src_w <= 0 || src_h <= 0) { c->src_w = c->src_h = 0; return 0; }
    const int dw = (int)std::nearbyint(src_w * 0.5), dh = (int)std::nearbyint(src_h * 0.5);   // cvRound
    if (dw != c->geom.w[0] || dh != c->geom.h[0])
        return fail(c, "source %dx%d halves to %dx%d, context is %dx%d", src_w, src_h, dw, dh, c->geom.w[0], c->geom.h[0]);
    c->src_w = src_w; c->src_h = src_h;
    return 0;
}

int svslam_pyramid_decimate_batch(svslam_ctx *c, int n, const int *slots, const void *const *imgs,
                                  const int *strides, int src_w, int src_h, int src_is_device)
{
    // cvRound(src*0.5): round half to even
    int dw = (int)std::nearbyint(src_w * 0.5), dh = (int)std::nearbyint(src_h * 0.5);
    if (dw != c->geom.w[0] || dh != c->geom.h[0])
        return fail(c, "decimate: source %dx%d halves to %dx%d, context is %dx%d", src_w, src_h, dw, dh,
                    c->geom.w[0], c->geom.h[0]);
    return pyramid_common(c, n, slots, imgs, strides, src_is_device, true, src_w, src_h, true);
}

int svslam_pyramid_read(svslam_ctx *c, int slot, int level, uint8_t *out, int *w, int *h)
{
    if (check_slot(c, slot)) return -1;
    if (level < 0 || level >= c->geom.nlevels) return fail(c, "level %d out of range", level);
    const PyrGeom &g = c->geom;
    if (w) *w = g.w[level];
    if (h) *h = g.h[level];
    if (!out) return 0;
    const uint8_t *src = c->d_pyr + (size_t)slot * g.slot_bytes + g.ofs[level] + (size_t)SVS_BORDER * g.pitch[level] + SVS_BORDER;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy2D(out, g.w[level], src, g.pitch[level], g.w[level], g.h[level], hipMemcpyDeviceToHost));
    return 0;
}

// test hook: one level WITH its stored 16-px REFLECT_101 border, tight rows of (w + 32) bytes
int svslam_pyramid_read_padded(svslam_ctx *c, int slot, int level, uint8_t *out)
{
    if (check_slot(c, slot)) return -1;
    if (level < 0 || level >= c->geom.nlevels) return fail(c, "level %d out of range", level);
    const PyrGeom &g = c->geom;
    const int pw = g.w[level] + 2 * SVS_BORDER, ph = g.h[level] + 2 * SVS_BORDER;
    const uint8_t *src = c->d_pyr + (size_t)slot * g.slot_bytes + g.ofs[level];
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy2D(out, pw, src, g.pitch[level], pw, ph, hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------ LK
static LkParams make_lk_params(const svslam_lk_params *p)
{
    LkParams k;
    k.max_level = p ? p->max_level : 3;
    int mc = p ? p->max_iter : 30;
    k.max_count = std::min(std::max(mc, 0), 100);
    double eps = p ? p->epsilon : 0.01;
    eps = std::min(std::max(eps, 0.), 10.);
    k.eps2 = eps * eps;
    k.min_eig_thr = p ? p->min_eig_thr : 1e-4;
    k.use_initial_flow = p ? p->use_initial_flow : 1;
    // k_lk tests the min eigenvalue as numerator < x* instead of (double)(numerator / 242.f) < thr: x* is the
    // smallest float whose quotient reaches the threshold (float division is monotone), found by stepping
    const float den = (float)(2 * LK_WIN * LK_WIN);
    float x = (float)(k.min_eig_thr * (double)den);
    int steps = 0;
    bool ok = std::isfinite(x) && k.min_eig_thr > 1e-30;
    while (ok && (double)(x / den) >= k.min_eig_thr) { x = std::nextafterf(x, -INFINITY); if (++steps > 64) ok = false; }
    while (ok && (double)(x / den) < k.min_eig_thr) { x = std::nextafterf(x, INFINITY); if (++steps > 128) ok = false; }
    k.eig_num_thr = x;
    k.eig_use_div = ok ? 0 : 1;
    k.njobs = 0; k.blocks_per_job = 1;
    return k;
}

static void launch_lk(svslam_ctx *c, int njobs, int maxn, const LkJob *jobs, const float2 *prev, float2 *next, uint8_t *stat,
                      float *err, const svslam_lk_params *p)
{
    LkParams k = make_lk_params(p);
    k.njobs = njobs; k.blocks_per_job = cdiv(maxn, LK_WAVES_PER_BLOCK);
    hipLaunchKernelGGL(k_lk, dim3(8 * k.blocks_per_job * cdiv(njobs, 8)), dim3(64 * LK_WAVES_PER_BLOCK), 0, c->stream, jobs,
                       c->d_pyr, c->geom, prev, next, stat, err, k);
}

int svslam_lk_batch(svslam_ctx *c, int njobs, const svslam_lk_job *jobs, int total_pts,
                    const float *prev_xy, float *next_xy, uint8_t *status, float *err,
                    const svslam_lk_params *p)
{
    if (njobs <= 0) return 0;
    if (njobs > c->lim.max_jobs) return fail(c, "lk: %d jobs > max_jobs", njobs);
    if (total_pts > c->lim.max_jobs * c->lim.max_pts) return fail(c, "lk: too many points");
    int maxn = 0;
    for (int i = 0; i < njobs; ++i) {
        if (check_slot(c, jobs[i].prev_slot) || check_slot(c, jobs[i].next_slot)) return -1;
        if (jobs[i].npts < 0 || jobs[i].pt_ofs < 0 || jobs[i].pt_ofs + jobs[i].npts > total_pts)
            return fail(c, "lk: job %d point range out of bounds", i);
        maxn = std::max(maxn, jobs[i].npts);
    }
    if (arena_busy(c)) return -1;
    c->ar.reset();
    size_t ojobs = c->ar.take(sizeof(LkJob) * njobs);
    size_t oprev = c->ar.take(sizeof(float) * 2 * total_pts);
    size_t onext = c->ar.take(sizeof(float) * 2 * total_pts);
    size_t in_end = c->ar.off;
    size_t ostat = c->ar.take(total_pts);
    size_t oerr = c->ar.take(sizeof(float) * total_pts);
    static_assert(sizeof(LkJob) == sizeof(svslam_lk_job), "job layout");
    memcpy(hp<void>(c, ojobs), jobs, sizeof(LkJob) * njobs);
    memcpy(hp<void>(c, oprev), prev_xy, sizeof(float) * 2 * total_pts);
    memcpy(hp<void>(c, onext), next_xy, sizeof(float) * 2 * total_pts);
    if (h2d(c, 0, in_end)) return -1;
    if (maxn > 0) {
        tm_begin(c, FAM_LK, total_pts);
        launch_lk(c, njobs, maxn, dp<LkJob>(c, ojobs), dp<float2>(c, oprev), dp<float2>(c, onext), dp<uint8_t>(c, ostat),
                  dp<float>(c, oerr), p);
        tm_end(c);
        HIPCHK(c, hipGetLastError());
    }
    if (d2h_sync(c, onext, c->ar.off)) return -1;
    memcpy(next_xy, hp<void>(c, onext), sizeof(float) * 2 * total_pts);
    memcpy(status, hp<void>(c, ostat), total_pts);
    if (err) memcpy(err, hp<void>(c, oerr), sizeof(float) * total_pts);
    return 0;
}

// ------------------------------------------------------------------ GFTT
static int launch_gftt(svslam_ctx *c, int njobs, const GfttJob *djobs, const float2 *drects,
                       int max_corners, double quality, double min_dist, float2 *dout, int *dn)
{
    const int w = c->geom.w[0], h = c->geom.h[0];
    tm_begin(c, c->timing_split ? FAM_DBG0 : FAM_GFTT, njobs);
    // one-dimensional grid, jobs dealt over the XCDs (see LkParams): the strips of an image share their halo
    // columns, rows and the 64-byte lines they straddle through one L2
    hipLaunchKernelGGL(k_gftt_eig3<false>, dim3(8 * cdiv(w, GE_COLS) * cdiv(h, GE_ROWS) * cdiv(njobs, 8)), dim3(64), 0, c->stream, djobs,
                       njobs, c->d_pyr, c->geom, c->gw, drects, quality, (float *)nullptr);
    if (c->timing_split) { tm_end(c); tm_begin(c, FAM_DBG1, njobs); }
    hipLaunchKernelGGL(k_gftt_select2, dim3(njobs), dim3(GS_THREADS), 0, c->stream, c->gw, w, h, max_corners, quality,
                       min_dist, dout, dn, max_corners);
    tm_end(c);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int svslam_gftt_batch(svslam_ctx *c, int njobs, const svslam_gftt_job *jobs, int total_rects,
                      const float *rect_xy, int max_corners, double quality, double min_dist,
                      float *out_xy, int *out_n)
{
    if (njobs <= 0) return 0;
    if (njobs > c->lim.max_jobs) return fail(c, "gftt: %d jobs > max_jobs", njobs);
    if (max_corners < 1 || max_corners > c->lim.max_corners) return fail(c, "gftt: max_corners %d out of [1,%d]", max_corners, c->lim.max_corners);
    if (total_rects > c->lim.max_jobs * c->lim.max_pts) return fail(c, "gftt: too many mask rects");
    for (int i = 0; i < njobs; ++i) {
        if (check_slot(c, jobs[i].slot)) return -1;
        if (jobs[i].nrect < 0 || jobs[i].rect_ofs < 0 || jobs[i].rect_ofs + jobs[i].nrect > total_rects)
            return fail(c, "gftt: job %d rect range out of bounds", i);
    }
    if (arena_busy(c)) return -1;
    c->ar.reset();
    size_t ojobs = c->ar.take(sizeof(GfttJob) * njobs);
    size_t orect = c->ar.take(sizeof(float) * 2 * std::max(total_rects, 1));
    size_t in_end = c->ar.off;
    size_t oout = c->ar.take(sizeof(float) * 2 * (size_t)max_corners * njobs);
    size_t on = c->ar.take(sizeof(int) * njobs);
    static_assert(sizeof(GfttJob) == sizeof(svslam_gftt_job), "job layout");
    memcpy(hp<void>(c, ojobs), jobs, sizeof(GfttJob) * njobs);
    if (total_rects > 0) memcpy(hp<void>(c, orect), rect_xy, sizeof(float) * 2 * total_rects);
    if (h2d(c, 0, in_end)) return -1;
    if (launch_gftt(c, njobs, dp<GfttJob>(c, ojobs), dp<float2>(c, orect), max_corners, quality,
                    min_dist, dp<float2>(c, oout), dp<int>(c, on))) return -1;
    if (d2h_sync(c, oout, c->ar.off)) return -1;
    memcpy(out_n, hp<void>(c, on), sizeof(int) * njobs);
    memcpy(out_xy, hp<void>(c, oout), sizeof(float) * 2 * (size_t)max_corners * njobs);
    return 0;
}

int svslam_gftt_eigmap(svslam_ctx *c, int slot, float *out)
{
    if (check_slot(c, slot)) return -1;
    if (arena_busy(c)) return -1;
    c->ar.reset();
    size_t ojobs = c->ar.take(sizeof(GfttJob));
    GfttJob *j = hp<GfttJob>(c, ojobs);
    j->slot = slot; j->rect_ofs = 0; j->nrect = 0;
    if (h2d(c, 0, c->ar.off)) return -1;
    const int w = c->geom.w[0], h = c->geom.h[0];
    float *d_eig = nullptr;
    HIPCHK(c, hipMalloc(&d_eig, sizeof(float) * (size_t)w * h));
    // the production kernel with its eigenvalue store compiled in (the product instantiation differs by
    // exactly that store), so the eig-map parity tests check what ships
    hipLaunchKernelGGL(k_gftt_eig3<true>, dim3(8 * cdiv(w, GE_COLS) * cdiv(h, GE_ROWS)), dim3(64), 0, c->stream,
                       dp<GfttJob>(c, ojobs), 1, c->d_pyr, c->geom, c->gw, (const float2 *)nullptr, 0.01, d_eig);
    hipError_t e1 = hipGetLastError();
    hipError_t e2 = hipMemsetAsync(c->gw.counters, 0, sizeof(unsigned int) * GF_CNT_STRIDE, c->stream);   // job 0's counters back to zero
    hipError_t e3 = hipStreamSynchronize(c->stream);
    hipError_t e4 = hipMemcpy(out, d_eig, sizeof(float) * (size_t)w * h, hipMemcpyDeviceToHost);
    (void)hipFree(d_eig);
    HIPCHK(c, e1); HIPCHK(c, e2); HIPCHK(c, e3); HIPCHK(c, e4);
    return 0;
}

// ------------------------------------------------------------------ triangulation
int svslam_triangulate_batch(svslam_ctx *c, int njobs, const svslam_tri_job *jobs, int total_pts,
                             const double cam_l[4], const double ext_l[7], const double cam_r[4],
                             const double ext_r[7], const float *uv_l, const float *uv_r,
                             double *out_xyz, uint8_t *out_ok)
{
    if (njobs <= 0 || total_pts <= 0) return 0;
    if (njobs > c->lim.max_jobs) return fail(c, "triangulate: %d jobs > max_jobs", njobs);
    if (total_pts > c->lim.max_jobs * c->lim.max_pts) return fail(c, "triangulate: too many points");
    int maxn = 0;
    for (int i = 0; i < njobs; ++i) {
        if (jobs[i].npts < 0 || jobs[i].pt_ofs < 0 || jobs[i].pt_ofs + jobs[i].npts > total_pts)
            return fail(c, "triangulate: job %d point range out of bounds", i);
        maxn = std::max(maxn, jobs[i].npts);
    }
    if (arena_busy(c)) return -1;
    c->ar.reset();
    static_assert(sizeof(TriJob) == sizeof(svslam_tri_job), "job layout");
    size_t ojobs = c->ar.take(sizeof(TriJob) * njobs);
    size_t ol = c->ar.take(sizeof(float) * 2 * total_pts);
    size_t orr = c->ar.take(sizeof(float) * 2 * total_pts);
    size_t in_end = c->ar.off;
    size_t oxyz = c->ar.take(sizeof(double) * 3 * total_pts);
    size_t ook = c->ar.take(total_pts);
    memcpy(hp<void>(c, ojobs), jobs, sizeof(TriJob) * njobs);
    memcpy(hp<void>(c, ol), uv_l, sizeof(float) * 2 * total_pts);
    memcpy(hp<void>(c, orr), uv_r, sizeof(float) * 2 * total_pts);
    TriCams cams;
    memcpy(cams.cam_l, cam_l, 32); memcpy(cams.ext_l, ext_l, 56);
    memcpy(cams.cam_r, cam_r, 32); memcpy(cams.ext_r, ext_r, 56);
    if (h2d(c, 0, in_end)) return -1;
    if (maxn > 0) {
        tm_begin(c, FAM_TRI, total_pts);
        hipLaunchKernelGGL(k_triangulate, dim3(njobs, cdiv(maxn, 64)), dim3(64), 0, c->stream, dp<TriJob>(c, ojobs),       // (job-major grid: k_geom.h)
                           cams, dp<float2>(c, ol), dp<float2>(c, orr), dp<double>(c, oxyz), dp<uint8_t>(c, ook));
        tm_end(c);
        HIPCHK(c, hipGetLastError());
    }
    if (d2h_sync(c, oxyz, c->ar.off)) return -1;
    memcpy(out_xyz, hp<void>(c, oxyz), sizeof(double) * 3 * total_pts);
    memcpy(out_ok, hp<void>(c, ook), total_pts);
    return 0;
}

// ------------------------------------------------------------------ pose-only
int svslam_pose_only_batch(svslam_ctx *c, int njobs, svslam_pose_job *jobs, int total_pts,
                           const double cam[4], const double *xyz, const float *uv,
                           uint8_t *outlier, double chi2_th, int rounds, int iters)
{
    if (njobs <= 0) return 0;
    if (njobs > c->lim.max_jobs) return fail(c, "pose_only: %d jobs > max_jobs", njobs);
    if (total_pts > c->lim.max_jobs * c->lim.max_pts) return fail(c, "pose_only: too many points");
    for (int i = 0; i < njobs; ++i) {
        if (jobs[i].npts < 0 || jobs[i].npts > c->lim.max_pts || jobs[i].pt_ofs < 0 ||
            jobs[i].pt_ofs + jobs[i].npts > total_pts)
            return fail(c, "pose_only: job %d point range out of bounds", i);
    }
    if (arena_busy(c)) return -1;
    c->ar.reset();
    static_assert(sizeof(PoseJob) == sizeof(svslam_pose_job), "job layout");
    size_t ocam = c->ar.take(32);
    size_t oxyz = c->ar.take(sizeof(double) * 3 * std::max(total_pts, 1));
    size_t ouv = c->ar.take(sizeof(float) * 2 * std::max(total_pts, 1));
    size_t ojobs = c->ar.take(sizeof(PoseJob) * njobs);
    size_t in_end = c->ar.off;
    size_t oout = c->ar.take(std::max(total_pts, 1));
    memcpy(hp<void>(c, ocam), cam, 32);
    if (total_pts > 0) {
        memcpy(hp<void>(c, oxyz), xyz, sizeof(double) * 3 * total_pts);
        memcpy(hp<void>(c, ouv), uv, sizeof(float) * 2 * total_pts);
    }
    memcpy(hp<void>(c, ojobs), jobs, sizeof(PoseJob) * njobs);
    if (h2d(c, 0, in_end)) return -1;
    tm_begin(c, FAM_POSE, njobs);
    launch_pose_only(c, njobs, dp<PoseJob>(c, ojobs), dp<double>(c, ocam), dp<double>(c, oxyz), dp<float2>(c, ouv),
                     nullptr, dp<uint8_t>(c, oout), chi2_th, rounds, iters);
    tm_end(c);
    HIPCHK(c, hipGetLastError());
    if (d2h_sync(c, ojobs, c->ar.off)) return -1;
    memcpy(jobs, hp<void>(c, ojobs), sizeof(PoseJob) * njobs);
    if (total_pts > 0) memcpy(outlier, hp<void>(c, oout), total_pts);
    return 0;
}

// ------------------------------------------------------------------ local BA
int svslam_local_ba_submit(svslam_ctx *c, int njobs, const svslam_ba_job *jobs, const double cam_l[4],
                           const double ext_l[7], const double cam_r[4], const double ext_r[7],
                           int total_kf, const double *poses, int total_lm, const double *pts, int total_obs,
                           const int *obs_kf, const int *obs_lm, const uint8_t *obs_is_right,
                           const float *obs_uv, double huber_delta, int iters)
{
    if (njobs <= 0) return 0;
    if (njobs > c->lim.max_jobs) return fail(c, "local_ba: %d jobs > max_jobs", njobs);
    if (njobs > c->bw_jobs) return fail(c, "local_ba: %d problems in one call, this context's solver scratch holds %d (a context with the map on the device takes 64 per host-side call)", njobs, c->bw_jobs);
    // the solver scratch (c->bw, the low-latency exchange area, the LM trace) is one per context: a deferred local BA of the
    // device map that is still running on the context's second stream owns it
    if (c->dmba.inflight) return fail(c, "local_ba: a deferred local BA of the device map is in flight on this context: call svslam_dmap_ba_collect first");
    for (int i = 0; i < njobs; ++i) {
        const svslam_ba_job &j = jobs[i];
        if (j.nkf < 0 || j.nkf > c->lim.max_kf || j.nlm < 0 || j.nlm > c->lim.max_lm || j.nobs < 0 ||
            j.nobs > c->lim.max_obs || j.kf_ofs < 0 || j.kf_ofs + j.nkf > total_kf || j.lm_ofs < 0 ||
            j.lm_ofs + j.nlm > total_lm || j.obs_ofs < 0 || j.obs_ofs + j.nobs > total_obs)
            return fail(c, "local_ba: job %d exceeds limits (kf %d/%d lm %d/%d obs %d/%d)", i, j.nkf,
                        c->lim.max_kf, j.nlm, c->lim.max_lm, j.nobs, c->lim.max_obs);
    }
    const long long t_prep0 = now_ns();
    if (arena_busy(c)) return -1;
    c->ar.reset();
    c->ba_eid = ba_ext_identity(ext_l, ext_r);
    static_assert(sizeof(BaJob) == sizeof(svslam_ba_job), "job layout");
    const size_t TO = (size_t)std::max(total_obs, 1);
    const bool use_ll = !c->ba_host_build && ba_ll_usable(c, njobs);
    const int tile_cap = use_ll ? ba_ll_tile_cap(c) : ba_tile_cap(c->lim.max_kf);     // (reservations: the smaller tiles need more room)
    // arena: cams | jobs | poses | points | [raw edges + order (device build)] | chi2 + flag (out) |
    //        records | aux  (records and aux are device-only when the device builds the structure)
    size_t ocams = c->ar.take(sizeof(BaCams));
    size_t opk_o = 0, ouv_o = 0, osrt_o = 0;
    if (!c->ba_host_build) {
        opk_o = c->ar.take(sizeof(unsigned int) * TO); osrt_o = c->ar.take(sizeof(int) * TO);
        ouv_o = c->ar.take(sizeof(float) * 2 * TO);
    }
    size_t ojobs = c->ar.take(sizeof(BaDev) * njobs);          // read back from here ...
    size_t oposes = c->ar.take(sizeof(double) * 7 * std::max(total_kf, 1));
    size_t opts = c->ar.take(sizeof(double) * 3 * std::max(total_lm, 1));
    size_t in_end = c->ar.off;
    size_t ochi = c->ar.take(sizeof(double) * TO);
    size_t oflag = c->ar.take(sizeof(int) * 4);
    size_t out_end = c->ar.off;
    size_t orecs = c->ar.take(sizeof(BaRec) * 2 * TO);   // landmark-major + pose-major edge records
    size_t oaux = c->ar.take(0);
    if (oaux > c->ar.cap) return fail(c, "local_ba: staging arena too small (%zu > %zu bytes)", oaux, c->ar.cap);
    BaCams *cams = hp<BaCams>(c, ocams);
    memcpy(cams->cam[0], cam_l, 32); memcpy(cams->cam[1], cam_r, 32);
    memcpy(cams->ext[0], ext_l, 56); memcpy(cams->ext[1], ext_r, 56);
    BaDev *dj = hp<BaDev>(c, ojobs);
    hp<int>(c, oflag)[0] = 0;
    size_t aux_total = 0;
    int max_nlm = 1, max_nobs = 1;
    bool all_sorted = true;
    if (c->ba_host_build) {
        // Host-side structure of every problem (edge records, blocks, pose-pair lists), built by the
        // pool with one scratch structure per thread (stays cache-hot) and written straight into the
        // arena; the aux space of a problem comes from a bump allocator, so its position depends on
        // thread timing but nothing else does (every problem carries its own offsets).
        std::vector<int> bad((size_t)njobs, 0);
        std::atomic<size_t> bump{ 0 };
        const size_t aux_cap_ints = (c->ar.cap - oaux) / sizeof(int);
        int *aux = hp<int>(c, oaux);
        BaRec *recs = hp<BaRec>(c, orecs);
        auto build_one = [&](int i) {
            static thread_local BaHostStruct hs;
            BaJob bj;
            memcpy(&bj, &jobs[i], sizeof(bj));
            if (!hs.build(bj, obs_kf, obs_lm, obs_is_right, obs_uv, tile_cap)) { bad[(size_t)i] = 1; return; }   // index out of range
            const size_t need = hs.aux_ints(bj);
            const size_t at = bump.fetch_add(need);
            if (at + need > aux_cap_ints) { bad[(size_t)i] = 2; return; }
            hs.write(bj, aux + at, recs + 2 * (size_t)bj.obs_ofs, dj[i]);
            dj[i].aux_ofs = (int)at;
            dj[i].rec_ofs = 2 * bj.obs_ofs;
        };
        if (c->pool && njobs > 1) c->pool->parallel_for(njobs, build_one);
        else for (int i = 0; i < njobs; ++i) build_one(i);
        for (int i = 0; i < njobs; ++i) {
            if (bad[(size_t)i] == 1) return fail(c, "local_ba: job %d has an edge index out of range", i);
            if (bad[(size_t)i] == 2) return fail(c, "local_ba: staging arena too small for the problem structure");
        }
        aux_total = bump.load();
        (void)c->ar.take(sizeof(int) * aux_total);
    } else {
        // The device builds the structure (k_ba_build).  One pass of the pool over the problems validates the
        // edge indices, packs them (landmark | keyframe << 16 | camera << 24: 4 bytes per edge instead of 9),
        // notes whether they arrive in (landmark, keyframe) order — the order the backend gathers them in
        // (src/backend.cpp:83-160); a stable sort otherwise — and copies the problem's measurements, poses and
        // positions into the staging arena: every input byte is touched once, by the thread that has it in cache.
        std::vector<int> bad((size_t)njobs, 0);
        int *srt = hp<int>(c, osrt_o);
        unsigned int *pk = hp<unsigned int>(c, opk_o);
        float *auv = hp<float>(c, ouv_o);
        double *aposes = hp<double>(c, oposes), *apts = hp<double>(c, opts);
        auto check_one = [&](int i) {
            const svslam_ba_job &j = jobs[i];
            const int *kf = obs_kf + j.obs_ofs, *lm = obs_lm + j.obs_ofs;
            const uint8_t *rt = obs_is_right + j.obs_ofs;
            unsigned int *pj = pk + j.obs_ofs;
            bool sorted = true;
            for (int e = 0; e < j.nobs; ++e) {
                const int k = kf[e], l = lm[e];
                if (k < 0 || k >= j.nkf || l < 0 || l >= j.nlm) { bad[(size_t)i] = 1; return; }
                if (e && !((lm[e - 1] < l) || (lm[e - 1] == l && kf[e - 1] <= k))) sorted = false;
                pj[e] = (unsigned int)l | ((unsigned int)k << 16) | ((rt[e] ? 1u : 0u) << 24);
            }
            if (!sorted) {
                int *sr = srt + j.obs_ofs;
                for (int e = 0; e < j.nobs; ++e) sr[e] = e;
                std::stable_sort(sr, sr + j.nobs, [&](int a, int b) { return lm[a] != lm[b] ? lm[a] < lm[b] : kf[a] < kf[b]; });
            }
            if (j.nobs) memcpy(auv + 2 * (size_t)j.obs_ofs, obs_uv + 2 * (size_t)j.obs_ofs, sizeof(float) * 2 * j.nobs);
            if (j.nkf) memcpy(aposes + 7 * (size_t)j.kf_ofs, poses + 7 * (size_t)j.kf_ofs, sizeof(double) * 7 * j.nkf);
            if (j.nlm) memcpy(apts + 3 * (size_t)j.lm_ofs, pts + 3 * (size_t)j.lm_ofs, sizeof(double) * 3 * j.nlm);
            dj[i].reserved = sorted ? 1 : 0;
        };
        if (c->pool && njobs > 1) c->pool->parallel_for(njobs, check_one);
        else for (int i = 0; i < njobs; ++i) check_one(i);
        size_t at = 0;
        for (int i = 0; i < njobs; ++i) {
            if (bad[(size_t)i]) return fail(c, "local_ba: job %d has an edge index out of range", i);
            const svslam_ba_job &j = jobs[i];
            BaDev &d = dj[i];
            d.kf_ofs = j.kf_ofs; d.nkf = j.nkf; d.lm_ofs = j.lm_ofs; d.nlm = j.nlm; d.obs_ofs = j.obs_ofs; d.nobs = j.nobs;
            d.nblk = d.na = d.ncontrib = d.ntile = 0; d.iters_done = 0; d.nmv = 0;
            d.rec_ofs = 2 * j.obs_ofs;
            d.lay_nblk = j.nobs; d.lay_na = j.nkf; d.lay_ntile = ba_tile_bound(j.nlm, j.nobs, j.nkf, tile_cap);
            d.aux_ofs = (int)at; d.lm_base = 0; d.shmask = 0; d.ntrial = 0;
            at += ba_aux_layout(j.nkf, j.nlm, j.nobs, d.lay_nblk, d.lay_na, 0, d.lay_ntile).total + ba_pitem_bound(j.nobs, j.nkf);
            if (use_ll) at += ba_split_aux_extra(j.nkf, c->ll.w);
            if (!d.reserved) all_sorted = false;
            max_nlm = std::max(max_nlm, j.nlm); max_nobs = std::max(max_nobs, j.nobs);
        }
        aux_total = at;
        (void)c->ar.take(sizeof(int) * aux_total);
        if (c->ar.off > c->ar.cap) return fail(c, "local_ba: staging arena too small for the problem structure (%zu > %zu bytes)", c->ar.off, c->ar.cap);
    }
    if (c->ba_host_build) {
        if (total_kf > 0) memcpy(hp<void>(c, oposes), poses, sizeof(double) * 7 * total_kf);
        if (total_lm > 0) memcpy(hp<void>(c, opts), pts, sizeof(double) * 3 * total_lm);
    }
    c->host_ns[4] += now_ns() - t_prep0;
    if (h2d(c, 0, in_end)) return -1;
    if (h2d(c, oflag, oflag + sizeof(int) * 4)) return -1;
    if (c->ba_host_build) { if (h2d(c, orecs, c->ar.off)) return -1; }
    if (c->d_lm_trace) HIPCHK(c, hipMemsetAsync(c->d_lm_trace, 0, sizeof(double) * LM_TRACE_STRIDE * (size_t)njobs, c->stream));
    tm_begin(c, c->timing_split ? FAM_DBG2 : FAM_BA, njobs);
    if (!c->ba_host_build)
        launch_ba_solver(c, njobs, use_ll && all_sorted, dp<BaDev>(c, ojobs), dp<BaCams>(c, ocams), dp<double>(c, oposes), dp<double>(c, opts),
                         dp<unsigned int>(c, opk_o), dp<float2>(c, ouv_o), dp<int>(c, osrt_o), dp<BaRec>(c, orecs), dp<int>(c, oaux),
                         dp<double>(c, ochi), dp<int>(c, oflag), max_nlm, max_nobs, huber_delta, iters, c->timing_split);
    else
        hipLaunchKernelGGL((k_local_ba_t<0, 1, false>), dim3(njobs), dim3(BA_THREADS), ba_lds_bytes(c->lim.max_kf), c->stream,
                           dp<BaDev>(c, ojobs), dp<BaCams>(c, ocams), dp<double>(c, oposes), dp<double>(c, opts),
                           dp<BaRec>(c, orecs), dp<int>(c, oaux), c->bw, huber_delta, iters, dp<double>(c, ochi), c->d_ba_prof,
                           tile_cap, SbaArgs{ 0, 0, 0.0, nullptr, c->d_lm_trace, 0, nullptr, nullptr, nullptr, 0, 0 });
    tm_end(c);
    HIPCHK(c, hipGetLastError());
    c->ba_pending.oflag = oflag;
    if (d2h_enqueue(c, ojobs, out_end)) return -1;
    c->ba_pending.active = true; c->ba_pending.njobs = njobs;
    c->ba_pending.total_kf = total_kf; c->ba_pending.total_lm = total_lm; c->ba_pending.total_obs = total_obs;
    c->ba_pending.ojobs = ojobs; c->ba_pending.oposes = oposes; c->ba_pending.opts = opts; c->ba_pending.ochi = ochi;
    c->ba_pending.ll_used = !c->ba_host_build && use_ll && all_sorted;
    if (c->ba_pending.ll_used) {
        c->ba_pending.ocams = ocams; c->ba_pending.opk = opk_o; c->ba_pending.ouv = ouv_o; c->ba_pending.osrt = osrt_o; c->ba_pending.orecs = orecs;
        c->ba_pending.oaux = oaux; c->ba_pending.out_end = out_end; c->ba_pending.max_nlm = max_nlm; c->ba_pending.max_nobs = max_nobs;
        c->ba_pending.iters = iters; c->ba_pending.delta = huber_delta;
        // the descriptors as uploaded, re-tiled for the batch solver (larger tiles: the aux room reserved for the smaller ones holds them)
        c->ll.saved.assign(dj, dj + njobs);        // (dj: the host arena; the read-back below overwrites it)
        for (int i = 0; i < njobs; ++i) c->ll.saved[(size_t)i].lay_ntile = ba_tile_bound(jobs[i].nlm, jobs[i].nobs, jobs[i].nkf, ba_tile_cap(c->lim.max_kf));
    }
    return 0;
}

int svslam_local_ba_collect(svslam_ctx *c, int njobs, svslam_ba_job *jobs, int total_kf, double *poses,
                            int total_lm, double *pts, int total_obs, double *edge_chi2)
{
    if (!c->ba_pending.active) return njobs <= 0 ? 0 : fail(c, "local_ba_collect: nothing submitted");
    if (njobs != c->ba_pending.njobs || total_kf != c->ba_pending.total_kf || total_lm != c->ba_pending.total_lm ||
        total_obs != c->ba_pending.total_obs)
        return fail(c, "local_ba_collect: sizes differ from the submitted batch");
    c->ba_pending.active = false;
    if (finish(c)) return -1;
    if (const int fl = hp<int>(c, c->ba_pending.oflag)[0])
        return fail(c, "local_ba: the device structure build overflowed a capacity (code %d)", fl);
    const BaDev *dj = hp<BaDev>(c, c->ba_pending.ojobs);
    std::vector<int> bad;
    for (int i = 0; i < njobs; ++i) if (dj[i].iters_done < 0) bad.push_back(i);
    std::vector<BaDev> res(dj, dj + njobs);        // descriptors as the first solve returned them
    if (!bad.empty()) {
        // a shard of the low-latency solver never became resident.  A problem that gives up writes nothing back, so ITS inputs are
        // untouched in the device arena; the problems that finished have their results there already (poses and positions are
        // updated in place) and must not be solved a second time (20 LM iterations instead of the reference's 10).
        // Only the problems that gave up go to the batch solver — one workgroup per problem, no barrier — as a compacted list:
        // every descriptor carries its own offsets into the arena.
        if (!c->ba_pending.ll_used || (int)c->ll.saved.size() != njobs) return fail(c, "local_ba: a problem reports iters_done < 0 outside the low-latency solver");
        const auto &bp = c->ba_pending;
        const int nb = (int)bad.size();
        BaDev *hj = hp<BaDev>(c, bp.ojobs);
        int max_nlm = 1, max_nobs = 1;
        for (int k = 0; k < nb; ++k) {
            hj[k] = c->ll.saved[(size_t)bad[(size_t)k]];
            max_nlm = std::max(max_nlm, hj[k].nlm); max_nobs = std::max(max_nobs, hj[k].nobs);
        }
        hp<int>(c, bp.oflag)[0] = 0;
        if (h2d(c, bp.ojobs, bp.ojobs + sizeof(BaDev) * (size_t)nb)) return -1;
        if (h2d(c, bp.oflag, bp.oflag + sizeof(int) * 4)) return -1;
        launch_ba_solver(c, nb, false, dp<BaDev>(c, bp.ojobs), dp<BaCams>(c, bp.ocams), dp<double>(c, bp.oposes), dp<double>(c, bp.opts),
                         dp<unsigned int>(c, bp.opk), dp<float2>(c, bp.ouv), dp<int>(c, bp.osrt), dp<BaRec>(c, bp.orecs), dp<int>(c, bp.oaux),
                         dp<double>(c, bp.ochi), dp<int>(c, bp.oflag), max_nlm, max_nobs, bp.delta, bp.iters, false);
        HIPCHK(c, hipGetLastError());
        if (d2h_sync(c, bp.ojobs, bp.out_end)) return -1;
        if (const int fl = hp<int>(c, bp.oflag)[0]) return fail(c, "local_ba: the device structure build overflowed a capacity (code %d)", fl);
        for (int k = 0; k < nb; ++k) res[(size_t)bad[(size_t)k]] = dj[k];
        c->ll.fallbacks += nb;
    }
    for (int i = 0; i < njobs; ++i) {
        const BaDev &d = res[(size_t)i];
        if (d.iters_done < 0) return fail(c, "local_ba: job %d reports iters_done < 0 from the batch solver", i);
        jobs[i].iters_done = d.iters_done;
        jobs[i].reserved = (int)(((unsigned)std::min(d.ntrial, 255) << 24) | ((unsigned)d.ncontrib & 0x00ffffffu));   // accounting (svslam.h)
    }
    if (total_kf > 0) memcpy(poses, hp<void>(c, c->ba_pending.oposes), sizeof(double) * 7 * total_kf);
    if (total_lm > 0) memcpy(pts, hp<void>(c, c->ba_pending.opts), sizeof(double) * 3 * total_lm);
    if (total_obs > 0) memcpy(edge_chi2, hp<void>(c, c->ba_pending.ochi), sizeof(double) * total_obs);
    return 0;
}

int svslam_local_ba_batch(svslam_ctx *c, int njobs, svslam_ba_job *jobs, const double cam_l[4],
                          const double ext_l[7], const double cam_r[4], const double ext_r[7],
                          int total_kf, double *poses, int total_lm, double *pts, int total_obs,
                          const int *obs_kf, const int *obs_lm, const uint8_t *obs_is_right,
                          const float *obs_uv, double huber_delta, int iters, double *edge_chi2)
{
    if (njobs <= 0) return 0;
    if (int rc = svslam_local_ba_submit(c, njobs, jobs, cam_l, ext_l, cam_r, ext_r, total_kf, poses, total_lm, pts,
                                        total_obs, obs_kf, obs_lm, obs_is_right, obs_uv, huber_delta, iters))
        return rc;
    return svslam_local_ba_collect(c, njobs, jobs, total_kf, poses, total_lm, pts, total_obs, edge_chi2);
}

// ------------------------------------------------------------------ shared-map BA (BASELINE config 5)
// One problem, its landmarks sharded over the ranks of a node: this rank opens its shard (all K poses, its
// landmarks and their edges), then drives the LM trial pieces of k_local_ba_t<1>; what has to be summed over
// the ranks goes through `io` (host memory; the caller all-reduces it with RCCL).  See k_ba.h:SbaArgs.
int svslam_sba_io_doubles(int nkf) { return (int)SBA_IO_DOUBLES(6 * nkf); }

int svslam_sba_open(svslam_ctx *c, const double cam_l[4], const double ext_l[7], const double cam_r[4], const double ext_r[7],
                    int nkf, const double *poses, int nlm, const double *pts, int nobs, const int *obs_kf, const int *obs_lm,
                    const uint8_t *obs_is_right, const float *obs_uv, double huber_delta)
{
    if (c->ba_pending.active || c->sba.open) return fail(c, "sba_open: the context already holds a BA batch");
    if (nkf < 1 || nkf > c->lim.max_kf || nlm < 1 || nlm > c->lim.max_lm || nobs < 1 || nobs > c->lim.max_obs)
        return fail(c, "sba_open: shard exceeds limits or is empty (kf %d/%d lm %d/%d obs %d/%d)", nkf, c->lim.max_kf, nlm, c->lim.max_lm, nobs, c->lim.max_obs);
    c->ar.reset();
    const int tile_cap = ba_tile_cap(c->lim.max_kf);
    const size_t TO = (size_t)nobs;
    size_t ocams = c->ar.take(sizeof(BaCams));
    size_t opk_o = c->ar.take(sizeof(unsigned int) * TO), osrt_o = c->ar.take(sizeof(int) * TO);
    size_t ouv_o = c->ar.take(sizeof(float) * 2 * TO);
    size_t ojobs = c->ar.take(sizeof(BaDev));
    size_t oposes = c->ar.take(sizeof(double) * 7 * nkf);
    size_t opts = c->ar.take(sizeof(double) * 3 * nlm);
    size_t in_end = c->ar.off;
    size_t ochi = c->ar.take(sizeof(double) * TO);
    size_t oflag = c->ar.take(sizeof(int) * 4);
    size_t oio = c->ar.take(sizeof(double) * SBA_IO_DOUBLES(6 * nkf));
    size_t orecs = c->ar.take(sizeof(BaRec) * 2 * TO);
    size_t oaux = c->ar.take(0);
    BaCams *cams = hp<BaCams>(c, ocams);
    memcpy(cams->cam[0], cam_l, 32); memcpy(cams->cam[1], cam_r, 32);
    memcpy(cams->ext[0], ext_l, 56); memcpy(cams->ext[1], ext_r, 56);
    int *srt = hp<int>(c, osrt_o);
    unsigned int *pk = hp<unsigned int>(c, opk_o);
    bool sorted = true;
    for (int e = 0; e < nobs; ++e) {
        const int k = obs_kf[e], l = obs_lm[e];
        if (k < 0 || k >= nkf || l < 0 || l >= nlm) return fail(c, "sba_open: edge %d index out of range", e);
        if (e && !((obs_lm[e - 1] < l) || (obs_lm[e - 1] == l && obs_kf[e - 1] <= k))) sorted = false;
        srt[e] = e;
        pk[e] = (unsigned int)l | ((unsigned int)k << 16) | ((obs_is_right[e] ? 1u : 0u) << 24);
    }
    if (!sorted) std::stable_sort(srt, srt + nobs, [&](int a, int b) { return obs_lm[a] != obs_lm[b] ? obs_lm[a] < obs_lm[b] : obs_kf[a] < obs_kf[b]; });
    BaDev &d = *hp<BaDev>(c, ojobs);
    d.kf_ofs = 0; d.nkf = nkf; d.lm_ofs = 0; d.nlm = nlm; d.obs_ofs = 0; d.nobs = nobs;
    d.nblk = d.na = d.ncontrib = d.ntile = 0; d.iters_done = 0; d.rec_ofs = 0; d.nmv = 0; d.reserved = sorted ? 1 : 0;
    d.lay_nblk = nobs; d.lay_na = nkf; d.lay_ntile = ba_tile_bound(nlm, nobs, nkf, tile_cap);
    d.aux_ofs = 0; d.lm_base = 0; d.shmask = 0; d.ntrial = 0;
    (void)c->ar.take(sizeof(int) * (ba_aux_layout(nkf, nlm, nobs, d.lay_nblk, d.lay_na, 0, d.lay_ntile).total + ba_pitem_bound(nobs, nkf)));
    if (c->ar.off > c->ar.cap) return fail(c, "sba_open: staging arena too small");
    memcpy(hp<void>(c, ouv_o), obs_uv, sizeof(float) * 2 * TO);
    memcpy(hp<void>(c, oposes), poses, sizeof(double) * 7 * nkf);
    memcpy(hp<void>(c, opts), pts, sizeof(double) * 3 * nlm);
    hp<int>(c, oflag)[0] = 0;
    if (h2d(c, 0, in_end)) return -1;
    if (h2d(c, oflag, oflag + sizeof(int) * 4)) return -1;
    hipLaunchKernelGGL(k_ba_build, dim3(1), dim3(BB_THREADS), bb_lds_bytes(nlm, nobs), c->stream, dp<BaDev>(c, ojobs), dp<unsigned int>(c, opk_o),
                       dp<float2>(c, ouv_o), dp<int>(c, osrt_o), dp<BaRec>(c, orecs),
                       dp<int>(c, oaux), tile_cap, nlm, dp<int>(c, oflag), 1 /* every keyframe active on every rank */,
                       bb_edge_cache_fits(nlm, nobs) ? 1 : 0, 0);
    HIPCHK(c, hipGetLastError());
    if (d2h_sync(c, oflag, oflag + sizeof(int) * 4)) return -1;
    if (hp<int>(c, oflag)[0]) return fail(c, "sba_open: the structure build overflowed a capacity");
    c->sba.open = true; c->sba.nkf = nkf; c->sba.nlm = nlm; c->sba.nobs = nobs; c->sba.np = 6 * nkf;
    c->sba.ojobs = ojobs; c->sba.ocams = ocams; c->sba.oposes = oposes; c->sba.opts = opts; c->sba.orecs = orecs; c->sba.oaux = oaux;
    c->sba.ochi = ochi; c->sba.oio = oio; c->sba.delta = huber_delta; c->sba.launches = 0;
    return 0;
}

// phase: 1 diagonal (lambda_0), 2 linearise at lambda, 3 solve with the reduced system in io + update + errors,
//        4 reject (restore), 5 finalise.  io: svslam_sba_io_doubles(nkf) doubles, in for phase 3, out for 1-3.
static int sba_launch(svslam_ctx *c, int phase, double lambda, int add_lambda)
{
    SbaArgs a{ phase, c->sba.launches == 0 ? 1 : 0, lambda, dp<double>(c, c->sba.oio), nullptr, add_lambda, nullptr, nullptr, nullptr, 0 };
    tm_begin(c, FAM_BA, 1);
    hipLaunchKernelGGL((k_local_ba_t<1, 1>), dim3(1), dim3(BA_THREADS), ba_lds_bytes(c->lim.max_kf), c->stream, dp<BaDev>(c, c->sba.ojobs),
                       dp<BaCams>(c, c->sba.ocams), dp<double>(c, c->sba.oposes), dp<double>(c, c->sba.opts), dp<BaRec>(c, c->sba.orecs),
                       dp<int>(c, c->sba.oaux), c->bw, c->sba.delta, 1, dp<double>(c, c->sba.ochi), (long long *)nullptr,
                       ba_tile_cap(c->lim.max_kf), a);
    tm_end(c);
    HIPCHK(c, hipGetLastError());
    c->sba.launches++;
    return 0;
}

int svslam_sba_phase(svslam_ctx *c, int phase, double lambda, double *io)
{
    if (!c->sba.open) return fail(c, "sba_phase: no shard is open");
    if (phase < 1 || phase > 5) return fail(c, "sba_phase: phase %d", phase);
    const size_t nio = SBA_IO_DOUBLES(c->sba.np) * sizeof(double);
    if (phase == 3) { memcpy(hp<void>(c, c->sba.oio), io, nio); if (h2d(c, c->sba.oio, c->sba.oio + nio)) return -1; }
    else HIPCHK(c, hipMemsetAsync(dp<void>(c, c->sba.oio), 0, nio, c->stream));
    if (sba_launch(c, phase, lambda, 0)) return -1;
    if (d2h_sync(c, c->sba.oio, c->sba.oio + nio)) return -1;
    if (io && phase <= 3) memcpy(io, hp<void>(c, c->sba.oio), nio);
    return 0;
}

// ---- the shared-map LM as a product path: the control flow of g2o's Levenberg-Marquardt (exactly the loop of
// k_local_ba_t<0> / shared_ba.py) in the library, the reduced camera system all-reduced IN PLACE on the device buffer
// by RCCL (ncclAllReduce on the context's stream: (6K)^2 + 3 (6K) + 1 doubles per trial, 3 781 at K = 10, plus two
// scalars), no host copy of the system, two launches and one 64-byte read-back per trial.
namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi *rccl_api(std::string *why)
{
    static RcclApi api;
    static std::once_flag once;
    static std::string err;
    std::call_once(once, [] {
        // RCCL must sit on the SAME HIP runtime as this library (a hipStream_t means nothing to another runtime), and
        // PyTorch-ROCm ships its own libamdhip64.so + librccl.so: in a process that imported torch first this library
        // is bound to torch's runtime, otherwise to the ROCm installation's.  So: the librccl next to the
        // libamdhip64 this library actually resolved (dladdr), then the generic names.  SVSLAM_RCCL_LIB overrides.
        if (const char *e = std::getenv("SVSLAM_RCCL_LIB")) api.lib = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
        Dl_info di;
        if (!api.lib && dladdr(reinterpret_cast<void *>(&hipStreamCreateWithFlags), &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t sl = dir.rfind('/');
            if (sl != std::string::npos) {
                dir.resize(sl + 1);
                for (const char *n : { "librccl.so.1", "librccl.so" }) {
                    api.lib = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_GLOBAL);
                    if (api.lib) break;
                }
            }
        }
        for (const char *n : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            if (api.lib) break;
            api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!api.lib) { err = std::string("librccl.so not found: ") + dlerror(); return; }
#define SVS_RCCL_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name)); if (!api.field) err = std::string("missing RCCL symbol ") + name
        SVS_RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); SVS_RCCL_SYM(CommInitRank, "ncclCommInitRank");
        SVS_RCCL_SYM(CommDestroy, "ncclCommDestroy"); SVS_RCCL_SYM(AllReduce, "ncclAllReduce");
        SVS_RCCL_SYM(GroupStart, "ncclGroupStart"); SVS_RCCL_SYM(GroupEnd, "ncclGroupEnd");
        SVS_RCCL_SYM(GetErrorString, "ncclGetErrorString"); SVS_RCCL_SYM(CommAbort, "ncclCommAbort");
#undef SVS_RCCL_SYM
    });
    if (!err.empty()) { if (why) *why = err; return nullptr; }
    return &api;
}
} // namespace

int svslam_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

// 128 bytes that rank 0 creates and hands to every rank (over any channel: torch.distributed, MPI, a file)
int svslam_sba_comm_unique_id(char out128[128])
{
    std::string why;
    RcclApi *r = rccl_api(&why);
    if (!r) return -1;
    ncclUniqueId id;
    static_assert(sizeof(id) == 128, "ncclUniqueId");
    if (r->GetUniqueId(&id) != ncclSuccess) return -2;
    memcpy(out128, &id, 128);
    return 0;
}
int svslam_sba_comm_init(svslam_ctx *c, int nranks, int rank, const char id128[128])
{
    if (c->sbac.comm) return fail(c, "sba_comm_init: the context already has a communicator");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(c, "sba_comm_init: rank %d of %d", rank, nranks);
    std::string why;
    RcclApi *r = rccl_api(&why);
    if (!r) return fail(c, "sba_comm_init: %s", why.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    const ncclResult_t rc = r->CommInitRank(&c->sbac.comm, nranks, id, rank);
    if (rc != ncclSuccess) { c->sbac.comm = nullptr; return fail(c, "ncclCommInitRank: %s", r->GetErrorString(rc)); }
    c->sbac.nranks = nranks; c->sbac.rank = rank;
    return 0;
}
int svslam_sba_comm_destroy(svslam_ctx *c)
{
    if (!c->sbac.comm) return 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    RcclApi *r = rccl_api(nullptr);
    if (r) (void)r->CommDestroy(c->sbac.comm);
    c->sbac.comm = nullptr; c->sbac.nranks = 1; c->sbac.rank = 0;
    return 0;
}

// The whole optimize(iters) of the open shard.  trace (optional): 6 doubles per LM trial as svslam_lm_trace;
// stats (optional, 4 doubles): trials, milliseconds in total, mean milliseconds per trial, all-reduced bytes per trial.
static int sba_solve_impl(svslam_ctx *c, int iters, int *iters_done, double *lambda_out, double *trace, int trace_cap,
                          int *n_trace, double *stats);
// A rank that fails inside the LM loop (launch, copy, collective) would leave its peers waiting in their next all-reduce:
// the communicator is aborted (ncclCommAbort), so the peers' collectives return with an error instead of hanging; the
// context needs a new communicator (svslam_sba_comm_init) before the next shared-map solve.
int svslam_sba_solve(svslam_ctx *c, int iters, int *iters_done, double *lambda_out, double *trace, int trace_cap,
                     int *n_trace, double *stats)
{
    const int rc = sba_solve_impl(c, iters, iters_done, lambda_out, trace, trace_cap, n_trace, stats);
    if (rc != 0 && c->sbac.comm) {
        RcclApi *r = rccl_api(nullptr);
        if (r && r->CommAbort) { (void)r->CommAbort(c->sbac.comm); c->sbac.comm = nullptr; c->err += " (shared-map communicator aborted)"; }
    }
    return rc;
}
static int sba_solve_impl(svslam_ctx *c, int iters, int *iters_done, double *lambda_out, double *trace, int trace_cap,
                          int *n_trace, double *stats)
{
    if (!c->sba.open) return fail(c, "sba_solve: no shard is open");
    RcclApi *r = c->sbac.comm ? rccl_api(nullptr) : nullptr;
    const int n = c->sba.np;
    const size_t oS = 0, ohd = (size_t)n * n + 2 * (size_t)n, osc = ohd + n;
    double *dio = dp<double>(c, c->sba.oio);
    double *hio = hp<double>(c, c->sba.oio);
    const size_t off_sc = c->sba.oio + osc * sizeof(double);
    auto allreduce = [&](double *buf, size_t cnt, ncclRedOp_t op) -> int {
        if (!r) return 0;                               // one rank, no communicator: the sums are the local values
        const ncclResult_t rc = r->AllReduce(buf, buf, cnt, ncclDouble, op, c->sbac.comm, c->stream);
        return rc == ncclSuccess ? 0 : fail(c, "ncclAllReduce: %s", r->GetErrorString(rc));
    };
    const long long t_begin = now_ns();
    HIPCHK(c, hipMemsetAsync(dio, 0, SBA_IO_DOUBLES(n) * sizeof(double), c->stream));
    double lam = 0, ni = 2, current = 0;
    bool have_current = false;
    int it_done = 0, ntr = 0, trials = 0;
    for (int it = 0; it < iters; ++it) {
        if (it == 0) {
            if (sba_launch(c, 1, 0.0, 0)) return -1;
            if (allreduce(dio + ohd, (size_t)n, ncclSum)) return -1;
            if (allreduce(dio + osc + 1, 1, ncclMax)) return -1;
            if (d2h_sync(c, c->sba.oio + ohd * sizeof(double), c->sba.oio + (osc + 8) * sizeof(double))) return -1;
            double md = hio[osc + 1];
            for (int i = 0; i < n; ++i) md = std::max(md, std::fabs(hio[ohd + i]));
            lam = 1e-5 * md; ni = 2;
            // the diagonal sits inside the range every trial all-reduces (S | bs | bp | hd | chi2) and no later phase rewrites
            // it: zeroed here, it stays zero under the sums instead of growing by the rank count per trial
            HIPCHK(c, hipMemsetAsync(dio + ohd, 0, sizeof(double) * (size_t)n, c->stream));
        }
        double rho = 0; int qmax = 0;
        for (;;) {
            if (sba_launch(c, 2, lam, 0)) return -1;
            if (allreduce(dio + oS, osc + 1, ncclSum)) return -1;               // S | bs | bp | (hd) | chi2
            if (sba_launch(c, 3, lam, 1)) return -1;                               // adds lambda I itself
            if (r) {                                                               // landmark part of the rho denominator, chi2 of the trial
                if (r->GroupStart() != ncclSuccess) return fail(c, "ncclGroupStart");
                if (allreduce(dio + osc + 3, 1, ncclSum) || allreduce(dio + osc + 5, 1, ncclSum)) return -1;
                if (r->GroupEnd() != ncclSuccess) return fail(c, "ncclGroupEnd");
            }
            if (d2h_sync(c, off_sc, off_sc + 8 * sizeof(double))) return -1;
            const double *sc = hio + osc;
            if (!have_current) { current = sc[0]; have_current = true; }
            const bool ok = sc[2] != 0.0;
            const double temp = ok ? sc[5] : 1.7976931348623157e308;
            const double scale = sc[3] + sc[4] + 1e-3;
            rho = (current - temp) / scale;
            const bool accept = rho > 0 && std::isfinite(temp);
            if (trace && ntr < trace_cap) {
                double *t = trace + 6 * (size_t)ntr++;
                t[0] = it; t[1] = lam; t[2] = current; t[3] = temp; t[4] = rho; t[5] = accept ? 1.0 : 0.0;
            }
            ++trials;
            if (accept) {
                const double t = 2 * rho - 1;
                const double alpha = std::min(1.0 - t * t * t, 2.0 / 3.0);
                lam *= std::max(1.0 / 3.0, alpha); ni = 2; current = temp;
            } else {
                lam *= ni; ni *= 2;
                if (sba_launch(c, 4, lam, 0)) return -1;
                if (!std::isfinite(lam)) break;
            }
            ++qmax;
            if (!(rho < 0 && qmax < 10)) break;
        }
        ++it_done;
        if (qmax == 10 || rho == 0 || !std::isfinite(lam)) break;
    }
    if (sba_launch(c, 5, lam, 0)) return -1;
    if (d2h_sync(c, off_sc, off_sc + 8 * sizeof(double))) return -1;
    if (iters_done) *iters_done = it_done;
    if (lambda_out) *lambda_out = lam;
    if (n_trace) *n_trace = ntr;
    if (stats) {
        const double ms = (now_ns() - t_begin) / 1e6;
        stats[0] = trials; stats[1] = ms; stats[2] = trials ? ms / trials : 0.0;
        stats[3] = (double)((osc + 1 + 2) * sizeof(double));
    }
    return 0;
}

// poses of all keyframes, the shard's landmark positions and per-edge chi2 (after phase 5); closes the shard
int svslam_sba_close(svslam_ctx *c, double *poses, double *pts, double *edge_chi2)
{
    if (!c->sba.open) return fail(c, "sba_close: no shard is open");
    c->sba.open = false;
    if (d2h_sync(c, c->sba.oposes, c->sba.ochi + sizeof(double) * (size_t)c->sba.nobs)) return -1;
    if (poses) memcpy(poses, hp<void>(c, c->sba.oposes), sizeof(double) * 7 * c->sba.nkf);
    if (pts) memcpy(pts, hp<void>(c, c->sba.opts), sizeof(double) * 3 * c->sba.nlm);
    if (edge_chi2) memcpy(edge_chi2, hp<void>(c, c->sba.ochi), sizeof(double) * c->sba.nobs);
    return 0;
}

// ------------------------------------------------------------------ the keyframe path on the device-resident map
// fb_mode: 0 the call as the caller made it; 1 / 2: the batch-solver repeat of local BAs the low-latency solver gave up on
// (dmap_ll_fallback) — optimise-only jobs; 1 also rebuilds the resident feature list of the job's keyframe (k_dmap_refresh ran
// before the outlier observations were removed), 2 refreshes positions only (a deferred BA that landed frames later)
static int dmap_keyframe_impl(svslam_ctx *c, int njobs, svslam_dmap_job *jobs, const void *const *left_imgs,
                              const void *const *right_imgs, const int *strides, int src_is_device,
                              const double cam_l[4], const double ext_l[7], const double cam_r[4], const double ext_r[7],
                              const svslam_dmap_params *p, int fb_mode)
{
    if (njobs <= 0) return 0;
    if (!c->dm_all) return fail(c, "dmap: context created without device_map");
    if (c->dmba.inflight) return fail(c, "dmap: a deferred local BA is in flight: call svslam_dmap_ba_collect first");
    static_assert(sizeof(DmJob) == sizeof(svslam_dmap_job), "job layout");
    static_assert(sizeof(DmEvicted) == sizeof(svslam_dmap_evicted_rec), "evicted record layout");
    c->evicted.clear();
    const DMap &m = c->dm;
    if (2 * njobs > c->lim.max_jobs) return fail(c, "dmap: %d jobs need max_jobs >= %d", njobs, 2 * njobs);
    if (njobs > SVSLAM_DMAP_CHUNK) return fail(c, "dmap: at most %d jobs per call", SVSLAM_DMAP_CHUNK);
    if (p->num_features < 1 || p->num_features > c->lim.max_corners) return fail(c, "dmap: num_features %d out of [1,%d]", p->num_features, c->lim.max_corners);
    if (p->num_active_keyframes + 1 > m.KW) return fail(c, "dmap: window of %d keyframes needs max_kf >= %d", p->num_active_keyframes, p->num_active_keyframes + 1);
    std::vector<int> slots; std::vector<const void *> imgs; std::vector<int> strd;
    // is_init == 2: no keyframe — ONE local BA over the stream's window as it is (Backend::UpdateMap from outside the
    // frontend, include/StereoVisionSLAM/backend.h:30).  A call holds keyframe jobs or such jobs, not both.
    int n_opt = 0;
    for (int i = 0; i < njobs; ++i) n_opt += jobs[i].is_init == 2 ? 1 : 0;
    if (n_opt != 0 && n_opt != njobs) return fail(c, "dmap: optimise-only jobs (is_init 2) cannot share a call with keyframe jobs");
    const bool opt_only = n_opt == njobs;
    for (int i = 0; i < njobs; ++i) {
        svslam_dmap_job &j = jobs[i];
        if (j.stream < 0 || j.stream >= c->lim.max_streams) return fail(c, "dmap: job %d stream %d out of range", i, j.stream);
        // two jobs of one call on the same stream would work on the same map arenas at once
        if (c->dm_seen.size() != (size_t)c->lim.max_streams) c->dm_seen.assign((size_t)c->lim.max_streams, 0u);
        if (i == 0 && ++c->dm_seen_gen == 0u) { std::fill(c->dm_seen.begin(), c->dm_seen.end(), 0u); c->dm_seen_gen = 1u; }
        if (c->dm_seen[(size_t)j.stream] == c->dm_seen_gen) return fail(c, "dmap: stream %d appears twice in one call (job %d)", j.stream, i);
        c->dm_seen[(size_t)j.stream] = c->dm_seen_gen;
        if (opt_only) {
            if (j.npts != c->rt_count[(size_t)j.stream]) return fail(c, "dmap: job %d says %d features, stream %d holds %d", i, j.npts, j.stream, c->rt_count[(size_t)j.stream]);
            continue;
        }
        if (check_slot(c, j.slot_cur) || check_slot(c, j.slot_right)) return -1;
        if (j.kf_slot < 0 || j.kf_slot >= m.KW || j.remove_slot >= m.KW) return fail(c, "dmap: job %d keyframe slot out of range", i);
        if (j.is_init ? j.npts != 0 : j.npts != c->rt_count[(size_t)j.stream])
            return fail(c, "dmap: job %d says %d features, stream %d holds %d", i, j.npts, j.stream, c->rt_count[(size_t)j.stream]);
        if (j.is_init) { slots.push_back(j.slot_cur); imgs.push_back(left_imgs[i]); strd.push_back(strides[i]); }
    }
    for (int i = 0; i < njobs && !opt_only; ++i) { slots.push_back(jobs[i].slot_right); imgs.push_back(right_imgs[i]); strd.push_back(strides[i]); }
    if (opt_only) {
        if (arena_busy(c)) return -1;
        c->ar.reset();
    } else {
        const bool dec = c->src_w > 0;
        if (pyramid_common(c, (int)slots.size(), slots.data(), imgs.data(), strd.data(), src_is_device, dec, dec ? c->src_w : c->geom.w[0],
                           dec ? c->src_h : c->geom.h[0], false)) return -1;
    }
    const int NF = m.NF, NL = m.NL, MO = c->lim.max_obs, MK = c->lim.max_kf, MC = p->num_features;
    c->ba_eid = ba_ext_identity(ext_l, ext_r);
    const bool use_ll = ba_ll_usable(c, njobs);
    const int tile_cap = use_ll ? ba_ll_tile_cap(c) : ba_tile_cap(MK);
    const size_t aux_stride = ba_aux_layout(MK, NL, MO, MO, MK, 0, ba_tile_bound(NL, MO, MK, tile_cap)).total + ba_pitem_bound(MO, MK) +
                              (use_ll ? ba_split_aux_extra(MK, c->ll.w) : 0);
    const size_t n = njobs, P = n * NF, E = n * MO;
    const size_t base = c->ar.off;
    size_t ojobs = c->ar.take(sizeof(DmJob) * n);
    size_t ogj = c->ar.take(sizeof(GfttJob) * n);
    size_t ocams = c->ar.take(sizeof(BaCams));
    size_t in_end = c->ar.off;
    // device-only scratch of this call
    size_t ocor = c->ar.take(sizeof(float2) * n * MC), oncor = c->ar.take(sizeof(int) * n);
    size_t olk = c->ar.take(sizeof(LkJob) * n), oprev = c->ar.take(sizeof(float2) * P), onext = c->ar.take(sizeof(float2) * P);
    size_t ostat = c->ar.take(P), oerr = c->ar.take(sizeof(float) * P);
    size_t otj = c->ar.take(sizeof(TriJob) * n), oul = c->ar.take(sizeof(float2) * P), our_ = c->ar.take(sizeof(float2) * P);
    size_t otidx = c->ar.take(sizeof(int) * P), oxyz = c->ar.take(sizeof(double) * 3 * P), ook = c->ar.take(P), oslot = c->ar.take(sizeof(int) * P);
    // the local-BA problem and the solver's scratch: in the arena, or — deferred — in the buffer that outlives this call
    const bool defer = p->ba_defer != 0 && p->ba_iters > 0 && !opt_only;
    Arena bar;                                    // offsets only (take()): base pointer chosen below
    bar.cap = ~(size_t)0;
    Arena &A = defer ? bar : c->ar;
    size_t obd = A.take(sizeof(BaDev) * n), oposes = A.take(sizeof(double) * 7 * MK * n), opts = A.take(sizeof(double) * 3 * NL * n);
    size_t opk = A.take(sizeof(unsigned int) * E), ouv = A.take(sizeof(float2) * E), oref = A.take(sizeof(int) * E);
    size_t olms = A.take(sizeof(int) * NL * n), ochi = A.take(sizeof(double) * E), oflag = A.take(sizeof(int) * 4);
    size_t orecs = A.take(sizeof(BaRec) * 2 * E), oaux = A.take(sizeof(int) * aux_stride * n);
    size_t obcams = 0, objobs = 0;
    if (defer) {
        obcams = A.take(sizeof(BaCams)); objobs = A.take(sizeof(DmJob) * n);
        if (!c->dmba.stream) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->dmba.stream, hipStreamNonBlocking));
            for (hipEvent_t *e : { &c->dmba.gathered, &c->dmba.solved }) HIPCHK(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
            for (hipEvent_t *e : { &c->dmba.t0, &c->dmba.t1 }) HIPCHK(c, hipEventCreate(e));
        }
        if (A.off > c->dmba.cap) {
            if (c->dmba.buf) { HIPCHK(c, hipStreamSynchronize(c->dmba.stream)); (void)hipFree(c->dmba.buf); c->dmba.buf = nullptr; c->dmba.cap = 0; }
            const size_t want = A.off + A.off / 4;
            HIPCHK(c, hipMalloc(&c->dmba.buf, want));
            c->dmba.cap = want;
        }
    }
    unsigned char *bab = defer ? c->dmba.buf : c->ar.d;          // base of the BA buffers
    auto bp_ = [&](size_t off) { return bab + off; };
    // (test hook SVSLAM_DMAP_EVICT_CAP: a smaller list for the whole call, to exercise the "waits for the next keyframe" path)
    static const int ev_cap_env = []{ const char *e = std::getenv("SVSLAM_DMAP_EVICT_CAP"); return e ? atoi(e) : 0; }();
    const int ev_cap = ev_cap_env > 0 ? std::max(njobs, std::min(ev_cap_env, njobs * SVSLAM_DMAP_EVICT_PER_JOB)) : njobs * SVSLAM_DMAP_EVICT_PER_JOB;
    size_t oev = c->ar.take(sizeof(DmEvicted) * (size_t)ev_cap);
    if (c->ar.off > c->ar.cap) return fail(c, "dmap: staging arena too small (%zu > %zu bytes); fewer jobs per call", c->ar.off, c->ar.cap);
    DmJob *hj = hp<DmJob>(c, ojobs);
    GfttJob *gj = hp<GfttJob>(c, ogj);
    memcpy(hj, jobs, sizeof(DmJob) * n);
    c->dm_stamp++;
    for (int i = 0; i < njobs; ++i) {
        hj[i].src_buf = c->rt_which[(size_t)hj[i].stream];
        hj[i].dst_buf = c->rt_which[(size_t)hj[i].stream];      // the survivors have been copied into the keyframe by then
        hj[i].stamp = c->dm_stamp;
        if (opt_only) {          // what k_dmap_begin would have initialised
            hj[i].ok = 1; hj[i].dead = 0; hj[i].flags = 0; hj[i].n_features = hj[i].npts; hj[i].n_corners = hj[i].n_right_ok = hj[i].n_tri_in = hj[i].n_tri_ok = 0;
            hj[i].ba_nkf = hj[i].ba_nlm = hj[i].ba_nobs = hj[i].ba_iters = hj[i].ba_npair = hj[i].ba_ntrial = 0; hj[i].ev_ofs = hj[i].ev_n = 0;
            if (fb_mode != 1) hj[i].kf_slot = -1;
            hj[i].remove_slot = -1; hj[i].pad0 = 0;
        }
        gj[i].slot = hj[i].slot_cur; gj[i].nrect = hj[i].npts;
        gj[i].rect_ofs = (int)(((size_t)hj[i].stream * m.KW + hj[i].kf_slot) * NF);
    }
    BaCams *cams = hp<BaCams>(c, ocams);
    memcpy(cams->cam[0], cam_l, 32); memcpy(cams->cam[1], cam_r, 32);
    memcpy(cams->ext[0], ext_l, 56); memcpy(cams->ext[1], ext_r, 56);
    DmParams prm;
    prm.num_features = p->num_features; prm.num_features_init = p->num_features_init; prm.num_active = p->num_active_keyframes;
    prm.zmax = p->max_triangulation_depth; prm.chi2_th = p->chi2_th;
    memcpy(prm.cam_l, cam_l, 32); memcpy(prm.cam_r, cam_r, 32);
    prm.max_obs = MO; prm.max_lm = NL; prm.w = c->geom.w[0]; prm.h = c->geom.h[0];
    TriCams tc;
    memcpy(tc.cam_l, cam_l, 32); memcpy(tc.ext_l, ext_l, 56); memcpy(tc.cam_r, cam_r, 32); memcpy(tc.ext_r, ext_r, 56);
    if (h2d(c, base, in_end)) return -1;
    // flag words: [0] BA structure build overflow, [1] evicted-list cursor.  Deferred: the build's word lives with the problem,
    // the cursor of this call's evicted list stays in the arena
    size_t oevc = oflag;
    if (defer) { oevc = c->ar.take(sizeof(int) * 4); if (c->ar.off > c->ar.cap) return fail(c, "dmap: staging arena too small"); }
    HIPCHK(c, hipMemsetAsync(bp_(oflag), 0, sizeof(int) * 4, c->stream));
    if (defer) HIPCHK(c, hipMemsetAsync(dp<void>(c, oevc), 0, sizeof(int) * 4, c->stream));
    int *d_evcur = defer ? dp<int>(c, oevc) + 1 : reinterpret_cast<int *>(bp_(oflag)) + 1;
    DmJob *dj = dp<DmJob>(c, ojobs);
    if (!opt_only) {
    // every job may hand over ev_cap / njobs landmarks per call: what a job evicts does not depend on the other jobs' timing
    hipLaunchKernelGGL(k_dmap_begin, dim3(njobs), dim3(DM_THREADS), 0, c->stream, dj, m, c->rt, dp<DmEvicted>(c, oev), d_evcur, std::max(1, ev_cap / njobs));
    if (launch_gftt(c, njobs, dp<GfttJob>(c, ogj), m.f_xy, MC, 0.01, 20.0, dp<float2>(c, ocor), dp<int>(c, oncor))) return -1;   // src/frontend.cpp:24
    hipLaunchKernelGGL(k_dmap_stereo_prep, dim3(njobs), dim3(DM_THREADS), 0, c->stream, dj, m, prm, dp<float2>(c, ocor), dp<int>(c, oncor), MC,
                       dp<LkJob>(c, olk), dp<float2>(c, oprev), dp<float2>(c, onext));
    {
        svslam_lk_params lp = { 3, 30, 0.01, 1e-4, 1 };                     // src/frontend.cpp:105-109
        tm_begin(c, FAM_LK, 0);
        launch_lk(c, njobs, NF, dp<LkJob>(c, olk), dp<float2>(c, oprev), dp<float2>(c, onext), dp<uint8_t>(c, ostat), dp<float>(c, oerr), &lp);
        tm_end(c);
    }
    hipLaunchKernelGGL(k_dmap_stereo_finish, dim3(njobs), dim3(DM_THREADS), 0, c->stream, dj, m, prm, dp<float2>(c, onext), dp<uint8_t>(c, ostat),
                       dp<TriJob>(c, otj), dp<float2>(c, oul), dp<float2>(c, our_), dp<int>(c, otidx));
    tm_begin(c, FAM_TRI, 0);
    hipLaunchKernelGGL(k_triangulate, dim3(njobs, cdiv(NF, 64)), dim3(64), 0, c->stream, dp<TriJob>(c, otj), tc, dp<float2>(c, oul), dp<float2>(c, our_),
                       dp<double>(c, oxyz), dp<uint8_t>(c, ook));
    tm_end(c);
    hipLaunchKernelGGL(k_dmap_commit, dim3(njobs), dim3(DM_THREADS), 0, c->stream, dj, m, dp<double>(c, oxyz), dp<uint8_t>(c, ook), dp<int>(c, otidx), dp<int>(c, oslot));
    }
    // Backend::UpdateMap (src/backend.cpp:14-18) only runs with a backend: a paused / absent one (ba_iters <= 0) means no
    // Optimize, so no outlier classification and no observation removed either — like the host-map path
    if (p->ba_iters > 0) {
        BaDev *b_bd = reinterpret_cast<BaDev *>(bp_(obd));
        double *b_poses = reinterpret_cast<double *>(bp_(oposes)), *b_pts = reinterpret_cast<double *>(bp_(opts)), *b_chi = reinterpret_cast<double *>(bp_(ochi));
        int *b_ref = reinterpret_cast<int *>(bp_(oref)), *b_lms = reinterpret_cast<int *>(bp_(olms));
        const BaCams *b_cams = defer ? reinterpret_cast<const BaCams *>(bp_(obcams)) : dp<BaCams>(c, ocams);
        hipStream_t main_stream = c->stream;
        if (!defer) tm_begin(c, FAM_BA, njobs);
        if (njobs <= SVSLAM_LL_MAX_PROBLEMS)
            hipLaunchKernelGGL(k_dmap_ba_gather<1024>, dim3(njobs), dim3(1024), dmg_lds_bytes_t<1024>(NL), c->stream, dj, m, prm, b_bd, b_poses, b_pts,
                               reinterpret_cast<unsigned int *>(bp_(opk)), reinterpret_cast<float2 *>(bp_(ouv)), b_ref, b_lms, MK, tile_cap, aux_stride);
        else
            hipLaunchKernelGGL(k_dmap_ba_gather<512>, dim3(njobs), dim3(512), dmg_lds_bytes_t<512>(NL), c->stream, dj, m, prm, b_bd, b_poses, b_pts,
                               reinterpret_cast<unsigned int *>(bp_(opk)), reinterpret_cast<float2 *>(bp_(ouv)), b_ref, b_lms, MK, tile_cap, aux_stride);
        if (defer) {
            // the solve leaves this call's stream: a copy of the jobs (the scatter needs streams, window slots) and of the
            // cameras goes with the problem; the second stream picks up behind the gather
            HIPCHK(c, hipMemcpyAsync(bp_(objobs), dj, sizeof(DmJob) * n, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(bp_(obcams), dp<void>(c, ocams), sizeof(BaCams), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipEventRecord(c->dmba.gathered, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->dmba.stream, c->dmba.gathered, 0));
            c->stream = c->dmba.stream;                       // (launch_ba_solver enqueues on the context's stream)
            (void)hipEventRecord(c->dmba.t0, c->stream);
        }
        launch_ba_solver(c, njobs, use_ll, b_bd, b_cams, b_poses, b_pts, reinterpret_cast<unsigned int *>(bp_(opk)),
                         reinterpret_cast<float2 *>(bp_(ouv)), b_ref /* order: identity, not read */, reinterpret_cast<BaRec *>(bp_(orecs)),
                         reinterpret_cast<int *>(bp_(oaux)), b_chi, reinterpret_cast<int *>(bp_(oflag)), NL, MO, p->chi2_th, p->ba_iters, false, !defer);
        if (defer) {
            (void)hipEventRecord(c->dmba.t1, c->stream);
            const hipError_t rec_rc = hipEventRecord(c->dmba.solved, c->stream);
            c->stream = main_stream;
            HIPCHK(c, rec_rc);
            c->dmba.inflight = true; c->dmba.njobs = njobs; c->dmba.ojobs = objobs; c->dmba.obd = obd; c->dmba.oposes = oposes; c->dmba.opts = opts;
            c->dmba.ochi = ochi; c->dmba.oref = oref; c->dmba.olms = olms; c->dmba.oflag = oflag; c->dmba.prm = prm; c->dmba.MK = MK;
            c->dmba.cams = *cams; c->dmba.ba_iters = p->ba_iters;
        } else {
            hipLaunchKernelGGL(k_dmap_ba_scatter, dim3(njobs), dim3(DM_THREADS), 0, c->stream, dj, m, prm, b_bd, b_poses, b_pts, b_chi, b_ref, b_lms, MK);
            tm_end(c);
        }
    }
    if (opt_only && fb_mode == 1) hipLaunchKernelGGL(k_dmap_refresh, dim3(njobs), dim3(DM_THREADS), 0, c->stream, dj, m, c->rt);
    else if (opt_only) {
        size_t ol3 = c->ar.take(sizeof(int) * 3 * n);
        if (c->ar.off > c->ar.cap) return fail(c, "dmap: staging arena too small");
        int *l3 = hp<int>(c, ol3);
        for (int i = 0; i < njobs; ++i) { l3[3 * i] = hj[i].stream; l3[3 * i + 1] = hj[i].src_buf; l3[3 * i + 2] = hj[i].npts; }
        if (h2d(c, ol3, ol3 + sizeof(int) * 3 * n)) return -1;
        hipLaunchKernelGGL(k_dmap_refresh_xyz, dim3(njobs), dim3(DM_THREADS), 0, c->stream, dp<int>(c, ol3), m, c->rt);
    } else hipLaunchKernelGGL(k_dmap_refresh, dim3(njobs), dim3(DM_THREADS), 0, c->stream, dj, m, c->rt);
    HIPCHK(c, hipGetLastError());
    // ONE wait for everything the host reads (round 4: three copies each behind its own wait cost a lone camera ~50 us per
    // keyframe): the jobs, the flag words and the head of the evicted list ride one completion; only a list longer than the
    // head is fetched with a second copy
    // (deferred: the build's overflow word is read by svslam_dmap_ba_collect; word 0 of the arena copy stays 0)
    const size_t ofw = defer ? oevc : oflag;
    const int ev_head = std::min(ev_cap, 256);
    HIPCHK(c, hipMemcpyAsync(hp<void>(c, ojobs), dp<void>(c, ojobs), sizeof(DmJob) * n, hipMemcpyDeviceToHost, c->stream));
    if (defer) HIPCHK(c, hipMemcpyAsync(hp<void>(c, ofw), dp<void>(c, ofw), sizeof(int) * 4, hipMemcpyDeviceToHost, c->stream));
    else HIPCHK(c, hipMemcpyAsync(hp<void>(c, ofw), bp_(ofw), sizeof(int) * 4, hipMemcpyDeviceToHost, c->stream));
    if (!opt_only && ev_head > 0) HIPCHK(c, hipMemcpyAsync(hp<void>(c, oev), dp<void>(c, oev), sizeof(DmEvicted) * (size_t)ev_head, hipMemcpyDeviceToHost, c->stream));
    if (d2h_sync(c, 0, 0)) return -1;
    const int *fw = hp<int>(c, ofw);
    if (const int fl = fw[0]) return fail(c, "dmap: the BA structure build overflowed a capacity (code %d)", fl);
    if (const int nev = fw[1]) {                // the landmarks this call freed (svslam_dmap_evicted)
        if (nev < 0 || nev > ev_cap) return fail(c, "dmap: evicted-list cursor %d out of [0,%d]", nev, ev_cap);
        if (nev > ev_head && d2h_sync(c, oev + sizeof(DmEvicted) * (size_t)ev_head, oev + sizeof(DmEvicted) * (size_t)nev)) return -1;
        c->evicted.assign(hp<DmEvicted>(c, oev), hp<DmEvicted>(c, oev) + nev);
    }
    memcpy(jobs, hj, sizeof(DmJob) * n);
    for (int i = 0; i < njobs && !opt_only; ++i) c->rt_count[(size_t)jobs[i].stream] = jobs[i].n_features;
    for (int i = 0; i < njobs; ++i)
        if (jobs[i].ba_iters < 0 && fb_mode != 0) return fail(c, "dmap: job %d: the batch solver reported a low-latency failure", i);
    return 0;
}

// Local BAs of `jobs` whose low-latency solve gave up (ba_iters < 0: a shard of the problem never became resident — the CUs were
// taken by something else; nothing was written back) are solved again with the batch solver, one workgroup per problem, which
// has no inter-workgroup barrier: optimise-only jobs over the same windows, their BA outputs copied into the caller's jobs.
static int dmap_ll_fallback(svslam_ctx *c, int njobs, svslam_dmap_job *jobs, const double cam_l[4], const double ext_l[7],
                            const double cam_r[4], const double ext_r[7], const svslam_dmap_params *p, int fb_mode)
{
    std::vector<int> bad;
    for (int i = 0; i < njobs; ++i) if (jobs[i].ba_iters < 0) bad.push_back(i);
    if (bad.empty()) return 0;
    const int nb = (int)bad.size();
    std::vector<svslam_dmap_job> oj((size_t)nb);
    for (int k = 0; k < nb; ++k) {
        oj[(size_t)k] = jobs[bad[(size_t)k]];
        oj[(size_t)k].is_init = 2;
        oj[(size_t)k].npts = c->rt_count[(size_t)oj[(size_t)k].stream];
    }
    std::vector<DmEvicted> keep;
    keep.swap(c->evicted);                         // (the repeat frees nothing; the caller still reads this call's list)
    svslam_dmap_params q = *p;
    q.ba_defer = 0;
    if (q.ba_iters <= 0) q.ba_iters = 10;
    std::vector<const void *> nul((size_t)nb, nullptr);
    std::vector<int> st((size_t)nb, 0);
    c->ll.force_batch = true;
    const int rc = dmap_keyframe_impl(c, nb, oj.data(), nul.data(), nul.data(), st.data(), 1, cam_l, ext_l, cam_r, ext_r, &q, fb_mode);
    c->ll.force_batch = false;
    c->evicted.swap(keep);
    if (rc) return rc;
    for (int k = 0; k < nb; ++k) {
        svslam_dmap_job &d = jobs[bad[(size_t)k]];
        const svslam_dmap_job &o = oj[(size_t)k];
        d.ba_nkf = o.ba_nkf; d.ba_nlm = o.ba_nlm; d.ba_nobs = o.ba_nobs; d.ba_iters = o.ba_iters; d.ba_npair = o.ba_npair; d.ba_ntrial = o.ba_ntrial;
        d.flags |= o.flags & 4;
        memcpy(d.win_pose, o.win_pose, sizeof(d.win_pose)); memcpy(d.win_slot, o.win_slot, sizeof(d.win_slot));
        if (fb_mode == 1) memcpy(d.pose, o.pose, sizeof(d.pose));
    }
    c->ll.fallbacks += nb;
    return 0;
}

int svslam_dmap_keyframe_batch(svslam_ctx *c, int njobs, svslam_dmap_job *jobs, const void *const *left_imgs,
                               const void *const *right_imgs, const int *strides, int src_is_device,
                               const double cam_l[4], const double ext_l[7], const double cam_r[4], const double ext_r[7],
                               const svslam_dmap_params *p)
{
    if (njobs <= 0) return 0;
    const bool was_inflight = c->dmba.inflight;
    const int rc = dmap_keyframe_impl(c, njobs, jobs, left_imgs, right_imgs, strides, src_is_device, cam_l, ext_l, cam_r, ext_r, p, 0);
    if (rc != 0) {
        // a failure behind the point where the deferred solve was armed must not leave it armed: the caller never learns
        // that it has to collect, and every later keyframe call would be refused
        if (!was_inflight && c->dmba.inflight) {
            const std::string keep_err = c->err;
            if (c->dmba.stream) (void)hipStreamSynchronize(c->dmba.stream);
            c->dmba.inflight = false;
            c->err = keep_err;
        }
        return rc;
    }
    return dmap_ll_fallback(c, njobs, jobs, cam_l, ext_l, cam_r, ext_r, p, 1);
}

int svslam_dmap_ba_collect(svslam_ctx *c, int njobs, svslam_dmap_job *jobs_out, int *njobs_inflight)
{
    if (njobs_inflight) *njobs_inflight = c->dmba.inflight ? c->dmba.njobs : 0;
    if (!c->dmba.inflight) return 0;
    if (njobs != c->dmba.njobs || !jobs_out) return fail(c, "dmap_ba_collect: %d jobs are in flight, the caller passed %d", c->dmba.njobs, njobs);
    if (arena_busy(c)) return -1;
    c->dmba.inflight = false;
    const DMap &m = c->dm;
    unsigned char *b = c->dmba.buf;
    const size_t n = (size_t)njobs;
    c->ar.reset();
    size_t ojobs = c->ar.take(sizeof(DmJob) * n), oflag = c->ar.take(sizeof(int) * 4), ol3 = c->ar.take(sizeof(int) * 3 * n);
    if (c->ar.off > c->ar.cap) return fail(c, "dmap_ba_collect: staging arena too small");
    // the solve is done when its event is: this stream goes on behind it
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->dmba.solved, 0));
    DmJob *dj = reinterpret_cast<DmJob *>(b + c->dmba.ojobs);
    tm_begin(c, FAM_BA, 0);
    hipLaunchKernelGGL(k_dmap_ba_scatter, dim3(njobs), dim3(DM_THREADS), 0, c->stream, dj, m, c->dmba.prm, reinterpret_cast<const BaDev *>(b + c->dmba.obd),
                       reinterpret_cast<const double *>(b + c->dmba.oposes), reinterpret_cast<const double *>(b + c->dmba.opts),
                       reinterpret_cast<const double *>(b + c->dmba.ochi), reinterpret_cast<const int *>(b + c->dmba.oref),
                       reinterpret_cast<const int *>(b + c->dmba.olms), c->dmba.MK);
    tm_end(c);
    // the jobs come back first: the streams whose resident lists are refreshed are named by them
    HIPCHK(c, hipMemcpyAsync(hp<void>(c, ojobs), dj, sizeof(DmJob) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hp<void>(c, oflag), b + c->dmba.oflag, sizeof(int) * 4, hipMemcpyDeviceToHost, c->stream));
    if (d2h_sync(c, 0, 0)) return -1;
    {   // the solver's time, measured on its own stream
        float ms = 0.f;
        if (c->timing && hipEventElapsedTime(&ms, c->dmba.t0, c->dmba.t1) == hipSuccess) { c->tm.ms[FAM_BA] += ms; c->tm.launches[FAM_BA] += 1; c->tm.units[FAM_BA] += njobs; }
    }
    if (const int fl = hp<int>(c, oflag)[0]) return fail(c, "dmap: the BA structure build overflowed a capacity (code %d)", fl);
    const DmJob *hj = hp<DmJob>(c, ojobs);
    int *l3 = hp<int>(c, ol3);
    for (int i = 0; i < njobs; ++i) {
        const int s_ = hj[i].stream;
        l3[3 * i] = s_; l3[3 * i + 1] = c->rt_which[(size_t)s_]; l3[3 * i + 2] = c->rt_count[(size_t)s_];
    }
    if (h2d(c, ol3, ol3 + sizeof(int) * 3 * n)) return -1;
    hipLaunchKernelGGL(k_dmap_refresh_xyz, dim3(njobs), dim3(DM_THREADS), 0, c->stream, dp<int>(c, ol3), m, c->rt);
    HIPCHK(c, hipGetLastError());
    if (d2h_sync(c, 0, 0)) return -1;
    memcpy(jobs_out, hj, sizeof(DmJob) * n);
    {   // problems the low-latency solver gave up on: once more with the batch solver (the map has not moved since the gather:
        // the host collects before the next keyframe)
        svslam_dmap_params q{};
        q.num_features = c->dmba.prm.num_features; q.num_features_init = c->dmba.prm.num_features_init; q.num_active_keyframes = c->dmba.prm.num_active;
        q.ba_iters = c->dmba.ba_iters; q.max_triangulation_depth = c->dmba.prm.zmax; q.chi2_th = c->dmba.prm.chi2_th;
        return dmap_ll_fallback(c, njobs, jobs_out, c->dmba.prm.cam_l, c->dmba.cams.ext[0], c->dmba.prm.cam_r, c->dmba.cams.ext[1], &q, 2);
    }
}

int svslam_dmap_evicted(svslam_ctx *c, const svslam_dmap_evicted_rec **recs, int *n)
{
    if (!c->dm_all) return fail(c, "dmap_evicted: context created without device_map");
    *recs = reinterpret_cast<const svslam_dmap_evicted_rec *>(c->evicted.data());
    *n = (int)c->evicted.size();
    return 0;
}

int svslam_dmap_read(svslam_ctx *c, int stream, long long *kf_frame, int *kf_id, double *kf_pose, int *kf_n, int *lm_id,
                     double *lm_pos, int *lm_obs, uint8_t *lm_state)
{
    if (!c->dm_all) return fail(c, "dmap_read: context created without device_map");
    if (stream < 0 || stream >= c->lim.max_streams) return fail(c, "dmap_read: stream %d out of range", stream);
    const DMap &m = c->dm;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t K = (size_t)stream * m.KW, L = (size_t)stream * m.NL;
    if (kf_frame) HIPCHK(c, hipMemcpy(kf_frame, m.kf_frame + K, sizeof(long long) * m.KW, hipMemcpyDeviceToHost));
    if (kf_id) HIPCHK(c, hipMemcpy(kf_id, m.kf_id + K, sizeof(int) * m.KW, hipMemcpyDeviceToHost));
    if (kf_pose) HIPCHK(c, hipMemcpy(kf_pose, m.kf_pose + K * 7, sizeof(double) * 7 * m.KW, hipMemcpyDeviceToHost));
    if (kf_n) HIPCHK(c, hipMemcpy(kf_n, m.kf_n + K, sizeof(int) * m.KW, hipMemcpyDeviceToHost));
    if (lm_id) HIPCHK(c, hipMemcpy(lm_id, m.lm_id + L, sizeof(int) * m.NL, hipMemcpyDeviceToHost));
    if (lm_pos) HIPCHK(c, hipMemcpy(lm_pos, m.lm_pos + L * 3, sizeof(double) * 3 * m.NL, hipMemcpyDeviceToHost));
    if (lm_obs) HIPCHK(c, hipMemcpy(lm_obs, m.lm_obs + L, sizeof(int) * m.NL, hipMemcpyDeviceToHost));
    if (lm_state) HIPCHK(c, hipMemcpy(lm_state, m.lm_st + L, m.NL, hipMemcpyDeviceToHost));
    return 0;
}

// number of host threads the library may use to prepare batched calls (BA structure
// building is per problem and independent); default 1
int svslam_set_host_threads(svslam_ctx *c, int n)
{
    c->pool.reset(n > 1 ? new svs::ThreadPool(n) : nullptr);
    return 0;
}

// test hook: host-side wall time per category since the last call (ns):
// 0 h2d enqueue, 1 d2h enqueue, 2 stream wait, 5 event collection; slot 6 is a count: local-BA problems solved by the
// low-latency path (one problem over several workgroups)
int svslam_debug_host_ns(svslam_ctx *c, long long *out8)
{
    c->host_ns[7] = c->ll.fallbacks; c->ll.fallbacks = 0;      // slot 7: problems the low-latency solver gave up on, repeated by the batch solver
    for (int i = 0; i < 8; ++i) { out8[i] = c->host_ns[i]; c->host_ns[i] = 0; }
    return 0;
}

// test hook: the shard descriptors of the last low-latency local-BA call (k_ba_split / k_ba_build): 8 ints per shard —
// landmarks, edges, blocks, tiles, solver (2: resident kernel k_ba_ll, 1: streaming k_local_ba_t<2>), active poses, mask of shards with edges, iterations
int svslam_debug_ll_limits(svslam_ctx *c, int *out4)
{
    out4[0] = c->ll.w; out4[1] = c->ll.max_problems; out4[2] = c->ll.cus; out4[3] = c->ll.blocks_per_cu;
    return 0;
}
int svslam_debug_ll_shards(svslam_ctx *c, int nproblems, int *out8, int *shards_per_problem)
{
    if (shards_per_problem) *shards_per_problem = c->ll.w;
    if (!c->ll.shards || nproblems < 1 || nproblems > SVSLAM_LL_MAX_PROBLEMS) return fail(c, "debug_ll_shards: no low-latency solver / bad count");
    std::vector<BaDev> h((size_t)nproblems * c->ll.w);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(h.data(), c->ll.shards, sizeof(BaDev) * h.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); ++i) {
        int *o = out8 + 8 * i;
        o[0] = h[i].nlm; o[1] = h[i].nobs; o[2] = h[i].nblk; o[3] = h[i].ntile; o[4] = h[i].reserved; o[5] = h[i].na; o[6] = h[i].shmask; o[7] = h[i].iters_done;
    }
    return 0;
}

// test hook: effective shader clock.  A single wave spins for ~`ms` milliseconds of
// wall_clock64 (100 MHz, constant) and reports the clock64() (shader cycles) it saw.
__global__ void k_clock_probe(long long *out, long long wall_ticks)
{
    long long w0 = wall_clock64(), c0 = clock64();
    long long w = w0;
    while (w - w0 < wall_ticks) w = wall_clock64();
    long long c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = w - w0; out[1] = c1 - c0; }
}
// test hook: `ncus` workgroups that each take a CU's whole LDS and spin for `ms` — the CUs they sit on cannot take a workgroup
// that needs LDS until they leave (what another process' kernels do to a partitioned or shared GPU).  Asynchronous: enqueued on
// the context's stream, returns at once; svslam_sync waits for it.
__global__ void k_hold_cu(long long wall_ticks)
{
    extern __shared__ unsigned char hold_lds[];
    if (threadIdx.x == 0) hold_lds[0] = 1;
    const long long w0 = wall_clock64();
    while (wall_clock64() - w0 < wall_ticks) __builtin_amdgcn_s_sleep(8);
}
int svslam_debug_hold_cus(svslam_ctx *c, int ncus, double ms)
{
    if (ncus <= 0) return 0;
    const int lds = 160 * 1024;
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_hold_cu), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(k_hold_cu, dim3(ncus), dim3(64), lds, c->stream, (long long)(ms * 1e5));
    HIPCHK(c, hipGetLastError());
    return 0;
}

int svslam_debug_clock_mhz(svslam_ctx *c, int blocks, double ms, double *mhz)
{
    long long *d = nullptr, h[2] = { 0, 0 };
    HIPCHK(c, hipMalloc(&d, 16));
    hipLaunchKernelGGL(k_clock_probe, dim3(blocks), dim3(64), 0, c->stream, d, (long long)(ms * 1e5));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    *mhz = h[0] > 0 ? 100.0 * (double)h[1] / (double)h[0] : 0.0;
    return 0;
}

// test hook: per-phase cycle counters of BA job 0 (wall_clock64 ticks, 100 MHz)
int svslam_ba_profile(svslam_ctx *c, int enable, long long *out12)
{
    if (enable && !c->d_ba_prof) {
        HIPCHK(c, hipMalloc(&c->d_ba_prof, sizeof(long long) * (BA_PROF_N + 4)));
        HIPCHK(c, hipMemset(c->d_ba_prof, 0, sizeof(long long) * (BA_PROF_N + 4)));
    }
    if (out12 && c->d_ba_prof) {
        long long all[BA_PROF_N + 4];
        HIPCHK(c, hipMemcpy(all, c->d_ba_prof, sizeof(all), hipMemcpyDeviceToHost));
        memcpy(out12, all, sizeof(long long) * BA_PROF_N);
        // development (-DBA_CHOL_PROF builds): wave 0's clocks inside the factorisation — wait at the top of a block column,
        // its share of the trailing update, factor + panel of the next column, wait at the column's barrier
        if (std::getenv("SVSLAM_BA_PROF_EXTRA"))
            fprintf(stderr, "[chol wave 0, us] top %.1f  trail %.1f  factor+panel %.1f  barrier %.1f\n", all[12] / 100.0, all[13] / 100.0, all[14] / 100.0, all[15] / 100.0);
        HIPCHK(c, hipMemset(c->d_ba_prof, 0, sizeof(long long) * (BA_PROF_N + 4)));
    }
    if (!enable && c->d_ba_prof) { (void)hipFree(c->d_ba_prof); c->d_ba_prof = nullptr; }
    return 0;
}

// test hook: the LM trajectory of the last pose-only / local-BA call (every trial, rejected ones included), job by job.
// enable = 1 allocates the buffer (the kernels then record), 0 frees it; out != NULL copies job `job`'s records
// (6 doubles each: iteration [pose-only: 16 round + iteration], lambda, chi2 before, chi2 of the trial, rho, accepted).
int svslam_lm_trace(svslam_ctx *c, int enable, int job, double *out, int cap_records, int *n_records)
{
    if (enable && !c->d_lm_trace) {
        HIPCHK(c, hipMalloc(&c->d_lm_trace, sizeof(double) * LM_TRACE_STRIDE * (size_t)c->lim.max_jobs));
        HIPCHK(c, hipMemset(c->d_lm_trace, 0, sizeof(double) * LM_TRACE_STRIDE * (size_t)c->lim.max_jobs));
    }
    if (out && c->d_lm_trace) {
        if (job < 0 || job >= c->lim.max_jobs) return fail(c, "lm_trace: job %d out of range", job);
        std::vector<double> t(LM_TRACE_STRIDE);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(t.data(), c->d_lm_trace + (size_t)job * LM_TRACE_STRIDE, sizeof(double) * LM_TRACE_STRIDE, hipMemcpyDeviceToHost));
        const int n = std::min((int)t[0], cap_records);
        if (n_records) *n_records = n;
        if (n > 0) memcpy(out, t.data() + 8, sizeof(double) * LM_TRACE_REC * (size_t)n);
    } else if (n_records) *n_records = 0;
    if (!enable && c->d_lm_trace) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_lm_trace); c->d_lm_trace = nullptr; }
    return 0;
}

// ------------------------------------------------------------------ fused tracking
int svslam_track_batch(svslam_ctx *c, int njobs, svslam_track_job *jobs, const void *const *next_imgs,
                       const int *strides, int src_is_device, int total_pts, const double cam[4],
                       const float *prev_xy, float *next_xy, const uint8_t *has_mp, const double *xyz,
                       uint8_t *status, uint8_t *outlier, const svslam_lk_params *p, double chi2_th)
{
    if (njobs <= 0) return 0;
    if (njobs > c->lim.max_jobs) return fail(c, "track: %d jobs > max_jobs", njobs);
    if (total_pts > c->lim.max_jobs * c->lim.max_pts) return fail(c, "track: too many points");
    std::vector<int> slots(njobs);
    for (int i = 0; i < njobs; ++i) {
        if (check_slot(c, jobs[i].prev_slot) || check_slot(c, jobs[i].next_slot)) return -1;
        if (jobs[i].npts < 0 || jobs[i].npts > c->lim.max_pts || jobs[i].pt_ofs < 0 ||
            jobs[i].pt_ofs + jobs[i].npts > total_pts)
            return fail(c, "track: job %d point range out of bounds", i);
        slots[i] = jobs[i].next_slot;
    }
    // 1. pyramids of the new left images (enqueue only)
    {
        const bool dec = c->src_w > 0;
        if (pyramid_common(c, njobs, slots.data(), next_imgs, strides, src_is_device, dec, dec ? c->src_w : c->geom.w[0],
                           dec ? c->src_h : c->geom.h[0], false)) return -1;
    }
    // 2. LK + pose-only back to back, single readback.  The arena region used by the
    //    pyramid job array stays live until the stream drains, so continue after it.
    int maxn = 0;
    for (int i = 0; i < njobs; ++i) maxn = std::max(maxn, jobs[i].npts);
    size_t base = c->ar.off;
    size_t olk = c->ar.take(sizeof(LkJob) * njobs);
    size_t ocam = c->ar.take(32);
    size_t oprev = c->ar.take(sizeof(float) * 2 * std::max(total_pts, 1));
    size_t omp = c->ar.take(std::max(total_pts, 1));
    size_t oxyz = c->ar.take(sizeof(double) * 3 * std::max(total_pts, 1));
    size_t opj = c->ar.take(sizeof(PoseJob) * njobs);
    size_t onext = c->ar.take(sizeof(float) * 2 * std::max(total_pts, 1));
    size_t in_end = c->ar.off;
    size_t ostat = c->ar.take(std::max(total_pts, 1));
    size_t oerr = c->ar.take(sizeof(float) * std::max(total_pts, 1));
    size_t oval = c->ar.take(std::max(total_pts, 1));
    size_t oout = c->ar.take(std::max(total_pts, 1));
    size_t ontr = c->ar.take(sizeof(int) * njobs);
    LkJob *lj = hp<LkJob>(c, olk);
    PoseJob *pj = hp<PoseJob>(c, opj);
    for (int i = 0; i < njobs; ++i) {
        lj[i].prev_slot = jobs[i].prev_slot; lj[i].next_slot = jobs[i].next_slot;
        lj[i].pt_ofs = jobs[i].pt_ofs; lj[i].npts = jobs[i].npts;
        pj[i].pt_ofs = jobs[i].pt_ofs; pj[i].npts = jobs[i].npts;
        memcpy(pj[i].pose, jobs[i].pose, 56);
        pj[i].n_inlier = 0; pj[i].pad = 0;
    }
    memcpy(hp<void>(c, ocam), cam, 32);
    if (total_pts > 0) {
        memcpy(hp<void>(c, oprev), prev_xy, sizeof(float) * 2 * total_pts);
        memcpy(hp<void>(c, onext), next_xy, sizeof(float) * 2 * total_pts);
        memcpy(hp<void>(c, omp), has_mp, total_pts);
        memcpy(hp<void>(c, oxyz), xyz, sizeof(double) * 3 * total_pts);
    }
    if (h2d(c, base, in_end)) return -1;
    if (maxn > 0) {
        tm_begin(c, FAM_LK, total_pts);
        launch_lk(c, njobs, maxn, dp<LkJob>(c, olk), dp<float2>(c, oprev), dp<float2>(c, onext), dp<uint8_t>(c, ostat),
                  dp<float>(c, oerr), p);
        tm_end(c);
        // status && in image && has map point -> pose-only edge (src/frontend.cpp:361-371, 443-444)
        hipLaunchKernelGGL(k_track_filter, dim3(njobs), dim3(256), 0, c->stream,
                           reinterpret_cast<const LkJobView *>(dp<LkJob>(c, olk)), dp<float2>(c, onext),
                           dp<uint8_t>(c, ostat), dp<uint8_t>(c, omp), dp<uint8_t>(c, oval), dp<int>(c, ontr),
                           c->geom.w[0], c->geom.h[0]);
    }
    tm_begin(c, FAM_POSE, njobs);
    launch_pose_only(c, njobs, dp<PoseJob>(c, opj), dp<double>(c, ocam), dp<double>(c, oxyz), dp<float2>(c, onext),
                     dp<uint8_t>(c, oval), dp<uint8_t>(c, oout), chi2_th, 4, 10);
    tm_end(c);
    HIPCHK(c, hipGetLastError());
    // readback: pose jobs .. n_tracked (next_xy sits between; one contiguous copy)
    if (d2h_sync(c, opj, c->ar.off)) return -1;
    const int *ntr = hp<int>(c, ontr);
    for (int i = 0; i < njobs; ++i) {
        memcpy(jobs[i].pose, pj[i].pose, 56);
        jobs[i].n_inlier = pj[i].n_inlier;
        jobs[i].n_tracked = maxn > 0 ? ntr[i] : 0;
    }
    if (total_pts > 0) {
        memcpy(next_xy, hp<void>(c, onext), sizeof(float) * 2 * total_pts);
        memcpy(status, hp<void>(c, ostat), total_pts);
        memcpy(outlier, hp<void>(c, oout), total_pts);
    }
    return 0;
}

int svslam_rtrack_batch(svslam_ctx *c, int njobs, svslam_rtrack_job *jobs, const void *const *next_imgs,
                        const int *strides, int src_is_device, int total_pts, const double cam[4],
                        float *out_xy, int *out_mp, const svslam_lk_params *p, double chi2_th)
{
    if (njobs <= 0) return 0;
    if (c->rt.max_pts <= 0) return fail(c, "rtrack: context created with max_streams = 0");
    if (njobs > c->lim.max_jobs) return fail(c, "rtrack: %d jobs > max_jobs", njobs);
    if (total_pts > c->lim.max_jobs * c->lim.max_pts) return fail(c, "rtrack: too many points");
    std::vector<int> slots(njobs);
    for (int i = 0; i < njobs; ++i) {
        const svslam_rtrack_job &j = jobs[i];
        if (check_slot(c, j.prev_slot) || check_slot(c, j.next_slot)) return -1;
        if (j.stream < 0 || j.stream >= c->lim.max_streams) return fail(c, "rtrack: job %d stream %d out of range", i, j.stream);
        if (j.npts != c->rt_count[(size_t)j.stream])
            return fail(c, "rtrack: job %d says %d features, stream %d holds %d", i, j.npts, j.stream, c->rt_count[(size_t)j.stream]);
        if (j.pt_ofs < 0 || j.pt_ofs + j.npts > total_pts) return fail(c, "rtrack: job %d point range out of bounds", i);
        slots[i] = j.next_slot;
    }
    int maxn = 0;
    for (int i = 0; i < njobs; ++i) maxn = std::max(maxn, jobs[i].npts);
    const size_t T = (size_t)std::max(total_pts, 1);
    size_t olk = 0, ocam = 0, opj = 0, ort = 0, in_end = 0, oxy = 0, ompo = 0, out_end = 0, oprev = 0, onext = 0, omp = 0, oxyz = 0, ostat = 0,
           oerr = 0, oout = 0;
    LkJob *lj = nullptr;
    PoseJob *pj = nullptr;
    RtJob *rj = nullptr;
    // this call's inputs are staged behind the pyramid's jobs and before its launch, so that the gather can ride on it
    PyrMore more;
    more.stage = [&](RtGatherArgs *ga) -> int {
        const size_t base = c->ar.off;
        olk = c->ar.take(sizeof(LkJob) * njobs);
        ocam = c->ar.take(32);
        opj = c->ar.take(sizeof(PoseJob) * njobs);
        ort = c->ar.take(sizeof(RtJob) * njobs);
        in_end = c->ar.off;
        oxy = c->ar.take(sizeof(float) * 2 * T);        // compacted survivors for the host
        ompo = c->ar.take(sizeof(int) * T);
        out_end = c->ar.off;
        // device-only scratch of this call
        oprev = c->ar.take(sizeof(float) * 2 * T);
        onext = c->ar.take(sizeof(float) * 2 * T);
        omp = c->ar.take(T);
        oxyz = c->ar.take(sizeof(double) * 3 * T);
        ostat = c->ar.take(T);
        oerr = c->ar.take(sizeof(float) * T);
        oout = c->ar.take(T);
        if (c->ar.off > c->ar.cap) return fail(c, "rtrack: staging arena too small (%zu > %zu bytes)", c->ar.off, c->ar.cap);
        lj = hp<LkJob>(c, olk);
        pj = hp<PoseJob>(c, opj);
        rj = hp<RtJob>(c, ort);
        for (int i = 0; i < njobs; ++i) {
            const svslam_rtrack_job &j = jobs[i];
            lj[i].prev_slot = j.prev_slot; lj[i].next_slot = j.next_slot; lj[i].pt_ofs = j.pt_ofs; lj[i].npts = j.npts;
            pj[i].pt_ofs = j.pt_ofs; pj[i].npts = j.npts;
            memcpy(pj[i].pose, j.pose, 56);
            pj[i].n_inlier = 0; pj[i].pad = 0;
            rj[i].stream = j.stream; rj[i].pt_ofs = j.pt_ofs; rj[i].npts = j.npts; rj[i].src_buf = c->rt_which[(size_t)j.stream];
            memcpy(rj[i].T_cam_w, j.T_cam_w, 56);
            rj[i].n_tracked = rj[i].n_edges = rj[i].n_outlier = 0; rj[i].pad = 0;
        }
        memcpy(hp<void>(c, ocam), cam, 32);
        if (!c->zero_copy && h2d(c, base, in_end)) return -1;
        if (maxn > 0)
            *ga = RtGatherArgs{ dpz<RtJob>(c, ort), c->rt, dpz<double>(c, ocam), dp<float2>(c, oprev), dp<float2>(c, onext), dp<uint8_t>(c, omp),
                                dp<double>(c, oxyz), njobs, cdiv(maxn, PF_THREADS) };
        return 0;
    };
    {
        const bool dec = c->src_w > 0;
        if (pyramid_common(c, njobs, slots.data(), next_imgs, strides, src_is_device, dec, dec ? c->src_w : c->geom.w[0],
                           dec ? c->src_h : c->geom.h[0], false, &more)) return -1;
    }
    if (maxn > 0) {
        if (!more.gathered)
            hipLaunchKernelGGL(k_rt_gather, dim3(cdiv(maxn, 256), njobs), dim3(256), 0, c->stream, dpz<RtJob>(c, ort), c->rt,
                               dpz<double>(c, ocam), dp<float2>(c, oprev), dp<float2>(c, onext), dp<uint8_t>(c, omp), dp<double>(c, oxyz));
        tm_begin(c, FAM_LK, total_pts);
        launch_lk(c, njobs, maxn, dpz<LkJob>(c, olk), dp<float2>(c, oprev), dp<float2>(c, onext), dp<uint8_t>(c, ostat),
                  dp<float>(c, oerr), p);
        tm_end(c);
    }
    // survivor filter, pose-only LM and the hand-over of the survivors in one launch (k_geom.h: k_pose_only<.., true>)
    const PoFuse fz = { dp<uint8_t>(c, ostat), dp<uint8_t>(c, omp), c->geom.w[0], c->geom.h[0], dpz<RtJob>(c, ort), c->rt,
                        dpz<float2>(c, oxy), dpz<int>(c, ompo) };
    tm_begin(c, FAM_POSE, njobs);
    launch_pose_only(c, njobs, dpz<PoseJob>(c, opj), dpz<double>(c, ocam), dp<double>(c, oxyz), dp<float2>(c, onext),
                     nullptr, dp<uint8_t>(c, oout), chi2_th, 4, 10, &fz);
    tm_end(c);
    HIPCHK(c, hipGetLastError());
    // pose jobs, rt jobs (+ the compacted survivors unless the caller keeps its map on the device and passes no buffers);
    // zero_copy: the kernels wrote them where the host reads them, only the completion is awaited
    if (c->zero_copy ? d2h_sync(c, 0, 0) : d2h_sync(c, opj, (out_xy || out_mp) ? out_end : in_end)) return -1;
    for (int i = 0; i < njobs; ++i) {
        svslam_rtrack_job &j = jobs[i];
        memcpy(j.pose, pj[i].pose, 56);
        j.n_tracked = rj[i].n_tracked; j.n_edges = rj[i].n_edges; j.n_outlier = rj[i].n_outlier;
        c->rt_which[(size_t)j.stream] ^= 1;
        c->rt_count[(size_t)j.stream] = j.n_tracked;
    }
    if (total_pts > 0) {
        if (out_xy) memcpy(out_xy, hp<void>(c, oxy), sizeof(float) * 2 * total_pts);
        if (out_mp) memcpy(out_mp, hp<void>(c, ompo), sizeof(int) * total_pts);
    }
    return 0;
}

int svslam_rtrack_upload(svslam_ctx *c, int n, const int *streams, const int *ofs, const int *counts,
                         const float *xy, const int *mp, const double *xyz)
{
    if (n <= 0) return 0;
    if (c->rt.max_pts <= 0) return fail(c, "rtrack_upload: context created with max_streams = 0");
    if (n > c->lim.max_jobs) return fail(c, "rtrack_upload: %d streams > max_jobs", n);
    int total = 0, maxc = 0;
    for (int i = 0; i < n; ++i) {
        if (streams[i] < 0 || streams[i] >= c->lim.max_streams) return fail(c, "rtrack_upload: stream %d out of range", streams[i]);
        if (counts[i] < 0 || counts[i] > c->lim.max_pts) return fail(c, "rtrack_upload: %d features > max_pts %d", counts[i], c->lim.max_pts);
        total = std::max(total, ofs[i] + counts[i]); maxc = std::max(maxc, counts[i]);
    }
    if (arena_busy(c)) return -1;
    c->ar.reset();
    const size_t T = (size_t)std::max(total, 1);
    size_t oj = c->ar.take(sizeof(RtUpJob) * n);
    size_t oxy = c->ar.take(sizeof(float) * 2 * T);
    size_t omp = c->ar.take(sizeof(int) * T);
    size_t oxyz = c->ar.take(sizeof(double) * 3 * T);
    if (c->ar.off > c->ar.cap) return fail(c, "rtrack_upload: staging arena too small");
    RtUpJob *uj = hp<RtUpJob>(c, oj);
    for (int i = 0; i < n; ++i) {
        uj[i].stream = streams[i]; uj[i].ofs = ofs[i]; uj[i].count = counts[i]; uj[i].dst_buf = c->rt_which[(size_t)streams[i]];
        c->rt_count[(size_t)streams[i]] = counts[i];
    }
    if (total > 0) {
        memcpy(hp<void>(c, oxy), xy, sizeof(float) * 2 * total);
        memcpy(hp<void>(c, omp), mp, sizeof(int) * total);
        memcpy(hp<void>(c, oxyz), xyz, sizeof(double) * 3 * total);
    }
    if (h2d(c, 0, c->ar.off)) return -1;
    if (maxc > 0)
        hipLaunchKernelGGL(k_rt_store, dim3(cdiv(maxc, 256), n), dim3(256), 0, c->stream, dp<RtUpJob>(c, oj), c->rt,
                           dp<float2>(c, oxy), dp<int>(c, omp), dp<double>(c, oxyz));
    HIPCHK(c, hipGetLastError());
    return d2h_sync(c, 0, 0);       // the staging memory is reused by the next call
}

} // extern "C"
