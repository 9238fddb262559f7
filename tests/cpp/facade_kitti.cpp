// facade_kitti.cpp — C++ test of the drop-in facade (host/slam_facade.h): reads a KITTI-layout
// sequence directory (calib.txt, image_0/, image_1/, PNG files) written by the Python side of the
// test, runs it through VisualOdometry / Frontend::AddFrame / Backend exactly as the reference's
// run_stereo_slam would, and prints one line per frame for the Python side to compare.
// -DFACADE_ORACLE: the oracle's kernel provider (CPU test); otherwise the HIP kernels (GPU test).
#include <cstdio>
#include <cstdlib>
#ifdef FACADE_ORACLE
#include "../../oracle/kernels_oracle.h"
#include "../../stereovision-slam_amd/host/slam_facade.h"
typedef svs::OracleKernels Provider;
#else
#include "../../stereovision-slam_amd/host/slam_facade_hip.h"
typedef svs::HipKernels Provider;
#endif

using namespace svs::facade;

#define CHECK_VOID(c) do { if (!(c)) { std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: facade_kitti <config.yaml> <out_dir>\n"); return 2; }
    VisualOdometryT<Provider> vo(argv[1]);
    CHECK(vo.initialize());
    // Dataset: calib.txt -> cameras with K halved and the baseline from t (src/dataset.cpp:41-77)
    Camera::Ptr c0 = vo.dataset()->GetCamera(0), c1 = vo.dataset()->GetCamera(1);
    std::printf("cam0 %.6f %.6f %.6f %.6f %.6f\n", c0->fx, c0->fy, c0->cx, c0->cy, c0->baseline);
    std::printf("cam1 %.6f %.6f %.6f %.6f %.6f t %.6f %.6f %.6f\n", c1->fx, c1->fy, c1->cx, c1->cy, c1->baseline, c1->pose.v[4], c1->pose.v[5], c1->pose.v[6]);
    int n_kf_hook = 0, n_view_hook = 0;
    vo.frontend()->SetLoopClosure([&](const Frame::Ptr &f) { ++n_kf_hook; CHECK_VOID(f->is_keyframe_); });
    vo.frontend()->SetViewer([&](const Frame::Ptr &) { ++n_view_hook; });
    CHECK(vo.GetFrontendStatus() == FrontendStatus::INITING);
    bool said_map = false;
    int n = 0, nkf = 0;
    const int pause_from = 8, pause_to = 12;
    while (true) {
        if (n == pause_from && vo.backend()) { vo.backend()->PauseRequest(); CHECK(vo.backend()->IsPaused()); }
        if (n == pause_to && vo.backend()) { vo.backend()->Resume(); CHECK(!vo.backend()->IsPaused()); }
        if (!vo.step()) break;
        if (!said_map) {
            std::printf("map: %s\n", vo.frontend()->pipeline()->MapOnDevice() ? "device" : "host");
            // the kernel shapes the facade selected (FrontendOptions::low_latency, default 1: one camera)
#ifdef FACADE_ORACLE
            std::printf("shape: %s\n", vo.frontend()->LowLatency() ? "low-latency" : "batch");
#else
            int lim[4] = { 0, 0, 0, 0 };
            svslam_debug_ll_limits(vo.frontend()->kernels()->ctx(), lim);
            std::printf("shape: %s; local BA over %d workgroups per problem, up to %d problems per call (%d CUs x %d resident)\n",
                        vo.frontend()->LowLatency() ? "low-latency" : "batch", lim[0], lim[1], lim[2], lim[3]);
#endif
            said_map = true;
        }
        Frame::Ptr f = vo.frontend()->GetLastFrame();
        CHECK(f && f->id_ == (unsigned long)n);
        nkf += f->is_keyframe_ ? 1 : 0;
        std::printf("frame %lu status %d kf %d kfid %lu feat %d inl %d pose %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", f->id_,
                    (int)vo.GetFrontendStatus(), (int)f->is_keyframe_, f->keyframe_id_, f->n_features_, f->n_inliers_, f->pose_.v[0],
                    f->pose_.v[1], f->pose_.v[2], f->pose_.v[3], f->pose_.v[4], f->pose_.v[5], f->pose_.v[6]);
        ++n;
    }
    CHECK(n > 0 && nkf >= 1 && n_kf_hook == nkf && n_view_hook == n);
    // Map views + Backend::UpdateMap() from outside (one more local BA over the active window)
    const auto kfs = vo.map()->GetAllKeyFrames();
    const auto lms = vo.map()->GetAllMapPoints();
    CHECK((int)kfs.size() == nkf && !lms.empty());
    CHECK(vo.map()->GetActiveKeyFrames().size() <= 10 && vo.map()->GetActiveMapPoints().size() <= lms.size());
    if (vo.backend()) {
        CHECK(vo.backend()->IsRunning());
        const svs::SE3 before = kfs.back().pose;
        vo.backend()->UpdateMap();
        const svs::SE3 after = vo.map()->GetAllKeyFrames().back().pose;
        double d = 0;
        for (int i = 0; i < 7; ++i) d += std::fabs(before.v[i] - after.v[i]);
        std::printf("update_map moved the last keyframe by %.3g\n", d);
        CHECK(d > 0 && d < 0.1);
        vo.backend()->Stop();
        CHECK(!vo.backend()->IsRunning());
    }
#ifndef FACADE_ORACLE
    {   // which solver took the keyframes' local BAs: slot 6 = problems the low-latency solver took, 7 = of those, repeated by the batch solver
        long long ns[8];
        svslam_debug_host_ns(vo.frontend()->kernels()->ctx(), ns);
        std::printf("local BA: %lld problems on the low-latency solver, %lld repeated by the batch solver\n", ns[6], ns[7]);
    }
#endif
    CHECK(vo.saveSLAMOutputInFile(argv[2]));
    // the shared Backend / Map handles outlive a frontend without dangling into it, and a backend
    // attached AFTER the first frame still optimises (the reference wires them in any order before run())
    {
        std::shared_ptr<Backend> be(new Backend());
        Map::Ptr mp(new Map());
        {
            FrontendT<Provider> fe;
            fe.SetCameras(vo.dataset()->GetCamera(0), vo.dataset()->GetCamera(1));
            fe.SetMap(mp);
            Frame::Ptr f0 = vo.dataset()->FrameById(0);
            CHECK(f0 && fe.AddFrame(f0));                       // pipeline exists, no backend yet
            fe.SetBackend(be);                                    // attached late
            CHECK(fe.pipeline()->BackendEnabled());
            be->PauseRequest();
            CHECK(!fe.pipeline()->BackendEnabled());
            be->Resume();
            CHECK(mp->GetAllKeyFrames().size() == 1);
        }
        be->UpdateMap();                                          // frontend gone: a no-op, not a dangling call
        CHECK(mp->GetAllKeyFrames().empty() && mp->GetAllMapPoints().empty());
    }
    // corrupt image files give an empty image like cv::imread, never an out-of-bounds read or a huge allocation
    {
        Image img;
        std::vector<uint8_t> sig = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
        auto chunk = [](std::vector<uint8_t> &f, const char *ty, const std::vector<uint8_t> &d) {
            const uint32_t n = (uint32_t)d.size();
            f.push_back(n >> 24); f.push_back(n >> 16); f.push_back(n >> 8); f.push_back(n);
            f.insert(f.end(), ty, ty + 4); f.insert(f.end(), d.begin(), d.end());
            f.insert(f.end(), 4, 0);                             // CRC (not checked)
        };
        std::vector<uint8_t> a = sig;                             // IHDR shorter than 13 bytes
        chunk(a, "IHDR", std::vector<uint8_t>(5, 0)); chunk(a, "IDAT", std::vector<uint8_t>(16, 1)); chunk(a, "IEND", {});
        CHECK(!read_png(a, img));
        std::vector<uint8_t> b = sig;                             // first chunk is not IHDR
        chunk(b, "IDAT", std::vector<uint8_t>(16, 1)); chunk(b, "IHDR", std::vector<uint8_t>(13, 0)); chunk(b, "IEND", {});
        CHECK(!read_png(b, img));
        std::vector<uint8_t> c = sig;                             // absurd size: 2^31 x 2^31
        chunk(c, "IHDR", { 0x80, 0, 0, 0, 0x80, 0, 0, 0, 8, 0, 0, 0, 0 }); chunk(c, "IDAT", std::vector<uint8_t>(16, 1)); chunk(c, "IEND", {});
        CHECK(!read_png(c, img));
        const std::string bad = std::string(argv[2]) + "/corrupt.png";
        { std::ofstream o(bad, std::ios::binary); o.write(reinterpret_cast<const char *>(c.data()), (long)c.size()); }
        CHECK(imread(bad).empty());
    }
    std::printf("frames %d keyframes %d landmarks %zu\nfacade ok\n", n, nkf, lms.size());
    return 0;
}
