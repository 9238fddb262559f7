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
    int n = 0, nkf = 0;
    const int pause_from = 8, pause_to = 12;
    while (true) {
        if (n == pause_from && vo.backend()) { vo.backend()->PauseRequest(); CHECK(vo.backend()->IsPaused()); }
        if (n == pause_to && vo.backend()) { vo.backend()->Resume(); CHECK(!vo.backend()->IsPaused()); }
        if (!vo.step()) break;
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
    CHECK(vo.saveSLAMOutputInFile(argv[2]));
    std::printf("frames %d keyframes %d landmarks %zu\nfacade ok\n", n, nkf, lms.size());
    return 0;
}
