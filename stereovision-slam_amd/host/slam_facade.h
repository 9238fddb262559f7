// slam_facade.h — the reference's public classes, by name and signature, over the batched host
// pipeline (slam_host.h) with ONE stream: what a user of farhad-dalirani/StereoVision-SLAM links
// against instead of its Frontend / Backend / Dataset / VisualOdometry when the hot path moves to
// the GPU.  Same method names, argument meaning and return values:
//
//   Frontend   include/StereoVisionSLAM/frontend.h:36-44   SetMap SetBackend SetLoopClosure SetViewer
//                                                           SetCameras GetLastFrame GetStatus AddFrame
//   Backend    include/StereoVisionSLAM/backend.h:28-43    UpdateMap Stop PauseRequest IsPaused
//                                                           IsRunning Resume SetMap SetCameras
//   Dataset    src/dataset.cpp:11-138                      initialize (calib.txt -> 4 cameras, K *= 0.5,
//                                                           baseline = |K^-1 t|)  NextFrame  FrameById
//                                                           GetCamera GetDataDir GetLeftCamIndex
//   VisualOdometry  src/visual_odometry.cpp:24-224          initialize step run GetFrontendStatus
//                                                           saveSLAMOutputInFile
//   Map        include/StereoVisionSLAM/map.h               read-only views (keyframes, landmarks)
//
// Differences, all forced by the missing third-party libraries: cv::Mat becomes facade::Image (u8, one
// channel), Sophus::SE3d becomes svs::SE3 (same 7 doubles), Frame::Ptr / Camera::Ptr stay shared_ptrs.
// The loop-closure and viewer objects of the reference are callbacks here (SetLoopClosure / SetViewer):
// they receive exactly what src/frontend.cpp:631-640 hands them, the new keyframe.  The backend
// "thread" is the pipeline's deterministic schedule: Backend::UpdateMap() from outside
// optimises at once.  The 1/2 INTER_NEAREST resize of Dataset::NextFrame (src/dataset.cpp:126-129)
// is not done on the host: frames keep their full resolution and the decimation is fused into the
// pyramid's level 0 on the device.
//
// Template over the kernel provider: the product instantiates HipKernels (kernels_hip.h); the test
// infrastructure instantiates the oracle's provider to test this file without a GPU.
#pragma once
#include <zlib.h>

#include <cstdio>
#include <fstream>
#include <functional>
#include <iomanip>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "slam_host.h"

namespace svs {
namespace facade {

struct SLAMException : std::runtime_error { using std::runtime_error::runtime_error; };   // slamexception.h

struct Image {                       // stand-in for cv::Mat CV_8UC1
    int cols = 0, rows = 0;
    std::vector<uint8_t> data;
    bool empty() const { return data.empty(); }
};

// ---- image files: 8-bit PNG (grey, grey+alpha, RGB, RGBA; non-interlaced) through zlib, and binary PGM.
// cv::imread(path, IMREAD_GRAYSCALE): colour becomes grey with OpenCV's fixed-point BT.601 weights.
inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline bool read_png(const std::vector<uint8_t> &f, Image &img)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (f.size() < 33 || std::memcmp(f.data(), sig, 8) != 0) return false;
    size_t p = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> z;
    bool first = true;
    while (p + 12 <= f.size()) {
        const uint32_t len = be32(&f[p]);
        const char *ty = reinterpret_cast<const char *>(&f[p + 4]);
        if (p + 12 + (size_t)len > f.size()) return false;
        const uint8_t *d = &f[p + 8];
        const bool ihdr = !std::memcmp(ty, "IHDR", 4);
        if (first != ihdr) return false;                       // IHDR is the first chunk and appears once
        first = false;
        if (ihdr) {
            if (len != 13) return false;
            w = be32(d); h = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
        }
        else if (!std::memcmp(ty, "IDAT", 4)) z.insert(z.end(), d, d + len);
        else if (!std::memcmp(ty, "IEND", 4)) break;
        p += 12 + (size_t)len;
    }
    // a corrupt file yields an empty image (what cv::imread returns), never an oversized allocation
    if (!w || !h || w > 16384 || h > 16384 || depth != 8 || interlace != 0 || z.empty()) return false;
    const int ch = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
    if (!ch) return false;
    const size_t stride = (size_t)w * ch;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, z.data(), (uLong)z.size()) != Z_OK || rawlen != raw.size()) return false;
    std::vector<uint8_t> cur(stride), prev(stride, 0);
    img.cols = (int)w; img.rows = (int)h; img.data.assign((size_t)w * h, 0);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *r = &raw[(stride + 1) * y];
        const int ft = r[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
            int v = r[1 + i];
            if (ft == 1) v += a;
            else if (ft == 2) v += b;
            else if (ft == 3) v += (a + b) >> 1;
            else if (ft == 4) { const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) return false;
            cur[i] = (uint8_t)v;
        }
        uint8_t *o = &img.data[(size_t)y * w];
        if (ch <= 2) for (uint32_t x = 0; x < w; ++x) o[x] = cur[(size_t)x * ch];
        else for (uint32_t x = 0; x < w; ++x) {       // cv::cvtColor RGB2GRAY: (R*4899 + G*9617 + B*1868 + 8192) >> 14
            const uint8_t *q = &cur[(size_t)x * ch];
            o[x] = (uint8_t)((q[0] * 4899 + q[1] * 9617 + q[2] * 1868 + 8192) >> 14);
        }
        prev.swap(cur);
    }
    return true;
}
inline bool read_pgm(const std::vector<uint8_t> &f, Image &img)
{
    if (f.size() < 7 || f[0] != 'P' || f[1] != '5') return false;
    size_t p = 2;
    int vals[3], n = 0;
    while (n < 3 && p < f.size()) {
        while (p < f.size() && std::isspace(f[p])) ++p;
        if (p < f.size() && f[p] == '#') { while (p < f.size() && f[p] != '\n') ++p; continue; }
        int v = 0; bool any = false;
        while (p < f.size() && std::isdigit(f[p])) { v = v * 10 + (f[p] - '0'); ++p; any = true; }
        if (!any) return false;
        vals[n++] = v;
    }
    ++p;                                         // the single whitespace after maxval
    if (n < 3 || vals[2] != 255 || vals[0] <= 0 || vals[1] <= 0 || vals[0] > 16384 || vals[1] > 16384 ||
        p + (size_t)vals[0] * vals[1] > f.size()) return false;
    img.cols = vals[0]; img.rows = vals[1];
    img.data.assign(f.begin() + (long)p, f.begin() + (long)p + (long)vals[0] * vals[1]);
    return true;
}
inline Image imread(const std::string &path)     // empty image when the file is missing or not understood, like cv::imread
{
    Image img;
    std::ifstream in(path, std::ios::binary);
    if (!in) return img;
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    try {
        if (!read_png(f, img) && !read_pgm(f, img)) img = Image();
    } catch (const std::exception &) { img = Image(); }        // bad_alloc / length_error on a corrupt header
    return img;
}

// ---- Config (src/config.cpp): the scalar keys of an OpenCV FileStorage YAML
class ConfigFile {
public:
    bool Load(const std::string &path)
    {
        std::ifstream in(path);
        if (!in) return false;
        std::string line;
        while (std::getline(in, line)) {
            const size_t h = line.find('#');
            if (h != std::string::npos) line.erase(h);
            const size_t c = line.find(':');
            if (c == std::string::npos || (!line.empty() && line[0] == '%')) continue;
            auto trim = [](std::string s) { const size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r"); return a == std::string::npos ? std::string() : s.substr(a, b - a + 1); };
            std::string val = trim(line.substr(c + 1));
            if (val.size() >= 2 && (val.front() == '"' || val.front() == '\'') && val.back() == val.front())
                val = val.substr(1, val.size() - 2);           // cv::FileStorage accepts quoted strings
            kv_.emplace_back(trim(line.substr(0, c)), val);
        }
        return true;
    }
    std::string Str(const std::string &k, const std::string &def = "") const { for (auto &e : kv_) if (e.first == k) return e.second; return def; }
    double Num(const std::string &k, double def) const { const std::string s = Str(k); return s.empty() ? def : std::atof(s.c_str()); }
private:
    std::vector<std::pair<std::string, std::string>> kv_;
};

// ---- Camera (include/StereoVisionSLAM/camera.h)
class Camera : public svs::Camera {
public:
    typedef std::shared_ptr<Camera> Ptr;
    Camera() {}
    Camera(double fx_, double fy_, double cx_, double cy_, double baseline_, const SE3 &pose_)
    { fx = fx_; fy = fy_; cx = cx_; cy = cy_; baseline = baseline_; pose = pose_; }
};

// ---- Frame (include/StereoVisionSLAM/frame.h)
class Frame {
public:
    typedef std::shared_ptr<Frame> Ptr;
    unsigned long id_ = 0, keyframe_id_ = 0;
    bool is_keyframe_ = false;
    SE3 pose_;                                    // T_cw
    Image left_img_, right_img_;
    double time_stamp_ = 0;
    int n_features_ = 0, n_inliers_ = 0;          // filled by AddFrame (sizes of feature_left_ / inlier count)
    SE3 Pose() const { return pose_; }
    void SetPose(const SE3 &p) { pose_ = p; }
    static Ptr CreateFrame()                      // src/frame.cpp:22-28
    {
        static unsigned long factory_id = 0;
        Ptr f(new Frame());
        f->id_ = factory_id++;
        return f;
    }
};

// ---- Map (read-only mirror of include/StereoVisionSLAM/map.h for consumers of the facade)
struct KeyframeView { unsigned long id, keyframe_id; SE3 pose; bool active; };
struct LandmarkView { unsigned long id; double pos[3]; int observed_times; bool active; };
class Map {
public:
    typedef std::shared_ptr<Map> Ptr;
    std::function<std::vector<KeyframeView>()> keyframes_fn;
    std::function<std::vector<LandmarkView>()> landmarks_fn;
    std::vector<KeyframeView> GetAllKeyFrames() const { return keyframes_fn ? keyframes_fn() : std::vector<KeyframeView>(); }
    std::vector<KeyframeView> GetActiveKeyFrames() const { std::vector<KeyframeView> o; for (auto &k : GetAllKeyFrames()) if (k.active) o.push_back(k); return o; }
    std::vector<LandmarkView> GetAllMapPoints() const { return landmarks_fn ? landmarks_fn() : std::vector<LandmarkView>(); }
    std::vector<LandmarkView> GetActiveMapPoints() const { std::vector<LandmarkView> o; for (auto &l : GetAllMapPoints()) if (l.active) o.push_back(l); return o; }
};

using FrontendStatus = svs::FrontendStatus;

// ---- Backend (include/StereoVisionSLAM/backend.h:28-43): control handle of the pipeline's backend
class Backend {
public:
    typedef std::shared_ptr<Backend> Ptr;
    Backend() : running_(true), pause_request_(false) {}
    void UpdateMap() { if (running_ && !pause_request_ && optimize_now_) optimize_now_(); }
    void Stop() { running_ = false; apply(); }
    void PauseRequest() { pause_request_ = true; apply(); }
    bool IsPaused() const { return pause_request_; }       // the pipeline's backend stops at the request itself
    bool IsRunning() const { return running_; }
    void Resume() { pause_request_ = false; apply(); }
    void SetMap(Map::Ptr m) { map_ = m; }
    void SetCameras(Camera::Ptr l, Camera::Ptr r) { cam_left_ = l; cam_right_ = r; }
    // wiring (Frontend::SetBackend)
    std::function<void()> optimize_now_;
    std::function<void(bool)> set_enabled_;
    bool enabled() const { return running_ && !pause_request_; }
private:
    void apply() { if (set_enabled_) set_enabled_(enabled()); }
    bool running_, pause_request_;
    Map::Ptr map_;
    Camera::Ptr cam_left_, cam_right_;
};

// ---- Frontend (include/StereoVisionSLAM/frontend.h:36-44)
struct FrontendOptions {            // the hyper-parameters Frontend::Frontend() reads from Config (src/frontend.cpp:10-34)
    int num_features = 150, num_features_init = 50, num_features_tracking = 50, num_features_tracking_bad = 20;
    int num_features_needed_for_keyframe = 80;
    double max_triangulation_depth = 300.0;
    int num_active_keyframes = 10;
    double chi2_th = 5.991;
    int device = 0;
    // 1 (default): the stream's map — window, keyframe features, landmarks, observation counts — lives in device memory
    // and a keyframe is ONE chain of kernels (svslam_dmap_keyframe_batch), the configuration every throughput figure of
    // this library uses; 0: the host keeps it (rounds 1-2).  Same results bit for bit (tests/test_facade_kitti.py).
    // Not a key of the reference's config files: `device_map: 0` in the YAML (or this field) selects the host map.
    // A kernel provider without a device map (the CPU twin of the tests) ignores it.
    int device_map = 1;
    // Kernel shapes for ONE camera (svslam_set_low_latency): pose-only LM on four waves, a keyframe's local BA dealt over
    // up to 16 workgroups.  The facade is one stream by construction, like the reference's VisualOdometry::Step
    // (src/visual_odometry.cpp:109-156), so this is its default; `low_latency: 0` in the YAML (or this field) selects the
    // batch shapes (least total work — what a host that runs hundreds of streams through one context wants).  The two
    // shapes sum in different orders: results agree to rounding, each is deterministic.
    int low_latency = 1;
    // Parameter tolerance of the pose-only LM (svslam_set_pose_only_xtol, include/svslam.h): a round of EstimateCurrentPose ends at
    // a stationary point instead of spending the rest of its ten iterations on trials that move the pose by rounding noise.
    // `pose_xtol: 0` in the YAML (or this field) = g2o's schedule to the last trial (src/frontend.cpp:482-493 as written);
    // 1e-9 trades the last digits of the pose for a fifth of the kernel's time.
    double pose_xtol = 1e-12;
    static FrontendOptions FromConfig(const ConfigFile &c)
    {
        FrontendOptions o;
        o.device_map = (int)c.Num("device_map", o.device_map);
        o.low_latency = (int)c.Num("low_latency", o.low_latency);
        o.pose_xtol = c.Num("pose_xtol", o.pose_xtol);
        o.num_features = (int)c.Num("num_features", o.num_features);
        o.num_features_init = (int)c.Num("num_features_init", o.num_features_init);
        o.num_features_tracking = (int)c.Num("num_features_tracking", o.num_features_tracking);
        o.num_features_tracking_bad = (int)c.Num("num_features_tracking_bad", o.num_features_tracking_bad);
        o.num_features_needed_for_keyframe = (int)c.Num("num_features_needed_for_keyframe", o.num_features_needed_for_keyframe);
        o.max_triangulation_depth = c.Num("max_triangulation_depth", o.max_triangulation_depth);
        o.num_active_keyframes = (int)c.Num("num_active_keyframes", o.num_active_keyframes);
        o.chi2_th = c.Num("chi2_th", o.chi2_th);
        if (c.Str("keypoint_feature_detector", "GFTT") != "GFTT") throw SLAMException("Only the GFTT keypoint detector is on the GPU path");
        return o;
    }
};

// does the kernel provider offer a device-resident map (HipKernels::kHasDeviceMap)?  The CPU twin's provider does not.
template <class K> constexpr auto provider_has_device_map(int) -> decltype(K::kHasDeviceMap) { return K::kHasDeviceMap; }
template <class K> constexpr bool provider_has_device_map(long) { return false; }

template <class K>
class FrontendT {
public:
    typedef std::shared_ptr<FrontendT> Ptr;
    explicit FrontendT(const FrontendOptions &opt = FrontendOptions()) : opt_(opt) {}
    ~FrontendT() { unwire(); }                    // the shared Backend / Map handles may outlive the frontend
    FrontendT(const FrontendT &) = delete;
    FrontendT &operator=(const FrontendT &) = delete;
    void SetMap(Map::Ptr map) { if (map_) { map_->keyframes_fn = nullptr; map_->landmarks_fn = nullptr; } map_ = map; wire_map(); }
    // The pipeline is created at the first AddFrame with local BA compiled in; a backend handle may be attached,
    // replaced or removed at any time (a frontend without one runs without optimisation, src/frontend.cpp:618-622).
    void SetBackend(std::shared_ptr<Backend> backend)
    {
        if (backend_) { backend_->optimize_now_ = nullptr; backend_->set_enabled_ = nullptr; }
        backend_ = backend;
        wire_backend();
    }
    // the reference's LoopClosure::AddNewKeyFrame / Viewer::UpdateMap call sites (src/frontend.cpp:631-640)
    void SetLoopClosure(std::function<void(const Frame::Ptr &)> on_new_keyframe) { loopclosure_ = std::move(on_new_keyframe); }
    void SetViewer(std::function<void(const Frame::Ptr &)> on_frame) { viewer_ = std::move(on_frame); }
    void SetCameras(Camera::Ptr left, Camera::Ptr right) { camera_left_ = left; camera_right_ = right; }
    Frame::Ptr GetLastFrame() { return last_frame_; }
    FrontendStatus GetStatus() const { return status_; }

    // Update Frontend when new frame (src/frontend.cpp:690-721); true like the reference's Track()
    bool AddFrame(Frame::Ptr frame)
    {
        if (!frame || frame->left_img_.empty() || frame->right_img_.empty()) return false;
        if (!pipe_) create(frame->left_img_.cols, frame->left_img_.rows);
        if (frame->left_img_.cols != src_w_ || frame->left_img_.rows != src_h_ || frame->right_img_.cols != src_w_ ||
            frame->right_img_.rows != src_h_)
            throw SLAMException("frame size changed");
        current_frame_ = frame;
        const void *l = frame->left_img_.data.data(), *r = frame->right_img_.data.data();
        FrameResult res;
        pipe_->step(&l, &r, nullptr, 0, &res);
        frame->pose_ = SE3(res.pose);
        frame->is_keyframe_ = res.is_keyframe != 0;
        if (frame->is_keyframe_) frame->keyframe_id_ = (unsigned long)res.keyframe_id;
        frame->n_features_ = res.n_features; frame->n_inliers_ = res.n_inliers;
        status_ = (FrontendStatus)res.status;
        if (frame->is_keyframe_ && loopclosure_) loopclosure_(frame);
        if (viewer_) viewer_(frame);
        last_frame_ = current_frame_;
        return true;
    }
    Pipeline<K> *pipeline() { return pipe_.get(); }
    K *kernels() { return kernels_.get(); }
    bool LowLatency() const { return opt_.low_latency != 0; }

private:
    void create(int w, int h)
    {
        if (!camera_left_ || !camera_right_) throw SLAMException("Frontend: SetCameras before the first AddFrame");
        src_w_ = w; src_h_ = h;
        // Dataset halves K (src/dataset.cpp:73); the images arrive at full resolution and are halved on the device
        const int dw = (int)std::nearbyint(w * 0.5), dh = (int)std::nearbyint(h * 0.5);
        Config cfg;
        cfg.num_features = opt_.num_features; cfg.num_features_init = opt_.num_features_init;
        cfg.num_features_tracking = opt_.num_features_tracking; cfg.num_features_tracking_bad = opt_.num_features_tracking_bad;
        cfg.num_features_needed_for_keyframe = opt_.num_features_needed_for_keyframe;
        cfg.max_triangulation_depth = opt_.max_triangulation_depth;
        cfg.num_active_keyframes = opt_.num_active_keyframes; cfg.chi2_th = opt_.chi2_th;
        cfg.backend_on = 1;                        // gated by SetBackendEnabled: see SetBackend
        cfg.width = dw; cfg.height = dh; cfg.src_width = w; cfg.src_height = h;
        const bool dmap = opt_.device_map != 0 && provider_has_device_map<K>(0) && cfg.num_active_keyframes + 1 <= 12;
        cfg.resident_track = dmap ? 1 : 0;         // host map: the host keeps the feature lists; device map: nothing per feature on the host
        cfg.device_map = dmap ? 1 : 0;
        cfg.cam_l = *camera_left_; cfg.cam_r = *camera_right_;
        cfg.max_pts = 512; cfg.max_kf = cfg.num_active_keyframes + 1; cfg.max_lm = 4096; cfg.max_obs = 16384;
        svslam_limits lim;
        std::memset(&lim, 0, sizeof(lim));
        lim.device = opt_.device; lim.width = dw; lim.height = dh; lim.max_slots = 3; lim.max_jobs = 2;
        lim.max_pts = cfg.max_pts; lim.max_corners = cfg.num_features;
        lim.max_kf = cfg.max_kf; lim.max_lm = cfg.max_lm; lim.max_obs = cfg.max_obs;
        lim.max_streams = dmap ? 1 : 0; lim.device_map = dmap ? 1 : 0;
        kernels_.reset(new K(lim));
        if (kernels_->set_source_size(w, h) != 0) throw SLAMException(std::string("source size: ") + kernels_->last_error());
        if (kernels_->set_low_latency(opt_.low_latency ? 1 : 0) != 0) throw SLAMException(std::string("low-latency shapes: ") + kernels_->last_error());
        if (kernels_->set_pose_only_xtol(opt_.pose_xtol) != 0) throw SLAMException(std::string("pose_xtol: ") + kernels_->last_error());
        pipe_.reset(new Pipeline<K>(cfg, *kernels_, 1, 1));
        wire_backend(); wire_map();
    }
    void unwire()
    {
        if (backend_) { backend_->optimize_now_ = nullptr; backend_->set_enabled_ = nullptr; }
        if (map_) { map_->keyframes_fn = nullptr; map_->landmarks_fn = nullptr; }
    }
    void wire_backend()
    {
        if (!backend_) { if (pipe_) pipe_->SetBackendEnabled(false); return; }
        backend_->optimize_now_ = [this]() { if (pipe_) pipe_->OptimizeNow(); };
        backend_->set_enabled_ = [this](bool on) { if (pipe_) pipe_->SetBackendEnabled(on); };
        if (pipe_) pipe_->SetBackendEnabled(backend_->enabled());
    }
    void wire_map()
    {
        if (!map_) return;
        map_->keyframes_fn = [this]() {
            std::vector<KeyframeView> o;
            if (!pipe_) return o;
            auto &m = pipe_->stream(0).map;
            for (const svs::Frame *f : m.keyframes_) {
                bool act = false;
                for (const svs::Frame *a : m.active_keyframes_) act |= (a == f);
                o.push_back(KeyframeView{ (unsigned long)f->id, (unsigned long)f->keyframe_id, f->pose, act });
            }
            return o;
        };
        map_->landmarks_fn = [this]() {
            std::vector<LandmarkView> o;
            if (!pipe_) return o;
            for (const svs::LandmarkRecord &p : pipe_->AllLandmarks(0))
                o.push_back(LandmarkView{ (unsigned long)p.id, { p.pos[0], p.pos[1], p.pos[2] }, p.observed_times, p.active });
            return o;
        };
    }
    FrontendOptions opt_;
    FrontendStatus status_ = FrontendStatus::INITING;
    Frame::Ptr current_frame_, last_frame_;
    Camera::Ptr camera_left_, camera_right_;
    Map::Ptr map_;
    std::shared_ptr<Backend> backend_;
    std::function<void(const Frame::Ptr &)> loopclosure_, viewer_;
    std::unique_ptr<K> kernels_;
    std::unique_ptr<Pipeline<K>> pipe_;
    int src_w_ = 0, src_h_ = 0;
};

// ---- Dataset (src/dataset.cpp)
class Dataset {
public:
    typedef std::shared_ptr<Dataset> Ptr;
    explicit Dataset(const std::string &dataset_path, int left_cam_index = 0, int right_cam_index = 1)
        : dataset_path_(dataset_path), left_cam_index_(left_cam_index), right_cam_index_(right_cam_index) {}

    bool initialize()                             // :24-86
    {
        std::ifstream fin(dataset_path_ + "/calib.txt");
        if (!fin) throw SLAMException("Cannot open KITTI camera parameters file (calib.txt).");
        cameras_.clear();
        for (int i = 0; i < 4; ++i) {
            char cam_name[3];
            fin >> cam_name[0] >> cam_name[1] >> cam_name[2];
            double pr[12];
            for (int j = 0; j < 12; ++j) fin >> pr[j];
            if (!fin) throw SLAMException("calib.txt: expected four 3x4 projection matrices");
            // rectified: P = [K | K t]  ->  t = K^-1 (p03, p13, p23); K is upper triangular
            const double fx = pr[0], sk = pr[1], cx = pr[2], fy = pr[5], cy = pr[6], k22 = pr[10];
            const double b2 = pr[11] / k22;
            const double b1 = (pr[7] - cy * b2) / fy;
            const double b0 = (pr[3] - sk * b1 - cx * b2) / fx;
            const double baseline = std::sqrt(b0 * b0 + b1 * b1 + b2 * b2);
            SE3 pose;
            pose.v[4] = b0; pose.v[5] = b1; pose.v[6] = b2;
            // K *= 0.5: the pipeline works on the images down-sampled by two (:73)
            cameras_.push_back(Camera::Ptr(new Camera(0.5 * fx, 0.5 * fy, 0.5 * cx, 0.5 * cy, baseline, pose)));
        }
        current_image_index_ = 0;
        return true;
    }
    Camera::Ptr GetCamera(int camera_id) const { return cameras_.at((size_t)camera_id); }
    std::string GetDataDir() const { return dataset_path_; }
    int GetLeftCamIndex() const { return left_cam_index_; }

    Frame::Ptr NextFrame()                        // :104-138 (the resize happens on the device, see the header)
    {
        Frame::Ptr f = FrameById((unsigned long)current_image_index_);
        if (f) ++current_image_index_;
        return f;
    }
    Frame::Ptr FrameById(unsigned long frame_id)  // :140-176
    {
        Image l = load(left_cam_index_, frame_id), r = load(right_cam_index_, frame_id);
        if (l.empty() || r.empty()) return nullptr;
        Frame::Ptr f = Frame::CreateFrame();
        f->left_img_ = std::move(l); f->right_img_ = std::move(r);
        return f;
    }

private:
    Image load(int cam, unsigned long id) const
    {
        std::ostringstream base;
        base << dataset_path_ << "/image_" << cam << "/" << std::setw(6) << std::setfill('0') << id;
        Image img = imread(base.str() + ".png");
        if (img.empty()) img = imread(base.str() + ".pgm");
        return img;
    }
    std::string dataset_path_;
    int left_cam_index_, right_cam_index_;
    std::vector<Camera::Ptr> cameras_;
    int current_image_index_ = 0;
};

// ---- VisualOdometry (src/visual_odometry.cpp): the wiring of the classes above
template <class K>
class VisualOdometryT {
public:
    explicit VisualOdometryT(const std::string &config_file_path) : config_file_path_(config_file_path) {}
    bool initialize()                             // :24-107
    {
        if (!config_.Load(config_file_path_)) return false;
        dataset_.reset(new Dataset(config_.Str("dataset_dir"), (int)config_.Num("left_cam_index", 0), (int)config_.Num("right_cam_index", 1)));
        if (!dataset_->initialize()) return false;
        frontend_.reset(new FrontendT<K>(FrontendOptions::FromConfig(config_)));
        map_.reset(new Map());
        if ((int)config_.Num("backend_on", 1) != 0) backend_.reset(new Backend());
        frontend_->SetMap(map_);
        frontend_->SetBackend(backend_);
        frontend_->SetCameras(dataset_->GetCamera(0), dataset_->GetCamera(1));   // src/visual_odometry.cpp:73: always cameras 0 / 1; only the image folders follow left/right_cam_index
        if (backend_) { backend_->SetMap(map_); backend_->SetCameras(dataset_->GetCamera(0), dataset_->GetCamera(1)); }
        return true;
    }
    bool step()                                   // :109-146
    {
        Frame::Ptr f = dataset_->NextFrame();
        if (!f) return false;
        return frontend_->AddFrame(f);
    }
    void run()                                    // :148-180
    {
        while (step()) {}
        if (backend_) backend_->Stop();
        saveSLAMOutputInFile();
    }
    FrontendStatus GetFrontendStatus() const { return frontend_->GetStatus(); }
    // keyframes.txt + landmarks.pcd under output_dir (:198-310; the reference adds a time-stamped folder)
    bool saveSLAMOutputInFile(const std::string &dir_override = "")
    {
        const std::string dir = dir_override.empty() ? config_.Str("output_dir", ".") : dir_override;
        if (!frontend_->pipeline()) return false;
        frontend_->pipeline()->Flush();
        return frontend_->pipeline()->SaveOutputs(0, dir, dataset_->GetDataDir(), dataset_->GetLeftCamIndex());
    }
    std::shared_ptr<FrontendT<K>> frontend() { return frontend_; }
    std::shared_ptr<Backend> backend() { return backend_; }
    Map::Ptr map() { return map_; }
    Dataset::Ptr dataset() { return dataset_; }

private:
    std::string config_file_path_;
    ConfigFile config_;
    Dataset::Ptr dataset_;
    std::shared_ptr<FrontendT<K>> frontend_;
    std::shared_ptr<Backend> backend_;
    Map::Ptr map_;
};

} // namespace facade
} // namespace svs
