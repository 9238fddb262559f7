"""Second, independent reading of the reference's HOST GLUE (test infrastructure, never imported by the product).

Rows a8 / f1 of SURVEY.md §8 — Frontend::AddFrame / StereoInit / BuildInitMap / Track / TrackLastFrame / EstimateCurrentPose's
bookkeeping / InsertKeyframe / SetObservationsForKeyFrame / DetectFeatures / FindFeaturesInRight / TriangulateNewPoints,
Map::InsertKeyFrame / RemoveOldKeyframe / CleanMap / InsertMapPoint, MapPoint::AddObservation / RemoveObservation and the
graph gather / outlier pass / write-back of Backend::Optimize — restated in plain Python, function by function, from the files
under /root/reference (cited per function) and from nothing else: in particular NOT from the product's
`stereovision-slam_amd/host/slam_host.h`, whose reading of the same files this module exists to cross-check
(the CPU twin instantiates slam_host.h itself, so twin-vs-HIP comparisons cannot see a misreading of the
glue).  The pointer graph of the reference (shared_ptr / weak_ptr, unordered_map) is kept as a Python object graph; nothing is
flattened the way the product flattens it.

The five third-party call sites (GFTT, pyramidal LK, triangulation's SVD, the two g2o optimisations) and the SE(3) arithmetic
go to the CPU oracle (tests/oracle_lib.py) exactly as the twin's do — those are rows a1–a7, checked elsewhere; what is under
test here is everything between them.

`tests/test_glue_second_reading.py` drives this and the C++ twin over the same frames and compares, frame by frame: status,
keyframe flag, feature and inlier counts, poses (bitwise), the active window's keyframe ids, the active landmarks' ids, every
active landmark's observation list and counter."""
import numpy as np

import oracle_lib as orc

INITING, TRACKING_GOOD, TRACKING_BAD, LOST = 0, 1, 2, 3      # include/StereoVisionSLAM/frontend.h: enum class FrontendStatus

IDENT = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)          # Sophus::SE3d(): unit quaternion (x, y, z, w), zero translation


class Feature:
    """include/StereoVisionSLAM/feature.h:19-36"""
    __slots__ = ("frame", "pt", "map_point", "outlier", "is_on_left_image")

    def __init__(self, frame, pt):
        self.frame = frame                       # weak_ptr<Frame>
        self.pt = np.array(pt, np.float32)       # cv::KeyPoint::pt (Point2f)
        self.map_point = None                    # weak_ptr<MapPoint>; map points are owned by Map::landmarks_ for ever, so
        #                                          "expired" == never linked or reset
        self.outlier = False
        self.is_on_left_image = True


class Frame:
    """include/StereoVisionSLAM/frame.h:17-77, src/frame.cpp"""

    def __init__(self, fid, left, right):
        self.id = fid
        self.keyframe_id = 0
        self.is_keyframe = False
        self.pose = IDENT.copy()                 # T_cw
        self.left, self.right = left, right
        self.feature_left = []
        self.feature_right = []                  # None where the left feature has no match
        self.prev_keyframe = None
        self.relative_pose_pkf = IDENT.copy()


class MapPoint:
    """include/StereoVisionSLAM/mappoint.h:19-53, src/mappoint.cpp"""

    def __init__(self, mid):
        self.id = mid
        self.is_outlier = False
        self.pos = np.zeros(3)
        self.observed_times = 0
        self.observations = []                   # std::list<std::weak_ptr<Feature>>

    def add_observation(self, feat):             # src/mappoint.cpp:22-36
        self.observations.append(feat)
        self.observed_times += 1

    def remove_observation(self, feat):          # src/mappoint.cpp:38-78 (first_valid_obs_ serves loop closure only: left out)
        for i, o in enumerate(self.observations):
            if o is feat:
                del self.observations[i]
                if feat.outlier:
                    feat.map_point = None
                self.observed_times -= 1
                break


class Map:
    """src/map.cpp; the unordered_maps become dicts — the one place where iteration order could matter (strict < / > against the
    running extremes in RemoveOldKeyframe) needs an exact tie between two pose distances to show"""

    def __init__(self, num_active_keyframes, events):
        self.num_active = num_active_keyframes
        self.landmarks, self.active_landmarks = {}, {}
        self.keyframes, self.active_keyframes = {}, {}
        self.current_frame = None
        self.ev = events

    def insert_keyframe(self, frame):            # src/map.cpp:53-67
        self.current_frame = frame
        self.keyframes[frame.keyframe_id] = frame
        self.active_keyframes[frame.keyframe_id] = frame
        if len(self.active_keyframes) > self.num_active:
            self.remove_old_keyframe()

    def insert_map_point(self, mp):              # src/map.cpp:69-74
        self.landmarks[mp.id] = mp
        self.active_landmarks[mp.id] = mp

    def remove_old_keyframe(self):               # src/map.cpp:76-181
        if self.current_frame is None:
            return
        max_dis, min_dis = 0.0, 999999.0
        max_kf_id = min_kf_id = 0
        Twc = orc.se3_inv(self.current_frame.pose)
        for kid, kf in self.active_keyframes.items():
            if kf is self.current_frame:
                continue
            dis = float(np.linalg.norm(orc.se3_log(orc.se3_mul(kf.pose, Twc))))
            if dis > max_dis:
                max_dis, max_kf_id = dis, kid
            if dis < min_dis:
                min_dis, min_kf_id = dis, kid
        if min_dis < 0.2:
            victim = self.active_keyframes[min_kf_id]
            self.ev["evict_nearest"] += 1
        else:
            victim = self.active_keyframes[max_kf_id]
            self.ev["evict_farthest"] += 1
        del self.active_keyframes[victim.keyframe_id]
        for feat in victim.feature_left:
            if feat.map_point is not None:
                feat.map_point.remove_observation(feat)
        for feat in victim.feature_right:
            if feat is None:
                continue
            if feat.map_point is not None:
                feat.map_point.remove_observation(feat)
        self.clean_map()
        victim.left = victim.right = None        # (loop closure off: both images released)

    def clean_map(self):                         # src/map.cpp:21-40
        for mid in [m for m, mp in self.active_landmarks.items() if mp.observed_times == 0]:
            del self.active_landmarks[mid]
            self.ev["cleaned_landmarks"] += 1


class Camera:
    """src/camera.cpp"""

    def __init__(self, k4, ext):
        self.fx, self.fy, self.cx, self.cy = (float(v) for v in k4)
        self.k4 = np.array(k4, np.float64)
        self.pose = np.array(ext, np.float64)    # stereo rig -> camera

    def world2pixel(self, p_w, T_c_w):           # :74-80 -> world2camera :27-37 (pose_ * T_c_w * p_w, left to right), camera2pixel :46-54
        p_c = orc.se3_act(orc.se3_mul(self.pose, T_c_w), p_w)
        return (self.fx * p_c[0] / p_c[2] + self.cx, self.fy * p_c[1] / p_c[2] + self.cy)


class Backend:
    """src/backend.cpp:9-246 around g2o's optimize(10) (the oracle's local BA; jac_mode 1: numeric Jacobians like g2o, 0: analytic
    ones like the HIP kernel — the choice the twin makes through SVS_ORACLE_BA_JAC)"""

    def __init__(self, cam_l, cam_r, chi2_th, events, iters=10, jac_mode=1):
        self.cam_l, self.cam_r, self.chi2_th, self.iters, self.ev, self.jac_mode = cam_l, cam_r, chi2_th, iters, events, jac_mode
        self.map = None
        self.last_problem = None

    def update_map(self):                        # :288-294 + BackendLoop :276-284, run synchronously
        self.optimize(dict(self.map.active_keyframes), dict(self.map.active_landmarks))

    def optimize(self, keyframes, landmarks):
        # vertices: g2o orders active vertices by id — poses by keyframe id, landmarks by id (their vertex id is id + max_kf_id + 1)
        kf_ids = sorted(keyframes)
        kf_index = {k: i for i, k in enumerate(kf_ids)}
        lm_ids, edges = [], []                   # edges: (landmark index, keyframe index, feature)
        lm_index = {}
        for mid in sorted(landmarks):
            mp = landmarks[mid]
            if mp.is_outlier:                    # :86-89
                continue
            for feat in list(mp.observations):   # GetObs(): a copy of the list
                if feat is None or feat.outlier:                     # :99-107
                    continue
                if feat.frame is None:           # :110-114 (keyframes are owned by Map::keyframes_: never expired here)
                    continue
                if mid not in lm_index:          # :118-130: the vertex is added before the frame is known to be active
                    lm_index[mid] = len(lm_ids)
                    lm_ids.append(mid)
                if feat.frame.keyframe_id in kf_index:               # :133
                    edges.append((lm_index[mid], kf_index[feat.frame.keyframe_id], feat))
        with_edges = {e[0] for e in edges}
        self.ev["ba_zero_edge_landmarks"] += len(lm_ids) - len(with_edges)
        poses = np.array([keyframes[k].pose for k in kf_ids]).reshape(-1, 7)
        pts = np.array([landmarks[m].pos for m in lm_ids]).reshape(-1, 3)
        okf = np.array([e[1] for e in edges], np.int32)
        olm = np.array([e[0] for e in edges], np.int32)
        ori = np.array([0 if e[2].is_on_left_image else 1 for e in edges], np.uint8)
        ouv = np.array([e[2].pt for e in edges], np.float32).reshape(-1, 2)
        self.last_problem = (len(kf_ids), len(lm_ids), len(edges))
        self.ev["ba_calls"] += 1
        poses, pts, chi2, _ = orc.local_ba(self.cam_l.k4, self.cam_l.pose, self.cam_r.k4, self.cam_r.pose, poses, pts, okf, olm,
                                           ori, ouv, huber_delta=self.chi2_th, iters=self.iters, jac_mode=self.jac_mode)
        # :167-193: the inlier threshold doubles until more than half of the edges are inliers, at most five times
        chi2_th = self.chi2_th
        iteration = 0
        while iteration < 5:
            cnt_outlier = int((chi2 > chi2_th).sum())
            cnt_inlier = len(chi2) - cnt_outlier
            inlier_ratio = cnt_inlier / float(cnt_inlier + cnt_outlier) if len(chi2) else float("nan")
            if inlier_ratio > 0.5:
                break
            chi2_th *= 2
            iteration += 1
        if iteration:
            self.ev["ba_threshold_doublings"] += iteration
        for (li, ki, feat), c2 in zip(edges, chi2):                  # :197-213
            if c2 > chi2_th:
                feat.outlier = True
                if feat.map_point is not None:
                    feat.map_point.remove_observation(feat)
                self.ev["ba_outlier_edges"] += 1
            else:
                feat.outlier = False
        for i, k in enumerate(kf_ids):                               # :224-227
            keyframes[k].pose = poses[i].copy()
        for i, m in enumerate(lm_ids):                               # :228-231
            landmarks[m].pos = pts[i].copy()
        for k in kf_ids:                                             # :235-246
            if k == 0:
                continue
            kf = keyframes[k]
            kf.relative_pose_pkf = orc.se3_mul(kf.pose, orc.se3_inv(kf.prev_keyframe.pose))


class Frontend:
    """src/frontend.cpp"""

    def __init__(self, cfg, cam_l, cam_r, the_map, backend, events):
        self.cfg, self.cam_l, self.cam_r, self.map, self.backend, self.ev = cfg, cam_l, cam_r, the_map, backend, events
        self.status = INITING
        self.current_frame = self.last_frame = None
        self.relative_motion = IDENT.copy()
        self.current_kf = self.prev_kf = None
        self.tracking_inliers = 0
        self.next_keyframe_id = 0                # Frame::SetKeyFrame's static counter, src/frame.cpp:28-33
        self.next_mappoint_id = 0                # MapPoint::CreateNewMappoint's static counter, src/mappoint.cpp:88-98

    # -- helpers ---------------------------------------------------------------------------------
    def _set_keyframe(self, frame):
        frame.is_keyframe = True
        frame.keyframe_id = self.next_keyframe_id
        self.next_keyframe_id += 1

    def _new_mappoint(self):
        mp = MapPoint(self.next_mappoint_id)
        self.next_mappoint_id += 1
        return mp

    @staticmethod
    def _inside(pt, img):
        rows, cols = img.shape
        return (pt[1] >= 0) and (pt[1] < rows) and (pt[0] >= 0) and (pt[0] < cols)

    # -- :36-70 ----------------------------------------------------------------------------------
    def detect_features(self):
        cur = self.current_frame
        rects = np.array([f.pt for f in cur.feature_left], np.float32).reshape(-1, 2)     # one 21x21 rectangle per existing feature
        kps = orc.gftt(cur.left, rects, max_corners=self.cfg["num_features"], quality=0.01, min_dist=20.0)
        for kp in kps:
            cur.feature_left.append(Feature(cur, kp))
        return len(kps)

    # -- :72-141 ---------------------------------------------------------------------------------
    def find_features_in_right(self):
        cur = self.current_frame
        kps_left = np.zeros((len(cur.feature_left), 2), np.float32)
        kps_right = np.zeros((len(cur.feature_left), 2), np.float32)
        for i, f in enumerate(cur.feature_left):
            kps_left[i] = f.pt
            if f.map_point is not None:
                px = self.cam_r.world2pixel(f.map_point.pos, cur.pose)
                kps_right[i] = (np.float32(px[0]), np.float32(px[1]))
            else:
                kps_right[i] = f.pt
        if len(kps_left):
            kps_right, status, _ = orc.lk(cur.left, cur.right, kps_left, kps_right)
        else:
            status = np.zeros(0, np.uint8)
        good = 0
        for i in range(len(status)):
            if status[i] and self._inside(kps_right[i], cur.right):
                feat = Feature(cur, kps_right[i])
                feat.is_on_left_image = False
                cur.feature_right.append(feat)
                good += 1
            else:
                cur.feature_right.append(None)
        return good

    def _triangulate(self, idx, T_wc, zmax):
        cur = self.current_frame
        if not idx:
            return np.zeros((0, 3)), np.zeros(0, np.uint8)
        uv_l = np.array([cur.feature_left[i].pt for i in idx], np.float32)
        uv_r = np.array([cur.feature_right[i].pt for i in idx], np.float32)
        return orc.triangulate(self.cam_l.k4, self.cam_l.pose, self.cam_r.k4, self.cam_r.pose, uv_l, uv_r, T_wc, zmax)

    # -- :143-214 --------------------------------------------------------------------------------
    def build_init_map(self):
        cur = self.current_frame
        idx = [i for i in range(len(cur.feature_left)) if cur.feature_right[i] is not None]
        xyz, ok = self._triangulate(idx, None, 0.0)          # world == stereo rig at initialisation; only pworld[2] > 0 is asked
        for j, i in enumerate(idx):
            if not ok[j]:
                continue
            mp = self._new_mappoint()
            mp.pos = xyz[j].copy()
            mp.add_observation(cur.feature_left[i])
            mp.add_observation(cur.feature_right[i])
            cur.feature_left[i].map_point = mp
            cur.feature_right[i].map_point = mp
            self.map.insert_map_point(mp)
        self._set_keyframe(cur)
        self.map.insert_keyframe(cur)
        if self.backend is not None:
            self.backend.update_map()
        return True

    # -- :216-249 --------------------------------------------------------------------------------
    def stereo_init(self):
        self.detect_features()
        num_good = self.find_features_in_right()
        if num_good < self.cfg["num_features_init"]:
            return False
        self.current_kf = self.current_frame
        if self.build_init_map():
            self.status = TRACKING_GOOD
            return True
        return False

    # -- :251-320 --------------------------------------------------------------------------------
    def triangulate_new_points(self):
        cur = self.current_frame
        T_wc = orc.se3_inv(cur.pose)
        idx = [i for i in range(len(cur.feature_left))
               if cur.feature_left[i].map_point is None and cur.feature_right[i] is not None]
        xyz, ok = self._triangulate(idx, T_wc, self.cfg["max_triangulation_depth"])
        n = 0
        for j, i in enumerate(idx):
            if not ok[j]:
                continue
            mp = self._new_mappoint()
            mp.pos = xyz[j].copy()
            mp.add_observation(cur.feature_left[i])
            mp.add_observation(cur.feature_right[i])
            cur.feature_left[i].map_point = mp
            cur.feature_right[i].map_point = mp
            self.map.insert_map_point(mp)
            n += 1
        return n

    # -- :322-392 --------------------------------------------------------------------------------
    def track_last_frame(self):
        cur, last = self.current_frame, self.last_frame
        n = len(last.feature_left)
        kps_last = np.zeros((n, 2), np.float32)
        kps_cur = np.zeros((n, 2), np.float32)
        for i, f in enumerate(last.feature_left):
            kps_last[i] = f.pt
            if f.map_point is not None:
                px = self.cam_l.world2pixel(f.map_point.pos, cur.pose)
                kps_cur[i] = (np.float32(px[0]), np.float32(px[1]))
            else:
                kps_cur[i] = f.pt
        if n:
            kps_cur, status, _ = orc.lk(last.left, cur.left, kps_last, kps_cur)
        else:
            status = np.zeros(0, np.uint8)
        good = 0
        for i in range(n):
            if status[i]:
                if not self._inside(kps_cur[i], cur.left):
                    continue
                feat = Feature(cur, kps_cur[i])
                cur.feature_left.append(feat)
                feat.map_point = last.feature_left[i].map_point
                good += 1
        return good

    # -- :394-558 --------------------------------------------------------------------------------
    def estimate_current_pose(self):
        cur = self.current_frame
        feats = [f for f in cur.feature_left if f.map_point is not None]
        xyz = np.array([f.map_point.pos for f in feats]).reshape(-1, 3)
        uv = np.array([f.pt for f in feats], np.float32).reshape(-1, 2)
        pose, outl, _ = orc.pose_only(self.cam_l.k4, cur.pose, xyz, uv, chi2_th=5.991, rounds=4, iters=10)
        cnt_outlier = int(np.count_nonzero(outl))
        cur.pose = pose
        for f, o in zip(feats, outl):
            if o:                                # :546-553 (the flag goes back to false: "maybe we can still use it in future")
                f.map_point = None
                f.outlier = False
                self.ev["pose_outliers_unlinked"] += 1
        return len(feats) - cnt_outlier

    # -- :560-574 --------------------------------------------------------------------------------
    def set_observations_for_keyframe(self):
        for f in self.current_frame.feature_left:
            if f.map_point is not None:
                f.map_point.add_observation(f)

    # -- :576-643 --------------------------------------------------------------------------------
    def insert_keyframe(self):
        if self.tracking_inliers >= self.cfg["num_features_needed_for_keyframe"]:
            return False
        cur = self.current_frame
        self._set_keyframe(cur)
        self.map.insert_keyframe(cur)
        self.prev_kf = self.current_kf
        self.current_kf = cur
        cur.prev_keyframe = self.prev_kf
        cur.relative_pose_pkf = orc.se3_mul(cur.pose, orc.se3_inv(self.prev_kf.pose))
        self.set_observations_for_keyframe()
        self.detect_features()
        self.find_features_in_right()
        self.triangulate_new_points()
        if self.backend is not None:
            self.backend.update_map()
        return True

    # -- :645-688 --------------------------------------------------------------------------------
    def track(self):
        cur = self.current_frame
        if self.last_frame is not None:
            cur.pose = orc.se3_mul(self.relative_motion, self.last_frame.pose)
        self.track_last_frame()
        self.tracking_inliers = self.estimate_current_pose()
        if self.tracking_inliers > self.cfg["num_features_tracking"]:
            self.status = TRACKING_GOOD
        elif self.tracking_inliers > self.cfg["num_features_tracking_bad"]:
            self.status = TRACKING_BAD
        else:
            self.status = LOST
        self.insert_keyframe()
        self.relative_motion = orc.se3_mul(cur.pose, orc.se3_inv(self.last_frame.pose))
        return True

    # -- :690-721 --------------------------------------------------------------------------------
    def add_frame(self, frame):
        self.current_frame = frame
        if self.status == INITING:
            self.stereo_init()
        elif self.status in (TRACKING_GOOD, TRACKING_BAD):
            self.track()
        else:
            pass                                 # LOST: Reset() is "not implemented" (:723-731)
        self.last_frame = self.current_frame
        return True


DEFAULT_CFG = dict(num_features=150, num_features_init=50, num_features_tracking=50, num_features_tracking_bad=20,
                   num_features_needed_for_keyframe=80, max_triangulation_depth=300.0, num_active_keyframes=10, backend_on=1,
                   chi2_th=5.991,                # config/stereo_slam_configs/config-00.yaml
                   ba_jac_mode=1)                # (not a reference key: which Jacobians the oracle's BA uses, see Backend)


class VisualOdometry:
    """src/visual_odometry.cpp:24-145 without viewer and loop closure; one stream"""

    def __init__(self, cam, baseline, cfg=None):
        self.cfg = dict(DEFAULT_CFG, **(cfg or {}))
        self.events = {k: 0 for k in ("evict_nearest", "evict_farthest", "cleaned_landmarks", "pose_outliers_unlinked",
                                      "ba_outlier_edges", "ba_zero_edge_landmarks", "ba_threshold_doublings", "ba_calls")}
        self.status_seen = {INITING: 0, TRACKING_GOOD: 0, TRACKING_BAD: 0, LOST: 0}
        # Dataset::Init, src/dataset.cpp:63-77: both cameras share K; the right camera sits at t = (-baseline, 0, 0)
        self.cam_l = Camera(cam, (0, 0, 0, 1, 0, 0, 0))
        self.cam_r = Camera(cam, (0, 0, 0, 1, -baseline, 0, 0))
        self.map = Map(self.cfg["num_active_keyframes"], self.events)
        self.backend = None
        if self.cfg["backend_on"]:
            self.backend = Backend(self.cam_l, self.cam_r, self.cfg["chi2_th"], self.events, jac_mode=self.cfg["ba_jac_mode"])
            self.backend.map = self.map
        self.frontend = Frontend(self.cfg, self.cam_l, self.cam_r, self.map, self.backend, self.events)
        self.next_frame_id = 0                   # Frame::CreateFrame's static counter, src/frame.cpp:20-26

    def step(self, left, right):
        """Dataset::NextFrame's frame (images already decimated) through Frontend::AddFrame; returns what the twin reports"""
        frame = Frame(self.next_frame_id, np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(right, np.uint8))
        self.next_frame_id += 1
        self.frontend.add_frame(frame)
        self.status_seen[self.frontend.status] += 1
        return {"pose": frame.pose.copy(), "status": self.frontend.status, "is_keyframe": int(frame.is_keyframe),
                "n_features": len(frame.feature_left), "n_inliers": self.frontend.tracking_inliers,
                "frame_id": frame.id, "keyframe_id": frame.keyframe_id if frame.is_keyframe else -1}

    def snapshot(self):
        """the map as the reference holds it after a frame: active window, active landmarks, their observation lists
        (keyframe id, camera, index of the feature in that keyframe's list) and counters"""
        act_kf = sorted(self.map.active_keyframes)
        index = {}                               # frame -> {id(feature): position in its list}, built on demand

        def pos_of(f):
            fr = f.frame
            if id(fr) not in index:
                index[id(fr)] = ({id(g): i for i, g in enumerate(fr.feature_left)},
                                 {id(g): i for i, g in enumerate(fr.feature_right) if g is not None})
            return index[id(fr)][0 if f.is_on_left_image else 1][id(f)]
        lms = []
        for mid in sorted(self.map.active_landmarks):
            mp = self.map.active_landmarks[mid]
            obs = tuple((f.frame.keyframe_id, 0 if f.is_on_left_image else 1, pos_of(f)) for f in mp.observations)
            lms.append((mid, mp.observed_times, obs, tuple(float(v) for v in mp.pos)))
        return {"active_keyframes": act_kf, "landmarks": lms, "n_landmarks_total": len(self.map.landmarks),
                "keyframe_poses": {k: self.map.active_keyframes[k].pose.copy() for k in act_kf}}
