"""Device-resident Tracking::Track over B independent camera streams (one frame per stream per step), software-pipelined.

The sequence of the reference's tracking thread for one RGB-D frame (src/Tracking.cc), every stage a batched HIP kernel behind the C ABI:

    Frame::Frame (src/Frame.cc:55-152)        ORBextractor, LineSegment::ExtractLineSegment, PlaneDetection (three streams, as its three
                                              threads :90-95), ComputeStereoFromRGBD
    Track (:248)                              TrackManhattanFrame(mLastRcm, surface normals, 3-D line directions)
    TranslationWithMotionModel (:1739-1790)   SearchByProjection(Cur, Last, 15) - LSDmatcher::SearchByDescriptor(refKF) - MatchORBPoints
                                              (the reference runs it only when < 50 projection matches; here for every frame: a superset) -
                                              PlaneMatcher::SearchMapByCoefficients - TranslationOptimization - discard outliers
    TrackLocalMap / SearchLocalPoints         Frame::isInFrustum for the local map points and lines - SearchByProjection(F, vpMapPoints, 3) -
    (:1954-2040)                              LSDmatcher::SearchByProjection - PoseOptimization - UnprojectStereo of the new frame's keypoints
                                              (the next frame's "last frame" map points)

Manhattan rotation (:250-253, :1778): mRotation_wc = (Rotation_cm * MF_can^T)^T is copied into the pose before TranslationOptimization, as the reference does
(planar_manhattan_pose_dev; manhattan_rotation=False keeps the last pose's rotation instead).  Rotation_cm is initialised as in :224-230: FindManhattan on the
stream's first frame (a host-side stand-in of Map::FindManhattan in build_map: Map is out of scope), refined by TrackManhattanFrame on the first tracked frame.

What is NOT the reference's code path and only stands in for the map it maintains (Map / KeyFrame / LocalMapping are out of scope):
the local map of a stream is the previous two frames' own back-projected keypoints, the reference key frame's lines and the map planes are
fixed per stream (set_map); every new "map point" has one observation, its own frame (MapPoint::UpdateNormalAndDepth: planar_update_normal_and_depth).

PyTorch supplies device memory, streams and events only.  Frame-batch parallelism: steps are pipelined `depth` deep - the tracking chain of
step i - depth runs on its own stream beside the extraction launches of step i (points on the main stream, lines and planes on theirs)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import KEYLINE_DTYPE, FrameView, LastFrameView, MapProbes, PoseBatch, TrackMatches, check, lib


class TrackPipeline:
    MAX_POSE_PLANES = 16

    def __init__(self, B, torch, device_index=0, depth=2, prio=(-1, 0, 0), cam=None, W=640, H=480, n_map_planes=8, n_plane_pts=128, n_normals=4096,
                 run_fallback_matcher=True, manhattan_rotation=True, dist_coef=None, work_sets=None, seq_cus=0, seq_which=3, seq_shared=True):
        from . import Context, ORBextractor, Optimizer, PlaneDetection
        from .lines import LineSegment
        from .planes import PlaneClouds, SurfaceNormals
        from .synth import TUM3
        self.torch, self.B, self.W, self.H, self.depth = torch, B, W, H, depth
        self.manhattan_rotation = manhattan_rotation
        self.cam = dict(cam or TUM3)
        # Camera.k1, k2, p1, p2, k3 (mDistCoef): Frame::UndistortKeyPoints runs after the extractor when k1 != 0 (Frame.cc:545-573); TUM3's are zero -> mvKeysUn = mvKeys
        self.dist_coef = None if dist_coef is None or float(dist_coef[0]) == 0.0 else np.ascontiguousarray(dist_coef, np.float32).reshape(5)
        self.dev = torch.device("cuda", device_index)
        self.NB = depth + 2                      # buffer sets: a step's extractor outputs live until the tracking chain `depth` steps later has used them as "last frame"
        # extractor sets (line / plane stream + the extractors' WORKSPACES, 18 MB per frame): only the outputs have to outlive the extraction, so there are as many as
        # extractions in flight (step i uses set i mod NW; the stream's own order guards the workspace), not one per buffer set
        self.NW = NW = min(self.NB, max(1, depth if work_sets is None else int(work_sets)))
        self.L = lib()
        self.stream = torch.cuda.Stream(device=device_index, priority=prio[0])
        self.ctx = Context(device_index, stream=self.stream.cuda_stream)
        # the tracking chain of step i - depth runs on its own stream, beside the point extraction of step i (they share nothing but buffers guarded by events)
        self.s_track = torch.cuda.Stream(device=device_index, priority=prio[3] if len(prio) > 3 else prio[0])
        self.ctx_t = Context(device_index, stream=self.s_track.cuda_stream)
        self.s_peacs = [torch.cuda.Stream(device=device_index, priority=prio[2]) for _ in range(NW)]
        self.s_lsds = [torch.cuda.Stream(device=device_index, priority=prio[1]) for _ in range(NW)]
        self.ctx_peacs = [Context(device_index, stream=q.cuda_stream) for q in self.s_peacs]
        self.ctx_lsds = [Context(device_index, stream=q.cuda_stream) for q in self.s_lsds]
        # CU partition (planar_ctx_set_seq_stream): the one-wavefront-per-frame kernels (PEAC clustering: seq_which & 1, LSD region growing: & 2) on streams masked to
        # seq_cus compute units - one shared by all sets (their launches then run in step order) or one per context
        self.seq_streams = []
        if seq_cus:
            from ._lib import cu_stream_create
            targets = (self.ctx_peacs if seq_which & 1 else []) + (self.ctx_lsds if seq_which & 6 else [])
            try:
                for c in targets:
                    own = (seq_which & 4) and c in self.ctx_lsds          # (& 4: region growing on the partition too, but on a masked stream per line context - not behind the clustering launches)
                    if own or not seq_shared or not self.seq_streams:
                        self.seq_streams.append(cu_stream_create(device_index, int(seq_cus)))
                        if not own and seq_shared:
                            shared = self.seq_streams[-1]
                    c.set_seq_stream(self.seq_streams[-1] if own or not seq_shared else shared)
            except Exception as e:      # a runtime that refuses CU masks: same results on the contexts' own streams (speed only)
                import sys
                print(f"TrackPipeline: no CU partition ({e}); the sequential kernels stay on their extractors' streams", file=sys.stderr)
                for c in targets:
                    c.set_seq_stream(None)
                self.seq_streams = []
        self.ex = ORBextractor(1000, 1.2, 8, 20, 7, width=W, height=H, max_batch=B, ctx=self.ctx)
        self.S = S = self.ex.kp_cap
        self.pds = [PlaneDetection(W, H, max_batch=B, ctx=c) for c in self.ctx_peacs]
        self.lss = [LineSegment(W, H, B, c, top_only=True) for c in self.ctx_lsds]      # (the 40 key lines are all the step reads)
        self.sns = [SurfaceNormals(W, H, B, c) for c in self.ctx_peacs]     # Frame::ComputePlanes: PEAC, then the surface normals, on the plane thread
        self.SN = self.sns[0].count
        self.pcs = [PlaneClouds(W, H, B, ctx=c) for c in self.ctx_peacs]       # ... and before them the voxel clouds + RANSAC refit of every plane (Frame.cc:655-692)
        self.plane_dist_th = 0.05                                               # Plane.DistanceThreshold (Examples/RGB-D/TUM*.yaml)
        self.opt = Optimizer(self.cam, ctx=self.ctx_t)
        self.PS = self.pds[0].max_planes
        self.run_fallback_matcher = run_fallback_matcher
        sf = np.asarray(self.ex.GetScaleFactors(), np.float32)
        self.sf = sf
        self.nlev = len(sf)
        self.lsf = float(np.float32(np.log(np.float32(sf[1]))))
        self.inv_sigma2 = np.asarray(self.ex.GetInverseScaleSigmaSquares(), np.float32)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.dev)
        t = torch
        NB = self.NB
        # ---- per-step buffers ----
        self.kps = [z((B, S, 7), t.float32) for _ in range(NB)]
        self.kpu = self.kps if self.dist_coef is None else [z((B, S, 7), t.float32) for _ in range(NB)]     # mvKeysUn
        self.desc = [z((B, S, 32), t.uint8) for _ in range(NB)]
        self.n = [z((B,), t.int32) for _ in range(NB)]
        self.ur = [z((B, S), t.float32) for _ in range(NB)]
        self.zd = [z((B, S), t.float32) for _ in range(NB)]
        self.lab = [z((B, H * W), t.int32) for _ in range(NB)]
        self.pls = [z((B, self.PS, 8), t.float64) for _ in range(NB)]
        self.npl = [z((B,), t.int32) for _ in range(NB)]
        self.snrm = [z((B, self.SN, 3), t.float32) for _ in range(NB)]
        # Frame::mnPlaneNum / mvPlaneCoefficients / mvPlanePoints (+ the detector plane every kept plane came from)
        self.pc = [dict(n=z((B,), t.int32), coef=z((B, self.PS, 4), t.float32), src=z((B, self.PS), t.int32), off=z((B, self.PS + 1), t.int32),
                        pts=z((B, self.pcs[0].max_points, 3), t.float32), status=z((B,), t.int32)) for _ in range(NB)]
        self.n_snrm = t.full((B,), self.SN, dtype=t.int32, device=self.dev)
        self.kls = [z((B * 40 * KEYLINE_DTYPE.itemsize,), t.uint8) for _ in range(NB)]
        self.ldesc = [z((B, 40, 32), t.uint8) for _ in range(NB)]
        self.leq = [z((B, 40, 3), t.float64) for _ in range(NB)]
        self.nl = [z((B,), t.int32) for _ in range(NB)]
        # Frame::isLineGood outputs: mvDepthLine, mvLines3D, the packed FrameLine directions (mVF3DLines) and their count
        self.l3 = [dict(depth_line=z((B, 40), t.float32), lines3d=z((B, 40, 6), t.float64), good=z((B, 40), t.uint8), direction=z((B, 40, 3), t.float64),
                        n_inliers=z((B, 40), t.int32), packed=z((B, 40, 3), t.float64), n_good=z((B,), t.int32), seeds=z((B,), t.int32)) for _ in range(NB)]
        self.seed_base = t.arange(B, dtype=t.int32, device=self.dev) * 64
        self.ev_in = [t.cuda.Event() for _ in range(NB)]
        self.ev_orb = [t.cuda.Event() for _ in range(NB)]
        self.join_p = [t.cuda.Event() for _ in range(NB)]
        self.join_l = [t.cuda.Event() for _ in range(NB)]
        self.done = [t.cuda.Event() for _ in range(NB)]
        # ---- per-stream state: the last two frames' back-projected keypoints (slot-major: one contiguous [B, S] array per slot) ----
        self.h_xw = z((2, B, S, 3), t.float32); self.h_valid = z((2, B, S), t.uint8); self.h_normal = z((2, B, S, 3), t.float32)
        self.h_mind = z((2, B, S), t.float32); self.h_maxd = z((2, B, S), t.float32); self.h_desc = z((2, B, S, 32), t.uint8)
        self.h_oct = z((2, B, S), t.int32); self.h_ang = z((2, B, S), t.float32); self.h_n = z((2, B), t.int32)
        self.pose = t.eye(4, dtype=t.float32, device=self.dev).reshape(1, 16).repeat(B, 1).contiguous()
        self.pose0 = self.pose.clone()          # ComputeStereoFromRGBD at extraction time needs no pose (its world points are discarded)
        self.Rcm = t.eye(3, dtype=t.float32, device=self.dev).reshape(1, 9).repeat(B, 1).contiguous()
        self.Rcm0 = self.Rcm.clone()                 # Rotation_cm: the stream's camera-to-Manhattan rotation at initialisation (set_map)
        self.pose_mf = self.pose.clone()             # the pose TranslationOptimization starts from: last pose with the Manhattan rotation of this frame
        self.ones_S = t.ones((B, S), dtype=t.uint8, device=self.dev)
        self.zeros_S = z((B, S), t.uint8)
        # ---- match / optimiser buffers (one set: the tracking chains run in step order on one stream) ----
        self.pm = z((B, S), t.int32); self.nm = z((B,), t.int32)
        self.cm2 = z((B, S), t.int32); self.npair = z((B,), t.int32)
        self.lm = z((B, 40), t.int32); self.nlm = z((B,), t.int32)
        self.plm = z((3, B, self.PS), t.int32); self.nplm = z((B,), t.int32)
        self.mm = z((B, S), t.int32); self.nmm = z((B,), t.int32)
        self.blocked = z((B, S), t.uint8); self.lblocked = z((B, 40), t.uint8)
        self.pm_all = z((B, S), t.int32)
        self.xw_all = z((B, 2 * S, 3), t.float32); self.valid_all = z((B, 2 * S), t.uint8)
        self.xw_tmp = z((B, S, 3), t.float32); self.valid_tmp = z((B, S), t.uint8)
        self.pr = dict(in_view=z((B, S), t.uint8), proj_x=z((B, S), t.float32), proj_y=z((B, S), t.float32), proj_xr=z((B, S), t.float32),
                       level=z((B, S), t.int32), view_cos=z((B, S), t.float32))
        self.lpr = dict(in_view=z((B, 40), t.uint8), proj=z((B, 40, 4), t.float32), level=z((B, 40), t.int32), view_cos=z((B, 40), t.float32))
        self.kept = z((B,), t.int32)
        self.Rcm_new = z((B, 9), t.float32)
        MP, ML, MM = S, 40, self.MAX_POSE_PLANES
        self.pb_arrays = []
        self.pbs = []
        for _ in range(2):      # [0] translation problem, [1] pose problem
            a = dict(n_points=z((B,), t.int32), n_lines=z((B,), t.int32), n_planes=z((B,), t.int32), pt_valid=z((B, MP), t.uint8), pt_xw=z((B, MP, 3), t.float32),
                     pt_obs=z((B, MP, 3), t.float32), pt_inv_sigma2=z((B, MP), t.float32), ln_valid=z((B, ML), t.uint8), ln_obs=z((B, ML, 3), t.float64),
                     ln_xw=z((B, ML, 6), t.float64), pl_meas=z((B, MM, 4), t.float32), pl_valid=z((B, MM, 3), t.uint8), pl_world=z((B, MM, 3, 4), t.float32),
                     Tcw_in=z((B, 16), t.float32), Tcw_out=z((B, 16), t.float32), pt_outlier=z((B, MP), t.uint8), ln_outlier=z((B, ML), t.uint8),
                     pl_outlier=z((B, MM, 3), t.uint8), n_inliers=z((B,), t.int32), lm_iters=z((B,), t.int32))
            pb = PoseBatch()
            pb.B, pb.max_points, pb.max_lines, pb.max_planes = B, MP, ML, MM
            for k, v in a.items():
                setattr(pb, k, v.data_ptr())
            self.pb_arrays.append(a); self.pbs.append(pb)
        # Frame::ComputeImageBounds (Frame.cc:573-598): the four image corners through cv::undistortPoints when k1 != 0; mfGridElement*Inv from them (:122-123)
        self.bounds = (0.0, float(W), 0.0, float(H))
        if self.dist_coef is not None:
            # a one-off of four points: the synchronous host-pointer entry point (numpy in / out), so nothing here depends on which torch stream is current
            cr = np.zeros((1, 4, 7), np.float32)
            cr[0, :, 0] = [0.0, W, 0.0, W]; cr[0, :, 1] = [0.0, 0.0, H, H]
            cu = np.zeros_like(cr); n4 = np.full((1,), 4, np.int32)
            c = self.cam
            check(self.L.planar_undistort_keypoints(self.ctx.h, 1, cr.ctypes.data, n4.ctypes.data, 4, c["fx"], c["fy"], c["cx"], c["cy"], self.dist_coef.ctypes.data, cu.ctypes.data))
            m = cu[0, :, :2]
            self.bounds = (float(min(m[0, 0], m[2, 0])), float(max(m[1, 0], m[3, 0])), float(min(m[0, 1], m[1, 1])), float(max(m[2, 1], m[3, 1])))
        self.pending = []
        self.map_set = False
        self.rcm0_set = False           # Rotation_cm is fixed by the first tracked frame
        self.step_count = 0
        self.capture_steps = set()      # tests: steps whose stage inputs / outputs are cloned (on the stream) into self.captured[j]
        self.captured = {}

    def close(self):
        """Releases the CU-masked side streams (after synchronising): the contexts fall back to their own streams."""
        if getattr(self, "seq_streams", None):
            self.torch.cuda.synchronize()
            for c in self.ctx_peacs + self.ctx_lsds:
                c.set_seq_stream(None)
            for q in self.seq_streams:
                self.L.planar_cu_stream_destroy(C.c_void_p(q))
            self.seq_streams = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------------------------------------
    def set_map(self, kf_lines: dict, map_planes: dict, normals: dict):
        """Per-stream stand-ins for the map (host numpy, uploaded once):
        kf_lines   : n [B], ldesc [B,40,32], xw6 [B,40,6] f64, normal [B,40,3] f64, min_dist, max_dist [B,40] f32 (reference key frame's map lines)
        map_planes : n [B], valid [B,M] u8, coef [B,M,4] f32 (world), npts [B,M] i32, pts [B,M,P,3] f32
        normals    : normals [B,SN,3] f32, n_normals [B], lines [B,40,3] f64, n_lines [B]  (Frame::vSurfaceNormal, mVF3DLines directions)"""
        t = self.torch
        up = lambda a, dt: t.from_numpy(np.ascontiguousarray(a, dt)).to(self.dev)
        self.kf = dict(n=up(kf_lines["n"], np.int32), ldesc=up(kf_lines["ldesc"], np.uint8), xw6=up(kf_lines["xw6"], np.float64), normal=up(kf_lines["normal"], np.float64),
                       min_dist=up(kf_lines["min_dist"], np.float32), max_dist=up(kf_lines["max_dist"], np.float32))
        self.kf["has_ml"] = t.ones((self.B, 40), dtype=t.uint8, device=self.dev)
        self.mp = dict(n=up(map_planes["n"], np.int32), valid=up(map_planes["valid"], np.uint8), coef=up(map_planes["coef"], np.float32), npts=up(map_planes["npts"], np.int32),
                       pts=up(map_planes["pts"], np.float32))
        self.sn = dict(normals=up(normals["normals"], np.float32), n_normals=up(normals["n_normals"], np.int32), lines=up(normals["lines"], np.float64),
                       n_lines=up(normals["n_lines"], np.int32))
        if normals.get("R_last") is not None:
            self.Rcm = up(np.asarray(normals["R_last"], np.float32).reshape(self.B, 9), np.float32)
            self.Rcm0 = self.Rcm.clone()
        self.plane_th = np.array([0.1, 0.86, 0.08716, 0.9962], np.float32)      # include/PlaneMatcher.h:19
        self.map_set = True
        self.rcm0_set = False           # a new map: Rotation_cm is refined again by the first tracked frame (src/Tracking.cc:227-230)

    # ------------------------------------------------------------------------------------------------------------------------------
    def _frame_view(self, k, Tcw, blocked=None):
        fv = FrameView()
        fv.B, fv.stride = self.B, self.S
        fv.n, fv.keys_un, fv.u_right, fv.desc = self.n[k].data_ptr(), self.kpu[k].data_ptr(), self.ur[k].data_ptr(), self.desc[k].data_ptr()
        fv.blocked = blocked.data_ptr() if blocked is not None else None
        fv.Tcw = Tcw.data_ptr()
        fv.min_x, fv.max_x, fv.min_y, fv.max_y = self.bounds
        fv.grid_w_inv = float(np.float32(64.0) / np.float32(np.float32(fv.max_x) - np.float32(fv.min_x)))
        fv.grid_h_inv = float(np.float32(48.0) / np.float32(np.float32(fv.max_y) - np.float32(fv.min_y)))
        c = self.cam
        fv.fx, fv.fy, fv.cx, fv.cy, fv.bf, fv.b = c["fx"], c["fy"], c["cx"], c["cy"], c["bf"], c["bf"] / c["fx"]
        for i, v in enumerate(self.sf):
            fv.scale_factors[i] = float(v)
        return fv

    def _stereo(self, ctx, k, Tcw, depth, ur, zd, xw, valid):
        c = self.cam
        check(self.L.planar_stereo_from_rgbd_dev(ctx.h, self.B, self.kps[k].data_ptr(), self.kpu[k].data_ptr(), self.n[k].data_ptr(), self.S, depth.data_ptr(),
                                                 self.W, self.W * self.H, float(np.float32(1.0 / 5000.0)), c["fx"], c["fy"], c["cx"], c["cy"], c["bf"], Tcw.data_ptr(),
                                                 ur.data_ptr(), zd.data_ptr(), xw.data_ptr(), valid.data_ptr()))

    def _assemble(self, which, k, pt_match, mp_xw, mp_valid, mp_stride, Tcw):
        m = TrackMatches()
        m.B, m.stride, m.mp_stride, m.n_levels = self.B, self.S, mp_stride, self.nlev
        m.n, m.keys_un, m.u_right = self.n[k].data_ptr(), self.kpu[k].data_ptr(), self.ur[k].data_ptr()
        m.pt_match, m.mp_xw, m.mp_valid = pt_match.data_ptr(), mp_xw.data_ptr(), mp_valid.data_ptr()
        for i, v in enumerate(self.inv_sigma2):
            m.inv_level_sigma2[i] = float(v)
        m.ln_stride, m.ml_stride = 40, 40
        m.n_lines, m.line_eq, m.ln_match, m.ml_xw6 = self.nl[k].data_ptr(), self.leq[k].data_ptr(), self.lm.data_ptr(), self.kf["xw6"].data_ptr()
        m.pl_stride, m.mpl_stride, m.mpl_shared = self.PS, self.mp["coef"].shape[1], 0
        m.n_planes, m.pl_coef, m.pl_match, m.mpl_coef = self.pc[k]["n"].data_ptr(), self.pc[k]["coef"].data_ptr(), self.plm.data_ptr(), self.mp["coef"].data_ptr()
        m.Tcw = Tcw.data_ptr()
        check(self.L.planar_pose_assemble_dev(self.ctx_t.h, C.byref(m), C.byref(self.pbs[which])))

    # ---- the extraction stages of one step, each on the stream the reference's thread of that name stands for (src/Frame.cc:90-95) ----
    def _lines_head(self, w, gray):
        """LineSegment::ExtractLineSegment, first half (smoothing, gradients, the pixel order) - line stream"""
        check(self.L.planar_lsd_preprocess_dev(self.lss[w].h, gray.data_ptr(), self.B, self.W, self.W * self.H))

    def _planes(self, w, k, depth):
        """PlaneDetection (PEAC) - plane stream"""
        self.pds[w].segment_dev(depth.data_ptr(), self.lab[k].data_ptr(), self.pls[k].data_ptr(), self.npl[k].data_ptr(), self.B,
                                K=(self.cam["fx"], self.cam["fy"], self.cam["cx"], self.cam["cy"]))      # PlaneDetection::readDepthImage(Depth, K, ...): the frame's intrinsics

    def _plane_clouds(self, w, k, depth):
        """Frame::ComputePlanes' voxel clouds + RANSAC refit (Frame.cc:655-692) - plane stream"""
        pc = self.pc[k]
        self.pcs[w].compute_dev(depth.data_ptr(), self.lab[k].data_ptr(), self.pls[k].data_ptr(), self.npl[k].data_ptr(), self.B, pc["n"].data_ptr(), pc["coef"].data_ptr(),
                                pc["src"].data_ptr(), pc["off"].data_ptr(), pc["pts"].data_ptr(), pc["status"].data_ptr(), dist_th=self.plane_dist_th,
                                K=(self.cam["fx"], self.cam["fy"], self.cam["cx"], self.cam["cy"]))

    def _normals(self, w, k, depth):
        """Frame::ComputePlanes' surface normals - plane stream"""
        self.sns[w].compute_dev(depth.data_ptr(), self.snrm[k].data_ptr(), self.B, K=(self.cam["fx"], self.cam["fy"], self.cam["cx"], self.cam["cy"]))

    def _lines_tail(self, i, w, k, depth):
        """ExtractLineSegment, second half (region growing, NFA, key lines, LBD), then Frame::isLineGood right behind it on the same thread - line stream"""
        L, B = self.L, self.B
        check(L.planar_lsd_detect_dev(self.lss[w].h, B, 40, self.kls[k].data_ptr(), self.ldesc[k].data_ptr(), self.leq[k].data_ptr(), self.nl[k].data_ptr()))
        l3 = self.l3[k]                                     # every (stream, step, line) has its own rand() seed
        check(L.planar_add_scalar_i32_dev(self.ctx_lsds[w].h, self.seed_base.data_ptr(), self.seed_base.numel(), (i * B * 64) & 0x3fffffff, l3["seeds"].data_ptr()))
        c = self.cam
        check(L.planar_is_line_good_dev(self.ctx_lsds[w].h, B, self.kls[k].data_ptr(), self.nl[k].data_ptr(), 40, depth.data_ptr(), self.W, self.H, self.W, self.W * self.H,
                                        float(np.float32(1.0 / 5000.0)), c["fx"], c["fy"], c["cx"], c["cy"], l3["seeds"].data_ptr(), l3["depth_line"].data_ptr(),
                                        l3["lines3d"].data_ptr(), l3["good"].data_ptr(), l3["direction"].data_ptr(), l3["n_inliers"].data_ptr(), l3["packed"].data_ptr(),
                                        l3["n_good"].data_ptr()))

    def _points(self, k, gray):
        """ORBextractor::operator() (+ Frame::UndistortKeyPoints for a distorting camera) - main stream"""
        self.ex.extract_dev(gray.data_ptr(), self.kps[k].data_ptr(), self.desc[k].data_ptr(), self.n[k].data_ptr(), self.B)
        if self.dist_coef is not None:
            c = self.cam
            check(self.L.planar_undistort_keypoints_dev(self.ctx.h, self.B, self.kps[k].data_ptr(), self.n[k].data_ptr(), self.S, c["fx"], c["fy"], c["cx"], c["cy"],
                                                        self.dist_coef.ctypes.data, self.kpu[k].data_ptr()))

    # ------------------------------------------------------------------------------------------------------------------------------
    def step(self, i, gray, depth, evs=None, side=None):
        """Enqueue step i: extraction of `gray` [B,H,W] u8 / `depth` [B,H,W] i16-as-u16 (device tensors that stay valid until the step's
        tracking chain has run) on the three streams, then the tracking chain of step i - depth.  evs / side: optional timing events."""
        t, L, B, k, w = self.torch, self.L, self.B, i % self.NB, i % self.NW
        sp, sl = self.s_peacs[w], self.s_lsds[w]
        st = self.stream
        st.wait_event(self.done[k])                         # the buffers of step i - NB have been consumed
        self.ev_in[k].record(st)                            # the caller gathered this step's frames on the main stream
        if evs: evs["start"].record(st)
        sl.wait_event(self.ev_in[k]); sp.wait_event(self.ev_in[k])
        if side: side[2].record(sl)
        self._lines_head(w, gray)
        if side: side[0].record(sp)
        self._planes(w, k, depth)
        self._plane_clouds(w, k, depth)
        self._normals(w, k, depth)
        self._lines_tail(i, w, k, depth)
        if side: side[1].record(sp); side[3].record(sl)
        self.join_p[k].record(sp); self.join_l[k].record(sl)
        self._points(k, gray)
        if evs: evs["orb"].record(st)
        # Frame::ComputeStereoFromRGBD: mvuRight / mvDepth of the new keypoints (the world points come after the pose is known)
        self._stereo(self.ctx, k, self.pose0, depth, self.ur[k], self.zd[k], self.xw_tmp, self.valid_tmp)
        if evs: evs["stereo"].record(st)
        self.ev_orb[k].record(st)
        self.pending.append((i, depth, evs))
        if len(self.pending) > self.depth:
            self._track(*self.pending.pop(0))
        self.step_count += 1

    def drain(self):
        while self.pending:
            self._track(*self.pending.pop(0))

    # ------------------------------------------------------------------------------------------------------------------------------
    def _track(self, j, depth, evs):
        with self.torch.cuda.stream(self.s_track):          # torch's element-wise glue ops follow the current stream
            self._track_on_stream(j, depth, evs)

    def _track_on_stream(self, j, depth, evs):
        t, L, B, S, k = self.torch, self.L, self.B, self.S, j % self.NB
        st = self.s_track
        st.wait_event(self.ev_orb[k])                       # this frame's keypoints / descriptors / mvuRight (main stream)
        l, o = (j - 1) % 2, j % 2                           # history slots: last frame / the one before (overwritten by this frame at the end)
        if evs: evs["wait0"].record(st)
        st.wait_event(self.join_p[k]); st.wait_event(self.join_l[k])
        if evs: evs["wait1"].record(st)
        cap = None
        if j in self.capture_steps:
            cap = self.captured[j] = {}
            snap = lambda name, x: cap.__setitem__(name, x.clone())
            for name, x in (("kps", self.kps[k]), ("kpu", self.kpu[k]), ("desc", self.desc[k]), ("n", self.n[k]), ("ur", self.ur[k]), ("zd", self.zd[k]), ("kls", self.kls[k]), ("ldesc", self.ldesc[k]),
                            ("leq", self.leq[k]), ("nl", self.nl[k]), ("lab", self.lab[k]), ("pls", self.pls[k]), ("npl", self.npl[k]), ("snrm", self.snrm[k]), ("l3_packed", self.l3[k]["packed"]), ("l3_n_good", self.l3[k]["n_good"]), ("l3_seeds", self.l3[k]["seeds"]),
                            ("l3_lines3d", self.l3[k]["lines3d"]), ("l3_depth_line", self.l3[k]["depth_line"]), ("pose_in", self.pose), ("Rcm_in", self.Rcm),
                            ("last_xw", self.h_xw[l]), ("last_valid", self.h_valid[l]), ("last_desc", self.h_desc[l]), ("last_oct", self.h_oct[l]), ("last_ang", self.h_ang[l]),
                            ("last_n", self.h_n[l]), ("old_xw", self.h_xw[o]), ("old_valid", self.h_valid[o]), ("old_desc", self.h_desc[o]), ("old_normal", self.h_normal[o]),
                            ("old_mind", self.h_mind[o]), ("old_maxd", self.h_maxd[o]), ("old_n", self.h_n[o])):
                snap(name, x)
        if j >= 2 and self.map_set:
            # ---- Track(): Manhattan frame ----
            # the frame's own surface normals (Frame::vSurfaceNormal) and 3-D line directions (Frame::mVF3DLines)
            check(L.planar_track_manhattan_frame_dev(self.ctx_t.h, B, self.Rcm.data_ptr(), self.snrm[k].data_ptr(), self.n_snrm.data_ptr(), self.SN,
                                                     self.l3[k]["packed"].data_ptr(), self.l3[k]["n_good"].data_ptr(), 40,
                                                     self.Rcm_new.data_ptr(), None, None, None))
            if not self.rcm0_set:                           # Rotation_cm = TrackManhattanFrame(FindManhattan(first frame), ...) (src/Tracking.cc:227-230): the first tracked frame
                check(L.planar_copy_rows_dev(self.ctx_t.h, self.Rcm0.data_ptr(), B * 36, self.Rcm_new.data_ptr(), B * 36, B * 36, 1))
                self.rcm0_set = True
            if evs: evs["manhattan"].record(st)
            # ---- TranslationWithMotionModel ----
            fv = self._frame_view(k, self.pose)             # zero-velocity motion model: predicted pose = last pose
            lv = LastFrameView()
            lv.stride = S
            lv.n, lv.Tcw, lv.usable, lv.xw = self.h_n[l].data_ptr(), self.pose.data_ptr(), self.h_valid[l].data_ptr(), self.h_xw[l].data_ptr()
            lv.octave, lv.angle, lv.mp_desc, lv.mp_observed = self.h_oct[l].data_ptr(), self.h_ang[l].data_ptr(), self.h_desc[l].data_ptr(), self.ones_S.data_ptr()
            check(L.planar_reset_matches_dev(self.ctx_t.h, self.pm.data_ptr(), self.pm.numel()))
            check(L.planar_search_by_projection_frame_dev(self.ctx_t.h, C.byref(fv), C.byref(lv), 15.0, 0, 1, self.pm.data_ptr(), self.nm.data_ptr()))
            if evs: evs["proj"].record(st)
            if cap is not None: snap("pm0", self.pm); snap("nm", self.nm)
            check(L.planar_reset_matches_dev(self.ctx_t.h, self.lm.data_ptr(), self.lm.numel()))
            check(L.planar_lsd_search_by_descriptor_dev(self.ctx_t.h, self.kf["ldesc"].data_ptr(), self.kf["n"].data_ptr(), 40, self.ldesc[k].data_ptr(), self.nl[k].data_ptr(), 40,
                                                        self.kf["has_ml"].data_ptr(), B, self.lm.data_ptr(), self.nlm.data_ptr()))
            if self.run_fallback_matcher:
                check(L.planar_reset_matches_dev(self.ctx_t.h, self.cm2.data_ptr(), self.cm2.numel()))
                check(L.planar_match_orb_points_dev(self.ctx_t.h, self.desc[k].data_ptr(), self.n[k].data_ptr(), S, self.h_desc[l].data_ptr(), self.h_n[l].data_ptr(), S,
                                                    self.h_valid[l].data_ptr(), self.zeros_S.data_ptr(), B, self.cm2.data_ptr(), self.npair.data_ptr()))
            if evs: evs["bf"].record(st)
            if cap is not None: snap("lm0", self.lm); snap("nlm0", self.nlm); snap("cm2", self.cm2); snap("npair", self.npair)
            # mvPlaneCoefficients / mnPlaneNum: the refitted coefficients of the planes Frame::ComputePlanes kept (planepost.hip, on the plane stream)
            pc = self.pc[k]
            check(L.planar_reset_matches_dev(self.ctx_t.h, self.plm.data_ptr(), self.plm.numel()))
            check(L.planar_plane_search_by_coefficients_dev(self.ctx_t.h, B, pc["n"].data_ptr(), self.PS, pc["coef"].data_ptr(), self.pose.data_ptr(), 0,
                                                            self.mp["n"].data_ptr(), self.mp["coef"].shape[1], self.mp["valid"].data_ptr(), self.mp["coef"].data_ptr(),
                                                            self.mp["npts"].data_ptr(), self.mp["pts"].shape[2], self.mp["pts"].data_ptr(), self.plane_th.ctypes.data,
                                                            self.plm[0].data_ptr(), self.plm[2].data_ptr(), self.plm[1].data_ptr(), self.nplm.data_ptr()))
            if evs: evs["planes"].record(st)
            if cap is not None: snap("pl_coef", pc["coef"]); snap("pl_n", pc["n"]); snap("pl_src", pc["src"]); snap("pl_off", pc["off"]); snap("pl_pts", pc["pts"]); snap("pl_status", pc["status"]); snap("plm", self.plm); snap("nplm", self.nplm); snap("Rcm_new", self.Rcm_new); snap("Rcm0", self.Rcm0)
            # mRotation_wc.copyTo(mCurrentFrame.mTcw.rowRange(0,3).colRange(0,3)) (:1778): the translation is optimised against the Manhattan rotation of THIS frame.
            pose_t = self.pose
            if self.manhattan_rotation:
                check(L.planar_manhattan_pose_dev(self.ctx_t.h, B, self.Rcm_new.data_ptr(), self.Rcm0.data_ptr(), self.pose.data_ptr(), self.pose_mf.data_ptr()))
                pose_t = self.pose_mf
            self._assemble(0, k, self.pm, self.h_xw[l], self.h_valid[l], S, pose_t)
            self.opt.enqueue_dev(self.pbs[0], 1, 4, 10)     # TranslationOptimization
            A0 = self.pb_arrays[0]
            if cap is not None: cap["pbT"] = {kk: v.clone() for kk, v in A0.items()}
            check(L.planar_discard_outliers_dev(self.ctx_t.h, B, self.n[k].data_ptr(), S, S, self.pm.data_ptr(), A0["pt_outlier"].data_ptr(), self.kept.data_ptr()))
            check(L.planar_discard_outliers_dev(self.ctx_t.h, B, self.nl[k].data_ptr(), 40, 40, self.lm.data_ptr(), A0["ln_outlier"].data_ptr(), None))
            if evs: evs["transl"].record(st)
            if cap is not None: snap("pm1", self.pm); snap("lm1", self.lm); snap("kept", self.kept)
            # ---- TrackLocalMap: SearchLocalPoints + PoseOptimization ----
            T1 = A0["Tcw_out"]
            check(L.planar_blocked_mask_dev(self.ctx_t.h, self.pm.data_ptr(), self.pm.numel(), self.blocked.data_ptr()))
            fv2 = self._frame_view(k, T1, self.blocked)
            pr = self.pr
            check(L.planar_is_in_frustum_points_dev(self.ctx_t.h, C.byref(fv2), self.lsf, self.nlev, self.h_n[o].data_ptr(), S, self.h_valid[o].data_ptr(), self.h_xw[o].data_ptr(),
                                                    self.h_normal[o].data_ptr(), self.h_mind[o].data_ptr(), self.h_maxd[o].data_ptr(), 0.5, pr["in_view"].data_ptr(),
                                                    pr["proj_x"].data_ptr(), pr["proj_y"].data_ptr(), pr["proj_xr"].data_ptr(), pr["level"].data_ptr(), pr["view_cos"].data_ptr()))
            mpv = MapProbes()
            mpv.stride = S
            mpv.n, mpv.in_view, mpv.proj_x, mpv.proj_y, mpv.proj_xr = self.h_n[o].data_ptr(), pr["in_view"].data_ptr(), pr["proj_x"].data_ptr(), pr["proj_y"].data_ptr(), pr["proj_xr"].data_ptr()
            mpv.level, mpv.view_cos, mpv.desc, mpv.observed = pr["level"].data_ptr(), pr["view_cos"].data_ptr(), self.h_desc[o].data_ptr(), self.ones_S.data_ptr()
            check(L.planar_reset_matches_dev(self.ctx_t.h, self.mm.data_ptr(), self.mm.numel()))
            check(L.planar_search_by_projection_map_dev(self.ctx_t.h, C.byref(fv2), C.byref(mpv), 3.0, 0.8, self.mm.data_ptr(), self.nmm.data_ptr()))
            lp = self.lpr
            check(L.planar_is_in_frustum_lines_dev(self.ctx_t.h, C.byref(fv2), self.lsf, self.kf["n"].data_ptr(), 40, self.kf["has_ml"].data_ptr(), self.kf["xw6"].data_ptr(),
                                                   self.kf["normal"].data_ptr(), self.kf["min_dist"].data_ptr(), self.kf["max_dist"].data_ptr(), 0.5, lp["in_view"].data_ptr(),
                                                   lp["proj"].data_ptr(), lp["level"].data_ptr(), lp["view_cos"].data_ptr()))
            check(L.planar_blocked_mask_dev(self.ctx_t.h, self.lm.data_ptr(), self.lm.numel(), self.lblocked.data_ptr()))
            check(L.planar_lsd_search_by_projection_dev(self.ctx_t.h, B, self.nl[k].data_ptr(), 40, self.kls[k].data_ptr(), self.ldesc[k].data_ptr(), self.lblocked.data_ptr(),
                                                        self.kf["n"].data_ptr(), 40, lp["in_view"].data_ptr(), lp["proj"].data_ptr(), lp["level"].data_ptr(), lp["view_cos"].data_ptr(),
                                                        self.kf["ldesc"].data_ptr(), self.kf["has_ml"].data_ptr(), self.sf.ctypes.data, self.nlev, 3.0, 0.6, self.lm.data_ptr(),
                                                        self.nlm.data_ptr()))
            if evs: evs["local"].record(st)
            if cap is not None:
                cap["pr"] = {kk: v.clone() for kk, v in self.pr.items()}; cap["lpr"] = {kk: v.clone() for kk, v in self.lpr.items()}
                snap("mm", self.mm); snap("nmm", self.nmm); snap("lm2", self.lm); snap("nlm2", self.nlm)
            # one index space for the optimiser: [last frame's points | the older frame's points]
            check(L.planar_merge_matches_dev(self.ctx_t.h, self.pm.data_ptr(), self.mm.data_ptr(), S, self.pm.numel(), self.pm_all.data_ptr()))
            rows = lambda dst, dpitch, src, spitch, nbytes: check(L.planar_copy_rows_dev(self.ctx_t.h, dst, dpitch, src, spitch, nbytes, B))
            xb = self.h_xw[l].element_size() * self.h_xw[l][0].numel()          # bytes of one stream's [S][3] block
            rows(self.xw_all.data_ptr(), 2 * xb, self.h_xw[l].data_ptr(), xb, xb); rows(self.xw_all.data_ptr() + xb, 2 * xb, self.h_xw[o].data_ptr(), xb, xb)
            rows(self.valid_all.data_ptr(), 2 * S, self.h_valid[l].data_ptr(), S, S); rows(self.valid_all.data_ptr() + S, 2 * S, self.h_valid[o].data_ptr(), S, S)
            self._assemble(1, k, self.pm_all, self.xw_all, self.valid_all, 2 * S, T1)
            self.opt.enqueue_dev(self.pbs[1], 0, 4, 10)     # PoseOptimization
            if cap is not None: cap["pbP"] = {kk: v.clone() for kk, v in self.pb_arrays[1].items()}; snap("pm_all", self.pm_all)
            check(L.planar_copy_rows_dev(self.ctx_t.h, self.pose.data_ptr(), B * 64, self.pb_arrays[1]["Tcw_out"].data_ptr(), B * 64, B * 64, 1))
            check(L.planar_copy_rows_dev(self.ctx_t.h, self.Rcm.data_ptr(), B * 36, self.Rcm_new.data_ptr(), B * 36, B * 36, 1))
            if evs: evs["pose"].record(st)
        elif evs:
            for name in ("manhattan", "proj", "bf", "planes", "transl", "local", "pose"):
                evs[name].record(st)
        # ---- the new frame becomes a "last frame": back-projected keypoints (UnprojectStereo) and what MapPoint::UpdateNormalAndDepth keeps ----
        self._stereo(self.ctx_t, k, self.pose, depth, self.ur[k], self.zd[k], self.h_xw[o], self.h_valid[o])
        check(L.planar_update_normal_and_depth_dev(self.ctx_t.h, B, self.n[k].data_ptr(), S, self.h_xw[o].data_ptr(), self.h_valid[o].data_ptr(), self.pose.data_ptr(),
                                                   self.kpu[k].data_ptr(), None, None, self.sf.ctypes.data, self.nlev, self.h_normal[o].data_ptr(), self.h_mind[o].data_ptr(),
                                                   self.h_maxd[o].data_ptr()))
        check(L.planar_keypoint_fields_dev(self.ctx_t.h, self.kps[k].data_ptr(), B * S, self.h_oct[o].data_ptr(), self.h_ang[o].data_ptr()))
        check(L.planar_copy_rows_dev(self.ctx_t.h, self.h_desc[o].data_ptr(), B * S * 32, self.desc[k].data_ptr(), B * S * 32, B * S * 32, 1))
        check(L.planar_copy_rows_dev(self.ctx_t.h, self.h_n[o].data_ptr(), B * 4, self.n[k].data_ptr(), B * 4, B * 4, 1))
        if cap is not None: snap("pose_out", self.pose); snap("new_xw", self.h_xw[o]); snap("new_valid", self.h_valid[o]); snap("new_ur", self.ur[k]); snap("new_normal", self.h_normal[o]); snap("new_mind", self.h_mind[o]); snap("new_maxd", self.h_maxd[o])
        if evs: evs["state"].record(st)
        self.done[k].record(st)

    # ------------------------------------------------------------------------------------------------------------------------------
    def check(self):
        """Capacity flags of the extractors (synchronises): raises if a frame overflowed an internal capacity."""
        from ._lib import check
        check(self.ex.L.planar_orb_check(self.ex.h))
        for q in self.pds:
            check(q.L.planar_peac_check(q.h, self.B))
        for q in self.lss:
            check(self.L.planar_lsd_check(q.h, self.B))
        for k, pc in enumerate(self.pc):
            bad = pc["status"].nonzero()
            if len(bad):
                raise RuntimeError(f"plane post-processing: buffer set {k}, frame {int(bad[0])} reported code {int(pc['status'][bad[0]])} "
                                   "(3 = more voxels than max_points / coordinate range, 4 = sampler table exhausted)")


def find_manhattan(normals, sizes, ver_th=0.08716):
    """Host-side stand-in of Map::FindManhattan (reference src/Map.cc:160-363; Map is out of scope), plane branch: the pair of frame planes whose
    normals are perpendicular within ver_th with the most points; their normals (signs: largest component positive) and the cross product, sorted by the axis
    each is closest to, orthonormalised by U * V^T of the SVD.  normals [n,3], sizes [n] -> Rotation_cm [3,3] f32 (identity without such a pair)."""
    best, max_size = None, 0
    n = len(normals)
    for i in range(n):
        for j in range(i + 1, n):
            angle = float(np.float32(normals[i][0]) * np.float32(normals[j][0]) + np.float32(normals[i][1]) * np.float32(normals[j][1]) + np.float32(normals[i][2]) * np.float32(normals[j][2]))
            if -ver_th < angle < ver_th and sizes[i] + sizes[j] > max_size:
                max_size = sizes[i] + sizes[j]
                best = (np.asarray(normals[i], np.float32).copy(), np.asarray(normals[j], np.float32).copy())
    R = np.eye(3, dtype=np.float32)
    if best is None:
        return R
    p1, p2 = best
    loc1 = int(np.argmax(np.abs(p1))); p1 = -p1 if p1[loc1] < 0 else p1
    loc2 = int(np.argmax(np.abs(p2))); p2 = -p2 if p2[loc2] < 0 else p2
    p3 = np.cross(p1, p2).astype(np.float32)
    loc3 = int(np.argmax(np.abs(p3))); p3 = -p3 if p3[loc3] < 0 else p3
    by_axis = {loc1: p1}
    by_axis[loc2] = p2
    by_axis[loc3] = p3
    if len(by_axis) < 3:
        return R
    M = np.stack([by_axis[0], by_axis[1], by_axis[2]], 1).astype(np.float32)       # columns: the x-, y-, z-like directions
    U, _, Vt = np.linalg.svd(M.astype(np.float64))
    return (U @ Vt).astype(np.float32)


def build_map(gray0, depth0, cam, torch_dev=None, seed=0, n_map_planes=8, n_plane_pts=128, n_normals=4096):
    """Host-side stand-in for the map of each stream, built from its first frame with the product extractors (numpy in, numpy out):
    the reference key frame's lines (end points back-projected with the depth at the end points, as Frame::isLineGood does when the depth is
    valid), the map planes (PEAC planes of the first frame at the identity pose with sampled member points) and synthetic surface normals."""
    from . import PlaneDetection
    from .lines import LineSegment
    from .synth import manhattan_scene
    B, H, W = gray0.shape
    CH = 256        # the extractors' workspaces here are scratch beside the pipeline's own (17 MB per frame of max_batch): a chunk of the streams at a time
    ls = LineSegment(W, H, min(B, CH))
    parts = [ls.ExtractLineSegment(gray0[a:a + CH]) for a in range(0, B, CH)]
    kl, ldesc, eq, nl = (np.concatenate([p[i] for p in parts]) for i in range(4))
    del ls, parts
    fx, fy, cx, cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    xw6 = np.zeros((B, 40, 6)); normal = np.zeros((B, 40, 3)); mind = np.zeros((B, 40), np.float32); maxd = np.zeros((B, 40), np.float32)
    dm = depth0.astype(np.float64) / 5000.0
    for b in range(B):
        for i in range(int(nl[b])):
            pts = []
            for (x, y) in ((kl[b, i]["start_x"], kl[b, i]["start_y"]), (kl[b, i]["end_x"], kl[b, i]["end_y"])):
                xi, yi = int(min(max(x, 0), W - 1)), int(min(max(y, 0), H - 1))
                zz = dm[b, yi, xi] if dm[b, yi, xi] > 0 else 2.0
                pts.append([(x - cx) * zz / fx, (y - cy) * zz / fy, zz])
            xw6[b, i] = np.concatenate(pts)
            mid = 0.5 * (np.array(pts[0]) + np.array(pts[1])); d = np.linalg.norm(mid)
            normal[b, i] = mid / max(d, 1e-9); maxd[b, i] = d * 1.2 ** 3; mind[b, i] = maxd[b, i] / 1.2 ** 7
    kf_lines = dict(n=nl.astype(np.int32), ldesc=ldesc, xw6=xw6, normal=normal, min_dist=mind, max_dist=maxd)
    pd = PlaneDetection(W, H, max_batch=min(B, CH))
    res = [r for a in range(0, B, CH) for r in pd.run(depth0[a:a + CH].astype(np.uint16), K=(fx, fy, cx, cy))]
    del pd
    M, P = n_map_planes, n_plane_pts
    mp = dict(n=np.zeros(B, np.int32), valid=np.zeros((B, M), np.uint8), coef=np.zeros((B, M, 4), np.float32), npts=np.zeros((B, M), np.int32), pts=np.zeros((B, M, P, 3), np.float32))
    ys, xs = np.mgrid[4:H:16, 4:W:16]
    for b in range(B):
        planes, labels = res[b]
        m = min(len(planes), M)
        mp["n"][b] = m
        for q in range(m):
            nrm, c = np.asarray(planes[q][1:4]), np.asarray(planes[q][4:7])
            mp["coef"][b, q] = [*nrm, -float(nrm @ c)]
            sel = labels[ys, xs] == q
            yy, xx = ys[sel][:P], xs[sel][:P]
            zz = dm[b, yy, xx]
            pp = np.stack([(xx - cx) * zz / fx, (yy - cy) * zz / fy, zz], -1)
            mp["npts"][b, q] = len(pp); mp["pts"][b, q, :len(pp)] = pp; mp["valid"][b, q] = 1
    sc = manhattan_scene(B=B, n_normals=n_normals, n_lines=40, seed=21 + seed)
    # Rotation_cm's seed: FindManhattan on the first frame's planes (refined by TrackManhattanFrame on the first tracked frame, TrackPipeline._track_on_stream)
    R_last = np.stack([find_manhattan([np.asarray(pl[1:4]) for pl in res[b][0]], [int((res[b][1] == q).sum()) for q in range(len(res[b][0]))]) for b in range(B)])
    return kf_lines, mp, dict(normals=sc["normals"], n_normals=sc["n_normals"], lines=sc["lines"], n_lines=sc["n_lines"], R_last=R_last)
