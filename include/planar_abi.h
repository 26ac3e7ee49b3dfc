/* planar_abi.h — C ABI of libplanar_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for PlanarSLAM's per-frame extract -> match -> pose-optimise
 * hot path.  POD only: plain pointers and sizes, no OpenCV /
 * Eigen / torch types.  Every entry point cites the reference interface it
 * replaces (paths relative to the PlanarSLAM tree).
 *
 * Conventions
 *  - return 0 (PLANAR_OK) on success, a negative PLANAR_E* code otherwise;
 *    never throws, never exits.  planar_last_error() gives a message (per thread).
 *  - "_dev" entry points take DEVICE pointers and only enqueue work on the
 *    context's stream (results valid after planar_ctx_sync / a stream sync);
 *    entry points without the suffix take HOST pointers, are synchronous, and
 *    are what the reference-signature adapters call.
 *  - the caller owns every buffer; a context is not re-entrant (one per host
 *    thread, as the reference's ORBextractor instance is, include/ORBextractor.h:85).
 *  - a batch is B independent frames; frame b of a per-frame array starts at
 *    b * <documented stride>.
 */
#ifndef PLANAR_ABI_H
#define PLANAR_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLANAR_OK 0
#define PLANAR_EINVAL (-1)   /* bad argument (null pointer, size out of range) */
#define PLANAR_ENOMEM (-2)   /* device/host allocation failed */
#define PLANAR_EDEVICE (-3)  /* HIP runtime error (no device, launch failure) */
#define PLANAR_ECAPACITY (-4) /* output capacity too small */
#define PLANAR_ESTATE (-5)   /* call out of order */

typedef struct planar_ctx planar_ctx;

/* Same 28-byte layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id),
 * so an adapter can memcpy into std::vector<cv::KeyPoint>. */
typedef struct planar_keypoint {
    float x, y;       /* level-0 pixel coordinates (level coords * scale[octave]), ORBextractor.cc:1094-1100 */
    float size;       /* (int)(31 * scale[octave]),                                  ORBextractor.cc:835,845 */
    float angle;      /* IC_Angle, degrees [0,360),                                  ORBextractor.cc:77-104 */
    float response;   /* FAST-9/16 corner score                                                              */
    int32_t octave;   /* pyramid level                                                                       */
    int32_t class_id; /* -1                                                                                  */
} planar_keypoint;

/* ---- context ------------------------------------------------------------------------- */
/* Creates a context on HIP device `device` with its own stream. */
int planar_ctx_create(planar_ctx** out, int device);
void planar_ctx_destroy(planar_ctx* ctx);
/* Use an external hipStream_t (e.g. torch's current stream) for all subsequent work; NULL
 * restores the context's own stream. */
int planar_ctx_set_stream(planar_ctx* ctx, void* hip_stream);
/* CU partition for the extractors' one-wavefront-per-frame kernels (no reference counterpart: the reference runs its three extractors on three CPU threads,
 * src/Frame.cc:90-95; this is the device-side analogue of pinning the sequential ones to some cores).  planar_cu_stream_create makes a HIP stream whose kernels
 * run only on the compute units whose bit is set in cu_mask (n_words x 32 bits, bit i = CU i in the driver's round-robin-over-XCDs numbering:
 * hipExtStreamCreateWithCUMask); planar_ctx_set_seq_stream(ctx, s) makes the context launch the PEAC clustering kernel and LSD's region-growing kernel on s
 * (forked from / joined into the context's stream by events, so results and ordering are unchanged); s = NULL restores the single-stream behaviour (default).
 * One such stream may be shared by several contexts.  The caller destroys it after the contexts. */
int planar_cu_stream_create(int device, const uint32_t* cu_mask, int n_words, void** out_stream);
void planar_cu_stream_destroy(void* stream);
int planar_ctx_set_seq_stream(planar_ctx* ctx, void* stream);
void* planar_ctx_get_stream(planar_ctx* ctx);
int planar_ctx_sync(planar_ctx* ctx);
const char* planar_last_error(void);
/* Library/ABI version: major*10000 + minor*100 + patch. */
int planar_abi_version(void);

/* ---- ORB extractor (replaces Planar_SLAM::ORBextractor, include/ORBextractor.h:45-112) -- */
typedef struct planar_orb planar_orb;

typedef struct planar_orb_params {
    int32_t nfeatures;    /* ORBextractor.nFeatures   (1000) */
    float scale_factor;   /* ORBextractor.scaleFactor (1.2)  */
    int32_t nlevels;      /* ORBextractor.nLevels     (8)    */
    int32_t ini_th_fast;  /* ORBextractor.iniThFAST   (20)   */
    int32_t min_th_fast;  /* ORBextractor.minThFAST   (7)    */
} planar_orb_params;

/* ORBextractor::ORBextractor (src/ORBextractor.cc:410-470) + all device workspace for
 * batches of up to max_batch frames of width x height 8-bit gray. */
int planar_orb_create(planar_ctx* ctx, const planar_orb_params* params, int width, int height, int max_batch,
                      planar_orb** out);
void planar_orb_destroy(planar_orb* orb);
/* Upper bound on keypoints per frame (nfeatures + small octree overshoot); the per-frame
 * stride of `kps`/`desc` below. */
int planar_orb_max_keypoints(const planar_orb* orb);
/* Getters of the reference class (include/ORBextractor.h:63-83): out[nlevels]. */
int planar_orb_get_scale_factors(const planar_orb* orb, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2);
int planar_orb_level_size(const planar_orb* orb, int level, int* w, int* h);
int planar_orb_features_per_level(const planar_orb* orb, int32_t* out);

/* ORBextractor::operator() (src/ORBextractor.cc:1043-1105) on B frames.
 *   gray   : B frames, row pitch `pitch` bytes, frame stride `frame_stride` bytes
 *   kps    : [B][planar_orb_max_keypoints]           desc : [B][max_keypoints][32]
 *   n_out  : [B] keypoints found per frame
 * Host-pointer version: copies in, runs, copies out, synchronous. */
int planar_orb_extract(planar_orb* orb, const uint8_t* gray, int B, int pitch, int64_t frame_stride,
                       planar_keypoint* kps, uint8_t* desc, int32_t* n_out);
/* Device-pointer version: enqueues on the context stream and returns. */
int planar_orb_extract_dev(planar_orb* orb, const uint8_t* d_gray, int B, int pitch, int64_t frame_stride,
                           planar_keypoint* d_kps, uint8_t* d_desc, int32_t* d_n_out);

/* Synchronises; PLANAR_ECAPACITY if a FAST cell produced more non-max-suppressed corners than its candidate slots since the last check
 * (the slot count is the mathematical bound on strict 8-neighbour maxima, so this reports a broken invariant, not a tuning limit).
 * The host-pointer entry point calls it itself. */
int planar_orb_check(planar_orb* orb);

/* Stage read-back for parity tests and for the adapter's public mvImagePyramid member
 * (include/ORBextractor.h:85).  Valid after an extract call; host output buffers. */
int planar_orb_read_level(planar_orb* orb, int frame, int level, uint8_t* out /* w*h */);
int planar_orb_read_blurred(planar_orb* orb, int frame, int level, uint8_t* out /* w*h */);
/* FAST survivors of one level in the reference's emission order (ORBextractor.cc:789-826):
 * out = n x {x, y, score} relative to (minBorderX, minBorderY); returns n or a negative code. */
int planar_orb_read_candidates(planar_orb* orb, int frame, int level, int32_t* out, int cap);

/* Per-launch timing with HIP events on the context stream (bench.py's roofline leg).
 * set_profiling(1) starts recording an event before/after every kernel launch of each
 * extract call; get_profile synchronises, returns the summed milliseconds per launch slot
 * (total_ms[planar_orb_profile_num_launches]) and the number of recorded calls, and resets. */
int planar_orb_set_profiling(planar_orb* orb, int enable);
int planar_orb_profile_num_launches(const planar_orb* orb);
const char* planar_orb_profile_launch_name(const planar_orb* orb, int i);
int planar_orb_get_profile(planar_orb* orb, double* total_ms, int64_t* calls);

/* ---- pose optimisation (replaces Optimizer::PoseOptimization, src/Optimizer.cc:550-1275, and
 *      Optimizer::TranslationOptimization, :2995-3738; include/Optimizer.h:37,43) ------------ */
#define PLANAR_POSE_FULL 0         /* PoseOptimization: point/line/plane/parallel/vertical edges, 6-DoF */
#define PLANAR_POSE_TRANSLATION 1  /* TranslationOptimization: ...OnlyTranslation edges, rotation kept  */

typedef struct planar_pose_params {
    float fx, fy, cx, cy, bf;      /* Frame::fx, fy, cx, cy, mbf                                         */
    /* raw Config values (src/Optimizer.cc:771-783): Plane.AngleInfo, Plane.DistanceInfo,
     * Plane.ParallelInfo, Plane.VerticalInfo, Plane.Chi, Plane.VPChi                                    */
    double angle_info, distance_info, parallel_info, vertical_info, plane_chi, vp_chi;
} planar_pose_params;

/* B frames, structure-of-arrays with per-frame strides max_points / max_lines / max_planes.
 * Inputs mirror the Frame fields the reference reads:                            */
typedef struct planar_pose_batch {
    int32_t B, max_points, max_lines, max_planes;
    const int32_t* n_points;       /* [B]  Frame::N                                                      */
    const int32_t* n_lines;        /* [B]  Frame::NL                                                     */
    const int32_t* n_planes;       /* [B]  Frame::mnPlaneNum                                             */
    const uint8_t* pt_valid;       /* [B][max_points]      mvpMapPoints[i] != NULL                       */
    const float* pt_xw;            /* [B][max_points][3]   MapPoint::GetWorldPos() (float32)             */
    const float* pt_obs;           /* [B][max_points][3]   mvKeysUn[i].pt.x, .pt.y, mvuRight[i] (<0: mono) */
    const float* pt_inv_sigma2;    /* [B][max_points]      mvInvLevelSigma2[mvKeysUn[i].octave]          */
    const uint8_t* ln_valid;       /* [B][max_lines]       mvpMapLines[i] != NULL                        */
    const double* ln_obs;          /* [B][max_lines][3]    mvKeyLineFunctions[i]                         */
    const double* ln_xw;           /* [B][max_lines][6]    MapLine::mWorldPos (start xyz, end xyz)       */
    const float* pl_meas;          /* [B][max_planes][4]   mvPlaneCoefficients[i]                        */
    const uint8_t* pl_valid;       /* [B][max_planes][3]   mvpMapPlanes / mvpParallelPlanes / mvpVerticalPlanes[i] != NULL */
    const float* pl_world;         /* [B][max_planes][3][4] GetWorldPos() of those three map planes      */
    const float* Tcw_in;           /* [B][16]              Frame::mTcw, row-major 4x4 float32            */
    /* outputs */
    float* Tcw_out;                /* [B][16]              pose passed to Frame::SetPose                 */
    uint8_t* pt_outlier;           /* [B][max_points]      mvbOutlier      (written only where pt_valid) */
    uint8_t* ln_outlier;           /* [B][max_lines]       mvbLineOutlier  (written only where ln_valid) */
    uint8_t* pl_outlier;           /* [B][max_planes][3]   mvbPlaneOutlier / mvbParPlaneOutlier / mvbVerPlaneOutlier */
    int32_t* n_inliers;            /* [B]                  the reference function's return value         */
    int32_t* lm_iters;             /* [B] or NULL          LM iterations run (diagnostic)                */
} planar_pose_batch;

/* rounds = 4, its = 10 reproduce the reference protocol (4 x optimize(10) with outlier
 * re-classification, src/Optimizer.cc:990-1000).  All pointers in `batch` are HOST pointers for
 * planar_pose_opt (synchronous) and DEVICE pointers for planar_pose_opt_dev (enqueue only; the
 * struct itself is always host memory, passed by value to the kernel). */
int planar_pose_opt(planar_ctx* ctx, const planar_pose_batch* batch, const planar_pose_params* params, int mode, int rounds, int its);
int planar_pose_opt_dev(planar_ctx* ctx, const planar_pose_batch* d_batch, const planar_pose_params* params, int mode, int rounds, int its);

/* ---- descriptor matchers ------------------------------------------------------------------
 * Descriptors are 32-byte rows (cv::Mat N x 32 CV_8U: Frame::mDescriptors, Frame::mLdesc).
 * Batched over B frame pairs: pair b uses rows [b*stride, b*stride + n[b]).                      */

/* cv::BFMatcher(cv::NORM_HAMMING).match (k = 1) / .knnMatch(..., 2) (k = 2), as called at
 * src/ORBmatcher.cc:1346-1347 and src/LSDmatcher.cpp:249-254.  idx/dist: [B][q_stride][k], ascending
 * distance, lowest train index on ties; a missing neighbour is idx -1 / dist INT32_MAX. */
int planar_hamming_knn(planar_ctx* ctx, const uint8_t* q, const int32_t* nq, int q_stride, const uint8_t* t, const int32_t* nt,
                       int t_stride, int B, int k, int32_t* idx, int32_t* dist);
int planar_hamming_knn_dev(planar_ctx* ctx, const uint8_t* d_q, const int32_t* d_nq, int q_stride, const uint8_t* d_t,
                           const int32_t* d_nt, int t_stride, int B, int k, int32_t* d_idx, int32_t* d_dist);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:259-324) for n_points map points at once: point p observes the descriptors
 * desc[off[p] .. off[p + 1]) (the rows pKF->mDescriptors.row(idx) of its non-bad key frames, in the order of the observation map); best[p] = index within that
 * list of the descriptor with the least median Hamming distance to all of them (first minimum; -1 without observations), median[p] (or NULL) that median.
 * _dev: max_obs = an upper bound of the observations per point (<= 2047; a point with more gets best = -2). */
int planar_distinctive_descriptors(planar_ctx* ctx, int n_points, const uint8_t* desc, const int32_t* off, int32_t* best, int32_t* median);
int planar_distinctive_descriptors_dev(planar_ctx* ctx, int n_points, const uint8_t* d_desc, const int32_t* d_off, int max_obs, int32_t* d_best, int32_t* d_median);

/* ORBmatcher::MatchORBPoints(Frame& Current, const Frame& Last) (src/ORBmatcher.cc:1332-1394).
 *   last_has_mp[j]   LastFrame.mvpMapPoints[j] != NULL        last_outlier[j]  LastFrame.mvbOutlier[j]
 *   cur_match[q]     (in/out) index j of the last-frame keypoint whose MapPoint the reference copies into
 *                    CurrentFrame.mvpMapPoints[q]; entries the reference does not assign keep their value
 *   npair[b]         the function's return value (number of good matches)                           */
int planar_match_orb_points(planar_ctx* ctx, const uint8_t* cur, const int32_t* n_cur, int cur_stride, const uint8_t* last,
                            const int32_t* n_last, int last_stride, const uint8_t* last_has_mp, const uint8_t* last_outlier, int B,
                            int32_t* cur_match, int32_t* npair);
int planar_match_orb_points_dev(planar_ctx* ctx, const uint8_t* d_cur, const int32_t* d_n_cur, int cur_stride, const uint8_t* d_last,
                                const int32_t* d_n_last, int last_stride, const uint8_t* d_last_has_mp, const uint8_t* d_last_outlier,
                                int B, int32_t* d_cur_match, int32_t* d_npair);

/* LSDmatcher::SearchByDescriptor(KeyFrame*, Frame&, vector<MapLine*>&) (src/LSDmatcher.cpp:242-279).
 *   kf_has_ml[i]     pKF->GetMapLineMatches()[i] != NULL
 *   cur_match[t]     (out) keyframe line index whose MapLine lands in vpMapLineMatches[t], or -1
 *   nmatches[b]      the function's return value                                                    */
int planar_lsd_search_by_descriptor(planar_ctx* ctx, const uint8_t* kf, const int32_t* n_kf, int kf_stride, const uint8_t* cur,
                                    const int32_t* n_cur, int cur_stride, const uint8_t* kf_has_ml, int B, int32_t* cur_match,
                                    int32_t* nmatches);
int planar_lsd_search_by_descriptor_dev(planar_ctx* ctx, const uint8_t* d_kf, const int32_t* d_n_kf, int kf_stride, const uint8_t* d_cur,
                                        const int32_t* d_n_cur, int cur_stride, const uint8_t* d_kf_has_ml, int B, int32_t* d_cur_match,
                                        int32_t* d_nmatches);

/* ---- guided (projection / vocabulary-node / coefficient) matchers ---------------------------
 * These reproduce the reference's SEQUENTIAL assignment semantics exactly: probes (map points, last-frame
 * points, key-frame features) are resolved in the reference's loop order and every assignment updates the
 * "already matched" state later probes see.  All arrays are batched over B independent frames with fixed
 * per-frame strides; pointers are HOST pointers for the plain entry points and DEVICE pointers for *_dev
 * (the view structs themselves are always host memory). */
#define PLANAR_MAX_LEVELS 16
#define PLANAR_GRID_COLS 64 /* FRAME_GRID_COLS include/Frame.h:38 */
#define PLANAR_GRID_ROWS 48 /* FRAME_GRID_ROWS include/Frame.h:37 */
#define PLANAR_MAX_FRAME_KEYS 4096 /* limit on Frame::N / probes per frame for the guided matchers */

/* The fields of the Frame being matched INTO that the guided matchers read (include/Frame.h). */
typedef struct planar_frame_view {
    int32_t B, stride;               /* frames, per-frame capacity of the arrays below                     */
    const int32_t* n;                /* [B]             Frame::N                                            */
    const planar_keypoint* keys_un;  /* [B][stride]     mvKeysUn (pt, angle, octave are read)               */
    const float* u_right;            /* [B][stride]     mvuRight                                            */
    const uint8_t* desc;             /* [B][stride][32] mDescriptors                                        */
    const uint8_t* blocked;          /* [B][stride]     mvpMapPoints[i] != NULL && ->Observations() > 0 on entry (NULL = none) */
    const float* Tcw;                /* [B][16]         mTcw (frame-to-frame variant only)                  */
    float min_x, max_x, min_y, max_y; /* mnMinX, mnMaxX, mnMinY, mnMaxY (src/Frame.cc:117-123)              */
    float grid_w_inv, grid_h_inv;    /* mfGridElementWidthInv / mfGridElementHeightInv                      */
    float fx, fy, cx, cy, bf, b;     /* Frame::fx.. , mbf, mb                                               */
    float scale_factors[PLANAR_MAX_LEVELS]; /* mvScaleFactors                                              */
} planar_frame_view;

/* LastFrame side of ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono). */
typedef struct planar_last_frame_view {
    int32_t stride;
    const int32_t* n;            /* [B]             LastFrame.N                                              */
    const float* Tcw;            /* [B][16]         LastFrame.mTcw                                           */
    const uint8_t* usable;       /* [B][stride]     mvpMapPoints[i] != NULL && !mvbOutlier[i]                */
    const float* xw;             /* [B][stride][3]  mvpMapPoints[i]->GetWorldPos()                           */
    const int32_t* octave;       /* [B][stride]     mvKeys[i].octave                                         */
    const float* angle;          /* [B][stride]     mvKeysUn[i].angle                                        */
    const uint8_t* mp_desc;      /* [B][stride][32] mvpMapPoints[i]->GetDescriptor()                         */
    const uint8_t* mp_observed;  /* [B][stride]     mvpMapPoints[i]->Observations() > 0                      */
} planar_last_frame_view;

/* ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) (src/ORBmatcher.cc:1396-1535).
 *   cur_match[b][i2] (in/out): index i of the LAST-frame keypoint whose MapPoint the reference stores in
 *       CurrentFrame.mvpMapPoints[i2]; -1 where the rotation check resets it to NULL; untouched otherwise.
 *   nmatches[b]: the function's return value. */
int planar_search_by_projection_frame(planar_ctx* ctx, const planar_frame_view* cur, const planar_last_frame_view* last, float th,
                                      int mono, int check_orientation, int32_t* cur_match, int32_t* nmatches);
int planar_search_by_projection_frame_dev(planar_ctx* ctx, const planar_frame_view* d_cur, const planar_last_frame_view* d_last, float th,
                                          int mono, int check_orientation, int32_t* d_cur_match, int32_t* d_nmatches);

/* The MapPoint tracking fields written by Frame::isInFrustum (src/Frame.cc:312-367) and read by
 * ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:46-130). */
typedef struct planar_map_probes {
    int32_t stride;
    const int32_t* n;            /* [B]             vpMapPoints.size()                                       */
    const uint8_t* in_view;      /* [B][stride]     mbTrackInView && !isBad()                                */
    const float* proj_x;         /* [B][stride]     mTrackProjX                                              */
    const float* proj_y;         /* [B][stride]     mTrackProjY                                              */
    const float* proj_xr;        /* [B][stride]     mTrackProjXR                                             */
    const int32_t* level;        /* [B][stride]     mnTrackScaleLevel                                        */
    const float* view_cos;       /* [B][stride]     mTrackViewCos                                            */
    const uint8_t* desc;         /* [B][stride][32] GetDescriptor()                                          */
    const uint8_t* observed;     /* [B][stride]     Observations() > 0                                       */
} planar_map_probes;

/* match[b][idx] (in/out): index iMP of the map point stored in F.mvpMapPoints[idx]. nn_ratio = mfNNratio. */
int planar_search_by_projection_map(planar_ctx* ctx, const planar_frame_view* frame, const planar_map_probes* probes, float th,
                                    float nn_ratio, int32_t* match, int32_t* nmatches);
int planar_search_by_projection_map_dev(planar_ctx* ctx, const planar_frame_view* d_frame, const planar_map_probes* d_probes, float th,
                                        float nn_ratio, int32_t* d_match, int32_t* d_nmatches);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (src/ORBmatcher.cc:160-292).
 * The DBoW2 FeatureVectors (node id -> feature indices) are passed as one node id per feature (-1: feature
 * is in no node), which is what DBoW2's transform(..., levelsup) produces (each feature falls in one node).
 *   kf_usable[i]  vpMapPointsKF[i] != NULL && !isBad()         kf_angle[i]  pKF->mvKeysUn[i].angle
 *   f_angle[i]    F.mvKeys[i].angle
 *   match[b][iF]  (out) index of the key-frame feature whose MapPoint lands in vpMapPointMatches[iF], else -1 */
int planar_search_by_bow(planar_ctx* ctx, int B, const int32_t* n_kf, int kf_stride, const int32_t* kf_node, const uint8_t* kf_usable,
                         const float* kf_angle, const uint8_t* kf_desc, const int32_t* n_f, int f_stride, const int32_t* f_node,
                         const float* f_angle, const uint8_t* f_desc, float nn_ratio, int check_orientation, int32_t* match,
                         int32_t* nmatches);
int planar_search_by_bow_dev(planar_ctx* ctx, int B, const int32_t* d_n_kf, int kf_stride, const int32_t* d_kf_node,
                             const uint8_t* d_kf_usable, const float* d_kf_angle, const uint8_t* d_kf_desc, const int32_t* d_n_f,
                             int f_stride, const int32_t* d_f_node, const float* d_f_angle, const uint8_t* d_f_desc, float nn_ratio,
                             int check_orientation, int32_t* d_match, int32_t* d_nmatches);

/* Frame::isInFrustum(MapPoint*, viewingCosLimit) (src/Frame.cc:312-367) for every local map point of B frames: fills the tracking
 * fields SearchByProjection(F, vpMapPoints, th) reads (the non-const twins of planar_map_probes' arrays).
 *   frame: Tcw, fx fy cx cy bf, min/max bounds are read (mRcw, mtcw, mOw are derived as Frame::UpdatePoseMatrices does, :301-306)
 *   log_scale_factor = Frame::mfLogScaleFactor, n_levels = mnScaleLevels (MapPoint::PredictScale, src/MapPoint.cc:419-434)
 *   valid[j] = vpMapPoints[j] usable; xw = GetWorldPos(), normal = GetNormal(), min_dist / max_dist = mfMinDistance / mfMaxDistance
 *   (GetMin/MaxDistanceInvariance apply the 0.8 / 1.2 factors, src/MapPoint.cc:390-400) */
int planar_is_in_frustum_points(planar_ctx* ctx, const planar_frame_view* frame, float log_scale_factor, int n_levels, const int32_t* n, int stride,
                                const uint8_t* valid, const float* xw, const float* normal, const float* min_dist, const float* max_dist,
                                float viewing_cos_limit, uint8_t* in_view, float* proj_x, float* proj_y, float* proj_xr, int32_t* level,
                                float* view_cos);
int planar_is_in_frustum_points_dev(planar_ctx* ctx, const planar_frame_view* d_frame, float log_scale_factor, int n_levels, const int32_t* d_n,
                                    int stride, const uint8_t* d_valid, const float* d_xw, const float* d_normal, const float* d_min_dist,
                                    const float* d_max_dist, float viewing_cos_limit, uint8_t* d_in_view, float* d_proj_x, float* d_proj_y,
                                    float* d_proj_xr, int32_t* d_level, float* d_view_cos);
/* Frame::isInFrustum(MapLine*, viewingCosLimit) (src/Frame.cc:369-438): xw6 = MapLine::GetWorldPos() (Vector6d: start xyz, end xyz),
 * normal = GetNormal() (Vector3d), level = MapLine::PredictScale (src/MapLine.cpp:381-390, NOT clamped to the pyramid, as in the reference);
 * proj[j] = {mTrackProjX1, Y1, X2, Y2}. */
int planar_is_in_frustum_lines(planar_ctx* ctx, const planar_frame_view* frame, float log_scale_factor, const int32_t* n, int stride,
                               const uint8_t* valid, const double* xw6, const double* normal, const float* min_dist, const float* max_dist,
                               float viewing_cos_limit, uint8_t* in_view, float* proj, int32_t* level, float* view_cos);
int planar_is_in_frustum_lines_dev(planar_ctx* ctx, const planar_frame_view* d_frame, float log_scale_factor, const int32_t* d_n, int stride,
                                   const uint8_t* d_valid, const double* d_xw6, const double* d_normal, const float* d_min_dist,
                                   const float* d_max_dist, float viewing_cos_limit, uint8_t* d_in_view, float* d_proj, int32_t* d_level,
                                   float* d_view_cos);

/* ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th) (src/ORBmatcher.cc:829-979), the SEARCH half: for every map point the
 * keypoint of the key frame it would be fused with (:847-951: projection, KeyFrame::IsInImage src/KeyFrame.cc:715-718, distance / viewing-angle
 * gates, MapPoint::PredictScale, KeyFrame::GetFeaturesInArea src/KeyFrame.cc:639-678, level / chi-square gates, smallest Hamming distance, TH_LOW).
 * The map edits that follow (:953-974: Replace / AddObservation / AddMapPoint) stay with the caller's map: a point's gates read only its own state
 * on entry, so they do not feed back into the search.  Batched over B key frames (LocalMapping::SearchInNeighbors fuses the same list into every
 * neighbour key frame: points_shared = 1).
 *   kf: keys_un, u_right, desc, Tcw, bounds, grid, fx fy cx cy bf, scale_factors are read (blocked is ignored)
 *   inv_level_sigma2[n_levels] = KeyFrame::mvInvLevelSigma2; log_scale_factor / n_levels = mfLogScaleFactor / mnScaleLevels
 *   usable[j] = vpMapPoints[j] != NULL && !isBad() && !IsInKeyFrame(pKF); xw / normal / desc = GetWorldPos() / GetNormal() / GetDescriptor();
 *   min_dist / max_dist = mfMinDistance / mfMaxDistance (the 0.8 / 1.2 factors of GetMin/MaxDistanceInvariance are applied here)
 *   fuse_idx[b][j] = bestIdx where bestDist <= TH_LOW (the reference then counts the point in its return value), else -1;
 *   fuse_dist[b][j] (may be NULL) = bestDist (256: no candidate); rows j >= n[b] read -1 / 256; n_fused[b] = the function's return value. */
int planar_fuse_search(planar_ctx* ctx, const planar_frame_view* kf, const float* inv_level_sigma2, float log_scale_factor, int n_levels,
                       const int32_t* n, int stride, int points_shared, const uint8_t* usable, const float* xw, const float* normal,
                       const float* min_dist, const float* max_dist, const uint8_t* desc, float th, int32_t* fuse_idx, int32_t* fuse_dist,
                       int32_t* n_fused);
int planar_fuse_search_dev(planar_ctx* ctx, const planar_frame_view* d_kf, const float* inv_level_sigma2 /* host */, float log_scale_factor, int n_levels,
                           const int32_t* d_n, int stride, int points_shared, const uint8_t* d_usable, const float* d_xw, const float* d_normal,
                           const float* d_min_dist, const float* d_max_dist, const uint8_t* d_desc, float th, int32_t* d_fuse_idx,
                           int32_t* d_fuse_dist, int32_t* d_n_fused);

/* cv::line_descriptor::KeyLine (opencv_contrib line_descriptor/descriptor.hpp), same field order, 68 bytes. */
typedef struct planar_keyline {
    float angle;
    int32_t class_id, octave;
    float pt_x, pt_y, response, size;
    float start_x, start_y, end_x, end_y;
    float s_oct_x, s_oct_y, e_oct_x, e_oct_y;
    float line_length;
    int32_t num_pixels;
} planar_keyline;

/* LSDmatcher::SearchByProjection(Frame& F, const vector<MapLine*>&, th) (src/LSDmatcher.cpp:141-211) with
 * Frame::GetLinesInArea (src/Frame.cc:491-524).
 *   keylines[b][i] mvKeylinesUn, ldesc mLdesc, blocked[i] = mvpMapLines[i] && ->Observations() > 0 on entry
 *   ml_in_view[j] = pML && !isBad() && mbTrackInView;  ml_proj[j] = {mTrackProjX1, Y1, X2, Y2};
 *   ml_level mnTrackScaleLevel, ml_view_cos mTrackViewCos, ml_desc GetDescriptor(), ml_observed Observations() > 0
 *   match[b][i] (in/out) index j of the map line stored in F.mvpMapLines[i] */
int planar_lsd_search_by_projection(planar_ctx* ctx, int B, const int32_t* n_lines, int line_stride, const planar_keyline* keylines,
                                    const uint8_t* ldesc, const uint8_t* blocked, const int32_t* n_ml, int ml_stride,
                                    const uint8_t* ml_in_view, const float* ml_proj, const int32_t* ml_level, const float* ml_view_cos,
                                    const uint8_t* ml_desc, const uint8_t* ml_observed, const float* scale_factors, int n_levels, float th,
                                    float nn_ratio, int32_t* match, int32_t* nmatches);
int planar_lsd_search_by_projection_dev(planar_ctx* ctx, int B, const int32_t* d_n_lines, int line_stride, const planar_keyline* d_keylines,
                                        const uint8_t* d_ldesc, const uint8_t* d_blocked, const int32_t* d_n_ml, int ml_stride,
                                        const uint8_t* d_ml_in_view, const float* d_ml_proj, const int32_t* d_ml_level,
                                        const float* d_ml_view_cos, const uint8_t* d_ml_desc, const uint8_t* d_ml_observed,
                                        const float* scale_factors /* host */, int n_levels, float th, float nn_ratio, int32_t* d_match,
                                        int32_t* d_nmatches);

/* LSDmatcher::Fuse(KeyFrame* pKF, const vector<MapLine*>& vpMapLines, th) (src/LSDmatcher.cpp:884-1015), the SEARCH half (:904-991): projection of both
 * end points, image-bounds / distance / viewing-angle gates, MapLine::PredictScale (src/MapLine.cpp:381-390, not clamped), KeyFrame::GetLinesInArea
 * (src/KeyFrame.cc:680-712), level gate, smallest Hamming distance, TH_LOW.  The map edits (:993-1010) stay with the caller, as for planar_fuse_search.
 *   kf: only B, Tcw, fx fy cx cy, min/max bounds and scale_factors are read (the key-point arrays may be NULL)
 *   keylines / ldesc = pKF->mvKeyLines / mLineDescriptors; usable[j] = vpMapLines[j] != NULL && !isBad(); xw6 / normal = GetWorldPos() / GetNormal()
 *   (doubles); min_dist / max_dist = mfMinDistance / mfMaxDistance.  A predicted level outside [0, n_levels) indexes mvScaleFactors out of
 *   bounds in the reference (undefined); such a line is skipped here.
 *   fuse_idx[b][j] = bestIdx where bestDist <= TH_LOW else -1; fuse_dist (may be NULL) = bestDist (INT_MAX: no candidate); rows j >= n_ml[b] read -1 / INT_MAX. */
int planar_lsd_fuse_search(planar_ctx* ctx, const planar_frame_view* kf, float log_scale_factor, int n_levels, const int32_t* n_lines, int line_stride,
                           const planar_keyline* keylines, const uint8_t* ldesc, const int32_t* n_ml, int ml_stride, int lines_shared,
                           const uint8_t* usable, const double* xw6, const double* normal, const float* min_dist, const float* max_dist,
                           const uint8_t* ml_desc, float th, int32_t* fuse_idx, int32_t* fuse_dist, int32_t* n_fused);
int planar_lsd_fuse_search_dev(planar_ctx* ctx, const planar_frame_view* d_kf, float log_scale_factor, int n_levels, const int32_t* d_n_lines, int line_stride,
                               const planar_keyline* d_keylines, const uint8_t* d_ldesc, const int32_t* d_n_ml, int ml_stride, int lines_shared,
                               const uint8_t* d_usable, const double* d_xw6, const double* d_normal, const float* d_min_dist, const float* d_max_dist,
                               const uint8_t* d_ml_desc, float th, int32_t* d_fuse_idx, int32_t* d_fuse_dist, int32_t* d_n_fused);

/* PlaneMatcher::SearchMapByCoefficients(Frame&, const vector<MapPlane*>&) (src/PlaneMatcher.cpp:10-66) with
 * Frame::ComputePlaneWorldCoeff (src/Frame.cc:815-820) and PointDistanceFromPlane (:67-79).
 *   pl_coef [B][pl_stride][4]  mvPlaneCoefficients[i] (camera frame)      Tcw [B][16]
 *   mp_*: the map planes seen by frame b (b = 0 for every frame when map_shared != 0):
 *     mp_valid !isBad(); mp_coef GetWorldPos() [4]; mp_pts mvPlanePoints xyz float, [.][mp_stride][pts_stride][3]
 *   th[4] = {dTh, aTh, verTh, parTh} (PlaneMatcher ctor, include/PlaneMatcher.h:19)
 *   match / ver / par [B][pl_stride] (in/out): index of the map plane stored in mvpMapPlanes[i] /
 *   mvpVerticalPlanes[i] / mvpParallelPlanes[i]; untouched where the reference does not assign. */
int planar_plane_search_by_coefficients(planar_ctx* ctx, int B, const int32_t* n_planes, int pl_stride, const float* pl_coef,
                                        const float* Tcw, int map_shared, const int32_t* n_mp, int mp_stride, const uint8_t* mp_valid,
                                        const float* mp_coef, const int32_t* mp_npts, int pts_stride, const float* mp_pts,
                                        const float* th, int32_t* match, int32_t* ver, int32_t* par, int32_t* nmatches);
int planar_plane_search_by_coefficients_dev(planar_ctx* ctx, int B, const int32_t* d_n_planes, int pl_stride, const float* d_pl_coef,
                                            const float* d_Tcw, int map_shared, const int32_t* d_n_mp, int mp_stride,
                                            const uint8_t* d_mp_valid, const float* d_mp_coef, const int32_t* d_mp_npts, int pts_stride,
                                            const float* d_mp_pts, const float* th /* host */, int32_t* d_match, int32_t* d_ver,
                                            int32_t* d_par, int32_t* d_nmatches);

/* ---- line extractor (replaces LineSegment::ExtractLineSegment, src/LSDextractor.cpp:12-39 /
 *      include/LSDextractor.h:344-352: cv::line_descriptor::LSDDetector::detect (1 octave, LSD_REFINE_ADV) +
 *      sort by response, keep max_lines + cv::line_descriptor::BinaryDescriptor::compute + line equations) --- */
typedef struct planar_lsd planar_lsd;
int planar_lsd_create(planar_ctx* ctx, int width, int height, int max_batch, planar_lsd** out);
void planar_lsd_destroy(planar_lsd* lsd);
int planar_lsd_max_segments(void);   /* raw LSD segments kept per frame before the top-`max_lines` cut (2048) */
int planar_lsd_scaled_size(planar_lsd* lsd, int* w, int* h);   /* the 0.8x working resolution */
int planar_lsd_set_tie_order(planar_lsd* lsd, int tie_order);   /* 0 (default): libstdc++ std::sort order, 1: raster order */
/* 1: the NFA stage (rect_improve) only for the regions that can end among the max_lines key lines LineSegment::ExtractLineSegment keeps (reference src/LSDextractor.cpp:17-27:
 * sort by response, resize to 40): the longest regions first, the rest only where that cannot settle the frame (fewer than max_lines + 1 accepted, a shorter region could still
 * reach the kept ones, or equal responses among them).  Key lines, descriptors and equations are those of the default mode, bit for bit; what planar_lsd_read_stage(.., 3, ..) returns
 * (the raw segments) is then only the part that was evaluated.  0 (default): every region. */
int planar_lsd_set_top_only(planar_lsd* lsd, int enable);
/* gray     : B frames of 8-bit gray (the `img` argument), pitch / frame_stride in bytes
 * max_lines: lsdNFeatures (40 in the reference); per-frame stride of the outputs
 * keylines : [B][max_lines] cv::line_descriptor::KeyLine records   ldesc: [B][max_lines][32] LBD bytes
 * line_eq  : [B][max_lines][3] keylineFunctions (sp x ep, normalised)   n_lines: [B] keylines.size()
 * Pixels of equal gradient bin are visited in the order libstdc++'s std::sort leaves them, as in the reference library
 * (lsd.cpp sorts with std::sort); planar_lsd_set_tie_order(lsd, 1) selects raster order inside a bin instead (the original
 * LSD's list order). */
int planar_lsd_extract(planar_lsd* lsd, const uint8_t* gray, int B, int pitch, int64_t frame_stride, int max_lines, planar_keyline* keylines,
                       uint8_t* ldesc, double* line_eq, int32_t* n_lines);
int planar_lsd_extract_dev(planar_lsd* lsd, const uint8_t* d_gray, int B, int pitch, int64_t frame_stride, int max_lines,
                           planar_keyline* d_keylines, uint8_t* d_ldesc, double* d_line_eq, int32_t* d_n_lines);
/* The same work as planar_lsd_extract_dev in two enqueues on the context's stream, for callers that overlap the sequential
 * detector with other stages: preprocess = Gaussians, 0.8x resample + gradients, Sobel, pixel ordering (throughput kernels);
 * detect = region growing / NFA, KeyLines, LBD.  planar_lsd_detect_dev returns PLANAR_ESTATE without a matching preprocess. */
int planar_lsd_preprocess_dev(planar_lsd* lsd, const uint8_t* d_gray, int B, int pitch, int64_t frame_stride);
int planar_lsd_detect_dev(planar_lsd* lsd, int B, int max_lines, planar_keyline* d_keylines, uint8_t* d_ldesc, double* d_line_eq, int32_t* d_n_lines);
/* Per-frame status of the last detect / extract call (synchronises the context's stream).  n_lines is never negative: a frame whose workspace overflowed (more
 * regions / segments than it holds) or whose std::sort order could not be reproduced delivers ZERO lines, and this call (which planar_lsd_extract makes itself)
 * returns PLANAR_ECAPACITY naming the frame; planar_last_error() has the reason.  The device entry points never synchronise: their callers check when they read back. */
int planar_lsd_check(planar_lsd* lsd, int B);
/* diagnostics (tests / profiling; stage 5 = cycle counters, see lsd.hip): stage 0 level-line angle float deg [w*h] (-1024 undefined), 1 squared gradient u32 [w*h],
 * 2 visiting order int32 (returns n), 3 raw segments 40 B each {x1,y1,x2,y2 float; width,p,nfa double} (returns count),
 * 4 number of grown regions int32[1] */
int planar_lsd_read_stage(planar_lsd* lsd, int frame, int stage, void* out, int64_t out_bytes);
/* Per-launch timing with HIP events on the context stream (bench.py's roofline leg), as planar_peac_set_profiling: get_profile synchronises and returns the
 * summed milliseconds of the recorded calls, total_ms[4] = preprocessing (blurs, gradient, Sobel), lsd_sort, lsd_detect, the rest, their number, and resets. */
int planar_lsd_set_profiling(planar_lsd* lsd, int enable);
int planar_lsd_get_profile(planar_lsd* lsd, double* total_ms, int64_t* calls);
#ifdef PLANAR_TEST_HOOKS
/* test hook, exported by the TEST build only (libplanar_hip_paranoid.so, `make paranoid`): device emulation of libstdc++ std::sort with the sort_lines_by_response
 * comparator (include/auxiliar.h:43-48) on raw keys */
int planar_debug_std_sort_desc(planar_ctx* ctx, float* keys, int32_t* perm, int n);
#endif

/* ---- plane extractor (replaces PlaneDetection::readDepthImage + runPlaneDetection,
 *      src/PlaneExtractor.cpp:26-65 / include/PlaneExtractor.h:36-56, i.e. ahc::PlaneFitter::run with
 *      PlanarSLAM's defaults, include/peac/AHCPlaneFitter.hpp:154-158,211) ------------------------- */
typedef struct planar_peac planar_peac;
int planar_peac_create(planar_ctx* ctx, int width, int height, int max_batch, planar_peac** out);
void planar_peac_destroy(planar_peac* peac);
int planar_peac_max_planes(void);   /* per-frame stride of `planes` (128) */
/* depth  : B frames of 16-bit depth (cv::Mat CV_16U as passed to readDepthImage), pitch / frame stride in PIXELS
 * fx..cy : K.at<float>(0,0), (1,1), (0,2), (1,2);   depth_factor : kScaleFactor (mDepthMapFactor, 1/5000)
 * labels : [B][H*W] int32, plane id of every pixel in the order of plane_vertices_ (= extractedPlanes), -1 = none;
 *          plane_vertices_[i] is the raster-ordered list of pixels with label i (src/Frame.cc:652-656)
 * planes : [B][max_planes][8] doubles: N, normal[3], center[3], mse of extractedPlanes[i] (src/Frame.cc:663-669)
 * n_planes : [B] plane_num_                                                                             */
int planar_peac_segment(planar_peac* peac, const uint16_t* depth, int B, int pitch_px, int64_t frame_stride_px, float fx, float fy,
                        float cx, float cy, float depth_factor, int32_t* labels, double* planes, int32_t* n_planes);
int planar_peac_segment_dev(planar_peac* peac, const uint16_t* d_depth, int B, int pitch_px, int64_t frame_stride_px, float fx, float fy,
                            float cx, float cy, float depth_factor, int32_t* d_labels, double* d_planes, int32_t* d_n_planes);
/* Synchronises and returns PLANAR_ECAPACITY if any of the last B frames overflowed an internal capacity
 * (more than max_planes planes, flood-fill queue, neighbour pool); the host-pointer entry point calls it itself. */
/* Debug aids (tools/peac_ab.py; not part of the drop-in surface): workspace layout {frame_bytes, NB, NB2, off_stats, off_geo, off_N, off_dsp, off_dss,
 * off_rid, off_nouse, off_hand, off_crec} and a raw read of one frame's workspace after the last call. */
int planar_peac_debug_layout(planar_peac* peac, int64_t* out /* [12] */);
int planar_peac_debug_read(planar_peac* peac, int frame, int64_t offset, int64_t bytes, void* out);
int planar_peac_check(planar_peac* peac, int B);
/* A/B aid (tools/peac_ab.py and the tests; nothing in the product calls it, and no environment variable selects a kernel): clustering 0 = product (fast attempt +
 * exact redo), 1 = exact heap only (anything else: PLANAR_EINVAL); wide_below = batch size up to which the refinement runs 1024 threads per frame (< 0: keep).
 * Every variant returns the same labels and planes. */
int planar_peac_set_variant(planar_peac* peac, int clustering, int wide_below);
/* Per-launch timing with HIP events on the context stream (bench.py's roofline leg), as planar_orb_set_profiling: get_profile synchronises and returns the
 * summed milliseconds of the four launches of the recorded calls (total_ms[4] = peac_blocks, peac_ahc, peac_order, peac_refine), their number, and resets. */
int planar_peac_set_profiling(planar_peac* peac, int enable);
int planar_peac_get_profile(planar_peac* peac, double* total_ms, int64_t* calls);
/* Profiling aid: per-frame record of the last call, out[B][48] (100 MHz ticks since kernel entry:
 * [1] graph edges, [2] heap built, [3] ahCluster, [4] seeds, [5] floodFill, [6] end; [7] evaluation phases << 40 | nodes evaluated << 20 | valid-record pops;
 * [8] flood-fill queue entries, [9] nodes, [10] pops of nodes whose bag lives in the pool; [16..39] shader-cycle buckets of the clustering kernel, zero
 * unless the library was built with -DPLANAR_PEAC_TIMING: tools/peac_ab.py names them). */
int planar_peac_read_timing(planar_peac* peac, int B, int64_t* out);

/* ---- local bundle adjustment (replaces the numerical core of Optimizer::LocalBundleAdjustment,
 *      src/Optimizer.cc:1853-2680 / include/Optimizer.h:35: optimize(5) -> outlier levels -> optimize(10) -> erase lists) ---- */
/* RCCL communicator for the one exchange step on the path: the all-reduce of the reduced camera system when landmarks
 * are partitioned over GPUs.  Rank 0 creates an id, every rank gets it out of band (e.g. a
 * torch.distributed broadcast), then all ranks call planar_comm_create. */
typedef struct planar_comm planar_comm;
typedef struct planar_comm_id { char internal[128]; } planar_comm_id;   /* == ncclUniqueId */
int planar_comm_unique_id(planar_comm_id* out);
int planar_comm_create(planar_ctx* ctx, const planar_comm_id* id, int nranks, int rank, planar_comm** out);
/* The same exchange over a transport the embedding program owns (MPI, gloo, shared memory ...): the library stages the buffer
 * through host memory and calls allreduce(user, buf, n, op) (op 0 = sum, 1 = max; in place; 0 = success) on every rank.
 * It is how several PROCESSES that share ONE GPU (which RCCL refuses) run the partitioned solve - tests/test_ba_gpu.py does that
 * with torch.distributed/gloo on the 1-GPU box - and a way to use the library where RCCL is not wanted. */
typedef int (*planar_allreduce_fn)(void* user, double* buf, size_t n, int op);
int planar_comm_create_hosted(planar_ctx* ctx, planar_allreduce_fn allreduce, void* user, int nranks, int rank, planar_comm** out);
void planar_comm_destroy(planar_comm* comm);

#define PLANAR_BA_MONO 0      /* g2o::EdgeSE3ProjectXYZ        (types_six_dof_expmap.cpp:103-139)  meas = u, v            */
#define PLANAR_BA_STEREO 1    /* g2o::EdgeStereoSE3ProjectXYZ  (:188-235)                          meas = u, v, ur        */
#define PLANAR_BA_LINE 2      /* EdgeLineProjectXYZ            (include/EdgeLine.h:53-153)         meas = line a, b, c    */
#define PLANAR_BA_PLANE 3     /* g2o::EdgePlane                (g2oAddition/EdgePlane.h:25-126)    meas = plane coeffs    */
#define PLANAR_BA_VERTICAL 4  /* g2o::EdgeVerticalPlane        (g2oAddition/EdgeVerticalPlane.h:21) meas = plane coeffs   */
#define PLANAR_BA_PARALLEL 5  /* g2o::EdgeParallelPlane        (g2oAddition/EdgeParallelPlane.h:21) meas = plane coeffs   */

/* The graph the reference assembles at src/Optimizer.cc:1985-2350, as arrays (all HOST pointers).  Keyframes in
 * ascending mnId; landmarks are 3-dof vertices: MapPoints and the two endpoints of every MapLine (type 0,
 * VertexSBAPointXYZ) and MapPlanes (type 1, VertexPlane).  The two edges of a line observation (start, end) must be
 * consecutive.  With a communicator every rank passes ITS landmarks and their edges (keyframes replicated). */
typedef struct planar_ba_problem {
    int32_t n_kf;
    const float* kf_Tcw;          /* [n_kf][16]  KeyFrame::GetPose(), row-major float32                              */
    const uint8_t* kf_fixed;      /* [n_kf]      setFixed(): mnId == 0 or a fixed camera (:1994, :2007)               */
    int32_t n_lm;
    const uint8_t* lm_type;       /* [n_lm]      0 = XYZ vertex, 1 = plane vertex                                    */
    const double* lm_init;        /* [n_lm][4]   x, y, z, 0  |  plane coefficients (GetWorldPos())                    */
    int32_t n_edges;
    const int32_t* e_kf;          /* [n_edges]   keyframe index (vertex 1)                                           */
    const int32_t* e_lm;          /* [n_edges]   landmark index (vertex 0)                                           */
    const uint8_t* e_type;        /* [n_edges]   PLANAR_BA_*                                                         */
    const double* e_meas;         /* [n_edges][4]                                                                    */
    const float* e_inv_sigma2;    /* [n_edges]   mvInvLevelSigma2[octave] for point edges (ignored otherwise)        */
} planar_ba_problem;

typedef struct planar_ba_result {
    float* kf_Tcw;                /* [n_kf][16]  optimised poses (what the reference passes to KeyFrame::SetPose)    */
    double* lm;                   /* [n_lm][4]   optimised landmarks                                                 */
    uint8_t* e_outlier;           /* [n_edges]   1 = the observation lands in the reference's erase lists (:2471-2575) */
    int32_t lm_iterations;        /* LM iterations run                                                               */
    int32_t stopped;              /* 1 if *stop_flag was seen (pbStopFlag, :1982-1983)                               */
} planar_ba_result;

/* its1 = 5, its2 = 10 reproduce the reference.  `params` supplies fx, fy, cx, cy, bf and the Plane.* configuration.
 * stop_flag (or NULL) is the reference's `bool* pbStopFlag` (one byte; the host samples it when LM steps are enqueued - with a flag at most two steps per chunk, so
 * a flag raised while the GPU is solving is seen at most two LM steps later - and once more between optimize(its1) and optimize(its2), as the reference's bDoMore
 * does; with several ranks every decision is taken on the all-reduced value).  comm may be NULL (single GPU).  Synchronous.
 * Non-fixed key frames: any number up to 128 (PLANAR_ECAPACITY above); up to 20 the reduced camera system stays in LDS, above that it is accumulated and
 * factorised in global memory (slower per LM step, same results): Optimizer::LocalBundleAdjustment has no cap on the covisible key frames. */
int planar_local_ba(planar_ctx* ctx, const planar_ba_problem* problem, const planar_pose_params* params, int its1, int its2,
                    planar_ba_result* result, const volatile unsigned char* stop_flag, planar_comm* comm);

/* ---- DBoW2 vocabulary transform (replaces ORBVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) as Frame::ComputeBoW and
 *      KeyFrame::ComputeBoW call it with levelsup = 4; Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1124-1180, :1203-1250; include/ORBVocabulary.h:31) ----
 * The vocabulary arrives as the rows of the text format TemplatedVocabulary::loadFromTextFile reads (ORBvoc.txt): for node 1..n in file order its
 * parent id, isLeaf flag, 32 descriptor bytes and weight; header values k and L.  TF_IDF weighting with L1 normalisation (header "k L 0 0").
 *   word / weight [B][stride]: WordId and idf weight of every feature (the per-feature transform);  node [B][stride]: the id of its ancestor at level
 *   L - levelsup (0 = root when levelsup >= L), -1 for a stopped word (weight 0), i.e. the FeatureVector as one node id per feature - the form
 *   planar_search_by_bow takes;  bow_word / bow_value [B][stride] + bow_n [B]: the BowVector, ascending word id, L1-normalised. */
typedef struct planar_vocab planar_vocab;
int planar_vocab_create(planar_ctx* ctx, int k, int L, int n_nodes, const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc /* [n][32] */,
                        const double* weight, planar_vocab** out);
void planar_vocab_destroy(planar_vocab* voc);
int planar_vocab_words(const planar_vocab* voc);
int planar_bow_transform(planar_vocab* voc, const uint8_t* desc, const int32_t* n, int B, int stride, int levelsup, int32_t* word, double* weight,
                         int32_t* node, int32_t* bow_word, double* bow_value, int32_t* bow_n);
/* the three BowVector outputs may be NULL together (FeatureVector only) */
int planar_bow_transform_dev(planar_vocab* voc, const uint8_t* d_desc, const int32_t* d_n, int B, int stride, int levelsup, int32_t* d_word,
                             double* d_weight, int32_t* d_node, int32_t* d_bow_word, double* d_bow_value, int32_t* d_bow_n);

/* ---- 3-D line back-projection (replaces Frame::isLineGood, src/Frame.cc:189-267, with compPt3dCov / extract3dline_mahdist / verify3dLine /
 *      mah_dist3d_pt_line / computeLine3d_svd of src/LineExtractor.cpp:1157-1470) ----
 * keylines [B][ln_stride] = Frame::mvKeylinesUn, depth = the u16 depth image (imDepth = value * depth_factor), fx.. = Frame::fx.. (K(0,0) of compPt3dCov is fx).
 * seeds [B]: line i of frame b draws from the glibc rand() stream of srand(seeds[b] + i) - the reference uses the process-global rand() state, which
 * is not reproducible; with this seeding a CPU run of the reference that calls srand(seed + i) before line i gives the same draws.
 *   depth_line [B][ln_stride] float   mvDepthLine (-1: no reliable 3-D line)          lines3d [B][ln_stride][6]  mvLines3D (A xyz, B xyz; zeros if none)
 *   good [B][ln_stride] u8            the line was pushed to mVF3DLines                 direction [B][ln_stride][3] FrameLine::direction
 *   n_inliers [B][ln_stride]          tmpLine.pts.size()
 *   packed_dirs [B][ln_stride][3] (or NULL in the _dev form) + n_good [B]: the directions of the good lines in line order = mVF3DLines, the
 *   line_dirs / n_lines arguments of planar_track_manhattan_frame. */
int planar_is_line_good(planar_ctx* ctx, int B, const planar_keyline* keylines, const int32_t* n_lines, int ln_stride, const uint16_t* depth, int width,
                        int height, int pitch_px, int64_t frame_stride_px, float depth_factor, float fx, float fy, float cx, float cy, const uint32_t* seeds,
                        float* depth_line, double* lines3d, uint8_t* good, double* direction, int32_t* n_inliers, double* packed_dirs, int32_t* n_good);
int planar_is_line_good_dev(planar_ctx* ctx, int B, const planar_keyline* d_keylines, const int32_t* d_n_lines, int ln_stride, const uint16_t* d_depth,
                            int width, int height, int pitch_px, int64_t frame_stride_px, float depth_factor, float fx, float fy, float cx, float cy,
                            const uint32_t* d_seeds, float* d_depth_line, double* d_lines3d, uint8_t* d_good, double* d_direction, int32_t* d_n_inliers,
                            double* d_packed_dirs, int32_t* d_n_good);

/* ---- surface normals (replaces the tail of Frame::ComputePlanes, src/Frame.cc:694-751: the depth image sampled every 3rd pixel ->
 *      pcl::IntegralImageNormalEstimation(AVERAGE_3D_GRADIENT, MaxDepthChangeFactor 0.05, NormalSmoothingSize 10) -> vSurfaceNormal) ----
 * Output per frame: planar_normals_count() entries in the reference's push_back order (odd rows m, odd columns n of the
 * ceil(W/3) x ceil(H/3) grid, row-major; 8 560 for 640x480): normals[b][i] = SurfaceNormal::normal (NaN where PCL yields none),
 * points[b][i] = SurfaceNormal::cameraPosition (or NULL); FramePosition is (3n, 3m).  The normals array is what
 * planar_track_manhattan_frame takes (n_normals = count, sn_stride = out_stride). */
typedef struct planar_normals planar_normals;
int planar_normals_create(planar_ctx* ctx, int width, int height, int max_batch, planar_normals** out);
void planar_normals_destroy(planar_normals* nrm);
int planar_normals_count(const planar_normals* nrm);
int planar_normals_grid(const planar_normals* nrm, int* grid_w, int* grid_h, int* out_w, int* out_h);
int planar_normals_compute(planar_normals* nrm, const uint16_t* depth, int B, int pitch_px, int64_t frame_stride_px, float fx, float fy, float cx,
                           float cy, float depth_factor, float* normals /* [B][count][3] */, float* points /* [B][count][3] or NULL */);
int planar_normals_compute_dev(planar_normals* nrm, const uint16_t* d_depth, int B, int pitch_px, int64_t frame_stride_px, float fx, float fy,
                               float cx, float cy, float depth_factor, float* d_normals, float* d_points, int out_stride);

/* ---- Manhattan-frame tracking (replaces Tracking::TrackManhattanFrame, src/Tracking.cc:963-1138, with its helpers
 *      ProjectSN2Conic :886-953, ProjectSN2MF :757-884 and MeanShift :1140-1157; called once per frame by Tracking::Track, :248) ----
 * R_last   : [B][9]   row-major 3x3 float, mLastRcm (camera <- Manhattan frame)
 * normals  : [B][sn_stride][3] float, Frame::vSurfaceNormal[i].normal; n_normals[b] of them are valid
 * line_dirs: [B][ln_stride][3] double, Frame::mVF3DLines[i].direction; n_lines[b] valid
 * R_out    : [B][9]   the returned rotation (U * Vt of the updated axes; the input, possibly with one column replaced, if fewer
 *                     than two directions were found - the reference returns that too)
 * member   : [B][sn_stride + ln_stride] or NULL: bit a-1 set iff the element was pushed to Frame::vSurfaceNormalx/y/z resp.
 *            vVanishingLinex/y/z by ProjectSN2MF for axis a (surface normals first, vanishing directions from offset sn_stride)
 * info     : [B][8] or NULL: numDirectionFound, found mask (bit a-1), numInCone[3], points given to MeanShift per axis [3]
 * density  : [B][3] or NULL: s_j_density of the axes that were found                                                            */
int planar_track_manhattan_frame(planar_ctx* ctx, int B, const float* R_last, const float* normals, const int32_t* n_normals, int sn_stride,
                                 const double* line_dirs, const int32_t* n_lines, int ln_stride, float* R_out, uint8_t* member, int32_t* info,
                                 float* density);
int planar_track_manhattan_frame_dev(planar_ctx* ctx, int B, const float* d_R_last, const float* d_normals, const int32_t* d_n_normals, int sn_stride,
                                     const double* d_line_dirs, const int32_t* d_n_lines, int ln_stride, float* d_R_out, uint8_t* d_member,
                                     int32_t* d_info, float* d_density);

/* ---- plane post-processing (planepost.hip; replaces the head of Frame::ComputePlanes, src/Frame.cc:652-692, with
 *      Frame::MaxPointDistanceFromPlane, src/Frame.cc:755-812) -----------------------------------------------------------------------
 * For every plane PlaneDetection extracted (labels / planes / n_planes exactly as planar_peac_segment delivers them): the member pixels' camera points
 * as float -> pcl::VoxelGrid(leaf) centroids -> coefficient (n, -n.c) in float -> a plane with a centroid farther than dist_th
 * (Plane.DistanceThreshold) is dropped -> pcl::SACSegmentation (plane, RANSAC, optimised coefficients, threshold dist_th) refits the coefficient.
 *   n_out  [B]                     Frame::mnPlaneNum: planes kept
 *   coef   [B][pl_stride][4]       mvPlaneCoefficients (pl_stride = planar_peac_max_planes())
 *   src    [B][pl_stride]          index of the detector plane each kept plane came from
 *   pt_off [B][pl_stride + 1]      mvPlanePoints[k] = points[b][pt_off[k] .. pt_off[k + 1])
 *   points [B][max_points][3]      the kept planes' voxel centroids, concatenated, each plane in PCL's output order
 *   status [B] (_dev only)         0, or 3 = more than max_points voxels / voxel coordinate out of range, 4 = sampler table exhausted: the frame's
 *                                  outputs are then empty; the host-pointer entry point turns it into PLANAR_ECAPACITY
 *   state / nvox / info (optional, per DETECTOR plane, stride pl_stride / pl_stride / pl_stride * 12): 0 kept, 1 distance, 2 no inliers; voxels;
 *                                  {RANSAC iterations, best count, best sample[3], inliers, inliers after the refit, sampler draws, model[4] bits}  */
typedef struct planar_plane_clouds planar_plane_clouds;
int planar_plane_clouds_create(planar_ctx* ctx, int width, int height, int max_batch, int max_points /* voxels per frame: a power of two <= 8192; the kernel holds 36 KB of LDS up to 4096, 64 KB at 8192; the key table is in the workspace */, planar_plane_clouds** out);
void planar_plane_clouds_destroy(planar_plane_clouds* pc);
int planar_plane_clouds_stride(const planar_plane_clouds* pc, int* pl_stride, int* max_points);
/* Plane window: the compute calls that follow only process the detector planes [first, first + count) of every frame (count < 0: all planes again).  pcl::VoxelGrid
 * has no voxel cap (reference src/Frame.cc:674-679); a frame's voxel table here holds max_points (<= 8192) voxels for all its planes together, and a compute call
 * reports PLANAR_ECAPACITY (frame status 3) beyond that.  Such a frame goes through once more plane by plane (window (i, 1) for every plane i, results appended in
 * plane order: exactly the sequence of Frame::ComputePlanes' loop) - include/planar_adapters.hpp and planarslam_amd.planes.PlaneClouds.compute do so; only a SINGLE
 * plane of more than max_points voxels (> 80 m^2 of surface at the 0.1 m leaf) is beyond the structure and is dropped with a message. */
int planar_plane_clouds_set_plane_window(planar_plane_clouds* pc, int first, int count);
/* HIP-event timing of the launches of planar_plane_clouds_compute_dev (bench.py's roofline leg), as planar_peac_set_profiling: total_ms [6] = plane_voxels,
 * plane_items, plane_sort_global, plane_sort_lds, plane_sort_heap, plane_tail over `calls` recorded calls */
int planar_plane_clouds_set_profiling(planar_plane_clouds* pc, int enable);
int planar_plane_clouds_get_profile(planar_plane_clouds* pc, double* total_ms, int64_t* calls);
/* diagnostics (synchronises): per frame of the last call, out [B][4] = {ranges that went through libstdc++'s heap-sort fallback (std::sort of
 * pcl::VoxelGrid::applyFilter on a plane whose introsort depth budget ran out), their elements, the longest, LDS-tier sort blocks} */
int planar_plane_clouds_sort_stats(planar_plane_clouds* pc, int B, int64_t* out);
/* Profiling aid, as planar_peac_read_timing: per-frame phase timestamps of the last call, out[B][16] (100 MHz ticks: [0] entry, [1] table cleared, [2] voxel sums,
 * [3] sorted + centroids, [4] refit, [5] end; [6] voxels, [7] planes; [8..11] shader-clock cycles of wavefront 0 in the voxel-sum phase: row loops, tile ends,
 * executions of the parked-run insertion and the cycles in it). */
int planar_plane_clouds_set_timing(planar_plane_clouds* pc, int enable);
int planar_plane_clouds_read_timing(planar_plane_clouds* pc, int B, int64_t* out);
int planar_plane_clouds_compute(planar_plane_clouds* pc, const uint16_t* depth, int B, int pitch_px, int64_t frame_stride_px, float fx, float fy, float cx,
                                float cy, float depth_factor, const int32_t* labels, const double* planes, const int32_t* n_planes, double dist_th, float leaf,
                                int32_t* n_out, float* coef, int32_t* src, int32_t* pt_off, float* points, int32_t* state, int32_t* nvox, int32_t* info);
int planar_plane_clouds_compute_dev(planar_plane_clouds* pc, const uint16_t* d_depth, int B, int pitch_px, int64_t frame_stride_px, float fx, float fy,
                                    float cx, float cy, float depth_factor, const int32_t* d_labels, const double* d_planes, const int32_t* d_n_planes,
                                    double dist_th, float leaf, int32_t* d_n_out, float* d_coef, int32_t* d_src, int32_t* d_pt_off, float* d_points,
                                    int32_t* d_status, int32_t* d_state, int32_t* d_nvox, int32_t* d_info);
/* Per-frame result codes of the last planar_plane_clouds_compute call on this handle (host-pointer entry point; the _dev entry point delivers them in d_status): 0 ok,
 * 3 = the frame's planes together hold more voxels than max_points (or a voxel index out of range) - pcl::VoxelGrid has no cap (src/Frame.cc:674-679), so callers redo
 * exactly those frames plane by plane (planar_plane_clouds_set_plane_window) -, 4 = sampler table exhausted, 5 = std::sort order not reproducible.  out [B]. */
int planar_plane_clouds_last_status(planar_plane_clouds* pc, int B, int32_t* out);
/* Frame::MaxPointDistanceFromPlane on given clouds (host pointers): planes [n_clouds][4] in/out (written when state == 0), cloud q = points[pt_off[q] ..
 * pt_off[q + 1]); state / info as above. */
int planar_plane_refit(planar_plane_clouds* pc, int n_clouds, const float* points, const int32_t* pt_off, double dist_th, float* planes, int32_t* state,
                       int32_t* info);
/* Map::FlagMatchedPlanePoints (src/Map.cc:366-393): flags[b][j] = 1 for every map point j within 0.5 of the WORLD coefficient (Frame::ComputePlaneWorldCoeff)
 * of a plane i < n_planes[b] with matched[b][i] != 0 (mvpMapPlanes[i] non-null).  xw: [B or 1][n_points][3]; flags are only ever set; n_matches (or NULL): nMatches. */
int planar_flag_matched_plane_points(planar_ctx* ctx, int B, const float* Tcw, const float* coef, const uint8_t* matched, const int32_t* n_planes, int pl_stride,
                                     const float* xw, int n_points, int points_shared, uint8_t* flags, int32_t* n_matches);
int planar_flag_matched_plane_points_dev(planar_ctx* ctx, int B, const float* d_Tcw, const float* d_coef, const uint8_t* d_matched, const int32_t* d_n_planes,
                                         int pl_stride, const float* d_xw, int n_points, int points_shared, uint8_t* d_flags, int32_t* d_n_matches);
/* The cloud half of MapPlane::UpdateCoefficientsAndPoints (src/MapPlane.cc:335-352): the frame plane's points through Twc (row-major 4x4 double, what
 * Converter::toSE3Quat(mTcw).inverse() yields; pcl::transformPointCloud: double arithmetic, float store), the map plane's points appended, VoxelGrid(leaf). */
int planar_merge_plane_points(planar_plane_clouds* pc, const double* Twc, const float* frame_points, int n_frame, const float* map_points, int n_map, float leaf,
                              float* out_points, int out_cap, int32_t* n_out);

/* ---- Frame-side glue of Tracking::Track (frame.hip) ---------------------------------------------------------------------------
 * Frame::ComputeStereoFromRGBD (src/Frame.cc:603-621) + Frame::UnprojectStereo (:623-634) for every keypoint of B frames.
 *   keys / keys_un : [B][stride] mvKeys (the depth image is read at the DISTORTED position, truncated to int) / mvKeysUn
 *   depth          : B frames of 16-bit depth, row pitch `pitch_px`, frame stride `frame_stride_px` (pixels); the float image the
 *                    reference indexes is depth * depth_factor in float (imDepth.convertTo(CV_32F, mDepthMapFactor), src/Tracking.cc:174)
 *   Tcw            : [B][16] the pose UnprojectStereo uses (mRwc, mOw are derived as Frame::UpdatePoseMatrices does)
 *   u_right, depth_out : [B][stride] mvuRight / mvDepth (-1 where the depth is not positive)
 *   xw             : [B][stride][3] UnprojectStereo(i) (zeros where it returns an empty Mat), valid[B][stride] = depth > 0          */
int planar_stereo_from_rgbd(planar_ctx* ctx, int B, const planar_keypoint* keys, const planar_keypoint* keys_un, const int32_t* n, int stride,
                            const uint16_t* depth, int pitch_px, int64_t frame_stride_px, float depth_factor, float fx, float fy, float cx, float cy,
                            float bf, const float* Tcw, float* u_right, float* depth_out, float* xw, uint8_t* valid);
int planar_stereo_from_rgbd_dev(planar_ctx* ctx, int B, const planar_keypoint* d_keys, const planar_keypoint* d_keys_un, const int32_t* d_n, int stride,
                                const uint16_t* d_depth, int pitch_px, int64_t frame_stride_px, float depth_factor, float fx, float fy, float cx,
                                float cy, float bf, const float* d_Tcw, float* d_u_right, float* d_depth_out, float* d_xw, uint8_t* d_valid);

/* MapPoint::UpdateNormalAndDepth (src/MapPoint.cc:347-388) for G groups of map points; group g = the points whose reference key frame (mpRefKF) has the pose
 * ref_Tcw[g] - e.g. the back-projected keypoints of one frame.  Its camera centre is formed as KeyFrame::SetPose does (src/KeyFrame.cc:85-86).
 *   n [G], xw [G][stride][3] (GetWorldPos), valid [G][stride] or NULL (0 = no point: its outputs are left alone)
 *   keys_un [G][stride]: the point's keypoint in the reference key frame (its octave selects mvScaleFactors[level])
 *   obs_off [G * stride + 1], obs_ow [obs_off[G * stride]][3]: camera centres of the key frames observing each point, in observation-map order; both NULL:
 *            every point is observed by its reference key frame only
 *   normal [G][stride][3] (mNormalVector), min_dist / max_dist [G][stride] (mfMinDistance / mfMaxDistance); a point without observations keeps its values */
int planar_update_normal_and_depth(planar_ctx* ctx, int G, const int32_t* n, int stride, const float* xw, const uint8_t* valid, const float* ref_Tcw,
                                   const planar_keypoint* keys_un, const int32_t* obs_off, const float* obs_ow, const float* scale_factors, int n_levels,
                                   float* normal, float* min_dist, float* max_dist);
int planar_update_normal_and_depth_dev(planar_ctx* ctx, int G, const int32_t* d_n, int stride, const float* d_xw, const uint8_t* d_valid, const float* d_ref_Tcw,
                                       const planar_keypoint* d_keys_un, const int32_t* d_obs_off, const float* d_obs_ow, const float* scale_factors,
                                       int n_levels, float* d_normal, float* d_min_dist, float* d_max_dist);

/* What Optimizer::PoseOptimization / TranslationOptimization read from the Frame once the matchers have filled mvpMapPoints / mvpMapLines /
 * mvpMapPlanes / mvpParallelPlanes / mvpVerticalPlanes (src/Optimizer.cc:593-668, 689-745, 859-981), as match INDICES into the arrays of
 * the matched-against objects.  -1 = no association.                                                                                 */
typedef struct planar_track_matches {
    int32_t B;
    /* points */
    int32_t stride, mp_stride, n_levels;
    const int32_t* n;                /* [B]                 Frame::N                                                              */
    const planar_keypoint* keys_un;  /* [B][stride]         mvKeysUn                                                              */
    const float* u_right;            /* [B][stride]         mvuRight                                                              */
    const int32_t* pt_match;         /* [B][stride]         index of the matched map point (what the matchers wrote), -1 = NULL   */
    const float* mp_xw;              /* [B][mp_stride][3]   GetWorldPos() of the candidate map points                             */
    const uint8_t* mp_valid;         /* [B][mp_stride] or NULL: the map point exists (e.g. planar_stereo_from_rgbd's `valid`)      */
    float inv_level_sigma2[PLANAR_MAX_LEVELS]; /* mvInvLevelSigma2                                                                 */
    /* lines (n_lines == NULL: no lines) */
    int32_t ln_stride, ml_stride;
    const int32_t* n_lines;          /* [B]                 Frame::NL                                                             */
    const double* line_eq;           /* [B][ln_stride][3]   mvKeyLineFunctions                                                    */
    const int32_t* ln_match;         /* [B][ln_stride]      index of the matched map line                                         */
    const double* ml_xw6;            /* [B][ml_stride][6]   MapLine::mWorldPos                                                    */
    /* planes (n_planes == NULL: no planes) */
    int32_t pl_stride, mpl_stride, mpl_shared;   /* mpl_shared != 0: one map-plane array for all frames                            */
    const int32_t* n_planes;         /* [B]                 Frame::mnPlaneNum                                                     */
    const float* pl_coef;            /* [B][pl_stride][4]   mvPlaneCoefficients                                                   */
    const int32_t* pl_match;         /* [3][B][pl_stride]   matched / parallel / vertical map plane (PlaneMatcher's three outputs) */
    const float* mpl_coef;           /* [B or 1][mpl_stride][4] MapPlane::GetWorldPos()                                           */
    const float* Tcw;                /* [B][16]             Frame::mTcw the optimiser starts from                                 */
} planar_track_matches;
/* Fills the INPUT arrays of `out` (n_points .. Tcw_in; they are written although the struct declares them const) for planar_pose_opt.
 * Frame b gets min(n[b], out->max_points) points, likewise lines / planes.  Host pointers / device pointers as usual.              */
int planar_pose_assemble(planar_ctx* ctx, const planar_track_matches* matches, const planar_pose_batch* out);
int planar_pose_assemble_dev(planar_ctx* ctx, const planar_track_matches* d_matches, const planar_pose_batch* d_out);

/* The loop that follows the optimiser in Tracking::TranslationWithMotionModel / TrackLocalMap (src/Tracking.cc:1784-1812): a match whose
 * outlier flag is set is dropped (match = -1, flag cleared); kept[b] (or NULL) = matches left.  outlier has row stride flag_stride.  */
int planar_discard_outliers(planar_ctx* ctx, int B, const int32_t* n, int stride, int flag_stride, int32_t* match, uint8_t* outlier, int32_t* kept);
int planar_discard_outliers_dev(planar_ctx* ctx, int B, const int32_t* d_n, int stride, int flag_stride, int32_t* d_match, uint8_t* d_outlier,
                                int32_t* d_kept);

/* ---- element-wise glue between the stages of Tracking::Track (src/Tracking.cc:240-253, 1739-1790, 1954-2040), enqueue-only on the context's stream: with these a
 *      host drives the whole per-frame chain through this header alone (planarslam_amd/track.py launches nothing else inside the timed region).
 *   reset_matches      fill(mvpMapPoints.begin(), mvpMapPoints.end(), NULL): every entry -1
 *   blocked_mask       mask[i] = match[i] >= 0                       (key points / lines that already have a map point are skipped by SearchByProjection)
 *   merge_matches      out = first >= 0 ? first : (second >= 0 ? second + offset : second)   (one index space for PoseOptimization: [last frame | older frame])
 *   manhattan_pose     Tcw_out = Tcw_in with its rotation block replaced by mRotation_wc = (Rotation_cm * MF_can^T)^T (:250-253), as :1778 does before
 *                      TranslationOptimization; Rcm_new = TrackManhattanFrame's result, Rcm0 = the stream's Rotation_cm; [B][9] / [B][16] row-major float32
 *   keypoint_fields    KeyPoint::octave / ::angle of n key points into flat arrays
 *   add_scalar_i32     dst = src + value (per-line rand() seeds of Frame::isLineGood: base + step offset)
 *   copy_rows          pitched device-to-device row copy */
/* Frame::UndistortKeyPoints (reference src/Frame.cc:545-573: cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK), five iterations in double): keys [B][stride]
 * -> keys_un [B][stride] (every field copied, pt undistorted); dist_coef: host array k1, k2, p1, p2, k3 (Camera.k1 .. Camera.k3 of the reference's yaml files);
 * k1 == 0 copies, as :546-549.  Frame::ComputeImageBounds (:575-598) is the same call on the four image corners. */
int planar_undistort_keypoints(planar_ctx* ctx, int B, const planar_keypoint* keys, const int32_t* n, int stride, float fx, float fy, float cx, float cy, const float* dist_coef,
                               planar_keypoint* keys_un);
int planar_undistort_keypoints_dev(planar_ctx* ctx, int B, const planar_keypoint* d_keys, const int32_t* d_n, int stride, float fx, float fy, float cx, float cy,
                                   const float* dist_coef, planar_keypoint* d_keys_un);
int planar_reset_matches_dev(planar_ctx* ctx, int32_t* d_match, int64_t n);
int planar_blocked_mask_dev(planar_ctx* ctx, const int32_t* d_match, int64_t n, uint8_t* d_mask);
int planar_merge_matches_dev(planar_ctx* ctx, const int32_t* d_first, const int32_t* d_second, int offset, int64_t n, int32_t* d_out);
int planar_manhattan_pose_dev(planar_ctx* ctx, int B, const float* d_Rcm_new, const float* d_Rcm0, const float* d_Tcw_in, float* d_Tcw_out);
int planar_keypoint_fields_dev(planar_ctx* ctx, const planar_keypoint* d_keys, int64_t n, int32_t* d_octave, float* d_angle);
int planar_add_scalar_i32_dev(planar_ctx* ctx, const int32_t* d_src, int64_t n, int32_t value, int32_t* d_dst);
int planar_copy_rows_dev(planar_ctx* ctx, void* d_dst, int64_t dst_pitch, const void* d_src, int64_t src_pitch, int64_t row_bytes, int64_t rows);


#ifdef __cplusplus
}
#endif
#endif /* PLANAR_ABI_H */
