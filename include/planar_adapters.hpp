// planar_adapters.hpp — drop-in C++ classes / member functions with the reference's own signatures over the C ABI (planar_abi.h).
//
// Build PlanarSLAM with these instead of src/ORBextractor.cc, src/PlaneExtractor.cpp, src/LSDextractor.cpp, src/ORBmatcher.cc,
// src/LSDmatcher.cpp, src/PlaneMatcher.cpp and the two pose functions of src/Optimizer.cc, and link libplanar_hip.so.
//   Planar_SLAM::ORBextractor   <- include/ORBextractor.h:45-112, src/ORBextractor.cc                 (always)
//   PlaneDetection              <- include/PlaneExtractor.h:36-56, src/PlaneExtractor.cpp             (always)
//   Planar_SLAM::LineSegment    <- include/LSDextractor.h:344-352, src/LSDextractor.cpp               (PLANAR_ADAPTERS_WITH_LINES)
//   ORBmatcher::Fuse / LSDmatcher::Fuse (search on the device, map edits as in the reference)         (PLANAR_ADAPTERS_WITH_FUSE, needs one accessor, see there)
//   ORBmatcher / LSDmatcher / PlaneMatcher / Optimizer member functions                                (PLANAR_ADAPTERS_WITH_TRACKING:
//       include this header AFTER the reference's Frame.h, KeyFrame.h, MapPoint.h, MapLine.h, MapPlane.h, ORBmatcher.h, LSDmatcher.h,
//       PlaneMatcher.h and Optimizer.h; it then DEFINES the member functions those headers declare - gather the Frame fields into flat
//       arrays, call the ABI, scatter the result back)
//
// Threading and lifetime.  The reference starts three std::threads per Frame (src/Frame.cc:90-95: ExtractLSD / ExtractORB / ComputePlanes),
// copies Frames (and the PlaneDetection inside them) by value (src/Tracking.cc:177) and calls LineSegment::ExtractLineSegment through a
// pointer it never initialises (include/Frame.h:123, src/Frame.cc:172).  Therefore NO adapter object owns a device resource: contexts and
// plans live in one process-wide runtime (planar_adapter::Runtime), one context + stream + mutex per ROLE (points, lines, planes,
// tracking), so the three extraction threads run concurrently and nothing is created or leaked per frame or per thread; plans are cached
// by image size (and ORB parameters).  The adapter classes hold only host data and are freely copyable.
#pragma once
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>

#include <cassert>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <tuple>
#include <vector>

#include "planar_abi.h"

namespace planar_adapter {

enum Role { POINTS = 0, LINES = 1, PLANES = 2, TRACKING = 3, NROLES = 4 };

class Runtime {
public:
    static Runtime& get() { static Runtime r; return r; }           // destroyed at process exit: plans first, then contexts
    static void complain(const char* what) { std::fprintf(stderr, "planar (%s): %s\n", what, planar_last_error()); }
    struct Lane { std::mutex mu; planar_ctx* ctx = nullptr; };
    // the role's context (created on first use) - hold lane(role).mu while a call on it is in flight
    Lane& lane(Role r) {
        Lane& l = lanes_[r];
        std::lock_guard<std::mutex> g(create_mu_);
        if (!l.ctx && planar_ctx_create(&l.ctx, device_) != PLANAR_OK) { l.ctx = nullptr; complain("planar_ctx_create"); }   // (no exceptions: this runs on the threads Frame's constructor spawns; the calls on a null context fail and degrade)
        return l;
    }
    typedef std::tuple<int, int, int, int, int, int, int> OrbKey;   // w, h, nfeatures, scale * 1e6, nlevels, ini, min
    planar_orb* orb(const planar_orb_params& p, int w, int h) {     // call with lane(POINTS).mu held
        const OrbKey k(w, h, p.nfeatures, (int)(p.scale_factor * 1e6f), p.nlevels, p.ini_th_fast, p.min_th_fast);
        auto it = orbs_.find(k);
        if (it != orbs_.end()) return it->second;
        planar_orb* o = nullptr;
        if (planar_orb_create(lanes_[POINTS].ctx, &p, w, h, 1, &o) != PLANAR_OK) { complain("planar_orb_create"); return nullptr; }
        return orbs_[k] = o;
    }
    planar_lsd* lsd(int w, int h) {                                 // call with lane(LINES).mu held
        auto it = lsds_.find({w, h});
        if (it != lsds_.end()) return it->second;
        planar_lsd* o = nullptr;
        if (planar_lsd_create(lanes_[LINES].ctx, w, h, 1, &o) != PLANAR_OK) { complain("planar_lsd_create"); return nullptr; }
        return lsds_[{w, h}] = o;
    }
    planar_peac* peac(int w, int h) {                               // call with lane(PLANES).mu held
        auto it = peacs_.find({w, h});
        if (it != peacs_.end()) return it->second;
        planar_peac* o = nullptr;
        // (a size the plane path does not support - more than ~10 900 blocks of 10x10: 1280x720 fits since round 6, 1920x1080 does not - is reported ONCE: the failed size is
        //  remembered as a null handle, later frames of that size track without planes without repeating the message; INTEGRATION.md "Frame sizes")
        if (planar_peac_create(lanes_[PLANES].ctx, w, h, 1, &o) != PLANAR_OK) { complain("planar_peac_create"); o = nullptr; }
        return peacs_[{w, h}] = o;
    }
    planar_plane_clouds* clouds(int w, int h) {                     // call with lane(PLANES).mu held
        auto it = clouds_.find({w, h});
        if (it != clouds_.end()) return it->second;
        planar_plane_clouds* o = nullptr;
        // 8192 voxels of 0.1 m per pass (a frame with more goes plane by plane); frames of more than 2^19 pixels (1280x720) carry 20 bits of pixel in a sort word: 4096 voxels per pass
        const int max_voxels = (long long)w * h > (1ll << 19) ? 4096 : 8192;
        if (planar_plane_clouds_create(lanes_[PLANES].ctx, w, h, 1, max_voxels, &o) != PLANAR_OK) { complain("planar_plane_clouds_create"); o = nullptr; }   // (sizes beyond 2^20 pixels: reported once)
        return clouds_[{w, h}] = o;
    }
    void set_device(int d) { device_ = d; }
private:
    Runtime() {}
    ~Runtime() {
        for (auto& kv : orbs_) planar_orb_destroy(kv.second);
        for (auto& kv : lsds_) planar_lsd_destroy(kv.second);
        for (auto& kv : peacs_) planar_peac_destroy(kv.second);
        for (auto& kv : clouds_) planar_plane_clouds_destroy(kv.second);
        for (Lane& l : lanes_) if (l.ctx) planar_ctx_destroy(l.ctx);
    }
    Runtime(const Runtime&) = delete;
    Runtime& operator=(const Runtime&) = delete;
    int device_ = 0;
    std::mutex create_mu_;
    Lane lanes_[NROLES];
    std::map<OrbKey, planar_orb*> orbs_;
    std::map<std::pair<int, int>, planar_lsd*> lsds_;
    std::map<std::pair<int, int>, planar_peac*> peacs_;
    std::map<std::pair<int, int>, planar_plane_clouds*> clouds_;
};

// The reference's convention is no exceptions: its extractors run on threads Frame's constructor spawns, its matchers / optimisers on the tracking and
// local-mapping threads, none with a handler - a throw would end in std::terminate.  A failing call says so on stderr and the adapter degrades to "nothing
// found" (no key points / lines / planes / matches, the pose left alone), which the reference's own state machine handles (tracking lost -> relocalisation).
inline bool ok(int rc, const char* where) {
    if (rc == PLANAR_OK) return true;
    std::fprintf(stderr, "planar (%s): %s - degraded to \"nothing found\"\n", where, planar_last_error());
    return false;
}

}  // namespace planar_adapter

namespace Planar_SLAM {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
        : nfeatures(nfeatures), scaleFactor(scaleFactor), nlevels(nlevels), iniThFAST(iniThFAST), minThFAST(minThFAST) {
        mvImagePyramid.resize(nlevels);
    }

    // Same contract as the reference operator() (src/ORBextractor.cc:1043-1105); mask is ignored there too.
    void operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors) {
        if (_image.empty()) return;
        cv::Mat image = _image.getMat();
        assert(image.type() == CV_8UC1);   // src/ORBextractor.cc:1050
        planar_adapter::Runtime& R = planar_adapter::Runtime::get();
        planar_adapter::Runtime::Lane& L = R.lane(planar_adapter::POINTS);
        std::lock_guard<std::mutex> g(L.mu);
        planar_orb* orb = R.orb(params(), image.cols, image.rows);
        if (!orb) { _keypoints.clear(); _descriptors.release(); return; }
        const int cap = planar_orb_max_keypoints(orb);
        std::vector<planar_keypoint> kps(cap);
        std::vector<uint8_t> desc((size_t)cap * 32);
        int32_t n = 0;
        if (!planar_adapter::ok(planar_orb_extract(orb, image.data, 1, (int)image.step, (int64_t)image.step * image.rows, kps.data(), desc.data(), &n), "ORBextractor")) n = 0;
        static_assert(sizeof(cv::KeyPoint) == sizeof(planar_keypoint), "cv::KeyPoint layout");
        _keypoints.resize(n);
        if (n) std::memcpy((void*)_keypoints.data(), kps.data(), (size_t)n * sizeof(planar_keypoint));
        if (n == 0) { _descriptors.release(); }
        else {
            _descriptors.create(n, 32, CV_8U);
            cv::Mat d = _descriptors.getMat();
            for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), &desc[(size_t)i * 32], 32);
        }
        for (int l = 0; l < nlevels; l++) {      // public member of the reference class (include/ORBextractor.h:85)
            int w, h;
            planar_orb_level_size(orb, l, &w, &h);
            mvImagePyramid[l].create(h, w, CV_8UC1);
            planar_orb_read_level(orb, 0, l, mvImagePyramid[l].data);
        }
    }

    int GetLevels() { return nlevels; }
    float GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> GetScaleFactors() { return factors(0); }
    std::vector<float> GetInverseScaleFactors() { return factors(1); }
    std::vector<float> GetScaleSigmaSquares() { return factors(2); }
    std::vector<float> GetInverseScaleSigmaSquares() { return factors(3); }

    std::vector<cv::Mat> mvImagePyramid;

protected:
    planar_orb_params params() const { return planar_orb_params{nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST}; }
    std::vector<float> factors(int which) {
        // the scale tables depend only on the constructor arguments; any cached plan with these parameters can answer
        planar_adapter::Runtime& R = planar_adapter::Runtime::get();
        planar_adapter::Runtime::Lane& L = R.lane(planar_adapter::POINTS);
        std::lock_guard<std::mutex> g(L.mu);
        planar_orb* orb = R.orb(params(), 640, 480);
        std::vector<float> v[4];
        for (auto& x : v) x.resize(nlevels);
        planar_orb_get_scale_factors(orb, v[0].data(), v[1].data(), v[2].data(), v[3].data());
        return v[which];
    }
    int nfeatures; double scaleFactor; int nlevels, iniThFAST, minThFAST;
};

}  // namespace Planar_SLAM

// ---- PlaneDetection (include/PlaneExtractor.h:36-56).  Frame::ComputePlanes (src/Frame.cc:647-672) reads plane_num_,
// plane_vertices_[i], cloud.vertices[j] and plane_filter.extractedPlanes[i]->normal/center; the same members exist here.
// A PlaneDetection is a by-value member of every Frame and is copied with it: it owns no device handle, and its
// extractedPlanes pointers are re-seated on copy.
struct PlanarPlaneSeg { double normal[3], center[3], mse; int N; };
struct PlanarPlaneFilter {
    std::vector<PlanarPlaneSeg*> extractedPlanes;
    std::vector<PlanarPlaneSeg> storage;
    PlanarPlaneFilter() {}
    PlanarPlaneFilter(const PlanarPlaneFilter& o) : storage(o.storage) { reseat(); }
    PlanarPlaneFilter& operator=(const PlanarPlaneFilter& o) { if (this != &o) { storage = o.storage; reseat(); } return *this; }
    void reseat() { extractedPlanes.resize(storage.size()); for (size_t i = 0; i < storage.size(); i++) extractedPlanes[i] = &storage[i]; }
};
struct PlanarVertex { double v[3]; double operator[](int i) const { return v[i]; } };
struct ImagePointCloud { std::vector<PlanarVertex> vertices; int w = 0, h = 0; };

class PlaneDetection {
public:
    ImagePointCloud cloud;
    PlanarPlaneFilter plane_filter;
    std::vector<std::vector<int>> plane_vertices_;
    cv::Mat seg_img_, color_img_;
    int plane_num_ = 0;

    bool readColorImage(cv::Mat RGBImg) { color_img_ = RGBImg; return !(color_img_.empty() || color_img_.depth() != CV_8U); }

    // src/PlaneExtractor.cpp:26-57: keeps the depth and the intrinsics; the XYZ cloud is produced with the same FP64 arithmetic.
    bool readDepthImage(cv::Mat depthImg, cv::Mat& K, float kScaleFactor) {
        if (depthImg.empty() || depthImg.depth() != CV_16U) { std::printf("WARNING: cannot read depth image. No such a file, or the image format is not 16UC1\n"); return false; }
        depth_ = depthImg; factor_ = kScaleFactor;
        fx_ = K.at<float>(0, 0); fy_ = K.at<float>(1, 1); cx_ = K.at<float>(0, 2); cy_ = K.at<float>(1, 2);
        cloud.w = depthImg.cols; cloud.h = depthImg.rows;
        cloud.vertices.resize((size_t)cloud.w * cloud.h);
        for (int i = 0; i < depthImg.rows; i++)
            for (int j = 0; j < depthImg.cols; j++) {
                const double z = (double)depthImg.at<unsigned short>(i, j) * kScaleFactor;
                PlanarVertex& p = cloud.vertices[(size_t)i * cloud.w + j];
                p.v[0] = ((double)j - cx_) * z / fx_; p.v[1] = ((double)i - cy_) * z / fy_; p.v[2] = z;
            }
        return true;
    }

    void runPlaneDetection(int H, int W) {   // src/PlaneExtractor.cpp:59-65
        std::vector<int32_t>& labels = labels_;
        std::vector<double>& planes = planes_;
        labels.assign((size_t)W * H, -1);
        planes.assign((size_t)planar_peac_max_planes() * 8, 0.0);
        int32_t n = 0;
        {
            planar_adapter::Runtime& R = planar_adapter::Runtime::get();
            planar_adapter::Runtime::Lane& L = R.lane(planar_adapter::PLANES);
            std::lock_guard<std::mutex> g(L.mu);
            planar_peac* pp = R.peac(W, H);
            if (!pp) { n = 0; labels.assign((size_t)W * H, -1); }          // (unsupported frame size: said once, at create)
            else if (!planar_adapter::ok(planar_peac_segment(pp, (const uint16_t*)depth_.data, 1, (int)(depth_.step / 2), (int64_t)(depth_.step / 2) * H, fx_, fy_, cx_, cy_,
                                                        factor_, labels.data(), planes.data(), &n), "PlaneDetection")) { n = 0; labels.assign((size_t)W * H, -1); }
        }
        plane_num_ = n;
        plane_vertices_.assign(n, std::vector<int>());
        for (size_t i = 0; i < labels.size(); i++) if (labels[i] >= 0) plane_vertices_[labels[i]].push_back((int)i);   // raster order, as :362-372
        plane_filter.storage.resize(n);
        for (int i = 0; i < n; i++) {
            PlanarPlaneSeg& s = plane_filter.storage[i];
            const double* p = &planes[(size_t)i * 8];
            s.N = (int)p[0]; for (int k = 0; k < 3; k++) { s.normal[k] = p[1 + k]; s.center[k] = p[4 + k]; } s.mse = p[7];
        }
        plane_filter.reseat();
        seg_img_ = cv::Mat(H, W, CV_8UC3);   // visualisation only (colours are out of scope)
    }

    // NOT a member of the reference class: the loop of Frame::ComputePlanes over the detected planes (src/Frame.cc:655-692: pcl::VoxelGrid(0.1) of every
    // plane's points, coefficient (n, -n.c), Frame::MaxPointDistanceFromPlane with its pcl::SACSegmentation refit) as one call.  In Frame::ComputePlanes:
    //     planeDetector.ComputePlaneClouds(Config::Get<double>("Plane.DistanceThreshold"), mvPlanePoints, mvPlaneCoefficients);
    // CloudT: pcl::PointCloud<pcl::PointXYZRGB> or anything with a `points` vector whose elements have float x, y, z.  Returns mnPlaneNum.
    template <class CloudT>
    int ComputePlaneClouds(double disTh, std::vector<CloudT>& planePoints, std::vector<cv::Mat>& planeCoefficients, float leaf = 0.1f) {
        const int W = cloud.w, H = cloud.h, PS = planar_peac_max_planes(), MP = 8192;
        std::vector<float> coef((size_t)PS * 4), pts((size_t)MP * 3);
        std::vector<int32_t> src(PS), off(PS + 1);
        int32_t n_in = plane_num_, n_out = 0;
        auto append = [&](int k, const float* cf, const int32_t* of, const float* pt) {     // kept plane k of a call's outputs -> mvPlanePoints / mvPlaneCoefficients
            CloudT c;
            c.points.resize(of[k + 1] - of[k]);
            for (int i = of[k]; i < of[k + 1]; i++) { auto& p = c.points[i - of[k]]; p.x = pt[(size_t)i * 3]; p.y = pt[(size_t)i * 3 + 1]; p.z = pt[(size_t)i * 3 + 2]; }
            planePoints.push_back(c);
            cv::Mat m(4, 1, CV_32F);
            for (int t = 0; t < 4; t++) m.at<float>(t, 0) = cf[(size_t)k * 4 + t];
            planeCoefficients.push_back(m);
        };
        {
            planar_adapter::Runtime& R = planar_adapter::Runtime::get();
            planar_adapter::Runtime::Lane& L = R.lane(planar_adapter::PLANES);
            std::lock_guard<std::mutex> g(L.mu);
            if (!R.clouds(W, H)) return 0;                                   // (unsupported frame size: said once, at create)
            const int rc = planar_plane_clouds_compute(R.clouds(W, H), (const uint16_t*)depth_.data, 1, (int)(depth_.step / 2), (int64_t)(depth_.step / 2) * H, fx_, fy_,
                                                       cx_, cy_, factor_, labels_.data(), planes_.data(), &n_in, disTh, leaf, &n_out, coef.data(), src.data(), off.data(),
                                                       pts.data(), nullptr, nullptr, nullptr);
            int32_t fst = 0;                                                  // the frame's result code (3 = more voxels than the table holds), not the message's text
            if (rc == PLANAR_ECAPACITY) planar_plane_clouds_last_status(R.clouds(W, H), 1, &fst);
            if (rc == PLANAR_ECAPACITY && fst == 3) {
                // pcl::VoxelGrid has no voxel cap (src/Frame.cc:674-679); the frame's voxel table here has (8192 for all planes together).  The frame goes through once
                // more PLANE BY PLANE (planar_plane_clouds_set_plane_window), results appended in plane order - the very sequence of Frame::ComputePlanes' loop, every
                // plane's cloud and refit what the one-pass call would have produced.  Only a single plane of more than 8192 voxels (> 80 m^2 at the 0.1 m leaf) is
                // beyond the structure: it alone is dropped, with a message (this runs on the thread Frame's constructor spawned: no exception may leave it).
                planar_plane_clouds* pc = R.clouds(W, H);
                int total = 0;
                for (int i = 0; i < n_in; i++) {
                    int32_t n1 = 0;
                    planar_plane_clouds_set_plane_window(pc, i, 1);
                    const int rc1 = planar_plane_clouds_compute(pc, (const uint16_t*)depth_.data, 1, (int)(depth_.step / 2), (int64_t)(depth_.step / 2) * H, fx_, fy_, cx_, cy_, factor_,
                                                                labels_.data(), planes_.data(), &n_in, disTh, leaf, &n1, coef.data(), src.data(), off.data(), pts.data(), nullptr,
                                                                nullptr, nullptr);
                    int32_t pst = 0;
                    if (rc1 == PLANAR_ECAPACITY) planar_plane_clouds_last_status(pc, 1, &pst);
                    if (rc1 == PLANAR_ECAPACITY && pst == 3) {
                        std::fprintf(stderr, "planar: plane %d of this frame alone has more than %d voxels: dropped (%s)\n", i, MP, planar_last_error());
                        continue;
                    }
                    if (!planar_adapter::ok(rc1, "ComputePlaneClouds (plane by plane)")) { total = 0; planePoints.clear(); planeCoefficients.clear(); break; }
                    if (n1 == 1) { append(0, coef.data(), off.data(), pts.data()); total++; }
                }
                planar_plane_clouds_set_plane_window(pc, 0, -1);
                return total;
            }
            if (!planar_adapter::ok(rc, "ComputePlaneClouds")) return 0;
        }
        for (int k = 0; k < n_out; k++) append(k, coef.data(), off.data(), pts.data());
        return n_out;
    }

private:
    cv::Mat depth_;
    float factor_ = 0, fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0;
    std::vector<int32_t> labels_;       // what planar_peac_segment delivered (plane_vertices_ / plane_filter are views of these)
    std::vector<double> planes_;
};

// ---- LineSegment (reference include/LSDextractor.h:344-352, src/LSDextractor.cpp:12-39).  Opt-in: define PLANAR_ADAPTERS_WITH_LINES
//      before including this header in a build that has opencv_contrib's line_descriptor and Eigen (the types of the signature). ------
#ifdef PLANAR_ADAPTERS_WITH_LINES
#include <opencv2/line_descriptor/descriptor.hpp>
#include <eigen3/Eigen/Core>
namespace Planar_SLAM {
class LineSegment {   // no data members: Frame calls this through an uninitialised pointer (include/Frame.h:123), which only works for a method that never touches `this`
public:
    // scale / numOctaves are accepted for signature compatibility; the reference passes 1.2f -> (int)1 and 1 (one octave, full resolution)
    void ExtractLineSegment(const cv::Mat& img, std::vector<cv::line_descriptor::KeyLine>& keylines, cv::Mat& ldesc,
                            std::vector<Eigen::Vector3d>& keylineFunctions, float scale = 1.2, int numOctaves = 1) {
        (void)scale; (void)numOctaves;
        static_assert(sizeof(cv::line_descriptor::KeyLine) == sizeof(planar_keyline), "cv::line_descriptor::KeyLine layout");
        static_assert(sizeof(Eigen::Vector3d) == 3 * sizeof(double), "Eigen::Vector3d layout");
        assert(img.type() == CV_8UC1);
        const int W = img.cols, H = img.rows;
        const int lsdNFeatures = 40;                                   // src/LSDextractor.cpp:18
        keylines.resize(lsdNFeatures);
        std::vector<unsigned char> desc((size_t)lsdNFeatures * 32);
        const size_t first = keylineFunctions.size();                  // the reference push_backs
        keylineFunctions.resize(first + lsdNFeatures);
        int32_t n = 0;
        {
            planar_adapter::Runtime& R = planar_adapter::Runtime::get();
            planar_adapter::Runtime::Lane& L = R.lane(planar_adapter::LINES);
            std::lock_guard<std::mutex> g(L.mu);
            if (!planar_adapter::ok(planar_lsd_extract(R.lsd(W, H), img.data, 1, (int)img.step, (int64_t)img.step * H, lsdNFeatures, (planar_keyline*)keylines.data(), desc.data(),
                                                       (double*)(keylineFunctions.data() + first), &n), "LineSegment")) n = 0;
        }
        keylines.resize(n);
        keylineFunctions.resize(first + n);
        if (n) { ldesc = cv::Mat(n, 32, CV_8UC1); std::memcpy(ldesc.data, desc.data(), (size_t)n * 32); }   // BinaryDescriptor::compute leaves ldesc untouched when there are no lines
    }
};
}  // namespace Planar_SLAM
#endif

// ---- matchers and pose optimisers: definitions of the member functions the reference headers declare -----------------------------------
#ifdef PLANAR_ADAPTERS_WITH_TRACKING
namespace planar_adapter {

// Frame fields of the guided matchers -> planar_frame_view (the vectors keep the storage alive)
struct FrameGather {
    int32_t n;
    std::vector<planar_keypoint> keys;
    std::vector<float> ur;
    std::vector<uint8_t> desc, blocked;
    float Tcw[16];
    planar_frame_view view;
    template <class FrameT> explicit FrameGather(FrameT& F, bool want_blocked) {
        n = F.N;
        keys.resize(n > 0 ? n : 1); ur.resize(keys.size()); desc.resize(keys.size() * 32); blocked.assign(keys.size(), 0);
        for (int i = 0; i < n; i++) {
            std::memcpy(&keys[i], &F.mvKeysUn[i], sizeof(planar_keypoint));
            ur[i] = F.mvuRight.empty() ? -1.f : F.mvuRight[i];
            std::memcpy(&desc[(size_t)i * 32], F.mDescriptors.ptr(i), 32);
            if (want_blocked && F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) blocked[i] = 1;
        }
        std::memset(&view, 0, sizeof(view));
        view.B = 1; view.stride = (int32_t)keys.size(); view.n = &n; view.keys_un = keys.data(); view.u_right = ur.data(); view.desc = desc.data();
        view.blocked = want_blocked ? blocked.data() : nullptr;
        if (!F.mTcw.empty()) { for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) Tcw[4 * r + c] = F.mTcw.template at<float>(r, c); view.Tcw = Tcw; }
        view.min_x = FrameT::mnMinX; view.max_x = FrameT::mnMaxX; view.min_y = FrameT::mnMinY; view.max_y = FrameT::mnMaxY;
        view.grid_w_inv = FrameT::mfGridElementWidthInv; view.grid_h_inv = FrameT::mfGridElementHeightInv;
        view.fx = FrameT::fx; view.fy = FrameT::fy; view.cx = FrameT::cx; view.cy = FrameT::cy; view.bf = F.mbf; view.b = F.mb;
        for (size_t l = 0; l < F.mvScaleFactors.size() && l < PLANAR_MAX_LEVELS; l++) view.scale_factors[l] = F.mvScaleFactors[l];
    }
};
inline Runtime::Lane& tracking_lane() { return Runtime::get().lane(TRACKING); }

}  // namespace planar_adapter

namespace Planar_SLAM {

// ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)  (src/ORBmatcher.cc:1396-1535)
inline int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) {
    planar_adapter::FrameGather cur(CurrentFrame, true);
    const int NL = LastFrame.N;
    int32_t nl = NL;
    std::vector<uint8_t> usable(NL > 0 ? NL : 1, 0), observed(usable.size(), 0), mpd(usable.size() * 32, 0);
    std::vector<float> xw(usable.size() * 3, 0.f), ang(usable.size(), 0.f);
    std::vector<int32_t> oct(usable.size(), 0);
    float Tl[16];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) Tl[4 * r + c] = LastFrame.mTcw.at<float>(r, c);
    for (int i = 0; i < NL; i++) {
        MapPoint* pMP = LastFrame.mvpMapPoints[i];
        oct[i] = LastFrame.mvKeys[i].octave; ang[i] = LastFrame.mvKeysUn[i].angle;
        if (pMP && !LastFrame.mvbOutlier[i]) {
            usable[i] = 1;
            cv::Mat x = pMP->GetWorldPos(), d = pMP->GetDescriptor();
            for (int k = 0; k < 3; k++) xw[3 * i + k] = x.at<float>(k);
            std::memcpy(&mpd[(size_t)i * 32], d.ptr(0), 32);
            observed[i] = pMP->Observations() > 0;
        }
    }
    planar_last_frame_view last;
    last.stride = (int32_t)usable.size(); last.n = &nl; last.Tcw = Tl; last.usable = usable.data(); last.xw = xw.data(); last.octave = oct.data(); last.angle = ang.data();
    last.mp_desc = mpd.data(); last.mp_observed = observed.data();
    const int32_t UNTOUCHED = -2;
    std::vector<int32_t> match(cur.keys.size(), UNTOUCHED);
    int32_t nmatches = 0;
    {
        planar_adapter::Runtime::Lane& L = planar_adapter::tracking_lane();
        std::lock_guard<std::mutex> g(L.mu);
        if (!planar_adapter::ok(planar_search_by_projection_frame(L.ctx, &cur.view, &last, th, bMono ? 1 : 0, mbCheckOrientation ? 1 : 0, match.data(), &nmatches), "planar_search_by_projection_frame")) return 0;
    }
    for (int i = 0; i < CurrentFrame.N; i++) {
        if (match[i] == UNTOUCHED) continue;
        CurrentFrame.mvpMapPoints[i] = match[i] >= 0 ? LastFrame.mvpMapPoints[match[i]] : static_cast<MapPoint*>(NULL);
    }
    return nmatches;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)  (src/ORBmatcher.cc:46-130)
inline int ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th) {
    planar_adapter::FrameGather fr(F, true);
    const size_t M = vpMapPoints.size();
    int32_t nm = (int32_t)M;
    std::vector<uint8_t> inview(M ? M : 1, 0), obs(inview.size(), 0), desc(inview.size() * 32, 0);
    std::vector<float> px(inview.size(), 0.f), py(inview.size(), 0.f), pxr(inview.size(), 0.f), vc(inview.size(), 0.f);
    std::vector<int32_t> lvl(inview.size(), 0);
    for (size_t j = 0; j < M; j++) {
        MapPoint* p = vpMapPoints[j];
        if (!p->mbTrackInView || p->isBad()) continue;
        inview[j] = 1; px[j] = p->mTrackProjX; py[j] = p->mTrackProjY; pxr[j] = p->mTrackProjXR; lvl[j] = p->mnTrackScaleLevel; vc[j] = p->mTrackViewCos;
        cv::Mat d = p->GetDescriptor();
        std::memcpy(&desc[j * 32], d.ptr(0), 32);
        obs[j] = p->Observations() > 0;
    }
    planar_map_probes pr;
    pr.stride = (int32_t)inview.size(); pr.n = &nm; pr.in_view = inview.data(); pr.proj_x = px.data(); pr.proj_y = py.data(); pr.proj_xr = pxr.data(); pr.level = lvl.data();
    pr.view_cos = vc.data(); pr.desc = desc.data(); pr.observed = obs.data();
    const int32_t UNTOUCHED = -2;
    std::vector<int32_t> match(fr.keys.size(), UNTOUCHED);
    int32_t nmatches = 0;
    {
        planar_adapter::Runtime::Lane& L = planar_adapter::tracking_lane();
        std::lock_guard<std::mutex> g(L.mu);
        if (!planar_adapter::ok(planar_search_by_projection_map(L.ctx, &fr.view, &pr, th, mfNNratio, match.data(), &nmatches), "planar_search_by_projection_map")) return 0;
    }
    for (int i = 0; i < F.N; i++) if (match[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[match[i]];
    return nmatches;
}

// ORBmatcher::MatchORBPoints(Frame&, const Frame&)  (src/ORBmatcher.cc:1332-1394)
inline int ORBmatcher::MatchORBPoints(Frame& CurrentFrame, const Frame& LastFrame) {
    int32_t nc = CurrentFrame.N, nl = LastFrame.N;
    std::vector<uint8_t> cd((size_t)(nc > 0 ? nc : 1) * 32), ld((size_t)(nl > 0 ? nl : 1) * 32), has(nl > 0 ? nl : 1, 0), outl(has.size(), 0);
    for (int i = 0; i < nc; i++) std::memcpy(&cd[(size_t)i * 32], CurrentFrame.mDescriptors.ptr(i), 32);
    for (int j = 0; j < nl; j++) {
        std::memcpy(&ld[(size_t)j * 32], LastFrame.mDescriptors.ptr(j), 32);
        has[j] = LastFrame.mvpMapPoints[j] != NULL;
        outl[j] = (size_t)j < LastFrame.mvbOutlier.size() && LastFrame.mvbOutlier[j];
    }
    const int32_t UNTOUCHED = -2;
    std::vector<int32_t> match(nc > 0 ? nc : 1, UNTOUCHED);
    int32_t npair = 0;
    {
        planar_adapter::Runtime::Lane& L = planar_adapter::tracking_lane();
        std::lock_guard<std::mutex> g(L.mu);
        if (!planar_adapter::ok(planar_match_orb_points(L.ctx, cd.data(), &nc, nc > 0 ? nc : 1, ld.data(), &nl, nl > 0 ? nl : 1, has.data(), outl.data(), 1, match.data(), &npair), "planar_match_orb_points")) return 0;
    }
    for (int i = 0; i < nc; i++) if (match[i] >= 0) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[match[i]];
    return npair;
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)  (src/ORBmatcher.cc:160-292).  A DBoW2::FeatureVector maps node id ->
// feature indices; every feature is in at most one node, so it is passed as one node id per feature.
inline int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
    const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL));
    int32_t nk = (int32_t)vpMapPointsKF.size(), nf = F.N;
    if (nk == 0 || nf == 0) return 0;
    std::vector<int32_t> knode(nk, -1), fnode(nf, -1);
    for (const auto& kv : pKF->mFeatVec) for (unsigned int i : kv.second) if ((int)i < nk) knode[i] = (int32_t)kv.first;
    for (const auto& kv : F.mFeatVec) for (unsigned int i : kv.second) if ((int)i < nf) fnode[i] = (int32_t)kv.first;
    std::vector<uint8_t> kus(nk, 0), kd((size_t)nk * 32), fd((size_t)nf * 32);
    std::vector<float> kang(nk), fang(nf);
    for (int i = 0; i < nk; i++) {
        kus[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();
        kang[i] = pKF->mvKeysUn[i].angle;
        std::memcpy(&kd[(size_t)i * 32], pKF->mDescriptors.ptr(i), 32);
    }
    for (int i = 0; i < nf; i++) { fang[i] = F.mvKeys[i].angle; std::memcpy(&fd[(size_t)i * 32], F.mDescriptors.ptr(i), 32); }
    std::vector<int32_t> match(nf, -1);
    int32_t nmatches = 0;
    {
        planar_adapter::Runtime::Lane& L = planar_adapter::tracking_lane();
        std::lock_guard<std::mutex> g(L.mu);
        if (!planar_adapter::ok(planar_search_by_bow(L.ctx, 1, &nk, nk, knode.data(), kus.data(), kang.data(), kd.data(), &nf, nf, fnode.data(), fang.data(), fd.data(), mfNNratio,
                                                   mbCheckOrientation ? 1 : 0, match.data(), &nmatches), "planar_search_by_bow")) return 0;
    }
    for (int i = 0; i < nf; i++) if (match[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[match[i]];
    return nmatches;
}

// LSDmatcher::SearchByDescriptor(KeyFrame*, Frame&, vector<MapLine*>&)  (src/LSDmatcher.cpp:242-279)
inline int LSDmatcher::SearchByDescriptor(KeyFrame* pKF, Frame& currentF, std::vector<MapLine*>& vpMapLineMatches) {
    const std::vector<MapLine*> vpMapLinesKF = pKF->GetMapLineMatches();
    int32_t nk = pKF->mLineDescriptors.rows, nc = currentF.mLdesc.rows;
    vpMapLineMatches = std::vector<MapLine*>(currentF.NL, static_cast<MapLine*>(NULL));
    if (nk == 0 || nc == 0) return 0;
    std::vector<uint8_t> kd((size_t)nk * 32), cd((size_t)nc * 32), has(nk, 0);
    for (int i = 0; i < nk; i++) { std::memcpy(&kd[(size_t)i * 32], pKF->mLineDescriptors.ptr(i), 32); has[i] = vpMapLinesKF[i] != NULL; }
    for (int i = 0; i < nc; i++) std::memcpy(&cd[(size_t)i * 32], currentF.mLdesc.ptr(i), 32);
    std::vector<int32_t> match(nc, -1);
    int32_t nmatches = 0;
    {
        planar_adapter::Runtime::Lane& L = planar_adapter::tracking_lane();
        std::lock_guard<std::mutex> g(L.mu);
        if (!planar_adapter::ok(planar_lsd_search_by_descriptor(L.ctx, kd.data(), &nk, nk, cd.data(), &nc, nc, has.data(), 1, match.data(), &nmatches), "planar_lsd_search_by_descriptor")) return 0;
    }
    for (int i = 0; i < nc && i < currentF.NL; i++) if (match[i] >= 0) vpMapLineMatches[i] = vpMapLinesKF[match[i]];
    return nmatches;
}

// LSDmatcher::SearchByProjection(Frame&, const vector<MapLine*>&, th)  (src/LSDmatcher.cpp:141-211)
inline int LSDmatcher::SearchByProjection(Frame& F, const std::vector<MapLine*>& vpMapLines, const float th) {
    int32_t nl = F.NL, nm = (int32_t)vpMapLines.size();
    if (nl == 0 || nm == 0) return 0;
    static_assert(sizeof(cv::line_descriptor::KeyLine) == sizeof(planar_keyline), "KeyLine layout");
    std::vector<uint8_t> ldesc((size_t)nl * 32), blocked(nl, 0), inview(nm, 0), obs(nm, 0), mdesc((size_t)nm * 32, 0);
    std::vector<float> proj((size_t)nm * 4, 0.f), vc(nm, 0.f);
    std::vector<int32_t> lvl(nm, 0);
    for (int i = 0; i < nl; i++) { std::memcpy(&ldesc[(size_t)i * 32], F.mLdesc.ptr(i), 32); blocked[i] = F.mvpMapLines[i] && F.mvpMapLines[i]->Observations() > 0; }
    for (int j = 0; j < nm; j++) {
        MapLine* p = vpMapLines[j];
        if (!p || p->isBad() || !p->mbTrackInView) continue;
        inview[j] = 1; proj[4 * j] = p->mTrackProjX1; proj[4 * j + 1] = p->mTrackProjY1; proj[4 * j + 2] = p->mTrackProjX2; proj[4 * j + 3] = p->mTrackProjY2;
        lvl[j] = p->mnTrackScaleLevel; vc[j] = p->mTrackViewCos;
        cv::Mat d = p->GetDescriptor();
        std::memcpy(&mdesc[(size_t)j * 32], d.ptr(0), 32);
        obs[j] = p->Observations() > 0;
    }
    const int32_t UNTOUCHED = -2;
    std::vector<int32_t> match(nl, UNTOUCHED);
    int32_t nmatches = 0;
    {
        planar_adapter::Runtime::Lane& L = planar_adapter::tracking_lane();
        std::lock_guard<std::mutex> g(L.mu);
        if (!planar_adapter::ok(planar_lsd_search_by_projection(L.ctx, 1, &nl, nl, (const planar_keyline*)F.mvKeylinesUn.data(), ldesc.data(), blocked.data(), &nm, nm, inview.data(),
                                                              proj.data(), lvl.data(), vc.data(), mdesc.data(), obs.data(), F.mvScaleFactors.data(), (int)F.mvScaleFactors.size(), th,
                                                              mfNNratio, match.data(), &nmatches), "planar_lsd_search_by_projection")) return 0;
    }
    for (int i = 0; i < nl; i++) if (match[i] >= 0) F.mvpMapLines[i] = vpMapLines[match[i]];
    return nmatches;
}

// ---- Fuse (LocalMapping::SearchInNeighbors).  Enabled with PLANAR_ADAPTERS_WITH_FUSE: the search needs the UNSCALED invariance distances of a map point / line
//      (MapPoint::PredictScale divides mfMaxDistance; GetMaxDistanceInvariance() returns 1.2f * it), which include/MapPoint.h / MapLine.h keep protected - add
//          void GetDistanceRange(float& mn, float& mx) { unique_lock<mutex> lock(mMutexPos); mn = mfMinDistance; mx = mfMaxDistance; }
//      to both classes.  The search half runs on the device; the map edits are the reference's own statements on its own objects.
#ifdef PLANAR_ADAPTERS_WITH_FUSE
// ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th)  (src/ORBmatcher.cc:829-979)
inline int ORBmatcher::Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th) {
    const int M = (int)vpMapPoints.size(), N = pKF->N;
    if (M == 0) return 0;
    // the key frame's side (what FrameGather collects of a Frame; KeyFrame keeps the bounds / grid constants per object and the pose behind GetPose())
    int32_t n = N;
    std::vector<planar_keypoint> keys(N > 0 ? N : 1);
    std::vector<float> ur(keys.size(), -1.f);
    std::vector<uint8_t> kdesc(keys.size() * 32, 0);
    for (int i = 0; i < N; i++) {
        std::memcpy(&keys[i], &pKF->mvKeysUn[i], sizeof(planar_keypoint));
        ur[i] = pKF->mvuRight.empty() ? -1.f : pKF->mvuRight[i];
        std::memcpy(&kdesc[(size_t)i * 32], pKF->mDescriptors.ptr(i), 32);
    }
    float Tcw[16];
    { cv::Mat T = pKF->GetPose(); for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) Tcw[4 * r + c] = T.at<float>(r, c); }
    planar_frame_view v;
    std::memset(&v, 0, sizeof(v));
    v.B = 1; v.stride = (int32_t)keys.size(); v.n = &n; v.keys_un = keys.data(); v.u_right = ur.data(); v.desc = kdesc.data(); v.Tcw = Tcw;
    v.min_x = pKF->mnMinX; v.max_x = pKF->mnMaxX; v.min_y = pKF->mnMinY; v.max_y = pKF->mnMaxY;
    v.grid_w_inv = pKF->mfGridElementWidthInv; v.grid_h_inv = pKF->mfGridElementHeightInv;
    v.fx = pKF->fx; v.fy = pKF->fy; v.cx = pKF->cx; v.cy = pKF->cy; v.bf = pKF->mbf; v.b = pKF->mb;
    for (size_t l = 0; l < pKF->mvScaleFactors.size() && l < PLANAR_MAX_LEVELS; l++) v.scale_factors[l] = pKF->mvScaleFactors[l];
    // the map points' side
    std::vector<uint8_t> usable(M, 0), desc((size_t)M * 32, 0);
    std::vector<float> xw((size_t)M * 3, 0.f), nrm((size_t)M * 3, 0.f), mn(M, 0.f), mx(M, 0.f);
    for (int j = 0; j < M; j++) {
        MapPoint* p = vpMapPoints[j];
        if (!p || p->isBad() || p->IsInKeyFrame(pKF)) continue;                     // :849-853
        usable[j] = 1;
        cv::Mat X = p->GetWorldPos(), nv = p->GetNormal(), d = p->GetDescriptor();
        for (int k = 0; k < 3; k++) { xw[3 * j + k] = X.at<float>(k); nrm[3 * j + k] = nv.at<float>(k); }
        std::memcpy(&desc[(size_t)j * 32], d.ptr(0), 32);
        p->GetDistanceRange(mn[j], mx[j]);
    }
    std::vector<int32_t> idx(M, -1);
    int32_t m = M, nFused = 0;
    {
        planar_adapter::Runtime::Lane& L = planar_adapter::tracking_lane();
        std::lock_guard<std::mutex> g(L.mu);
        if (!planar_adapter::ok(planar_fuse_search(L.ctx, &v, pKF->mvInvLevelSigma2.data(), pKF->mfLogScaleFactor, pKF->mnScaleLevels, &m, M, 0, usable.data(), xw.data(),
                                                 nrm.data(), mn.data(), mx.data(), desc.data(), th, idx.data(), nullptr, &nFused), "planar_fuse_search")) return 0;
    }
    for (int j = 0; j < M; j++) {                                                   // :953-974
        if (idx[j] < 0) continue;
        MapPoint* pMP = vpMapPoints[j];
        MapPoint* pMPinKF = pKF->GetMapPoint(idx[j]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                else pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, idx[j]);
            pKF->AddMapPoint(pMP, idx[j]);
        }
    }
    return nFused;
}

// LSDmatcher::Fuse(KeyFrame*, const vector<MapLine*>&, th)  (src/LSDmatcher.cpp:884-1015)
inline int LSDmatcher::Fuse(KeyFrame* pKF, const std::vector<MapLine*>& vpMapLines, const float th) {
    const int M = (int)vpMapLines.size();
    int32_t nl = (int32_t)pKF->mvKeyLines.size();
    if (M == 0 || nl == 0) return 0;
    static_assert(sizeof(cv::line_descriptor::KeyLine) == sizeof(planar_keyline), "KeyLine layout");
    std::vector<uint8_t> ldesc((size_t)nl * 32);
    for (int i = 0; i < nl; i++) std::memcpy(&ldesc[(size_t)i * 32], pKF->mLineDescriptors.ptr(i), 32);
    float Tcw[16];
    { cv::Mat T = pKF->GetPose(); for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) Tcw[4 * r + c] = T.at<float>(r, c); }
    planar_frame_view v;
    std::memset(&v, 0, sizeof(v));
    v.B = 1; v.stride = 1; v.Tcw = Tcw;
    v.min_x = pKF->mnMinX; v.max_x = pKF->mnMaxX; v.min_y = pKF->mnMinY; v.max_y = pKF->mnMaxY;
    v.fx = pKF->fx; v.fy = pKF->fy; v.cx = pKF->cx; v.cy = pKF->cy; v.bf = pKF->mbf; v.b = pKF->mb;
    const int n_levels = (int)std::min<size_t>(pKF->mvScaleFactors.size(), PLANAR_MAX_LEVELS);
    for (int l = 0; l < n_levels; l++) v.scale_factors[l] = pKF->mvScaleFactors[l];
    std::vector<uint8_t> usable(M, 0), desc((size_t)M * 32, 0);
    std::vector<double> xw6((size_t)M * 6, 0.0), nrm((size_t)M * 3, 0.0);
    std::vector<float> mn(M, 0.f), mx(M, 0.f);
    for (int j = 0; j < M; j++) {
        MapLine* p = vpMapLines[j];
        if (!p || p->isBad()) continue;                                             // :906-907
        usable[j] = 1;
        const Vector6d P = p->GetWorldPos();
        const Eigen::Vector3d Pn = p->GetNormal();
        for (int k = 0; k < 6; k++) xw6[6 * j + k] = P(k);
        for (int k = 0; k < 3; k++) nrm[3 * j + k] = Pn(k);
        cv::Mat d = p->GetDescriptor();
        std::memcpy(&desc[(size_t)j * 32], d.ptr(0), 32);
        p->GetDistanceRange(mn[j], mx[j]);
    }
    std::vector<int32_t> idx(M, -1);
    int32_t m = M, nFused = 0;
    {
        planar_adapter::Runtime::Lane& L = planar_adapter::tracking_lane();
        std::lock_guard<std::mutex> g(L.mu);
        if (!planar_adapter::ok(planar_lsd_fuse_search(L.ctx, &v, pKF->mfLogScaleFactor, n_levels, &nl, nl, (const planar_keyline*)pKF->mvKeyLines.data(), ldesc.data(), &m, M, 0,
                                                     usable.data(), xw6.data(), nrm.data(), mn.data(), mx.data(), desc.data(), th, idx.data(), nullptr, &nFused), "planar_lsd_fuse_search")) return 0;
    }
    for (int j = 0; j < M; j++) {                                                   // :993-1010
        if (idx[j] < 0) continue;
        MapLine* pML = vpMapLines[j];
        MapLine* pMLinKF = pKF->GetMapLine(idx[j]);
        if (pMLinKF) {
            if (!pMLinKF->isBad()) {
                if (pMLinKF->Observations() > pML->Observations()) pML->Replace(pMLinKF);
                else pMLinKF->Replace(pML);
            }
        } else {
            pML->AddObservation(pKF, idx[j]);
            pKF->AddMapLine(pML, idx[j]);
        }
    }
    return nFused;
}
#endif   // PLANAR_ADAPTERS_WITH_FUSE

// PlaneMatcher::SearchMapByCoefficients(Frame&, const vector<MapPlane*>&)  (src/PlaneMatcher.cpp:10-66)
inline int PlaneMatcher::SearchMapByCoefficients(Frame& pF, const std::vector<MapPlane*>& vpMapPlanes) {
    pF.mbNewPlane = false;
    int32_t np = pF.mnPlaneNum, nm = (int32_t)vpMapPlanes.size();
    if (np == 0) return 0;
    std::vector<float> coef((size_t)np * 4), T(16), mcoef((size_t)(nm ? nm : 1) * 4, 0.f);
    for (int i = 0; i < np; i++) for (int k = 0; k < 4; k++) coef[4 * i + k] = pF.mvPlaneCoefficients[i].template at<float>(k);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T[4 * r + c] = pF.mTcw.template at<float>(r, c);
    std::vector<uint8_t> valid(nm ? nm : 1, 0);
    std::vector<int32_t> npts(nm ? nm : 1, 0);
    int maxp = 1;
    for (int j = 0; j < nm; j++) if (!vpMapPlanes[j]->isBad() && vpMapPlanes[j]->mvPlanePoints) maxp = std::max(maxp, (int)vpMapPlanes[j]->mvPlanePoints->points.size());
    std::vector<float> pts((size_t)(nm ? nm : 1) * maxp * 3, 0.f);
    for (int j = 0; j < nm; j++) {
        MapPlane* p = vpMapPlanes[j];
        if (p->isBad()) continue;
        valid[j] = 1;
        cv::Mat c = p->GetWorldPos();
        for (int k = 0; k < 4; k++) mcoef[4 * j + k] = c.at<float>(k);
        if (p->mvPlanePoints) {
            npts[j] = (int32_t)p->mvPlanePoints->points.size();
            for (int k = 0; k < npts[j]; k++) { const auto& q = p->mvPlanePoints->points[k]; float* o = &pts[((size_t)j * maxp + k) * 3]; o[0] = q.x; o[1] = q.y; o[2] = q.z; }
        }
    }
    const float th[4] = {dTh, aTh, verTh, parTh};
    std::vector<int32_t> a(np, -1), v(np, -1), par(np, -1);
    int32_t nmatches = 0;
    {
        planar_adapter::Runtime::Lane& L = planar_adapter::tracking_lane();
        std::lock_guard<std::mutex> g(L.mu);
        if (!planar_adapter::ok(planar_plane_search_by_coefficients(L.ctx, 1, &np, np, coef.data(), T.data(), 0, &nm, nm ? nm : 1, valid.data(), mcoef.data(), npts.data(), maxp, pts.data(),
                                                                  th, a.data(), v.data(), par.data(), &nmatches), "planar_plane_search_by_coefficients")) return 0;
    }
    for (int i = 0; i < np; i++) {
        if (a[i] >= 0) pF.mvpMapPlanes[i] = vpMapPlanes[a[i]];
        if (v[i] >= 0) pF.mvpVerticalPlanes[i] = vpMapPlanes[v[i]];
        if (par[i] >= 0) pF.mvpParallelPlanes[i] = vpMapPlanes[par[i]];
    }
    return nmatches;
}

// Optimizer::PoseOptimization(Frame*) (src/Optimizer.cc:550-1275) / TranslationOptimization(Frame*) (:2995-3738).  cfg = the six Plane.* values
// Config::Get<double> returns (Plane.AngleInfo, DistanceInfo, ParallelInfo, VerticalInfo, Chi, VPChi).
namespace planar_detail {
template <class FrameT> inline int pose_opt(FrameT* pFrame, int mode) {
    const int N = pFrame->N, NL = pFrame->NL, M = pFrame->mnPlaneNum;
    const int MP = N > 0 ? N : 1, ML = NL > 0 ? NL : 1, MM = M > 0 ? M : 1;
    int32_t n_points = N, n_lines = NL, n_planes = M, n_inliers = 0;
    std::vector<uint8_t> pt_valid(MP, 0), ln_valid(ML, 0), pl_valid((size_t)MM * 3, 0), pt_out(MP, 0), ln_out(ML, 0), pl_out((size_t)MM * 3, 0);
    std::vector<float> pt_xw((size_t)MP * 3, 0.f), pt_obs((size_t)MP * 3, 0.f), pt_is2(MP, 1.f), pl_meas((size_t)MM * 4, 0.f), pl_world((size_t)MM * 12, 0.f), Tin(16), Tout(16);
    std::vector<double> ln_obs((size_t)ML * 3, 0.0), ln_xw((size_t)ML * 6, 0.0);
    for (int i = 0; i < N; i++) {
        const cv::KeyPoint& kp = pFrame->mvKeysUn[i];
        pt_obs[3 * i] = kp.pt.x; pt_obs[3 * i + 1] = kp.pt.y; pt_obs[3 * i + 2] = pFrame->mvuRight[i];
        pt_is2[i] = pFrame->mvInvLevelSigma2[kp.octave];
        if (MapPoint* p = pFrame->mvpMapPoints[i]) { pt_valid[i] = 1; cv::Mat x = p->GetWorldPos(); for (int k = 0; k < 3; k++) pt_xw[3 * i + k] = x.at<float>(k); }
    }
    for (int i = 0; i < NL; i++) {
        for (int k = 0; k < 3; k++) ln_obs[3 * i + k] = pFrame->mvKeyLineFunctions[i][k];
        if (MapLine* p = pFrame->mvpMapLines[i]) { ln_valid[i] = 1; for (int k = 0; k < 6; k++) ln_xw[6 * i + k] = p->mWorldPos[k]; }
    }
    for (int i = 0; i < M; i++) {
        for (int k = 0; k < 4; k++) pl_meas[4 * i + k] = pFrame->mvPlaneCoefficients[i].template at<float>(k);
        MapPlane* three[3] = {pFrame->mvpMapPlanes[i], pFrame->mvpParallelPlanes[i], pFrame->mvpVerticalPlanes[i]};
        for (int j = 0; j < 3; j++) if (three[j]) { pl_valid[3 * i + j] = 1; cv::Mat c = three[j]->GetWorldPos(); for (int k = 0; k < 4; k++) pl_world[(3 * i + j) * 4 + k] = c.at<float>(k); }
    }
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) Tin[4 * r + c] = pFrame->mTcw.template at<float>(r, c);
    planar_pose_batch pb;
    std::memset(&pb, 0, sizeof(pb));
    pb.B = 1; pb.max_points = MP; pb.max_lines = ML; pb.max_planes = MM;
    pb.n_points = &n_points; pb.n_lines = &n_lines; pb.n_planes = &n_planes; pb.pt_valid = pt_valid.data(); pb.pt_xw = pt_xw.data(); pb.pt_obs = pt_obs.data();
    pb.pt_inv_sigma2 = pt_is2.data(); pb.ln_valid = ln_valid.data(); pb.ln_obs = ln_obs.data(); pb.ln_xw = ln_xw.data(); pb.pl_meas = pl_meas.data(); pb.pl_valid = pl_valid.data();
    pb.pl_world = pl_world.data(); pb.Tcw_in = Tin.data(); pb.Tcw_out = Tout.data(); pb.pt_outlier = pt_out.data(); pb.ln_outlier = ln_out.data(); pb.pl_outlier = pl_out.data();
    pb.n_inliers = &n_inliers; pb.lm_iters = nullptr;
    planar_pose_params prm;
    prm.fx = FrameT::fx; prm.fy = FrameT::fy; prm.cx = FrameT::cx; prm.cy = FrameT::cy; prm.bf = pFrame->mbf;
    prm.angle_info = Config::Get<double>("Plane.AngleInfo"); prm.distance_info = Config::Get<double>("Plane.DistanceInfo");
    prm.parallel_info = Config::Get<double>("Plane.ParallelInfo"); prm.vertical_info = Config::Get<double>("Plane.VerticalInfo");
    prm.plane_chi = Config::Get<double>("Plane.Chi"); prm.vp_chi = Config::Get<double>("Plane.VPChi");
    {
        planar_adapter::Runtime::Lane& L = planar_adapter::tracking_lane();
        std::lock_guard<std::mutex> g(L.mu);
        if (!planar_adapter::ok(planar_pose_opt(L.ctx, &pb, &prm, mode, 4, 10), "planar_pose_opt")) return 0;
    }
    // the reference writes the flags of the correspondences it used and the pose through Frame::SetPose
    for (int i = 0; i < N; i++) if (pt_valid[i]) pFrame->mvbOutlier[i] = pt_out[i] != 0;
    for (int i = 0; i < NL; i++) if (ln_valid[i]) pFrame->mvbLineOutlier[i] = ln_out[i] != 0;
    for (int i = 0; i < M; i++) {
        if (pl_valid[3 * i]) pFrame->mvbPlaneOutlier[i] = pl_out[3 * i] != 0;
        if (mode == PLANAR_POSE_FULL) {
            if (pl_valid[3 * i + 1]) pFrame->mvbParPlaneOutlier[i] = pl_out[3 * i + 1] != 0;
            if (pl_valid[3 * i + 2]) pFrame->mvbVerPlaneOutlier[i] = pl_out[3 * i + 2] != 0;
        }
    }
    cv::Mat pose(4, 4, CV_32F);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) pose.at<float>(r, c) = Tout[4 * r + c];
    pFrame->SetPose(pose);
    return n_inliers;
}
}  // namespace planar_detail
inline int Optimizer::PoseOptimization(Frame* pFrame) { return planar_detail::pose_opt(pFrame, PLANAR_POSE_FULL); }
inline int Optimizer::TranslationOptimization(Frame* pFrame) { return planar_detail::pose_opt(pFrame, PLANAR_POSE_TRANSLATION); }

}  // namespace Planar_SLAM
#endif   // PLANAR_ADAPTERS_WITH_TRACKING

// ---- Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*)  (src/Optimizer.cc:1853-2680): the graph the reference assembles from the covisibility
//      list and the observation maps becomes a planar_ba_problem, planar_local_ba solves it, the erase lists and the optimised values go back.
//      Define PLANAR_ADAPTERS_WITH_LOCAL_BA after including KeyFrame.h, MapPoint.h, MapLine.h, MapPlane.h, Map.h, Optimizer.h, Config.h. --------------
#ifdef PLANAR_ADAPTERS_WITH_LOCAL_BA
#include <list>
namespace Planar_SLAM {
inline void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap) {
    // local key frames, local landmarks, fixed cameras: the reference's traversal (:1855-1974), marks included
    std::vector<KeyFrame*> kfs;
    kfs.push_back(pKF);
    pKF->mnBALocalForKF = pKF->mnId;
    for (KeyFrame* k : pKF->GetVectorCovisibleKeyFrames()) { k->mnBALocalForKF = pKF->mnId; if (!k->isBad()) kfs.push_back(k); }
    const size_t n_local = kfs.size();
    std::vector<MapPoint*> pts; std::vector<MapLine*> lns; std::vector<MapPlane*> pls;
    for (size_t i = 0; i < n_local; i++) {
        for (MapPoint* p : kfs[i]->GetMapPointMatches()) if (p && !p->isBad() && p->mnBALocalForKF != pKF->mnId) { pts.push_back(p); p->mnBALocalForKF = pKF->mnId; }
    }
    for (size_t i = 0; i < n_local; i++) {
        for (MapLine* p : kfs[i]->GetMapLineMatches()) if (p && !p->isBad() && p->mnBALocalForKF != pKF->mnId) { lns.push_back(p); p->mnBALocalForKF = pKF->mnId; }
    }
    for (size_t i = 0; i < n_local; i++) {
        for (MapPlane* p : kfs[i]->GetMapPlaneMatches()) if (p && !p->isBad() && p->mnBALocalForKF != pKF->mnId) { pls.push_back(p); p->mnBALocalForKF = pKF->mnId; }
    }
    auto fix_observers = [&](const std::map<KeyFrame*, size_t>& obs) {
        for (const auto& o : obs) {
            KeyFrame* k = o.first;
            if (k->mnBALocalForKF != pKF->mnId && k->mnBAFixedForKF != pKF->mnId) { k->mnBAFixedForKF = pKF->mnId; if (!k->isBad()) kfs.push_back(k); }
        }
    };
    for (MapPoint* p : pts) fix_observers(p->GetObservations());
    for (MapLine* p : lns) fix_observers(p->GetObservations());
    for (MapPlane* p : pls) fix_observers(p->GetObservations());
    const int K = (int)kfs.size();
    std::map<KeyFrame*, int> kf_index;
    std::vector<float> kf_Tcw((size_t)K * 16);
    std::vector<uint8_t> kf_fixed(K);
    unsigned long maxKFid = 0;
    for (int k = 0; k < K; k++) {
        kf_index[kfs[k]] = k;
        cv::Mat T = kfs[k]->GetPose();
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) kf_Tcw[(size_t)k * 16 + 4 * r + c] = T.at<float>(r, c);
        kf_fixed[k] = (size_t)k >= n_local || kfs[k]->mnId == 0;
        if (kfs[k]->mnId > maxKFid) maxKFid = kfs[k]->mnId;
    }
    // landmarks and edges
    std::vector<uint8_t> lm_type, e_type;
    std::vector<double> lm_init, e_meas;
    std::vector<int32_t> e_kf, e_lm;
    std::vector<float> e_is2;
    struct Assoc { KeyFrame* kf; void* lm; };                 // what an erased edge removes
    std::vector<Assoc> assoc;
    auto add_lm = [&](int type, double a, double b, double c, double d) { lm_type.push_back((uint8_t)type); lm_init.insert(lm_init.end(), {a, b, c, d}); return (int32_t)lm_type.size() - 1; };
    auto add_edge = [&](int kf, int lm, int type, double m0, double m1, double m2, double m3, float is2, KeyFrame* okf, void* olm) {
        e_kf.push_back(kf); e_lm.push_back(lm); e_type.push_back((uint8_t)type); e_meas.insert(e_meas.end(), {m0, m1, m2, m3}); e_is2.push_back(is2); assoc.push_back(Assoc{okf, olm});
    };
    std::vector<int32_t> pt_lm(pts.size()), ln_lm(lns.size()), pl_lm(pls.size());
    for (size_t i = 0; i < pts.size(); i++) {
        cv::Mat X = pts[i]->GetWorldPos();
        pt_lm[i] = add_lm(0, X.at<float>(0), X.at<float>(1), X.at<float>(2), 0);
        const std::map<KeyFrame*, size_t> obs = pts[i]->GetObservations();
        for (const auto& o : obs) {
            KeyFrame* k = o.first;
            if (k->isBad()) continue;
            const cv::KeyPoint& kp = k->mvKeysUn[o.second];
            const float ur = k->mvuRight[o.second], is2 = k->mvInvLevelSigma2[kp.octave];
            if (ur < 0) add_edge(kf_index.at(k), pt_lm[i], PLANAR_BA_MONO, kp.pt.x, kp.pt.y, 0, 0, is2, k, pts[i]);
            else add_edge(kf_index.at(k), pt_lm[i], PLANAR_BA_STEREO, kp.pt.x, kp.pt.y, ur, 0, is2, k, pts[i]);
        }
    }
    for (size_t i = 0; i < lns.size(); i++) {
        const auto W = lns[i]->GetWorldPos();                  // Vector6d: start xyz, end xyz
        ln_lm[i] = add_lm(0, W[0], W[1], W[2], 0);
        add_lm(0, W[3], W[4], W[5], 0);
        const std::map<KeyFrame*, size_t> obs = lns[i]->GetObservations();
        for (const auto& o : obs) {
            if (o.first->isBad()) continue;
            // both end-point edges hang on the CURRENT key frame's vertex and use its line function at the observer's slot (:2170-2194)
            const auto f = pKF->mvKeyLineFunctions[o.second];
            add_edge(0, ln_lm[i], PLANAR_BA_LINE, f[0], f[1], f[2], 0, 1.f, o.first, lns[i]);
            add_edge(0, ln_lm[i] + 1, PLANAR_BA_LINE, f[0], f[1], f[2], 0, 1.f, o.first, lns[i]);
        }
    }
    for (size_t i = 0; i < pls.size(); i++) {
        cv::Mat C = pls[i]->GetWorldPos();
        pl_lm[i] = add_lm(1, C.at<float>(0), C.at<float>(1), C.at<float>(2), C.at<float>(3));
        auto plane_edges = [&](const std::map<KeyFrame*, size_t>& obs, int type) {
            for (const auto& o : obs) {
                KeyFrame* k = o.first;
                if (k->isBad() || k->mnId > maxKFid) continue;
                auto it = kf_index.find(k);
                if (it == kf_index.end()) continue;            // (the reference would dereference a null vertex here)
                const cv::Mat& m = k->mvPlaneCoefficients[o.second];
                add_edge(it->second, pl_lm[i], type, m.at<float>(0), m.at<float>(1), m.at<float>(2), m.at<float>(3), 1.f, k, pls[i]);
            }
        };
        plane_edges(pls[i]->GetObservations(), PLANAR_BA_PLANE);
        plane_edges(pls[i]->GetVerObservations(), PLANAR_BA_VERTICAL);
        plane_edges(pls[i]->GetParObservations(), PLANAR_BA_PARALLEL);
    }
    if (pbStopFlag && *pbStopFlag) return;
    const int NL = (int)lm_type.size(), NE = (int)e_type.size();
    std::vector<float> out_T((size_t)K * 16);
    std::vector<double> out_lm((size_t)std::max(NL, 1) * 4);
    std::vector<uint8_t> out_e(std::max(NE, 1));
    planar_ba_problem P{K, kf_Tcw.data(), kf_fixed.data(), NL, lm_type.data(), lm_init.data(), NE, e_kf.data(), e_lm.data(), e_type.data(), e_meas.data(), e_is2.data()};
    planar_ba_result R{out_T.data(), out_lm.data(), out_e.data(), 0, 0};
    planar_pose_params prm;
    prm.fx = pKF->fx; prm.fy = pKF->fy; prm.cx = pKF->cx; prm.cy = pKF->cy; prm.bf = pKF->mbf;
    prm.angle_info = Config::Get<double>("Plane.AngleInfo"); prm.distance_info = Config::Get<double>("Plane.DistanceInfo");
    prm.parallel_info = Config::Get<double>("Plane.ParallelInfo"); prm.vertical_info = Config::Get<double>("Plane.VerticalInfo");
    prm.plane_chi = Config::Get<double>("Plane.Chi"); prm.vp_chi = Config::Get<double>("Plane.VPChi");
    static_assert(sizeof(bool) == 1, "bool* pbStopFlag is passed as a byte flag");
    {
        planar_adapter::Runtime::Lane& L = planar_adapter::Runtime::get().lane(planar_adapter::TRACKING);
        std::lock_guard<std::mutex> g(L.mu);
        if (!planar_adapter::ok(planar_local_ba(L.ctx, &P, &prm, 5, 10, &R, reinterpret_cast<const volatile unsigned char*>(pbStopFlag), nullptr), "planar_local_ba")) return;
    }
    // erase lists (:2471-2620) and write-back (:2622-2680) under the map mutex
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
    for (int e = 0; e < NE; e++) {
        if (!out_e[e]) continue;
        KeyFrame* k = assoc[e].kf;
        switch (e_type[e]) {
            case PLANAR_BA_MONO: case PLANAR_BA_STEREO: { MapPoint* p = (MapPoint*)assoc[e].lm; if (p->isBad()) break; k->EraseMapPointMatch(p); p->EraseObservation(k); break; }
            case PLANAR_BA_LINE: {
                if (e > 0 && e_type[e - 1] == PLANAR_BA_LINE && assoc[e - 1].lm == assoc[e].lm && assoc[e - 1].kf == k && e_lm[e - 1] + 1 == e_lm[e]) break;   // the end-point twin
                MapLine* p = (MapLine*)assoc[e].lm; if (p->isBad()) break; k->EraseMapLineMatch(p); p->EraseObservation(k); break;
            }
            case PLANAR_BA_PLANE: { MapPlane* p = (MapPlane*)assoc[e].lm; if (p->isBad()) break; k->EraseMapPlaneMatch(p); p->EraseObservation(k); break; }
            case PLANAR_BA_VERTICAL: { MapPlane* p = (MapPlane*)assoc[e].lm; if (p->isBad()) break; k->EraseMapVerticalPlaneMatch(p); p->EraseVerObservation(k); break; }
            default: { MapPlane* p = (MapPlane*)assoc[e].lm; if (p->isBad()) break; k->EraseMapParallelPlaneMatch(p); p->EraseParObservation(k); break; }
        }
    }
    for (size_t k = 0; k < n_local; k++) {
        cv::Mat T(4, 4, CV_32F);
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T.at<float>(r, c) = out_T[k * 16 + 4 * r + c];
        kfs[k]->SetPose(T);
    }
    for (size_t i = 0; i < pts.size(); i++) {
        cv::Mat X(3, 1, CV_32F);
        for (int j = 0; j < 3; j++) X.at<float>(j) = (float)out_lm[(size_t)pt_lm[i] * 4 + j];
        pts[i]->SetWorldPos(X); pts[i]->UpdateNormalAndDepth();
    }
    for (size_t i = 0; i < lns.size(); i++) {
        auto W = lns[i]->GetWorldPos();
        for (int j = 0; j < 3; j++) { W[j] = (double)(float)out_lm[(size_t)ln_lm[i] * 4 + j]; W[3 + j] = (double)(float)out_lm[(size_t)(ln_lm[i] + 1) * 4 + j]; }   // through Converter::toCvMat (float)
        lns[i]->SetWorldPos(W); lns[i]->UpdateAverageDir();
    }
    for (size_t i = 0; i < pls.size(); i++) {
        cv::Mat C(4, 1, CV_32F);
        for (int j = 0; j < 4; j++) C.at<float>(j) = (float)out_lm[(size_t)pl_lm[i] * 4 + j];
        pls[i]->SetWorldPos(C); pls[i]->UpdateCoefficientsAndPoints();
    }
}
}  // namespace Planar_SLAM
#endif   // PLANAR_ADAPTERS_WITH_LOCAL_BA

