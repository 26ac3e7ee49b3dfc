// planar_adapters.hpp — drop-in C++ classes with the reference's own signatures over the C ABI (planar_abi.h).
//
// Build PlanarSLAM with these instead of src/ORBextractor.cc / src/PlaneExtractor.cpp and link libplanar_hip.so.
// Requires the OpenCV headers PlanarSLAM already uses (cv::Mat, cv::KeyPoint); nothing else.
//   Planar_SLAM::ORBextractor   <- include/ORBextractor.h:45-112, src/ORBextractor.cc
//   PlaneDetection              <- include/PlaneExtractor.h:36-56,  src/PlaneExtractor.cpp
// The matcher / optimizer entry points take Frame*; their gather/scatter glue is shown in INTEGRATION.md.
#pragma once
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>

#include <cassert>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "planar_abi.h"

namespace planar_adapter {
inline planar_ctx* shared_ctx() {   // one context per host thread (the ABI's re-entrancy rule)
    static thread_local planar_ctx* ctx = nullptr;
    if (!ctx && planar_ctx_create(&ctx, 0) != PLANAR_OK) throw std::runtime_error(planar_last_error());
    return ctx;
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
    ~ORBextractor() { if (orb_) planar_orb_destroy(orb_); }

    // Same contract as the reference operator() (src/ORBextractor.cc:1043-1105); mask is ignored there too.
    void operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors) {
        if (_image.empty()) return;
        cv::Mat image = _image.getMat();
        assert(image.type() == CV_8UC1);   // src/ORBextractor.cc:1050
        ensure(image.cols, image.rows);
        const int cap = planar_orb_max_keypoints(orb_);
        kps_.resize(cap);
        desc_.resize((size_t)cap * 32);
        int32_t n = 0;
        if (planar_orb_extract(orb_, image.data, 1, (int)image.step, (int64_t)image.step * image.rows, kps_.data(), desc_.data(), &n) != PLANAR_OK)
            throw std::runtime_error(planar_last_error());
        static_assert(sizeof(cv::KeyPoint) == sizeof(planar_keypoint), "cv::KeyPoint layout");
        _keypoints.resize(n);
        if (n) std::memcpy((void*)_keypoints.data(), kps_.data(), (size_t)n * sizeof(planar_keypoint));
        if (n == 0) { _descriptors.release(); }
        else {
            _descriptors.create(n, 32, CV_8U);
            cv::Mat d = _descriptors.getMat();
            for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), &desc_[(size_t)i * 32], 32);
        }
        for (int l = 0; l < nlevels; l++) {      // public member of the reference class (include/ORBextractor.h:85)
            int w, h;
            planar_orb_level_size(orb_, l, &w, &h);
            mvImagePyramid[l].create(h, w, CV_8UC1);
            planar_orb_read_level(orb_, 0, l, mvImagePyramid[l].data);
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
    void ensure(int w, int h) {
        if (orb_ && w == w_ && h == h_) return;
        if (orb_) planar_orb_destroy(orb_);
        planar_orb_params p{nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST};
        if (planar_orb_create(planar_adapter::shared_ctx(), &p, w, h, 1, &orb_) != PLANAR_OK) { orb_ = nullptr; throw std::runtime_error(planar_last_error()); }
        w_ = w; h_ = h;
    }
    std::vector<float> factors(int which) {
        // the scale tables depend only on the constructor arguments; a 64x64 plan is enough to read them
        if (!orb_) ensure(640, 480);
        std::vector<float> v[4];
        for (auto& x : v) x.resize(nlevels);
        planar_orb_get_scale_factors(orb_, v[0].data(), v[1].data(), v[2].data(), v[3].data());
        return v[which];
    }
    int nfeatures; double scaleFactor; int nlevels, iniThFAST, minThFAST;
    planar_orb* orb_ = nullptr;
    int w_ = 0, h_ = 0;
    std::vector<planar_keypoint> kps_;
    std::vector<uint8_t> desc_;
};

}  // namespace Planar_SLAM

// ---- PlaneDetection (include/PlaneExtractor.h:36-56).  Frame::ComputePlanes (src/Frame.cc:647-672) reads plane_num_,
// plane_vertices_[i], cloud.vertices[j] and plane_filter.extractedPlanes[i]->normal/center; the same members exist here.
struct PlanarPlaneSeg { double normal[3], center[3], mse; int N; };
struct PlanarPlaneFilter { std::vector<PlanarPlaneSeg*> extractedPlanes; std::vector<PlanarPlaneSeg> storage; };
struct PlanarVertex { double v[3]; double operator[](int i) const { return v[i]; } };
struct ImagePointCloud { std::vector<PlanarVertex> vertices; int w = 0, h = 0; };

class PlaneDetection {
public:
    ImagePointCloud cloud;
    PlanarPlaneFilter plane_filter;
    std::vector<std::vector<int>> plane_vertices_;
    cv::Mat seg_img_, color_img_;
    int plane_num_ = 0;

    ~PlaneDetection() { if (peac_) planar_peac_destroy(peac_); }
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
        if (!peac_ || W != w_ || H != h_) {
            if (peac_) planar_peac_destroy(peac_);
            if (planar_peac_create(planar_adapter::shared_ctx(), W, H, 1, &peac_) != PLANAR_OK) { peac_ = nullptr; throw std::runtime_error(planar_last_error()); }
            w_ = W; h_ = H;
        }
        std::vector<int32_t> labels((size_t)W * H);
        std::vector<double> planes((size_t)planar_peac_max_planes() * 8);
        int32_t n = 0;
        if (planar_peac_segment(peac_, (const uint16_t*)depth_.data, 1, (int)(depth_.step / 2), (int64_t)(depth_.step / 2) * H, fx_, fy_, cx_, cy_, factor_,
                                labels.data(), planes.data(), &n) != PLANAR_OK)
            throw std::runtime_error(planar_last_error());
        plane_num_ = n;
        plane_vertices_.assign(n, std::vector<int>());
        for (size_t i = 0; i < labels.size(); i++) if (labels[i] >= 0) plane_vertices_[labels[i]].push_back((int)i);   // raster order, as :362-372
        plane_filter.storage.resize(n); plane_filter.extractedPlanes.resize(n);
        for (int i = 0; i < n; i++) {
            PlanarPlaneSeg& s = plane_filter.storage[i];
            const double* p = &planes[(size_t)i * 8];
            s.N = (int)p[0]; for (int k = 0; k < 3; k++) { s.normal[k] = p[1 + k]; s.center[k] = p[4 + k]; } s.mse = p[7];
            plane_filter.extractedPlanes[i] = &s;
        }
        seg_img_ = cv::Mat(H, W, CV_8UC3);   // visualisation only (colours are out of scope)
    }

private:
    planar_peac* peac_ = nullptr;
    int w_ = 0, h_ = 0;
    cv::Mat depth_;
    float factor_ = 0, fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0;
};

// ---- LineSegment (reference include/LSDextractor.h:344-352, src/LSDextractor.cpp:12-39).  Opt-in: define PLANAR_ADAPTERS_WITH_LINES
//      before including this header in a build that has opencv_contrib's line_descriptor and Eigen (the types of the signature). ------
#ifdef PLANAR_ADAPTERS_WITH_LINES
#include <opencv2/line_descriptor/descriptor.hpp>
#include <eigen3/Eigen/Core>
namespace Planar_SLAM {
class LineSegment {
public:
    LineSegment() {}
    ~LineSegment() { if (lsd_) planar_lsd_destroy(lsd_); }
    LineSegment(const LineSegment&) = delete;
    LineSegment& operator=(const LineSegment&) = delete;

    // scale / numOctaves are accepted for signature compatibility; the reference passes 1.2f -> (int)1 and 1 (one octave, full resolution)
    void ExtractLineSegment(const cv::Mat& img, std::vector<cv::line_descriptor::KeyLine>& keylines, cv::Mat& ldesc,
                            std::vector<Eigen::Vector3d>& keylineFunctions, float scale = 1.2, int numOctaves = 1) {
        (void)scale; (void)numOctaves;
        static_assert(sizeof(cv::line_descriptor::KeyLine) == sizeof(planar_keyline), "cv::line_descriptor::KeyLine layout");
        static_assert(sizeof(Eigen::Vector3d) == 3 * sizeof(double), "Eigen::Vector3d layout");
        assert(img.type() == CV_8UC1);
        const int W = img.cols, H = img.rows;
        if (!lsd_ || W != w_ || H != h_) {
            if (lsd_) planar_lsd_destroy(lsd_);
            if (planar_lsd_create(planar_adapter::shared_ctx(), W, H, 1, &lsd_) != PLANAR_OK) { lsd_ = nullptr; throw std::runtime_error(planar_last_error()); }
            w_ = W; h_ = H;
        }
        const int lsdNFeatures = 40;                                   // src/LSDextractor.cpp:18
        keylines.resize(lsdNFeatures);
        std::vector<unsigned char> desc((size_t)lsdNFeatures * 32);
        const size_t first = keylineFunctions.size();                  // the reference push_backs
        keylineFunctions.resize(first + lsdNFeatures);
        int32_t n = 0;
        if (planar_lsd_extract(lsd_, img.data, 1, (int)img.step, (int64_t)img.step * H, lsdNFeatures, (planar_keyline*)keylines.data(), desc.data(),
                               (double*)(keylineFunctions.data() + first), &n) != PLANAR_OK)
            throw std::runtime_error(planar_last_error());
        keylines.resize(n);
        keylineFunctions.resize(first + n);
        if (n) { ldesc = cv::Mat(n, 32, CV_8UC1); std::memcpy(ldesc.data, desc.data(), (size_t)n * 32); }   // BinaryDescriptor::compute leaves ldesc untouched when there are no lines
    }

private:
    planar_lsd* lsd_ = nullptr;
    int w_ = 0, h_ = 0;
};
}  // namespace Planar_SLAM
#endif
