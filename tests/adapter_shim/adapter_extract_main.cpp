// tests/adapter_shim/adapter_extract_main.cpp — TEST INFRASTRUCTURE.
// The way the reference's Frame constructor uses the three extractors (src/Frame.cc:90-97): three std::threads per frame, Frames (and
// the PlaneDetection inside) copied by value, LineSegment called without a usable object - here through include/planar_adapters.hpp.
//   adapter_extract <in.bin> <out.bin>    in: int32 W, H, nframes; per frame gray u8 [H*W], depth u16 [H*W]
//   out per frame: int32 nkp, kps (28 B each), desc; int32 nlines, keylines (68 B each), ldesc, eqs (3 doubles each);
//                  int32 nplanes, per plane {int32 npix; double normal[3], center[3]}, labels int32 [H*W] (from plane_vertices_);
//                  int32 mnPlaneNum, per kept plane {float coef[4]; int32 npts; float xyz[npts][3]} (the Frame::ComputePlanes loop, on the COPY)
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#define PLANAR_ADAPTERS_WITH_LINES
#include "planar_adapters.hpp"

struct CloudLike { struct P { float x, y, z; }; std::vector<P> points; };   // stands in for pcl::PointCloud<pcl::PointXYZRGB>

struct FrameLike {   // the members of Planar_SLAM::Frame the extraction threads write
    std::vector<cv::KeyPoint> mvKeys;
    cv::Mat mDescriptors, mLdesc;
    std::vector<cv::line_descriptor::KeyLine> mvKeylinesUn;
    std::vector<Eigen::Vector3d> mvKeyLineFunctions;
    PlaneDetection planeDetector;
    std::vector<CloudLike> mvPlanePoints;
    std::vector<cv::Mat> mvPlaneCoefficients;
};

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    FILE* fi = std::fopen(argv[1], "rb");
    FILE* fo = std::fopen(argv[2], "wb");
    if (!fi || !fo) return 2;
    int32_t hdr[3];
    if (std::fread(hdr, 4, 3, fi) != 3) return 2;
    const int W = hdr[0], H = hdr[1], NF = hdr[2];
    Planar_SLAM::ORBextractor* orb = new Planar_SLAM::ORBextractor(1000, 1.2f, 8, 20, 7);
    cv::Mat K(3, 3, CV_32F);
    K.at<float>(0, 0) = 535.4f; K.at<float>(1, 1) = 539.2f; K.at<float>(0, 2) = 320.1f; K.at<float>(1, 2) = 247.6f;
    std::vector<FrameLike> history;   // copies, as mLastFrame = Frame(mCurrentFrame)
    for (int f = 0; f < NF; f++) {
        cv::Mat gray(H, W, CV_8UC1), depth(H, W, CV_16U);
        if (std::fread(gray.data, 1, (size_t)W * H, fi) != (size_t)W * H) return 2;
        if (std::fread(depth.data, 2, (size_t)W * H, fi) != (size_t)W * H) return 2;
        FrameLike F;
        Planar_SLAM::LineSegment* mpLineSegment = nullptr;   // include/Frame.h:123 - never initialised by the reference
        std::thread threadLines([&] { mpLineSegment->ExtractLineSegment(gray, F.mvKeylinesUn, F.mLdesc, F.mvKeyLineFunctions); });
        std::thread threadPoints([&] { (*orb)(gray, cv::Mat(), F.mvKeys, F.mDescriptors); });
        std::thread threadPlanes([&] {        // Frame::ComputePlanes
            F.planeDetector.readDepthImage(depth, K, 1.0f / 5000); F.planeDetector.runPlaneDetection(H, W);
            F.planeDetector.ComputePlaneClouds(0.05, F.mvPlanePoints, F.mvPlaneCoefficients);
        });
        threadPoints.join(); threadLines.join(); threadPlanes.join();
        history.push_back(F);                 // by-value copy
        const FrameLike& G = history.back();  // everything below reads the COPY
        int32_t n = (int32_t)G.mvKeys.size();
        std::fwrite(&n, 4, 1, fo);
        std::fwrite(G.mvKeys.data(), sizeof(cv::KeyPoint), n, fo);
        for (int i = 0; i < n; i++) std::fwrite(G.mDescriptors.ptr(i), 1, 32, fo);
        n = (int32_t)G.mvKeylinesUn.size();
        std::fwrite(&n, 4, 1, fo);
        std::fwrite(G.mvKeylinesUn.data(), sizeof(cv::line_descriptor::KeyLine), n, fo);
        for (int i = 0; i < n; i++) std::fwrite(G.mLdesc.ptr(i), 1, 32, fo);
        std::fwrite(G.mvKeyLineFunctions.data(), 24, n, fo);
        n = G.planeDetector.plane_num_;
        std::fwrite(&n, 4, 1, fo);
        std::vector<int32_t> labels((size_t)W * H, -1);
        for (int i = 0; i < n; i++) {
            const int32_t npix = (int32_t)G.planeDetector.plane_vertices_[i].size();
            std::fwrite(&npix, 4, 1, fo);
            std::fwrite(G.planeDetector.plane_filter.extractedPlanes[i]->normal, 8, 3, fo);
            std::fwrite(G.planeDetector.plane_filter.extractedPlanes[i]->center, 8, 3, fo);
            for (int v : G.planeDetector.plane_vertices_[i]) labels[v] = i;
        }
        std::fwrite(labels.data(), 4, labels.size(), fo);
        n = (int32_t)G.mvPlanePoints.size();
        std::fwrite(&n, 4, 1, fo);
        for (int i = 0; i < n; i++) {
            for (int t = 0; t < 4; t++) { const float c = G.mvPlaneCoefficients[i].at<float>(t, 0); std::fwrite(&c, 4, 1, fo); }
            const int32_t np = (int32_t)G.mvPlanePoints[i].points.size();
            std::fwrite(&np, 4, 1, fo);
            std::fwrite(G.mvPlanePoints[i].points.data(), 12, np, fo);
        }
    }
    history.clear();
    delete orb;
    std::fclose(fi); std::fclose(fo);
    return 0;
}
