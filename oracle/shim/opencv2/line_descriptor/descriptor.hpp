// oracle/shim: cv::line_descriptor::KeyLine stand-in (TEST INFRASTRUCTURE): the record layout of opencv_contrib 3.4
#pragma once
#include "../../cvshim.hpp"
namespace cv { namespace line_descriptor {
struct KeyLine {
    float angle = 0;
    int class_id = -1, octave = 0;
    Point2f pt;
    float response = 0, size = 0;
    float startPointX = 0, startPointY = 0, endPointX = 0, endPointY = 0;
    float sPointInOctaveX = 0, sPointInOctaveY = 0, ePointInOctaveX = 0, ePointInOctaveY = 0;
    float lineLength = 0;
    int numOfPixels = 0;
    Point2f getStartPoint() const { return Point2f(startPointX, startPointY); }
    Point2f getEndPoint() const { return Point2f(endPointX, endPointY); }
};
static_assert(sizeof(KeyLine) == 68, "cv::line_descriptor::KeyLine layout");
} }
#ifdef CVSHIM_LSD
// LSDDetector / BinaryDescriptor stand-ins for the real src/LSDextractor.cpp: they forward to the restatement in
// oracle/lsd_oracle.cpp (these two classes ARE the un-vendored third-party part; only the wrapper around them is pinned).
#include <memory>
#include <vector>
#include "../../../../include/planar_abi.h"
namespace orc {
void lsd_detect_keylines(const uint8_t* img, int w, int h, int step, int tie_order, std::vector<planar_keyline>& out);
void lbd_compute_keylines(const uint8_t* img, int w, int h, int step, const std::vector<planar_keyline>& kls, uint8_t* desc);
extern int g_tie_order;
}
namespace cv {
template <typename T> struct Ptr : std::shared_ptr<T> { Ptr() {} Ptr(T* p) : std::shared_ptr<T>(p) {} };
namespace line_descriptor {
class LSDDetector {
public:
    static Ptr<LSDDetector> createLSDDetector() { return Ptr<LSDDetector>(new LSDDetector()); }
    void detect(const Mat& image, std::vector<KeyLine>& keylines, int scale, int numOctaves, const Mat& = Mat()) {
        assert(scale == 1 && numOctaves == 1 && image.type() == CV_8UC1);   // what the reference's float 1.2 / int 1 arrive as
        std::vector<planar_keyline> k;
        orc::lsd_detect_keylines(image.data, image.cols, image.rows, (int)(size_t)image.step, orc::g_tie_order, k);
        keylines.resize(k.size());
        if (!k.empty()) std::memcpy((void*)keylines.data(), k.data(), k.size() * sizeof(KeyLine));
    }
};
class BinaryDescriptor {
public:
    static Ptr<BinaryDescriptor> createBinaryDescriptor() { return Ptr<BinaryDescriptor>(new BinaryDescriptor()); }
    void compute(const Mat& image, std::vector<KeyLine>& keylines, Mat& descriptors) {
        if (keylines.empty()) return;   // "Error: keypoint list is empty" in the library
        std::vector<planar_keyline> k(keylines.size());
        std::memcpy((void*)k.data(), keylines.data(), k.size() * sizeof(KeyLine));
        descriptors = Mat((int)k.size(), 32, CV_8UC1);
        orc::lbd_compute_keylines(image.data, image.cols, image.rows, (int)(size_t)image.step, k, descriptors.data);
    }
};
} }
#endif
