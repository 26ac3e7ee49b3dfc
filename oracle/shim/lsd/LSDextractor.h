// oracle/shim/lsd/LSDextractor.h — TEST INFRASTRUCTURE.  Found before the reference's include/LSDextractor.h when the REAL
// src/LSDextractor.cpp is compiled into oracle/_ref/ref_lsd: the real header pulls in imgcodecs / highgui / levmar prototypes and
// dozens of unrelated structs; the translation unit itself only needs the class it defines (include/LSDextractor.h:344-352).
#pragma once
#include <opencv2/core/core.hpp>
#include <opencv2/line_descriptor/descriptor.hpp>
#include <eigen3/Eigen/Core>
#include "auxiliar.h"
namespace Planar_SLAM {
class LineSegment {
public:
    LineSegment();
    ~LineSegment() = default;
    void ExtractLineSegment(const cv::Mat& img, std::vector<cv::line_descriptor::KeyLine>& keylines, cv::Mat& ldesc,
                            std::vector<Eigen::Vector3d>& keylineFunctions, float scale = 1.2, int numOctaves = 1);
};
}  // namespace Planar_SLAM
