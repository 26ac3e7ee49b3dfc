// oracle/shim/line3d_post.hpp — TEST INFRASTRUCTURE.  Follows the extracted RandomPoint3d / RandomLine3d definitions: prototypes (the reference declares them in
// include/LSDextractor.h, which cannot be included as a whole), a FrameLine with the fields Frame::isLineGood fills, and a data-holder Frame.
#pragma once
cv::Point3d projectPt3d2Ln3d(const cv::Point3d& P, const cv::Point3d& mid, const cv::Point3d& drct);
cv::Mat array2mat(double a[], int n);
cv::Point3d mat2cvpt3d(cv::Mat m);
void computeLine3d_svd(const vector<RandomPoint3d>& pts, const vector<int>& idx, cv::Point3d& mean, cv::Point3d& drct);
double depthStdDev(double d);
RandomPoint3d compPt3dCov(cv::Point3d pt, cv::Mat K, double time_diff_sec);
RandomLine3d extract3dline_mahdist(const vector<RandomPoint3d>& pts);
bool verify3dLine(const vector<RandomPoint3d>& pts, const cv::Point3d& A, const cv::Point3d& B);
double mah_dist3d_pt_line(const RandomPoint3d& pt, const cv::Point3d& q1, const cv::Point3d& q2);

struct FrameLine {   // include/LSDextractor.h:141-183, the members isLineGood writes
    cv::Point2d p, q;
    cv::Point3d direction, direct1, direct2;
    bool haveDepth = false;
    std::vector<RandomPoint3d> rndpts3d;
};

namespace Planar_SLAM {
class Frame {
public:
    std::vector<float> mvDepthLine;
    std::vector<Vector6d> mvLines3D;
    std::vector<KeyLine> mvKeylinesUn;
    std::vector<FrameLine> mVF3DLines;
    static float cx, cy, invfx, invfy;
    void isLineGood(const cv::Mat& imGray, const cv::Mat& imDepth, cv::Mat K);
};
}  // namespace Planar_SLAM
