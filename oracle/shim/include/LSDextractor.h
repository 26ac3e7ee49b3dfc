// oracle/shim/include/LSDextractor.h — TEST INFRASTRUCTURE, not product code.
// include/peac/AHCPlaneFitter.hpp:49 includes "include/LSDextractor.h" only for the SurfaceNormal
// type of an unused member (surfaceNormals).  The real header drags in opencv_contrib line_descriptor;
// this stand-in (found first on the include path of the oracle/_ref build) declares just that type.
#pragma once
#include <opencv2/core/core.hpp>
class SurfaceNormal {
public:
    cv::Point3f normal;
    cv::Point3f cameraPosition;
    cv::Point2i FramePosition;
    SurfaceNormal() {}
};
