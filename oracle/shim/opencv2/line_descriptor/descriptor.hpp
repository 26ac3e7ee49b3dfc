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
