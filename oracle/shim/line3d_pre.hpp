// oracle/shim/line3d_pre.hpp — TEST INFRASTRUCTURE.  What precedes the line ranges of the REAL reference that oracle/Makefile extracts into
// oracle/_ref/gen/line3d_extract.cpp (include/LSDextractor.h:57-132, 239-251; src/LineExtractor.cpp:278-292, 305-314, 1157-1470; src/Frame.cc:189-267).
#pragma once
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "cvshim.hpp"
#include "eigenshim.hpp"
#include "opencv2/line_descriptor/descriptor.hpp"

#define EPS (1e-10)   // include/LSDextractor.h:30
using namespace std;
using namespace cv;
using namespace cv::line_descriptor;
typedef Eigen::Matrix<double, 6, 1> Vector6d;   // include/auxiliar.h
