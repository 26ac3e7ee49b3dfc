// oracle/shim: <cv.h> forwards to the minimal OpenCV stand-in (TEST INFRASTRUCTURE; see cvshim.hpp)
#include "cvshim.hpp"
