// oracle/shim: forwards to the minimal OpenCV stand-in (TEST INFRASTRUCTURE)
#include "../cvshim.hpp"
