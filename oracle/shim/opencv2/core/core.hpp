// oracle/shim: forwards to the minimal OpenCV stand-in (TEST INFRASTRUCTURE; see ../cvshim.hpp)
#include "../../cvshim.hpp"
#ifdef CVSHIM_FILESTORAGE
#include "../../filestorage_stub.hpp"
#endif
