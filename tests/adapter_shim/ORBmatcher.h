// TEST INFRASTRUCTURE: stands where the reference's include/ORBmatcher.h would be (see adapter_decls.hpp)
#pragma once
#include "adapter_decls.hpp"
