// TEST INFRASTRUCTURE: stands where the reference's include/LSDmatcher.h would be (see adapter_decls.hpp)
#pragma once
#include "adapter_decls.hpp"
