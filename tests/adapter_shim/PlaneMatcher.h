// TEST INFRASTRUCTURE: stands where the reference's include/PlaneMatcher.h would be (see adapter_decls.hpp)
#pragma once
#include "adapter_decls.hpp"
