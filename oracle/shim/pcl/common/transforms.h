#pragma once
#include "../point_types.h"
