// see ../core/core.hpp: cv::KeyPoint lives there in this stand-in
#pragma once
#include "../core/core.hpp"
