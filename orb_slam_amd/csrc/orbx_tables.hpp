// orbx_tables.hpp -- host-side parameter tables of the extractor.
#pragma once
#include <vector>
#include "orbx_common.hpp"

namespace orbx {

// Constructor tables of ORBextractor (src/ORBextractor.cc:967-1026).
struct Params {
    int nfeatures = 0, nlevels = 0, iniTh = 0, minTh = 0;
    double scaleFactor = 0;  // kept as double like the reference member (include/ORBextractor.h:163)
    float scale[kMaxLevels], inv_scale[kMaxLevels], sigma2[kMaxLevels], inv_sigma2[kMaxLevels];
    int nfeat[kMaxLevels];
    int umax[kHalfPatch + 2];
    // circular patch as a flat (du, dv) list, row by row (dv = -15..15)
    int8_t patch_du[kPatchPixels], patch_dv[kPatchPixels];
};

// returns false if the arguments are outside what the kernels support
bool build_params(Params& p, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh);

void level_size(const Params& p, int w, int h, int level, int* lw, int* lh);

// Geometry of every level for a w x h input, and the cv::resize tables.
// Returns 0 or a negative ORBX error code; msg explains a failure.
struct ResizeTables {
    std::vector<XTab> xtab;
    std::vector<YTab> ytab;
    std::vector<XTab4> xtab4;
    std::vector<YTab2> ytab2;
    std::vector<FastCell> fast_cells;   // [g.ncells]
};
int build_geometry(const Params& p, int w, int h, FrameGeom& g, ResizeTables& t, const char** msg);

// 7-tap Gaussian (sigma 2) as 8.8 fixed point: [18,34,49,55,49,34,18]
void gauss7_taps(int taps[7]);

} // namespace orbx
