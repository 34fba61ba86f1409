// orbx_common.hpp -- geometry and parameter tables shared by host code and kernels.
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace orbx {

constexpr int kMaxLevels = 8;          // pyramid levels supported by one launch descriptor
constexpr int kEdge = 19;              // EDGE_THRESHOLD, src/ORBextractor.cc:72
constexpr int kHalfPatch = 15;         // HALF_PATCH_SIZE, src/ORBextractor.cc:71
constexpr int kPatch = 31;             // PATCH_SIZE, src/ORBextractor.cc:70
constexpr int kDetBorder = 16;         // minBorderX/Y = EDGE_THRESHOLD-3, src/ORBextractor.cc:1036
constexpr int kPadX = 32;              // left/right padding of a pyramid plane row (>= kEdge, keeps rows dword-friendly)
constexpr int kPadY = kEdge;           // top/bottom padding of a pyramid plane
constexpr int kMaxNodes = 2048;        // quadtree nodes one workgroup can hold in LDS
constexpr int kMaxNodesBig = 16384;    // ... and in its slice of global memory (14-bit node index)
constexpr int kPatchPixels = 749;      // pixels of the circular orientation patch

// One pyramid level of one frame geometry (identical for every frame of a batch).
struct LevelGeom {
    int w, h;              // level image size
    int pitch;             // bytes per plane row (w + 2*kPadX rounded up to 64)
    int rows;              // plane rows (h + 2*kPadY)
    uint32_t plane_off;    // byte offset of the plane inside one frame's pyramid block
    // detection region [16, w-16) x [16, h-16) and its cell grid (src/ORBextractor.cc:1036-1050)
    int width, height;     // region size; 0 if the level is too small to be searched
    int nCols, nRows, wCell, hCell;
    uint32_t rcp_ncols;    // ceil(2^32 / nCols) for nCols > 1: cell / nCols = umulhi(cell, rcp_ncols)
    int cell_base;         // index of this level's first cell in the per-frame cell array
    int cell_cap;          // candidate slots per cell
    uint32_t cand_base;    // first candidate slot of this level inside one frame (cells * cell_cap)
    uint32_t key_base;     // first compacted key of this level inside one frame
    uint32_t key_cap;      // keys this level can hold (= ncells * cell_cap)
    // quadtree (src/ORBextractor.cc:488-512)
    int nfeat;             // N = mnFeaturesPerLevel[level]
    int nIni;
    float hX;
    int kp_base;           // first retained-keypoint slot of this level inside one frame
    int kp_cap;            // retained-keypoint slots: max(N+3, 4*nIni)
    // resize tables of this level (index into the table buffers), level > 0 only
    uint32_t xtab_off;     // per destination column: xofs (u16) | pad, alpha0 (i16), alpha1 (i16)
    uint32_t ytab_off;     // per destination row: y0 (u16), y1 (u16), beta0 (i16), beta1 (i16)
    uint32_t xtab4_off;    // per destination plane dword column (XTab4), border columns included
    uint32_t ytab2_off;    // per destination plane row (YTab2), border rows included
    float scale;           // mvScaleFactor[level]
    float patch_size;      // (int)(31 * scale) as float
};

struct FrameGeom {
    int nlevels;
    int iniTh, minTh;
    int ncells;                 // cells per frame, all levels
    uint32_t pyr_frame_bytes;   // bytes of one frame's pyramid block
    uint32_t cand_per_frame;    // candidate slots per frame
    uint32_t keys_per_frame;    // compacted key slots per frame
    int kp_per_frame;           // retained-keypoint slots per frame
    int max_nodes;              // largest kp_cap+4 over levels (quadtree LDS sizing)
    int resize_slide_ok;        // 1 if every XTab4 window fits in 8 source bytes
    LevelGeom lv[kMaxLevels];
};

struct XTab { uint16_t xofs; uint16_t pad; int16_t a0, a1; };   // 8 bytes
struct YTab { uint16_t y0, y1; int16_t b0, b1; };                // 8 bytes
// Four adjacent destination columns of one plane dword: the 8 source bytes at `base` (relative to
// source image column 0) hold every pixel they need; sel[k] is a v_perm_b32 selector that yields
// (left pixel, right pixel) of column k as two zero-extended u16, a01[k] their weights as two u16.
struct XTab4 { uint32_t base; uint32_t sel[4]; uint32_t a01[4]; };   // 36 bytes
// One destination plane row: byte offsets of its two source rows inside the source plane
// (pointing at source image column 0) and the vertical weights.
struct YTab2 { uint32_t off0, off1; int16_t b0, b1; };               // 12 bytes

// Everything the per-cell FAST kernel needs to know about one cell of one frame geometry, so
// that a wavefront starts with one 32-byte scalar load instead of a level search and divisions.
struct FastCell {
    uint32_t src_off;    // byte offset inside a frame of the tile's first dword (3-pixel ring included)
    uint32_t cand_off;   // first candidate slot of this cell inside a frame
    uint32_t rcp_cw;     // ceil(2^18 / cw): i / cw for i < 2^12
    uint32_t pad32;
    uint16_t pitch;      // plane pitch
    uint16_t cell_cap;   // candidate slots
    uint8_t cw, ch;      // tested area (0: empty cell)
    uint8_t ox;          // bytes between the tile's first dword and its first pixel
    uint8_t ndw;         // dwords per tile row
    uint8_t lpr;         // lanes per cell row in the pretest: ceil(cw / 4), four pixels per lane
    uint8_t fast_rows;   // cell rows one 64-lane round of the pretest covers: 64 / lpr
    uint8_t pad8[2];
    uint32_t rcp_lpr;    // ceil(2^16 / lpr): lane / lpr = lane * rcp_lpr >> 16 for lane < 64
};
static_assert(sizeof(FastCell) == 32, "one s_load_dwordx8");

} // namespace orbx
