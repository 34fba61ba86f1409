// orbx_kernels.hpp -- host-callable launchers of the extraction kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/orbx.h"
#include "orbx_common.hpp"

namespace orbx {

// work items of the sliding-window kernels: (level, 64-dword column block, row band)
struct SlideMap {
    int base[kMaxLevels + 1];   // first item of each level; [kMaxLevels] = total
    int ndw[kMaxLevels];        // dword columns of the level image
    uint32_t rcp_ndw[kMaxLevels];   // ceil(2^32 / ndw): flat / ndw by one mul_hi
    int nrb[kMaxLevels];        // row bands per level
    int rows[kMaxLevels];       // rows per band (every band has this many; the last one is shifted up to fit)
};

// LDS layout of the per-cell FAST kernel, sized for the largest cell of the geometry
struct FastLds {
    int tile_pitch, score_pitch;   // bytes per row of the pixel tile / score tile
    int score_off, queue_off;      // byte offsets inside the dynamic LDS block
    int queue_cap;                 // entries of the pretest queue (multiple of 64)
    int total;                     // bytes of dynamic LDS per workgroup
    int fixed_pitch;               // 1: the pitches are the compile-time ones of the specialised kernel
};

// levels of a geometry whose quadtree runs on node counts (k_qt_bins + quadtree_counts_body inside the quadtree kernels)
struct CountLevels {
    int n;                      // how many
    int level[kMaxLevels];      // which
    int slot[kMaxLevels];       // level -> its place in level[] / depth[], -1 for the other levels
    int depth[kMaxLevels];      // depth of the leaves
    int max_cells;              // most FAST cells on one of them
    int max_blocks;             // most leaf blocks (workgroups of k_qt_bins) on one of them
    int min_ratio;              // a level takes this form only with at least min_ratio x nfeat candidates
    size_t stride;              // bytes of scratch per (frame, listed level)
};

struct ScaleTab { float v[kMaxLevels]; };

// Level 0 read from the caller's image instead of a copy of it in the pyramid block (resident batches): the FAST cells, the blur, the resize into level 1 and the orientation patches of level 0 take
// their pixels from `base`; only the blur ever needs pixels outside the image (3 of them, REFLECT_101), and forms them
// itself.  on == 0: every level is read from the pyramid block.
struct RawL0 {
    const uint8_t* base = nullptr;         // first frame
    const FastCell* cells0 = nullptr;      // the level-0 entries of the FAST cell table, addressed into the image
    const YTab2* ytab2_l1 = nullptr;       // the row table of level 1, addressed into the image
    unsigned long long frame_stride = 0;   // bytes between frames
    uint32_t frame_bytes = 0;              // h * stride: what a buffer descriptor of one frame covers
    int stride = 0;                        // bytes per image row
    int ncells0 = 0;                       // FAST cells of level 0 (the first cells of a frame)
    int on = 0;
    int any_alignment = 0;                 // base, strides or width are not multiples of 4: byte-addressed loads
};

int upload_constant_tables(const int* gauss7, const int* umax16, hipStream_t s);

void launch_pyr_level0(const uint8_t* in, int stride, size_t frame_stride, uint8_t* pyr, const FrameGeom& g,
                       int nframes, hipStream_t s);
void launch_pyr_resize(uint8_t* pyr, const FrameGeom& g, int level, const XTab* xtab, const YTab* ytab,
                       int nframes, hipStream_t s);
void launch_resize_slide(uint8_t* pyr, const FrameGeom& g, int level, const XTab4* xtab, const YTab2* ytab,
                         int nframes, hipStream_t s, const RawL0& raw = RawL0());
void make_slide_map(const FrameGeom& g, int target_rows, SlideMap& sm);
void launch_fast_blur(const uint8_t* pyr, uint8_t* blur, const FrameGeom& g, const FastLds& fl, const SlideMap& sm,
                      const FastCell* cells, uint32_t* cell_cand, int* cell_count, int nframes, hipStream_t s,
                      const RawL0& raw = RawL0());
void launch_blur_slide(const uint8_t* pyr, uint8_t* blur, const FrameGeom& g, const SlideMap& sm, int nframes,
                       hipStream_t s, const RawL0& raw = RawL0());
void make_fast_lds(const FrameGeom& g, FastLds& fl);
void launch_fast_cells_wave(const uint8_t* pyr, const FrameGeom& g, const FastLds& fl, const FastCell* cells, uint32_t* cell_cand,
                            int* cell_count, int nframes, hipStream_t s, const RawL0& raw = RawL0());
size_t quadtree_lds_bytes(int max_nodes);
int quadtree_prepare(int max_nodes);
void launch_quadtree(const FrameGeom& g, const uint32_t* cell_cand, const int* cell_count, int* cell_off,
                     uint32_t* key_xy, uint8_t* key_resp, uint16_t* key_node, void* kp_stage, int* level_count,
                     int* level_ncand, int nframes, unsigned char* gscratch, const CountLevels* cl, unsigned char* count_scratch,
                     hipStream_t s);
void quadtree_count_levels(const FrameGeom& g, CountLevels& cl);
size_t quadtree_scratch_stride(const FrameGeom& g);
void launch_orient_desc(const FrameGeom& g, const uint8_t* pyr, const uint8_t* blur, const void* kp_stage,
                        const int* level_count, orbx_keypoint* out_kps, uint8_t* out_desc, int cap, int* n_out,
                        int nframes, hipStream_t s, const RawL0& raw = RawL0());

// matching (orbx_match.hip)
void launch_match_bruteforce(const uint32_t* d_q, int Q, int q_stride, const int32_t* d_nq,
                             const uint32_t* d_t, int T, int t_stride, const int32_t* d_nt, int nbatch,
                             int tsplit, uint32_t* d_partial, uint32_t* d_expand, int32_t* d_best_idx, int32_t* d_best_dist,
                             int32_t* d_second_dist, hipStream_t s);
size_t match_expand_bytes(int q_stride, int T, int t_stride, int nbatch);   // room d_expand must have; 0: pass nullptr
int match_bruteforce_tsplit(int Q, int T, int nbatch);
// static device scratch of the entry point that allocates nothing (one array per GPU, orbx_match.hip)
constexpr size_t kMatchStaticScratchBytes = 4u << 20;
uint32_t* match_static_scratch(int device, hipError_t* err);
void launch_match_windowed(const uint32_t* d_q, int Q, const uint32_t* d_t, const int32_t* d_off,
                           const int32_t* d_idx, const int32_t* d_toct, int32_t* d_best_idx, int32_t* d_best_dist,
                           int32_t* d_second_dist, int32_t* d_best_lvl, int32_t* d_second_lvl, hipStream_t s);
void launch_hamming_csr(const uint32_t* d_q, int Q, const uint32_t* d_t, const int32_t* d_off,
                        const int32_t* d_idx, uint16_t* d_out, hipStream_t s);
// one launch of the row-band stereo search (device pointers; optional ones may be NULL)
struct StereoLaunch {
    const orbx_keypoint *d_kl, *d_kr;
    const uint32_t *d_dl, *d_dr;
    const int32_t *d_nl, *d_nr;
    const uint8_t* d_active;
    const float *d_min_d, *d_max_d;
    int32_t* d_listed;
    int32_t* d_rows;               // stereo_rows_ints(nbatch, stride, nrows) ints of scratch for the row ordering
    int32_t *d_best_idx, *d_best_dist;
    float *d_u_right, *d_depth;
    int32_t* d_nmatched;
    int nbatch, stride, nrows, th_orb_dist, init_outputs, median_cut;
    float mbf, mb, min_x;
};
constexpr int kStereoMaxRows = 16000;   // image rows the row-band search accepts (two ints of LDS per row)
void launch_stereo_match(const StereoLaunch& L, const float* scale8, hipStream_t s);
size_t stereo_rows_ints(int nbatch, int stride, int nrows);
size_t stereo_classic_ints(int nbatch, int stride, int nrows);

void launch_stereo_classic(const orbx_keypoint* d_kl, const uint32_t* d_dl, const int32_t* d_nl,
                           const orbx_keypoint* d_kr, const uint32_t* d_dr, const int32_t* d_nr,
                           int nbatch, int stride, const uint8_t* pyrL, const uint8_t* pyrR, const FrameGeom& g,
                           const float* scale8, const float* inv_scale8, float mbf, float mb,
                           int32_t* d_best_idx, int32_t* d_best_dist, float* d_u_right, float* d_depth,
                           int32_t* d_nmatched, int32_t* d_rows, hipStream_t s);

constexpr int kFrameGridCells = 64 * 48;   // FRAME_GRID_COLS x FRAME_GRID_ROWS, include/Frame.h:92-93
// camera model of Frame::UndistortKeyPoints: K, distortion coefficients, RR = P(:, 0:3) * R
struct UndistortModel {
    double fx, fy, cx, cy, ifx, ify;
    double k[12];
    double RR[9];
    int ndist, fisheye;
};
void launch_undistort(const orbx_keypoint* d_in, const int32_t* d_n, int n_fixed, int cap, int nframes,
                      const UndistortModel& m, orbx_keypoint* d_out, hipStream_t s);

void launch_grid_lists(const orbx_keypoint* d_kps, const uint32_t* d_desc, const int* d_cell_start, const int* d_cell_list,
                       int stride, int nbatch, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                       const int32_t* qmax, const uint32_t* qdesc, const int32_t* d_nq, int Q, int q_stride, float minX, float minY,
                       float invW, float invH, int32_t* ncand, const int32_t* cand_off, int32_t* cand_idx, uint16_t* cand_dist,
                       hipStream_t s);
void launch_grid_build(const orbx_keypoint* d_kps, const int32_t* d_n, int N, int stride, int nbatch, float minX,
                       float minY, float invW, float invH, int* d_cell_start, int* d_cell_list, hipStream_t s);
void launch_grid_search(const orbx_keypoint* d_kps, const uint32_t* d_desc, const int* d_cell_start,
                        const int* d_cell_list, int stride, int nbatch, const float* qx, const float* qy, const float* qr,
                        const int32_t* qmin, const int32_t* qmax, const uint32_t* qdesc, const int32_t* d_nq, int Q,
                        int q_stride, float minX, float minY, float invW, float invH, int32_t* best_idx,
                        int32_t* best_dist, int32_t* second_dist, int32_t* best_lvl, int32_t* second_lvl, int32_t* ncand,
                        const int32_t* cand_off, int32_t* cand_idx, hipStream_t s, uint16_t* cand_dist = nullptr);

} // namespace orbx
