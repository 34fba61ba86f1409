// orbx_api.cpp -- C ABI of liborbx (see include/orbx.h): context, device memory,
// launch sequences.  Host code only; every result is computed by the kernels in
// orbx_kernels.hip / orbx_match.hip.  There is no CPU path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbx.h"
#include "orbx_common.hpp"
#include "orbx_kernels.hpp"
#include "orbx_tables.hpp"

using namespace orbx;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need)
    {
        if (need <= bytes) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; bytes = 0; if (e != hipSuccess) return e; }
        // grow geometrically so that slowly growing batches do not reallocate every call
        size_t want = need + need / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        bytes = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

enum Stage { ST_PYR0 = 0, ST_RESIZE, ST_FAST, ST_BLUR, ST_QUADTREE, ST_DESC, ST_COUNT };   // launch order
const char* kStageNames[ST_COUNT] = {"pyr_level0", "pyr_resize", "fast_cells", "blur", "quadtree", "orient_desc"};
constexpr int kEventSets = 64;
// stage-timing events only order work on this device: no system-scope cache flush per event
constexpr unsigned kStageEventFlags = hipEventDisableSystemFence;

struct EventSet {
    hipEvent_t start[ST_COUNT], stop[ST_COUNT];
    bool pending = false;
    bool chained = false;   // stages ran back to back on one stream: stop[i-1] doubles as start[i]
    bool only_fast = false; // only the FAST (+ blur) launch was bracketed
};

} // namespace

struct orbx_ctx {
    Params p;
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t aux_stream = nullptr;   // the blur runs here, beside FAST + quadtree
    hipEvent_t ev_pyr = nullptr, ev_blur = nullptr;
    std::vector<hipEvent_t> ev_up;      // one per chunk of a pipelined host-fed batch
    bool aux_busy = false;              // uploads were queued on aux_stream since the last finish()
    std::string err;
    uint32_t* static_scratch = nullptr;   // this device's copy of the static scratch array (orbx_match.hip)

    // geometry of the current frame size
    int gw = 0, gh = 0;
    FrameGeom g;
    SlideMap blur_map, blur_map_batch;   // row bands of the blur: short for a few frames (latency), longer for batches
    FastLds fast_lds;
    int mk_w = -1, mk_h = -1, mk_val = 0;   // last orbx_max_keypoints answer
    // pinned staging arena of the host-buffer entry points
    struct PinPending { void* dst; size_t off, bytes; };
    uint8_t* pin = nullptr;
    size_t pin_cap = 0, pin_used = 0;
    std::vector<PinPending> pin_pending;
    DevBuf xtab, ytab, xtab4, ytab2, fast_cells;
    // Level 0 straight from the caller's image (RawL0, orbx_kernels.hpp): the tables that depend on the image's row
    // stride, built from these host copies when a stride is first seen; and the image the level-0 plane of the
    // pyramid block is still owed from (filled in by materialise_level0 when somebody reads that plane).
    std::vector<FastCell> h_cells0;
    std::vector<YTab2> h_ytab2_l1;
    DevBuf raw_cells0, raw_ytab2;
    int raw_stride = -1;
    struct Level0Owed { const uint8_t* img = nullptr; int stride = 0, nframes = 0; size_t frame_stride = 0; } l0_owed;

    // per-batch working set
    int cap_frames = 0;
    DevBuf pyr, blur, cell_cand, cell_count, cell_off, key_xy, key_resp, key_node, kp_stage, level_count, level_ncand, qt_scratch;
    DevBuf qt_counts;               // quadtree on node counts: per-launch scratch (leaves, sums and best keys)
    // The launch sequence of a small host-fed call, recorded once as a HIP graph and replayed: a lone frame is a
    // chain of a dozen dependent launches of a few microseconds each, and issuing them one by one leaves the
    // device waiting for the host.  Valid for one set of arguments (`graph_key`); rebuilt when any of them changes.
    struct GraphKey {
        const void *base = nullptr, *stream = nullptr;
        const void *pyr = nullptr, *blur = nullptr, *cand = nullptr, *kp_stage = nullptr, *counts = nullptr;
        int nframes = 0, w = 0, h = 0, cap = 0;
        unsigned long long switches = 0;      // the diagnostic switches that run_batch reads per call
        bool operator==(const GraphKey& o) const
        {
            return switches == o.switches && base == o.base && stream == o.stream && pyr == o.pyr && blur == o.blur && cand == o.cand &&
                   kp_stage == o.kp_stage && counts == o.counts && nframes == o.nframes && w == o.w && h == o.h && cap == o.cap;
        }
    };
    GraphKey graph_key, graph_seen;
    hipGraphExec_t graph_exec = nullptr;
    bool graph_broken = false;      // a capture or an instantiation failed once: plain launches from then on
    CountLevels count_levels{};

    // staging for the host-buffer entry points
    DevBuf in, out_kps, out_desc, out_n;
    // scratch for matching
    DevBuf partial, match_expand, m0, m1, m2, m3, m4, m5, m6, m7, m8, m9;
    DevBuf grid_start, grid_list, gq[12], stereo_listed, stereo_rows;

    int last_nframes = 0;
    bool have_pyramid = false, have_features = false;
    uint8_t* h_pyr = nullptr;      // pinned host mirror of one frame's pyramid (orbx_pyramid_level)
    size_t h_pyr_bytes = 0;
    int h_pyr_frame = -1;

    // stage timing
    EventSet evs[kEventSets];
    bool events_ok = false;
    int stage_timing = 0;          // 0 off, 1 every stage, 2 only the FAST (+ blur) launch; events cost about 10 us per call: opt-in
    int ev_next = 0;
    double stage_ms[ST_COUNT] = {0};
    int stage_launches[ST_COUNT] = {0};
};

namespace {

int fail(orbx_ctx* c, int code, const char* what)
{
    if (c) c->err = what;
    return code;
}

int fail_hip(orbx_ctx* c, const char* what, hipError_t e)
{
    if (c) {
        c->err = std::string(what) + ": " + hipGetErrorString(e);
    }
    return ORBX_ERR_HIP;
}

#define HIPCHECK(c, x)                                              \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) return fail_hip((c), #x, e_);         \
    } while (0)

void harvest(orbx_ctx* c, EventSet& es)
{
    if (!es.pending) return;
    if (es.only_fast) {
        (void)hipEventSynchronize(es.stop[ST_FAST]);
        float ms = 0;
        if (hipEventElapsedTime(&ms, es.start[ST_FAST], es.stop[ST_FAST]) == hipSuccess) {
            c->stage_ms[ST_FAST] += ms;
            c->stage_launches[ST_FAST] += 1;
        }
        es.pending = false;
        return;
    }
    (void)hipEventSynchronize(es.stop[ST_DESC]);
    (void)hipEventSynchronize(es.stop[ST_BLUR]);
    for (int s = 0; s < ST_COUNT; s++) {
        float ms = 0;
        hipEvent_t from = (es.chained && s > 0) ? es.stop[s - 1] : es.start[s];
        if (hipEventElapsedTime(&ms, from, es.stop[s]) == hipSuccess) {
            c->stage_ms[s] += ms;
            c->stage_launches[s] += 1;
        }
    }
    es.pending = false;
}

void drop_graph(orbx_ctx* c);

// A recorded launch sequence (run_small_batch) holds the addresses of the geometry tables and of every per-batch
// buffer: whoever frees or re-sizes one of them drops the recording, whatever entry point the change came through.
int ensure_geometry(orbx_ctx* c, int w, int h)
{
    if (c->gw == w && c->gh == h) return ORBX_OK;
    drop_graph(c);
    FrameGeom g;
    ResizeTables rt;
    std::vector<XTab>& xt = rt.xtab;
    std::vector<YTab>& yt = rt.ytab;
    const char* msg = "";
    int rc = build_geometry(c->p, w, h, g, rt, &msg);
    if (rc != ORBX_OK) return fail(c, rc, msg);
    if (quadtree_prepare(g.max_nodes) != 0) return fail(c, ORBX_ERR_UNSUPPORTED, "quadtree needs more LDS than one CU has");
    HIPCHECK(c, hipStreamSynchronize(c->stream));   // old tables may still be in use
    HIPCHECK(c, c->xtab.ensure(xt.size() * sizeof(XTab)));
    HIPCHECK(c, c->ytab.ensure(yt.size() * sizeof(YTab)));
    HIPCHECK(c, hipMemcpyAsync(c->xtab.p, xt.data(), xt.size() * sizeof(XTab), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(c->ytab.p, yt.data(), yt.size() * sizeof(YTab), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, c->xtab4.ensure(rt.xtab4.size() * sizeof(XTab4)));
    HIPCHECK(c, c->ytab2.ensure(rt.ytab2.size() * sizeof(YTab2)));
    HIPCHECK(c, hipMemcpyAsync(c->xtab4.p, rt.xtab4.data(), rt.xtab4.size() * sizeof(XTab4), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(c->ytab2.p, rt.ytab2.data(), rt.ytab2.size() * sizeof(YTab2), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, c->fast_cells.ensure(rt.fast_cells.size() * sizeof(FastCell) + 32));
    if (!rt.fast_cells.empty())
        HIPCHECK(c, hipMemcpyAsync(c->fast_cells.p, rt.fast_cells.data(), rt.fast_cells.size() * sizeof(FastCell), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));   // the tables are stack-owned
    c->g = g;
    {   // what the stride-dependent tables of RawL0 are made from: level 0's FAST cells, level 1's row table
        const int n0 = g.lv[0].width > 0 ? g.lv[0].nCols * g.lv[0].nRows : 0;
        c->h_cells0.assign(rt.fast_cells.begin(), rt.fast_cells.begin() + n0);
        c->h_ytab2_l1.clear();
        if (g.nlevels > 1 && g.resize_slide_ok)
            c->h_ytab2_l1.assign(rt.ytab2.begin() + g.lv[1].ytab2_off, rt.ytab2.begin() + g.lv[1].ytab2_off + g.lv[1].rows);
        c->raw_stride = -1;
        c->l0_owed = orbx_ctx::Level0Owed{};
    }
    // short bands: the kernel is bound by the latency of its dependent row groups, and many short
    // wavefronts hide it better than few long ones (measured optimum 12 rows, 6 of warm-up included)
    make_slide_map(c->g, getenv("ORBX_BLUR_ROWS") ? atoi(getenv("ORBX_BLUR_ROWS")) : 12, c->blur_map);
    // in a batch the FAST wavefronts of the same launch hide the blur's load latency, and every band pays six
    // warm-up rows of horizontal sums: longer bands there (measured: 24 rows, 2 % off the fused launch)
    make_slide_map(c->g, getenv("ORBX_BLUR_ROWS") ? atoi(getenv("ORBX_BLUR_ROWS")) : 24, c->blur_map_batch);
    make_fast_lds(c->g, c->fast_lds);
    quadtree_count_levels(c->g, c->count_levels);
    c->gw = w;
    c->gh = h;
    c->cap_frames = 0;   // per-frame sizes changed
    c->have_pyramid = c->have_features = false;
    return ORBX_OK;
}

int ensure_frames(orbx_ctx* c, int nframes)
{
    if (nframes <= c->cap_frames) return ORBX_OK;
    const FrameGeom& g = c->g;
    const size_t n = (size_t)nframes;
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    drop_graph(c);
    HIPCHECK(c, c->pyr.ensure(n * g.pyr_frame_bytes + 256));
    HIPCHECK(c, c->blur.ensure(n * g.pyr_frame_bytes + 256));
    HIPCHECK(c, c->cell_cand.ensure((n * g.cand_per_frame + 64) * 4));
    HIPCHECK(c, c->cell_count.ensure((n * g.ncells + 64) * 4));
    HIPCHECK(c, c->cell_off.ensure((n * g.ncells + 64) * 4));
    HIPCHECK(c, c->key_xy.ensure((n * g.keys_per_frame + 64) * 4));
    HIPCHECK(c, c->key_resp.ensure(n * g.keys_per_frame + 64));
    HIPCHECK(c, c->key_node.ensure((n * g.keys_per_frame + 64) * 2));
    HIPCHECK(c, c->kp_stage.ensure((n * (size_t)g.kp_per_frame + 64) * 8));
    HIPCHECK(c, c->level_count.ensure(n * kMaxLevels * 4));
    HIPCHECK(c, c->level_ncand.ensure(n * kMaxLevels * 4));
    HIPCHECK(c, c->qt_scratch.ensure(n * kMaxLevels * quadtree_scratch_stride(g)));   // only for levels beyond 2048 nodes
    // the blurred planes are only written inside the level images; give the rest a defined value once
    HIPCHECK(c, hipMemsetAsync(c->blur.p, 0, c->blur.bytes, c->stream));
    c->cap_frames = nframes;
    return ORBX_OK;
}

// the launch sequence; everything asynchronous on c->stream
// f0: index of the first of these frames inside the context's per-frame buffers (a batch fed from the
// host runs in chunks; the caller has sized the buffers for all of them before the first chunk)
// src_on_device: d_imgs is device memory (a resident batch, or the upload buffer of a host-fed one): level 0 can then be
// read where it lies.  Page-locked host memory the device reads over the bus is copied into the pyramid block once instead.
int run_batch(orbx_ctx* c, const uint8_t* d_imgs, int nframes, int w, int h, int stride, size_t frame_stride,
              orbx_keypoint* d_kps, uint8_t* d_desc, int cap, int32_t* d_n_out, bool pyramid_only, int f0 = 0,
              bool src_on_device = false)
{
    int rc = ensure_geometry(c, w, h);
    if (rc != ORBX_OK) return rc;
    rc = ensure_frames(c, f0 + nframes);
    if (rc != ORBX_OK) return rc;
    RawL0 raw;
    {
        // ORBX_NO_RAW0: always copy level 0 into the pyramid block (read once; at every call in a process started with
        // ORBX_SWITCHES_PER_CALL, as the tests are)
        static const bool per_call = getenv("ORBX_SWITCHES_PER_CALL") != nullptr;
        static const bool no_raw_at_start = getenv("ORBX_NO_RAW0") != nullptr;
        const bool no_raw = per_call ? getenv("ORBX_NO_RAW0") != nullptr : no_raw_at_start;
        const FrameGeom& g0 = c->g;
        const bool aligned = (((uintptr_t)d_imgs | (uintptr_t)frame_stride | (uintptr_t)stride | (uintptr_t)w) & 3u) == 0;
        // every consumer of level 0 has its image form only on the ordinary path: sliding-window resize, some FAST cells
        // on level 0, an image tall and wide enough for a single reflection, offsets that fit 32 bits and 16 bits
        if (src_on_device && !no_raw && !pyramid_only && g0.resize_slide_ok && g0.nlevels > 1 && !c->h_cells0.empty() &&
            w >= 64 && h >= 64 && stride <= 65532 && (size_t)stride * (size_t)h < ((size_t)1 << 31)) {
            if (c->raw_stride != stride) {
                // the tables addressed into an image with this row stride (plane offsets -> image offsets)
                const LevelGeom& L0 = g0.lv[0];
                std::vector<FastCell> cells = c->h_cells0;
                for (FastCell& fc : cells) {
                    if (fc.cw == 0) continue;
                    const uint32_t rel = fc.src_off - L0.plane_off;
                    const uint32_t prow = rel / (uint32_t)L0.pitch, pcol = rel - prow * (uint32_t)L0.pitch;
                    fc.src_off = (prow - (uint32_t)kPadY) * (uint32_t)stride + (pcol - (uint32_t)kPadX);
                    fc.pitch = (uint16_t)stride;
                }
                std::vector<YTab2> yt = c->h_ytab2_l1;
                for (YTab2& y : yt) {
                    y.off0 = ((y.off0 - (uint32_t)kPadX) / (uint32_t)L0.pitch - (uint32_t)kPadY) * (uint32_t)stride;
                    y.off1 = ((y.off1 - (uint32_t)kPadX) / (uint32_t)L0.pitch - (uint32_t)kPadY) * (uint32_t)stride;
                }
                HIPCHECK(c, hipStreamSynchronize(c->stream));          // earlier launches may still read the old tables
                HIPCHECK(c, c->raw_cells0.ensure(cells.size() * sizeof(FastCell) + 32));
                HIPCHECK(c, c->raw_ytab2.ensure(yt.size() * sizeof(YTab2)));
                HIPCHECK(c, hipMemcpy(c->raw_cells0.p, cells.data(), cells.size() * sizeof(FastCell), hipMemcpyHostToDevice));
                HIPCHECK(c, hipMemcpy(c->raw_ytab2.p, yt.data(), yt.size() * sizeof(YTab2), hipMemcpyHostToDevice));
                c->raw_stride = stride;
            }
            raw.base = d_imgs;
            raw.cells0 = c->raw_cells0.as<FastCell>();
            raw.ytab2_l1 = c->raw_ytab2.as<YTab2>();
            raw.frame_stride = frame_stride;
            raw.frame_bytes = (uint32_t)((size_t)stride * (size_t)h);
            raw.stride = stride;
            raw.ncells0 = (int)c->h_cells0.size();
            raw.on = 1;
            raw.any_alignment = aligned ? 0 : 1;
        }
    }
    const FrameGeom& g = c->g;
    const size_t F = (size_t)f0;
    uint8_t* const pyr = c->pyr.as<uint8_t>() + F * g.pyr_frame_bytes;
    uint8_t* const blur = c->blur.as<uint8_t>() + F * g.pyr_frame_bytes;
    uint32_t* const cell_cand = c->cell_cand.as<uint32_t>() + F * g.cand_per_frame;
    int* const cell_count = c->cell_count.as<int>() + F * g.ncells;
    int* const cell_off = c->cell_off.as<int>() + F * g.ncells;
    uint32_t* const key_xy = c->key_xy.as<uint32_t>() + F * g.keys_per_frame;
    uint8_t* const key_resp = c->key_resp.as<uint8_t>() + F * g.keys_per_frame;
    uint16_t* const key_node = c->key_node.as<uint16_t>() + F * g.keys_per_frame;
    uint8_t* const kp_stage = static_cast<uint8_t*>(c->kp_stage.p) + F * (size_t)g.kp_per_frame * 8;
    int* const level_count = c->level_count.as<int>() + F * kMaxLevels;
    int* const level_ncand = c->level_ncand.as<int>() + F * kMaxLevels;
    hipStream_t s = c->stream;
    EventSet* es = nullptr;
    if (c->events_ok && c->stage_timing && !pyramid_only) {
        es = &c->evs[c->ev_next];
        c->ev_next = (c->ev_next + 1) % kEventSets;
        harvest(c, *es);
    }
    static const bool want_overlap = getenv("ORBX_OVERLAP") && atoi(getenv("ORBX_OVERLAP")) != 0;
    const bool overlap = c->aux_stream && want_overlap && !pyramid_only;
    const bool only_fast = c->stage_timing == 2;
    if (es) { es->chained = !overlap && !only_fast; es->only_fast = only_fast; }
    auto t0 = [&](int i, hipStream_t q) {
        if (es && (only_fast ? i == ST_FAST : (i == 0 || !es->chained))) (void)hipEventRecord(es->start[i], q);
    };
    auto t1 = [&](int i, hipStream_t q) { if (es && (!only_fast || i == ST_FAST)) (void)hipEventRecord(es->stop[i], q); };

    t0(ST_PYR0, s);
    if (raw.on) {
        // nobody on this path reads the level-0 plane; whoever asks for it later (orbx_pyramid_level, the classic stereo
        // search) gets it from the image, which the caller keeps until then (chunks of one batch are contiguous)
        c->l0_owed.img = d_imgs - F * frame_stride;
        c->l0_owed.stride = stride;
        c->l0_owed.frame_stride = frame_stride;
        c->l0_owed.nframes = f0 + nframes;
    } else {
        if (f0 == 0) c->l0_owed = orbx_ctx::Level0Owed{};
        launch_pyr_level0(d_imgs, stride, frame_stride, pyr, g, nframes, s);
    }
    t1(ST_PYR0, s);
    t0(ST_RESIZE, s);
    static const bool old_resize = getenv("ORBX_OLD_RESIZE") != nullptr;
    const bool slide = g.resize_slide_ok && !old_resize;
    for (int l = 1; l < g.nlevels; l++) {
        if (slide) launch_resize_slide(pyr, g, l, c->xtab4.as<XTab4>(), c->ytab2.as<YTab2>(), nframes, s, l == 1 ? raw : RawL0());
        else launch_pyr_resize(pyr, g, l, c->xtab.as<XTab>(), c->ytab.as<YTab>(), nframes, s);
    }
    t1(ST_RESIZE, s);
    c->last_nframes = f0 + nframes;
    c->have_pyramid = true;
    c->have_features = false;
    c->h_pyr_frame = -1;
    if (pyramid_only) {
        HIPCHECK(c, hipGetLastError());
        return ORBX_OK;
    }
    // The blur rides in the FAST launch: both only need the finished pyramid, one is bound by vector
    // issue and the other by memory, and a lone frame saves a dependent launch.  ORBX_FUSE_MAX (read per
    // call: a test switches it) limits the fusion to batches of at most that many frames.
    const char* e_fuse = getenv("ORBX_FUSE_MAX");
    const bool fused = (!e_fuse || nframes <= atoi(e_fuse)) && !overlap && g.ncells > 0;
    t0(ST_FAST, s);
    if (fused)
        launch_fast_blur(pyr, blur, g, c->fast_lds, nframes >= 32 ? c->blur_map_batch : c->blur_map, c->fast_cells.as<FastCell>(),
                         cell_cand, cell_count, nframes, s, raw);
    else
        launch_fast_cells_wave(pyr, g, c->fast_lds, c->fast_cells.as<FastCell>(), cell_cand, cell_count, nframes, s, raw);
    t1(ST_FAST, s);
    // The blur only feeds the descriptors and the quadtree is a chain of workgroup barriers that
    // leaves the memory system idle, so the two run side by side on two streams (ORBX_OVERLAP=0
    // puts everything back on one stream; per-stage durations are only separable then).
    hipStream_t bs = overlap ? c->aux_stream : s;
    if (overlap) {
        HIPCHECK(c, hipEventRecord(c->ev_pyr, s));
        HIPCHECK(c, hipStreamWaitEvent(bs, c->ev_pyr, 0));
    }
    t0(ST_BLUR, bs);
    if (!fused) launch_blur_slide(pyr, blur, g, c->blur_map, nframes, bs, raw);
    t1(ST_BLUR, bs);
    if (overlap) HIPCHECK(c, hipEventRecord(c->ev_blur, bs));
    t0(ST_QUADTREE, s);
    // levels with very many candidates (large frames) run their quadtree on node counts, if the scratch for it is moderate
    const CountLevels* cl = nullptr;
    if (c->count_levels.n > 0) {
        const size_t need_bytes = (size_t)nframes * c->count_levels.n * c->count_levels.stride;
        if (need_bytes <= ((size_t)1 << 30)) {
            if (need_bytes > c->qt_counts.bytes) drop_graph(c);
            HIPCHECK(c, c->qt_counts.ensure(need_bytes));
            cl = &c->count_levels;
        }
    }
    launch_quadtree(g, cell_cand, cell_count, cell_off, key_xy, key_resp, key_node, kp_stage, level_count, level_ncand,
                    nframes, c->qt_scratch.as<unsigned char>() + F * kMaxLevels * quadtree_scratch_stride(g), cl,
                    c->qt_counts.as<unsigned char>(), s);
    t1(ST_QUADTREE, s);
    if (overlap) HIPCHECK(c, hipStreamWaitEvent(s, c->ev_blur, 0));
    t0(ST_DESC, s);
    launch_orient_desc(g, pyr, blur, kp_stage, level_count, d_kps, d_desc, cap, d_n_out, nframes, s, raw);
    t1(ST_DESC, s);
    if (es) es->pending = true;
    HIPCHECK(c, hipGetLastError());
    c->have_features = true;
    return ORBX_OK;
}

// The level-0 plane of the pyramid block, when the last extraction read level 0 from the image instead of copying it:
// made now, from that image, for a caller that reads the plane (ORBextractor::mvImagePyramid[0], the SAD windows of the
// classic stereo search).
int materialise_level0(orbx_ctx* c)
{
    if (!c->l0_owed.img) return ORBX_OK;
    launch_pyr_level0(c->l0_owed.img, c->l0_owed.stride, c->l0_owed.frame_stride, c->pyr.as<uint8_t>(), c->g, c->l0_owed.nframes,
                      c->stream);
    HIPCHECK(c, hipGetLastError());
    c->l0_owed = orbx_ctx::Level0Owed{};
    return ORBX_OK;
}

// ---- host <-> device staging ------------------------------------------------------------------
// Copies between the device and ordinary (pageable) host memory make the runtime pin the pages on
// every call, about a millisecond each -- ten times one frame's GPU time.  All host-buffer entry
// points therefore go through one pinned arena owned by the context: uploads are copied into it and
// sent from there, downloads land in it and are copied out by finish() after the stream has drained.
// One core moves about 30 GB/s between pageable memory and the staging area, less than the link to
// the device: copies of several megabytes are split over a few threads.
void big_memcpy(void* dst, const void* src, size_t bytes)
{
    constexpr size_t kSplitMin = 4u << 20;
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned nt = bytes < kSplitMin ? 1u : std::min(4u, std::max(1u, hw));
    if (nt <= 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t part = ((bytes / nt) + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (unsigned i = 1; i < nt; i++) {
        const size_t o = std::min(bytes, i * part), e = std::min(bytes, (i + 1) * part);
        if (e > o) th.emplace_back([=] { std::memcpy(static_cast<char*>(dst) + o, static_cast<const char*>(src) + o, e - o); });
    }
    std::memcpy(dst, src, std::min(bytes, part));
    for (auto& t : th) t.join();
}

hipError_t arena_reserve(orbx_ctx* c, size_t bytes, size_t* off)
{
    const size_t need = (c->pin_used + bytes + 63) & ~(size_t)63;
    if (need > c->pin_cap) {
        // everything enqueued so far refers to the old block: drain it first
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess && c->aux_busy) e = hipStreamSynchronize(c->aux_stream);
        if (e != hipSuccess) return e;
        for (const auto& pd : c->pin_pending) big_memcpy(pd.dst, c->pin + pd.off, pd.bytes);
        c->pin_pending.clear();
        c->pin_used = 0;
        const size_t want = std::max(((bytes + 63) & ~(size_t)63) * 2, c->pin_cap * 2);
        if (want > c->pin_cap || !c->pin) {
            if (c->pin) (void)hipHostFree(c->pin);
            c->pin = nullptr;
            c->pin_cap = 0;
            void* p = nullptr;
            e = hipHostMalloc(&p, want, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            c->pin = static_cast<uint8_t*>(p);
            c->pin_cap = want;
        }
    }
    *off = c->pin_used;
    c->pin_used = (c->pin_used + bytes + 63) & ~(size_t)63;
    return hipSuccess;
}

// Large caller buffers that are page-locked already (hipHostMalloc, hipHostRegister, a pinned torch
// tensor) are copied from / to directly: staging them would only add a pass over host memory, which
// for a batch of frames costs more than the upload itself.  Every host-buffer entry point drains the
// stream before it returns, so the caller's memory is not touched after the call.
constexpr size_t kDirectCopyMin = 64 * 1024;
bool host_is_pinned(const void* p, size_t bytes)
{
    if (bytes < kDirectCopyMin) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();          // an ordinary pointer is reported as an error by older runtimes
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

hipError_t h2d(orbx_ctx* c, void* dev, const void* host, size_t bytes)
{
    if (bytes == 0) return hipSuccess;
    if (host_is_pinned(host, bytes)) return hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream);
    size_t off;
    hipError_t e = arena_reserve(c, bytes, &off);
    if (e != hipSuccess) return e;
    std::memcpy(c->pin + off, host, bytes);
    return hipMemcpyAsync(dev, c->pin + off, bytes, hipMemcpyHostToDevice, c->stream);
}

// h rows of w bytes, source rows `stride` apart, packed tightly on the device
hipError_t h2d_rows(orbx_ctx* c, void* dev, const uint8_t* host, size_t stride, size_t w, size_t h, hipStream_t q = nullptr)
{
    if (w == 0 || h == 0) return hipSuccess;
    if (!q) q = c->stream;
    if (host_is_pinned(host, w * h))
        return stride == w ? hipMemcpyAsync(dev, host, w * h, hipMemcpyHostToDevice, q)
                           : hipMemcpy2DAsync(dev, w, host, stride, w, h, hipMemcpyHostToDevice, q);
    size_t off;
    hipError_t e = arena_reserve(c, w * h, &off);
    if (e != hipSuccess) return e;
    if (stride == w) big_memcpy(c->pin + off, host, w * h);
    else for (size_t r = 0; r < h; r++) std::memcpy(c->pin + off + r * w, host + r * stride, w);
    return hipMemcpyAsync(dev, c->pin + off, w * h, hipMemcpyHostToDevice, q);
}

hipError_t d2h(orbx_ctx* c, void* host, const void* dev, size_t bytes)
{
    if (bytes == 0) return hipSuccess;
    if (host_is_pinned(host, bytes)) return hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream);
    size_t off;
    hipError_t e = arena_reserve(c, bytes, &off);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(c->pin + off, dev, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) return e;
    c->pin_pending.push_back({host, off, bytes});
    return hipSuccess;
}

// drains the stream and delivers the downloads
hipError_t finish(orbx_ctx* c)
{
    hipError_t e = hipStreamSynchronize(c->stream);
    if (c->aux_busy) {      // normally complete already (the first stream waited for every upload); matters after an error
        const hipError_t e2 = hipStreamSynchronize(c->aux_stream);
        if (e == hipSuccess) e = e2;
        c->aux_busy = false;
    }
    if (e == hipSuccess)
        for (const auto& pd : c->pin_pending) big_memcpy(pd.dst, c->pin + pd.off, pd.bytes);
    c->pin_pending.clear();
    c->pin_used = 0;
    return e;
}

void drop_graph(orbx_ctx* c)
{
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    c->graph_exec = nullptr;
    c->graph_key = orbx_ctx::GraphKey{};
    c->graph_seen = orbx_ctx::GraphKey{};
}

// run_batch for a small host-fed call on the staging block `base`: as a recorded graph from the second call with the
// same arguments on.  The first call with new arguments runs the plain launches (they may size buffers, which a
// capture must not), the second one records them, later ones replay.
int run_small_batch(orbx_ctx* c, uint8_t* base, size_t o_in, size_t o_k, size_t o_d, int32_t* hn, int nframes, int w, int h,
                    size_t fbytes, int cap)
{
    static const bool no_graph = getenv("ORBX_NO_GRAPH") != nullptr;
    auto plain = [&]() {
        return run_batch(c, base + o_in, nframes, w, h, w, fbytes, reinterpret_cast<orbx_keypoint*>(base + o_k), base + o_d, cap,
                         hn, false);
    };
    if (no_graph || c->graph_broken || c->stage_timing) return plain();
    orbx_ctx::GraphKey key;
    key.base = base; key.stream = c->stream; key.nframes = nframes; key.w = w; key.h = h; key.cap = cap;
    for (const char* name : {"ORBX_FUSE_MAX", "ORBX_QT_WIDE_MAX", "ORBX_QT_MID_MAX", "ORBX_OVERLAP", "ORBX_OLD_RESIZE"}) {
        const char* v = getenv(name);
        key.switches = key.switches * 1099511628211ull + 1469598103934665603ull;
        for (; v && *v; v++) key.switches = (key.switches ^ (unsigned char)*v) * 1099511628211ull;
    }
    auto fill = [&](orbx_ctx::GraphKey& k) {
        k.pyr = c->pyr.p; k.blur = c->blur.p; k.cand = c->cell_cand.p; k.kp_stage = c->kp_stage.p; k.counts = c->qt_counts.p;
    };
    fill(key);
    if (c->graph_exec && c->graph_key == key && c->gw == w && c->gh == h) {
        HIPCHECK(c, hipGraphLaunch(c->graph_exec, c->stream));
        c->last_nframes = nframes;
        c->have_pyramid = c->have_features = true;
        c->h_pyr_frame = -1;
        return ORBX_OK;
    }
    if (!(c->graph_seen == key) || c->gw != w || c->gh != h) {
        // new arguments: plain launches now, a recording next time if they come again unchanged
        if (c->graph_exec) drop_graph(c);
        const int rc = plain();
        fill(key);                                  // the call may have sized the buffers
        c->graph_seen = key;
        return rc;
    }
    if (c->graph_exec) drop_graph(c);
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        c->graph_broken = true;
        return plain();
    }
    const int rc = plain();
    const hipError_t e_end = hipStreamEndCapture(c->stream, &graph);
    hipGraphExec_t exec = nullptr;
    const bool ok = rc == ORBX_OK && e_end == hipSuccess && graph &&
                    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess && exec;
    if (graph) (void)hipGraphDestroy(graph);
    if (!ok) {
        (void)hipGetLastError();
        if (exec) (void)hipGraphExecDestroy(exec);
        c->graph_broken = true;
        if (rc != ORBX_OK) return rc;
        return plain();                             // nothing has run yet: the capture only recorded
    }
    c->graph_exec = exec;
    c->graph_key = key;
    HIPCHECK(c, hipGraphLaunch(c->graph_exec, c->stream));
    return ORBX_OK;
}

int check_frame_args(orbx_ctx* c, int nframes, int w, int h, int stride)
{
    if (!c) return ORBX_ERR_INVALID;
    if (nframes < 1 || nframes > 65535) return fail(c, ORBX_ERR_INVALID, "nframes must be in [1, 65535]");
    if (w < 1 || h < 1 || stride < w) return fail(c, ORBX_ERR_INVALID, "bad image size or stride");
    return ORBX_OK;
}

} // namespace

extern "C" {

const char* orbx_version(void) { return "orbx 0.1 (gfx950)"; }

int orbx_create(orbx_ctx** out, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device)
{
    if (!out) return ORBX_ERR_INVALID;
    *out = nullptr;
    if (device < 0) return ORBX_ERR_NO_DEVICE;   // there is no CPU path
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) return ORBX_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ORBX_ERR_NO_DEVICE;   // code objects are gfx950 only
    orbx_ctx* c = new (std::nothrow) orbx_ctx();
    if (!c) return ORBX_ERR_INVALID;
    if (!build_params(c->p, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)) {
        delete c;
        return ORBX_ERR_INVALID;
    }
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return ORBX_ERR_HIP;
    }
    c->stream = c->own_stream;
    {
        hipError_t se;
        c->static_scratch = match_static_scratch(device, &se);   // nullptr: only the _static entry point is affected
        if (!c->static_scratch) (void)hipGetLastError();
    }
    int taps[7];
    gauss7_taps(taps);
    if (upload_constant_tables(taps, c->p.umax, c->stream) != 0) {
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return ORBX_ERR_HIP;
    }
    c->events_ok = true;
    for (int i = 0; i < kEventSets && c->events_ok; i++)
        for (int j = 0; j < ST_COUNT; j++)
            if (hipEventCreateWithFlags(&c->evs[i].start[j], kStageEventFlags) != hipSuccess ||
                hipEventCreateWithFlags(&c->evs[i].stop[j], kStageEventFlags) != hipSuccess) {
                c->events_ok = false;
                break;
            }
    if (hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_pyr, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_blur, hipEventDisableTiming) != hipSuccess)
        c->aux_stream = nullptr;   // no overlap, still correct
    *out = c;
    return ORBX_OK;
}

void orbx_destroy(orbx_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)finish(c);
    DevBuf* bufs[] = {&c->raw_cells0, &c->raw_ytab2, &c->xtab, &c->ytab, &c->xtab4, &c->ytab2, &c->fast_cells, &c->pyr, &c->blur, &c->cell_cand, &c->cell_count, &c->cell_off, &c->key_xy,
                      &c->key_resp, &c->key_node, &c->kp_stage, &c->level_count, &c->level_ncand, &c->qt_scratch, &c->qt_counts, &c->in, &c->out_kps,
                      &c->out_desc, &c->out_n, &c->partial, &c->match_expand, &c->m0, &c->m1, &c->m2, &c->m3, &c->m4, &c->m5, &c->m6,
                      &c->m7, &c->m8, &c->m9, &c->grid_start, &c->grid_list, &c->gq[0], &c->gq[1], &c->gq[2],
                      &c->gq[3], &c->gq[4], &c->gq[5], &c->gq[6], &c->gq[7], &c->gq[8], &c->gq[9], &c->gq[10], &c->gq[11],
                      &c->stereo_listed, &c->stereo_rows};
    drop_graph(c);
    for (DevBuf* b : bufs) b->release();
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->h_pyr) (void)hipHostFree(c->h_pyr);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    if (c->events_ok)
        for (int i = 0; i < kEventSets; i++)
            for (int j = 0; j < ST_COUNT; j++) { (void)hipEventDestroy(c->evs[i].start[j]); (void)hipEventDestroy(c->evs[i].stop[j]); }
    for (hipEvent_t ev : c->ev_up) (void)hipEventDestroy(ev);
    if (c->ev_pyr) (void)hipEventDestroy(c->ev_pyr);
    if (c->ev_blur) (void)hipEventDestroy(c->ev_blur);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* orbx_last_error(const orbx_ctx* c) { return c ? c->err.c_str() : "null context"; }

int orbx_set_stream(orbx_ctx* c, void* hip_stream)
{
    if (!c) return ORBX_ERR_INVALID;
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, finish(c));
    c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
    return ORBX_OK;
}

int orbx_get_params(const orbx_ctx* c, int* nfeatures, float* scaleFactor, int* nlevels, int* iniThFAST, int* minThFAST)
{
    if (!c) return ORBX_ERR_INVALID;
    if (nfeatures) *nfeatures = c->p.nfeatures;
    if (scaleFactor) *scaleFactor = (float)c->p.scaleFactor;
    if (nlevels) *nlevels = c->p.nlevels;
    if (iniThFAST) *iniThFAST = c->p.iniTh;
    if (minThFAST) *minThFAST = c->p.minTh;
    return ORBX_OK;
}

int orbx_get_tables(const orbx_ctx* c, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int32_t* nfeat_per_level)
{
    if (!c) return ORBX_ERR_INVALID;
    for (int i = 0; i < c->p.nlevels; i++) {
        if (scale) scale[i] = c->p.scale[i];
        if (inv_scale) inv_scale[i] = c->p.inv_scale[i];
        if (sigma2) sigma2[i] = c->p.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = c->p.inv_sigma2[i];
        if (nfeat_per_level) nfeat_per_level[i] = c->p.nfeat[i];
    }
    return ORBX_OK;
}

int orbx_max_keypoints(orbx_ctx* c, int w, int h, int* max_kps)
{
    if (!c || !max_kps) return ORBX_ERR_INVALID;
    // per-frame callers ask this before every extraction: answer from the geometry in use, or from
    // the last size asked about, before building tables for a new size
    if (w == c->gw && h == c->gh) { *max_kps = c->g.kp_per_frame; return ORBX_OK; }
    if (w == c->mk_w && h == c->mk_h) { *max_kps = c->mk_val; return ORBX_OK; }
    FrameGeom g;
    ResizeTables rt;
    const char* msg = "";
    int rc = build_geometry(c->p, w, h, g, rt, &msg);
    if (rc != ORBX_OK) return fail(c, rc, msg);
    c->mk_w = w; c->mk_h = h; c->mk_val = g.kp_per_frame;
    *max_kps = g.kp_per_frame;
    return ORBX_OK;
}

int orbx_level_size(const orbx_ctx* c, int w, int h, int level, int* lw, int* lh)
{
    if (!c || level < 0 || level >= c->p.nlevels || !lw || !lh) return ORBX_ERR_INVALID;
    level_size(c->p, w, h, level, lw, lh);
    return ORBX_OK;
}

int orbx_extract_batch_device(orbx_ctx* c, const uint8_t* d_imgs, int nframes, int w, int h, int stride,
                              size_t frame_stride, orbx_keypoint* d_kps, uint8_t* d_desc, int cap_per_frame,
                              int32_t* d_n_out)
{
    int rc = check_frame_args(c, nframes, w, h, stride);
    if (rc != ORBX_OK) return rc;
    if (!d_imgs || !d_kps || !d_desc || !d_n_out || cap_per_frame < 1) return fail(c, ORBX_ERR_INVALID, "null buffer or cap < 1");
    if (frame_stride < (size_t)stride * (size_t)h) return fail(c, ORBX_ERR_INVALID, "frame_stride smaller than one frame");
    HIPCHECK(c, hipSetDevice(c->device));
    return run_batch(c, d_imgs, nframes, w, h, stride, frame_stride, d_kps, d_desc, cap_per_frame, d_n_out, false, 0, true);
}

int orbx_extract_batch(orbx_ctx* c, const uint8_t* const* imgs, int nframes, int w, int h, int stride,
                       orbx_keypoint* kps, uint8_t* desc, int cap_per_frame, int32_t* n_out)
{
    int rc = check_frame_args(c, nframes, w, h, stride);
    if (rc != ORBX_OK) return rc;
    if (!imgs || !kps || !desc || !n_out || cap_per_frame < 1) return fail(c, ORBX_ERR_INVALID, "null buffer or cap < 1");
    HIPCHECK(c, hipSetDevice(c->device));
    const size_t fbytes = (size_t)w * h, n = (size_t)nframes;
    HIPCHECK(c, finish(c));
    HIPCHECK(c, c->in.ensure(n * fbytes));
    HIPCHECK(c, c->out_kps.ensure(n * cap_per_frame * sizeof(orbx_keypoint)));
    HIPCHECK(c, c->out_desc.ensure(n * cap_per_frame * 32));
    HIPCHECK(c, c->out_n.ensure(n * 4));
    for (int f = 0; f < nframes; f++)
        if (!imgs[f]) return fail(c, ORBX_ERR_INVALID, "null frame pointer");
    // A handful of frames is bound by latency, and every copy between host and device is a dependent step of
    // 10-20 us.  The staging area is page-locked memory the device can address, so the kernels read the
    // image from it and write the features into it directly: no copy commands at all, and only the
    // keypoints that exist are handed to the caller.
    static const bool zero_copy = getenv("ORBX_NO_ZERO_COPY") == nullptr;
    // (measured: 174 -> 142 us for a stereo pair, 205 -> 191 us for 4 frames, no gain from 8 frames on)
    static const int zero_copy_max = getenv("ORBX_ZERO_COPY_MAX") ? atoi(getenv("ORBX_ZERO_COPY_MAX")) : 4;
    if (nframes <= zero_copy_max && zero_copy) {
        const size_t kb = (size_t)cap_per_frame * sizeof(orbx_keypoint), db = (size_t)cap_per_frame * 32;
        const size_t o_in = 0, o_n = (n * fbytes + 63) & ~(size_t)63, o_k = o_n + ((4 * n + 63) & ~(size_t)63),
                     o_d = (o_k + n * kb + 63) & ~(size_t)63;
        size_t off;
        HIPCHECK(c, arena_reserve(c, o_d + n * db, &off));
        uint8_t* base = c->pin + off;
        for (int f = 0; f < nframes; f++)
            for (int r = 0; r < h; r++)
                std::memcpy(base + o_in + f * fbytes + (size_t)r * w, imgs[f] + (size_t)r * stride, (size_t)w);
        int32_t* hn = reinterpret_cast<int32_t*>(base + o_n);
        for (int f = 0; f < nframes; f++) hn[f] = 0;
        rc = run_small_batch(c, base, o_in, o_k, o_d, hn, nframes, w, h, fbytes, cap_per_frame);
        if (rc != ORBX_OK) return rc;
        HIPCHECK(c, finish(c));
        // every frame is delivered (the first cap_per_frame keypoints of one that overflows) before
        // the overflow is reported, as on the copy path below
        bool overflow = false;
        for (int f = 0; f < nframes; f++) {
            n_out[f] = hn[f];
            const size_t m = (size_t)std::min(hn[f], cap_per_frame);
            overflow = overflow || hn[f] > cap_per_frame;
            std::memcpy(kps + (size_t)f * cap_per_frame, base + o_k + f * kb, m * sizeof(orbx_keypoint));
            std::memcpy(desc + (size_t)f * db, base + o_d + f * db, m * 32);
        }
        if (overflow) return fail(c, ORBX_ERR_CAPACITY, "cap_per_frame smaller than the number of keypoints");
        return ORBX_OK;
    }
    // uploads frames [f_lo, f_hi) on stream q; frames that lie back to back in host memory (one array
    // holding the batch) go up as one copy: a copy per 361 KB frame spends as long on its set-up as on
    // its bytes
    auto upload = [&](int f_lo, int f_hi, hipStream_t q) -> hipError_t {
        for (int f = f_lo; f < f_hi;) {
            int e = f + 1;
            if (stride == w)
                while (e < f_hi && imgs[e] == imgs[e - 1] + fbytes) e++;
            hipError_t err = h2d_rows(c, c->in.as<uint8_t>() + f * fbytes, imgs[f], stride, w, (size_t)h * (e - f), q);
            if (err != hipSuccess) return err;
            f = e;
        }
        return hipSuccess;
    };
    // A large batch is transfer-bound, so it runs in chunks: the upload of one chunk (second stream)
    // overlaps the kernels of the previous one.  Buffers and the staging area are sized for the whole
    // batch first, so that nothing is reallocated while copies are in flight.
    constexpr int kChunk = 64;
    const int nchunks = (nframes + kChunk - 1) / kChunk;
    static const bool no_pipe = getenv("ORBX_NO_PIPELINE") != nullptr;
    bool pipelined = nframes >= 2 * kChunk && c->aux_stream && !no_pipe;
    if (pipelined) {
        while ((int)c->ev_up.size() < nchunks) {
            hipEvent_t ev;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { pipelined = false; break; }
            c->ev_up.push_back(ev);
        }
    }
    if (pipelined) {
        rc = ensure_geometry(c, w, h);
        if (rc != ORBX_OK) return rc;
        rc = ensure_frames(c, nframes);
        if (rc != ORBX_OK) return rc;
        {   // staging for everything that is not page-locked, reserved (and released again) in one piece
            size_t need = 4 * n + 4096;
            if (!host_is_pinned(imgs[0], fbytes)) need += n * (fbytes + 64);
            if (!host_is_pinned(kps, n * cap_per_frame * sizeof(orbx_keypoint))) need += n * cap_per_frame * sizeof(orbx_keypoint);
            if (!host_is_pinned(desc, n * cap_per_frame * 32)) need += n * cap_per_frame * 32;
            size_t off;
            HIPCHECK(c, arena_reserve(c, need, &off));
            c->pin_used = off;
        }
        // the second stream starts after everything already queued on the first one (table uploads, memsets)
        HIPCHECK(c, hipEventRecord(c->ev_up[0], c->stream));
        HIPCHECK(c, hipStreamWaitEvent(c->aux_stream, c->ev_up[0], 0));
        c->aux_busy = true;
        for (int k = 0; k < nchunks; k++) {
            const int f0 = k * kChunk, m = std::min(kChunk, nframes - f0);
            HIPCHECK(c, upload(f0, f0 + m, c->aux_stream));
            HIPCHECK(c, hipEventRecord(c->ev_up[k], c->aux_stream));
            HIPCHECK(c, hipStreamWaitEvent(c->stream, c->ev_up[k], 0));
            rc = run_batch(c, c->in.as<uint8_t>() + (size_t)f0 * fbytes, m, w, h, w, fbytes,
                           c->out_kps.as<orbx_keypoint>() + (size_t)f0 * cap_per_frame,
                           c->out_desc.as<uint8_t>() + (size_t)f0 * cap_per_frame * 32, cap_per_frame,
                           c->out_n.as<int32_t>() + f0, false, f0, true);
            if (rc != ORBX_OK) return rc;
        }
    } else {
        HIPCHECK(c, upload(0, nframes, c->stream));
        rc = run_batch(c, c->in.as<uint8_t>(), nframes, w, h, w, fbytes, c->out_kps.as<orbx_keypoint>(),
                       c->out_desc.as<uint8_t>(), cap_per_frame, c->out_n.as<int32_t>(), false, 0, true);
        if (rc != ORBX_OK) return rc;
    }
    HIPCHECK(c, d2h(c, n_out, c->out_n.p, n * 4));
    HIPCHECK(c, d2h(c, kps, c->out_kps.p, n * cap_per_frame * sizeof(orbx_keypoint)));
    HIPCHECK(c, d2h(c, desc, c->out_desc.p, n * cap_per_frame * 32));
    HIPCHECK(c, finish(c));
    for (int f = 0; f < nframes; f++)
        if (n_out[f] > cap_per_frame) return fail(c, ORBX_ERR_CAPACITY, "cap_per_frame smaller than the number of keypoints");
    return ORBX_OK;
}

int orbx_extract(orbx_ctx* c, const uint8_t* img, int w, int h, int stride, orbx_keypoint* kps, uint8_t* desc, int cap,
                 int* n_out)
{
    if (!c || !n_out) return ORBX_ERR_INVALID;
    *n_out = 0;
    if (!img || w <= 0 || h <= 0) return ORBX_OK;   // empty image: nothing happens (src/ORBextractor.cc:1130-1131)
    int32_t n = 0;
    const uint8_t* imgs[1] = {img};
    int rc = orbx_extract_batch(c, imgs, 1, w, h, stride, kps, desc, cap, &n);
    *n_out = n;
    return rc;
}

int orbx_compute_pyramid(orbx_ctx* c, const uint8_t* img, int w, int h, int stride)
{
    int rc = check_frame_args(c, 1, w, h, stride);
    if (rc != ORBX_OK) return rc;
    if (!img) return fail(c, ORBX_ERR_INVALID, "null image");
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, finish(c));
    HIPCHECK(c, c->in.ensure((size_t)w * h));
    HIPCHECK(c, h2d_rows(c, c->in.as<uint8_t>(), img, stride, w, h));
    rc = run_batch(c, c->in.as<uint8_t>(), 1, w, h, w, (size_t)w * h, nullptr, nullptr, 0, nullptr, true);
    if (rc != ORBX_OK) return rc;
    HIPCHECK(c, finish(c));
    return ORBX_OK;
}

int orbx_pyramid_level(orbx_ctx* c, int frame, int level, const uint8_t** data, int* w, int* h, int* stride)
{
    if (!c || !data) return ORBX_ERR_INVALID;
    if (!c->have_pyramid) return fail(c, ORBX_ERR_STATE, "no pyramid has been computed on this context");
    if (frame < 0 || frame >= c->last_nframes || level < 0 || level >= c->g.nlevels)
        return fail(c, ORBX_ERR_INVALID, "frame or level out of range");
    HIPCHECK(c, hipSetDevice(c->device));
    {
        const int rc0 = materialise_level0(c);
        if (rc0 != ORBX_OK) return rc0;
    }
    if (c->h_pyr_frame != frame) {
        // the mirror is pinned memory of its own: one direct copy, nothing staged
        if (c->h_pyr_bytes < c->g.pyr_frame_bytes) {
            HIPCHECK(c, finish(c));
            if (c->h_pyr) (void)hipHostFree(c->h_pyr);
            c->h_pyr = nullptr;
            c->h_pyr_bytes = 0;
            void* p = nullptr;
            HIPCHECK(c, hipHostMalloc(&p, c->g.pyr_frame_bytes, hipHostMallocDefault));
            c->h_pyr = static_cast<uint8_t*>(p);
            c->h_pyr_bytes = c->g.pyr_frame_bytes;
        }
        HIPCHECK(c, hipMemcpyAsync(c->h_pyr, c->pyr.as<uint8_t>() + (size_t)frame * c->g.pyr_frame_bytes,
                                   c->g.pyr_frame_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, finish(c));
        c->h_pyr_frame = frame;
    }
    const LevelGeom& L = c->g.lv[level];
    *data = c->h_pyr + L.plane_off + (size_t)kPadY * L.pitch + kPadX;
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (stride) *stride = L.pitch;
    return ORBX_OK;
}

int orbx_debug_candidates(orbx_ctx* c, int frame, int level, int32_t* xyr, int cap, int* n)
{
    if (!c || !n) return ORBX_ERR_INVALID;
    if (!c->have_features) return fail(c, ORBX_ERR_STATE, "no extraction has run on this context");
    if (frame < 0 || frame >= c->last_nframes || level < 0 || level >= c->g.nlevels)
        return fail(c, ORBX_ERR_INVALID, "frame or level out of range");
    HIPCHECK(c, hipSetDevice(c->device));
    const FrameGeom& g = c->g;
    const LevelGeom& L = g.lv[level];
    // the candidates of the level in the reference's append order (cells row by row, each cell's list in its own
    // order), put together from the per-cell lists the FAST launch wrote
    if (L.width == 0) { *n = 0; return ORBX_OK; }
    const int ncell = L.nCols * L.nRows;
    std::vector<int> counts(ncell);
    HIPCHECK(c, d2h(c, counts.data(), c->cell_count.as<int>() + (size_t)frame * g.ncells + L.cell_base, (size_t)ncell * 4));
    HIPCHECK(c, finish(c));
    long long cnt = 0;
    for (int i = 0; i < ncell; i++) cnt += counts[i];
    *n = (int)cnt;
    if (cnt > cap) return fail(c, ORBX_ERR_CAPACITY, "candidate buffer too small");
    if (cnt == 0) return ORBX_OK;
    if (!xyr) return ORBX_ERR_INVALID;
    std::vector<uint32_t> slots((size_t)ncell * L.cell_cap);
    HIPCHECK(c, d2h(c, slots.data(), c->cell_cand.as<uint32_t>() + (size_t)frame * g.cand_per_frame + L.cand_base, slots.size() * 4));
    HIPCHECK(c, finish(c));
    int o = 0;
    for (int cell = 0; cell < ncell; cell++) {
        const int ci = cell / L.nCols, cj = cell - ci * L.nCols;
        for (int j = 0; j < counts[cell]; j++, o++) {
            const uint32_t u = slots[(size_t)cell * L.cell_cap + j];
            xyr[3 * o] = cj * L.wCell + 3 + (int32_t)((u >> 8) & 0xff);
            xyr[3 * o + 1] = ci * L.hCell + 3 + (int32_t)(u >> 16);
            xyr[3 * o + 2] = (int32_t)(u & 0xff);
        }
    }
    return ORBX_OK;
}

int orbx_debug_level_counts(orbx_ctx* c, int frame, int32_t* counts)
{
    if (!c || !counts) return ORBX_ERR_INVALID;
    if (!c->have_features) return fail(c, ORBX_ERR_STATE, "no extraction has run on this context");
    if (frame < 0 || frame >= c->last_nframes) return fail(c, ORBX_ERR_INVALID, "frame out of range");
    HIPCHECK(c, hipSetDevice(c->device));
    int32_t tmp[kMaxLevels];
    HIPCHECK(c, d2h(c, tmp, c->level_count.as<int>() + (size_t)frame * kMaxLevels, sizeof(tmp)));
    HIPCHECK(c, finish(c));
    for (int l = 0; l < c->g.nlevels; l++) counts[l] = tmp[l];
    return ORBX_OK;
}

int orbx_debug_blurred_level(orbx_ctx* c, int frame, int level, uint8_t* out, size_t out_size)
{
    if (!c || !out) return ORBX_ERR_INVALID;
    if (!c->have_features) return fail(c, ORBX_ERR_STATE, "no extraction has run on this context");
    if (frame < 0 || frame >= c->last_nframes || level < 0 || level >= c->g.nlevels)
        return fail(c, ORBX_ERR_INVALID, "frame or level out of range");
    const LevelGeom& L = c->g.lv[level];
    if (out_size < (size_t)L.w * L.h) return fail(c, ORBX_ERR_CAPACITY, "output buffer too small");
    HIPCHECK(c, hipSetDevice(c->device));
    const uint8_t* src = c->blur.as<uint8_t>() + (size_t)frame * c->g.pyr_frame_bytes + L.plane_off +
                         (size_t)kPadY * L.pitch + kPadX;
    HIPCHECK(c, hipMemcpy2DAsync(out, L.w, src, L.pitch, L.w, L.h, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, finish(c));
    return ORBX_OK;
}

int orbx_stage_times(orbx_ctx* c, const char** names, float* total_ms, int32_t* launches, int cap)
{
    if (!c) return ORBX_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)finish(c);
    for (int i = 0; i < kEventSets; i++) harvest(c, c->evs[i]);
    int n = ST_COUNT < cap ? ST_COUNT : cap;
    for (int s = 0; s < n; s++) {
        if (names) names[s] = kStageNames[s];
        if (total_ms) total_ms[s] = (float)c->stage_ms[s];
        if (launches) launches[s] = c->stage_launches[s];
    }
    return n;
}

int orbx_reset_stage_times(orbx_ctx* c)
{
    if (!c) return ORBX_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)finish(c);
    for (int i = 0; i < kEventSets; i++) harvest(c, c->evs[i]);
    for (int s = 0; s < ST_COUNT; s++) { c->stage_ms[s] = 0; c->stage_launches[s] = 0; }
    return ORBX_OK;
}

int orbx_set_stage_timing(orbx_ctx* c, int enable)
{
    if (!c) return ORBX_ERR_INVALID;
    c->stage_timing = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    return ORBX_OK;
}

// ---------------------------------------------------------------------------
// matching
// ---------------------------------------------------------------------------
int orbx_hamming(const uint32_t a[8], const uint32_t b[8])
{
    int d = 0;
    for (int i = 0; i < 8; i++) d += __builtin_popcount(a[i] ^ b[i]);
    return d;
}

int orbx_match_bruteforce_device(orbx_ctx* c, const uint32_t* d_q, int Q, int q_stride, const int32_t* d_nq,
                                 const uint32_t* d_t, int T, int t_stride, const int32_t* d_nt, int nbatch,
                                 int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist)
{
    if (!c) return ORBX_ERR_INVALID;
    if (!d_q || !d_best_idx || !d_best_dist || !d_second_dist) return fail(c, ORBX_ERR_INVALID, "null buffer");
    if (Q < 0 || T < 0 || nbatch < 1 || nbatch > 65535 || q_stride < Q || (t_stride != 0 && t_stride < T))
        return fail(c, ORBX_ERR_INVALID, "bad sizes");
    if (T > 0 && !d_t) return fail(c, ORBX_ERR_INVALID, "null train set");
    if (T >= (1 << 22)) return fail(c, ORBX_ERR_UNSUPPORTED, "train set of 2^22 descriptors or more");
    if (q_stride == 0) return ORBX_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    // Dense batches against one shared train set are one long list of queries: taken as a single batch, so that only
    // the list's last tile of 256 queries can be short instead of every batch's (the outputs lie the same way).
    if (t_stride == 0 && !d_nq && !d_nt && q_stride == Q && nbatch > 1 && (long long)Q * nbatch < (1ll << 30)) {
        Q *= nbatch;
        q_stride = Q;
        nbatch = 1;
    }
    int tsplit = match_bruteforce_tsplit(q_stride, T, nbatch);
    if (tsplit > 1) {
        size_t need = (size_t)nbatch * tsplit * q_stride * 8;
        if (need > c->partial.bytes) {
            HIPCHECK(c, finish(c));
            HIPCHECK(c, c->partial.ensure(need));
        }
    }
    // many tiles of queries against one train set: the set is expanded to its matrix form once, in memory
    const size_t expand = match_expand_bytes(q_stride, T, t_stride, nbatch);
    if (expand > c->match_expand.bytes) {
        HIPCHECK(c, finish(c));
        HIPCHECK(c, c->match_expand.ensure(expand));
    }
    launch_match_bruteforce(d_q, Q, q_stride, d_nq, d_t, T, t_stride, d_nt, nbatch, tsplit, c->partial.as<uint32_t>(),
                            expand ? c->match_expand.as<uint32_t>() : nullptr, d_best_idx, d_best_dist, d_second_dist, c->stream);
    HIPCHECK(c, hipGetLastError());
    return ORBX_OK;
}

int orbx_match_bruteforce_device_static(orbx_ctx* c, const uint32_t* d_q, int Q, int q_stride, const int32_t* d_nq,
                                        const uint32_t* d_t, int T, int t_stride, const int32_t* d_nt, int nbatch,
                                        int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist)
{
    if (!c) return ORBX_ERR_INVALID;
    if (!d_q || !d_best_idx || !d_best_dist || !d_second_dist) return fail(c, ORBX_ERR_INVALID, "null buffer");
    if (Q < 0 || T < 0 || nbatch < 1 || nbatch > 65535 || q_stride < Q || (t_stride != 0 && t_stride < T))
        return fail(c, ORBX_ERR_INVALID, "bad sizes");
    if (T > 0 && !d_t) return fail(c, ORBX_ERR_INVALID, "null train set");
    if (T >= (1 << 22)) return fail(c, ORBX_ERR_UNSUPPORTED, "train set of 2^22 descriptors or more");
    if (q_stride == 0) return ORBX_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    if (!c->static_scratch) return fail(c, ORBX_ERR_HIP, "the static scratch array of this device could not be located at orbx_create");
    // The preferred number of splits of the train set, or as many as the static scratch holds partial results for
    // (nbatch * tsplit * q_stride * 8 bytes); a single split needs no scratch at all, so no size is refused for it.
    int tsplit = match_bruteforce_tsplit(q_stride, T, nbatch);
    const size_t per_split = (size_t)nbatch * (size_t)q_stride * 8;
    if (tsplit > 1 && per_split * (size_t)tsplit > kMatchStaticScratchBytes)
        tsplit = (int)std::max<size_t>(1, kMatchStaticScratchBytes / per_split);
    // every slot k_match_merge reads is written by this call's sweep first: nothing to clear
    launch_match_bruteforce(d_q, Q, q_stride, d_nq, d_t, T, t_stride, d_nt, nbatch, tsplit, c->static_scratch, nullptr,
                            d_best_idx, d_best_dist, d_second_dist, c->stream);
    HIPCHECK(c, hipGetLastError());
    return ORBX_OK;
}

int orbx_match_bruteforce(orbx_ctx* c, const uint32_t* q, int Q, const uint32_t* t, int T, int32_t* best_idx,
                          int32_t* best_dist, int32_t* second_dist)
{
    if (!c) return ORBX_ERR_INVALID;
    if (Q < 0 || T < 0) return fail(c, ORBX_ERR_INVALID, "negative size");
    if (Q == 0) return ORBX_OK;
    if (!q || !best_idx || !best_dist || !second_dist || (T > 0 && !t)) return fail(c, ORBX_ERR_INVALID, "null buffer");
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, finish(c));
    HIPCHECK(c, c->m0.ensure((size_t)Q * 32));
    HIPCHECK(c, c->m1.ensure((size_t)(T > 0 ? T : 1) * 32));
    HIPCHECK(c, c->m2.ensure((size_t)Q * 4));
    HIPCHECK(c, c->m3.ensure((size_t)Q * 4));
    HIPCHECK(c, c->m4.ensure((size_t)Q * 4));
    HIPCHECK(c, h2d(c, c->m0.p, q, (size_t)Q * 32));
    if (T > 0) HIPCHECK(c, h2d(c, c->m1.p, t, (size_t)T * 32));
    int rc = orbx_match_bruteforce_device(c, c->m0.as<uint32_t>(), Q, Q, nullptr, c->m1.as<uint32_t>(), T, T, nullptr, 1,
                                          c->m2.as<int32_t>(), c->m3.as<int32_t>(), c->m4.as<int32_t>());
    if (rc != ORBX_OK) return rc;
    HIPCHECK(c, d2h(c, best_idx, c->m2.p, (size_t)Q * 4));
    HIPCHECK(c, d2h(c, best_dist, c->m3.p, (size_t)Q * 4));
    HIPCHECK(c, d2h(c, second_dist, c->m4.p, (size_t)Q * 4));
    HIPCHECK(c, finish(c));
    return ORBX_OK;
}

static int upload_csr(orbx_ctx* c, const uint32_t* q, int Q, const uint32_t* t, int T, const int32_t* cand_off,
                      const int32_t* cand_idx, const int32_t* t_octave, int* ncand_out)
{
    if (!q || !cand_off || (T > 0 && !t)) return fail(c, ORBX_ERR_INVALID, "null buffer");
    const int ncand = cand_off[Q];
    if (cand_off[0] != 0 || ncand < 0) return fail(c, ORBX_ERR_INVALID, "cand_off must start at 0 and be non-decreasing");
    for (int i = 0; i < Q; i++) {
        if (cand_off[i + 1] < cand_off[i]) return fail(c, ORBX_ERR_INVALID, "cand_off must be non-decreasing");
        if (cand_off[i + 1] - cand_off[i] >= (1 << 22)) return fail(c, ORBX_ERR_UNSUPPORTED, "candidate list of 2^22 entries or more");
    }
    if (ncand > 0 && !cand_idx) return fail(c, ORBX_ERR_INVALID, "null candidate list");
    for (int k = 0; k < ncand; k++)
        if (cand_idx[k] < 0 || cand_idx[k] >= T) return fail(c, ORBX_ERR_INVALID, "candidate index out of range");
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, finish(c));
    HIPCHECK(c, c->m0.ensure((size_t)Q * 32));
    HIPCHECK(c, c->m1.ensure((size_t)(T > 0 ? T : 1) * 32));
    HIPCHECK(c, c->m5.ensure((size_t)(Q + 1) * 4));
    HIPCHECK(c, c->m6.ensure((size_t)(ncand > 0 ? ncand : 1) * 4));
    HIPCHECK(c, c->m7.ensure((size_t)(T > 0 ? T : 1) * 4));
    HIPCHECK(c, h2d(c, c->m0.p, q, (size_t)Q * 32));
    if (T > 0) HIPCHECK(c, h2d(c, c->m1.p, t, (size_t)T * 32));
    HIPCHECK(c, h2d(c, c->m5.p, cand_off, (size_t)(Q + 1) * 4));
    if (ncand > 0) HIPCHECK(c, h2d(c, c->m6.p, cand_idx, (size_t)ncand * 4));
    if (t_octave && T > 0) HIPCHECK(c, h2d(c, c->m7.p, t_octave, (size_t)T * 4));
    *ncand_out = ncand;
    return ORBX_OK;
}

int orbx_match_windowed(orbx_ctx* c, const uint32_t* q, int Q, const uint32_t* t, int T, const int32_t* cand_off,
                        const int32_t* cand_idx, const int32_t* t_octave, int32_t* best_idx, int32_t* best_dist,
                        int32_t* second_dist, int32_t* best_lvl, int32_t* second_lvl)
{
    if (!c) return ORBX_ERR_INVALID;
    if (Q < 0 || T < 0) return fail(c, ORBX_ERR_INVALID, "negative size");
    if (Q == 0) return ORBX_OK;
    if (!best_idx || !best_dist || !second_dist) return fail(c, ORBX_ERR_INVALID, "null buffer");
    int ncand = 0;
    int rc = upload_csr(c, q, Q, t, T, cand_off, cand_idx, t_octave, &ncand);
    if (rc != ORBX_OK) return rc;
    HIPCHECK(c, c->m2.ensure((size_t)Q * 4));
    HIPCHECK(c, c->m3.ensure((size_t)Q * 4));
    HIPCHECK(c, c->m4.ensure((size_t)Q * 4));
    HIPCHECK(c, c->m8.ensure((size_t)Q * 4));
    HIPCHECK(c, c->m9.ensure((size_t)Q * 4));
    launch_match_windowed(c->m0.as<uint32_t>(), Q, c->m1.as<uint32_t>(), c->m5.as<int32_t>(), c->m6.as<int32_t>(),
                          t_octave ? c->m7.as<int32_t>() : nullptr, c->m2.as<int32_t>(), c->m3.as<int32_t>(),
                          c->m4.as<int32_t>(), c->m8.as<int32_t>(), c->m9.as<int32_t>(), c->stream);
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, d2h(c, best_idx, c->m2.p, (size_t)Q * 4));
    HIPCHECK(c, d2h(c, best_dist, c->m3.p, (size_t)Q * 4));
    HIPCHECK(c, d2h(c, second_dist, c->m4.p, (size_t)Q * 4));
    if (best_lvl) HIPCHECK(c, d2h(c, best_lvl, c->m8.p, (size_t)Q * 4));
    if (second_lvl) HIPCHECK(c, d2h(c, second_lvl, c->m9.p, (size_t)Q * 4));
    HIPCHECK(c, finish(c));
    return ORBX_OK;
}

int orbx_hamming_csr(orbx_ctx* c, const uint32_t* q, int Q, const uint32_t* t, int T, const int32_t* cand_off,
                     const int32_t* cand_idx, uint16_t* dist_out)
{
    if (!c) return ORBX_ERR_INVALID;
    if (Q < 0 || T < 0) return fail(c, ORBX_ERR_INVALID, "negative size");
    if (Q == 0) return ORBX_OK;
    int ncand = 0;
    int rc = upload_csr(c, q, Q, t, T, cand_off, cand_idx, nullptr, &ncand);
    if (rc != ORBX_OK) return rc;
    if (ncand == 0) return ORBX_OK;
    if (!dist_out) return fail(c, ORBX_ERR_INVALID, "null buffer");
    HIPCHECK(c, c->m2.ensure((size_t)ncand * 2));
    launch_hamming_csr(c->m0.as<uint32_t>(), Q, c->m1.as<uint32_t>(), c->m5.as<int32_t>(), c->m6.as<int32_t>(),
                       c->m2.as<uint16_t>(), c->stream);
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, d2h(c, dist_out, c->m2.p, (size_t)ncand * 2));
    HIPCHECK(c, finish(c));
    return ORBX_OK;
}

// shared by the stereo entry points: argument checks + the two launches
static int stereo_launch(orbx_ctx* c, const orbx_keypoint* d_kl, const uint32_t* d_dl, const int32_t* d_nl,
                         const orbx_keypoint* d_kr, const uint32_t* d_dr, const int32_t* d_nr, int nbatch, int stride,
                         int nrows, float mbf, float mb, float min_x, const uint8_t* d_active, const float* d_min_d,
                         const float* d_max_d, int th_orb_dist, int median_cut, int init_outputs, int32_t* d_listed,
                         int32_t* d_best_idx, int32_t* d_best_dist, float* d_u_right, float* d_depth, int32_t* d_nmatched)
{
    if (!d_kl || !d_dl || !d_nl || !d_kr || !d_dr || !d_nr || !d_best_idx || !d_best_dist || !d_u_right || !d_depth || !d_nmatched)
        return fail(c, ORBX_ERR_INVALID, "null buffer");
    if (nbatch < 1 || nbatch > 65535 || stride < 1 || nrows < 1) return fail(c, ORBX_ERR_INVALID, "bad sizes");
    if (stride >= (1 << 22)) return fail(c, ORBX_ERR_UNSUPPORTED, "2^22 keypoints per image or more");
    if (!(mb > 0.0f)) return fail(c, ORBX_ERR_INVALID, "baseline must be positive");
    if (th_orb_dist < 0 || th_orb_dist > 256) return fail(c, ORBX_ERR_INVALID, "th_orb_dist must be in [0, 256]");
    if (median_cut && !d_listed) return fail(c, ORBX_ERR_INVALID, "the median cut needs the listed-distance array");
    if (nrows > kStereoMaxRows) return fail(c, ORBX_ERR_UNSUPPORTED, "more than 16000 image rows");
    HIPCHECK(c, hipSetDevice(c->device));
    {   // scratch for the row ordering of the right keypoints
        const size_t need = stereo_rows_ints(nbatch, stride, nrows) * 4;
        if (need > c->stereo_rows.bytes) {
            HIPCHECK(c, finish(c));
            HIPCHECK(c, c->stereo_rows.ensure(need));
        }
    }
    StereoLaunch L;
    L.d_rows = c->stereo_rows.as<int32_t>();
    L.d_kl = d_kl; L.d_kr = d_kr; L.d_dl = d_dl; L.d_dr = d_dr; L.d_nl = d_nl; L.d_nr = d_nr;
    L.d_active = d_active; L.d_min_d = d_min_d; L.d_max_d = d_max_d; L.d_listed = d_listed;
    L.d_best_idx = d_best_idx; L.d_best_dist = d_best_dist; L.d_u_right = d_u_right; L.d_depth = d_depth;
    L.d_nmatched = d_nmatched;
    L.nbatch = nbatch; L.stride = stride; L.nrows = nrows; L.th_orb_dist = th_orb_dist;
    L.init_outputs = init_outputs; L.median_cut = median_cut;
    L.mbf = mbf; L.mb = mb; L.min_x = min_x;
    launch_stereo_match(L, c->p.scale, c->stream);
    HIPCHECK(c, hipGetLastError());
    return ORBX_OK;
}

int orbx_stereo_match_device(orbx_ctx* c, const orbx_keypoint* d_kl, const uint32_t* d_dl, const int32_t* d_nl,
                             const orbx_keypoint* d_kr, const uint32_t* d_dr, const int32_t* d_nr, int nbatch, int stride,
                             int nrows, float mbf, float mb, float min_x, int32_t* d_best_idx, int32_t* d_best_dist,
                             float* d_u_right, float* d_depth, int32_t* d_nmatched)
{
    if (!c) return ORBX_ERR_INVALID;
    if (nbatch < 1 || nbatch > 65535 || stride < 1) return fail(c, ORBX_ERR_INVALID, "bad sizes");
    HIPCHECK(c, hipSetDevice(c->device));
    // mvDistIdx of a Frame that is matched in one piece: scratch of the context
    const size_t need = (size_t)nbatch * stride * 4;
    if (need > c->stereo_listed.bytes) {
        HIPCHECK(c, finish(c));
        HIPCHECK(c, c->stereo_listed.ensure(need));
    }
    return stereo_launch(c, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, nbatch, stride, nrows, mbf, mb, min_x, nullptr, nullptr,
                         nullptr, 75, 1, 1, c->stereo_listed.as<int32_t>(), d_best_idx, d_best_dist, d_u_right, d_depth,
                         d_nmatched);
}

int orbx_stereo_match_ex_device(orbx_ctx* c, const orbx_keypoint* d_kl, const uint32_t* d_dl, const int32_t* d_nl,
                                const orbx_keypoint* d_kr, const uint32_t* d_dr, const int32_t* d_nr, int nbatch,
                                int stride, int nrows, float mbf, float mb, float min_x, const uint8_t* d_active,
                                const float* d_min_d, const float* d_max_d, int th_orb_dist, int median_cut,
                                int32_t* d_listed_dist, int32_t* d_best_idx, int32_t* d_best_dist, float* d_u_right,
                                float* d_depth, int32_t* d_nmatched)
{
    if (!c) return ORBX_ERR_INVALID;
    if (!d_listed_dist) return fail(c, ORBX_ERR_INVALID, "null buffer");
    return stereo_launch(c, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, nbatch, stride, nrows, mbf, mb, min_x, d_active, d_min_d,
                         d_max_d, th_orb_dist, median_cut, 0, d_listed_dist, d_best_idx, d_best_dist, d_u_right, d_depth,
                         d_nmatched);
}

// one pair from host buffers; ex: in/out state arrays as orbx_stereo_match_ex documents them
static int stereo_host(orbx_ctx* c, const orbx_keypoint* kl, const uint32_t* dl, int nl, const orbx_keypoint* kr,
                       const uint32_t* dr, int nr, int nrows, float mbf, float mb, float min_x, bool ex,
                       const uint8_t* active, const float* min_d, const float* max_d, int th_orb_dist, int median_cut,
                       int32_t* listed, int32_t* best_idx, int32_t* best_dist, float* u_right, float* depth, int* nmatched)
{
    if (!c || !nmatched) return ORBX_ERR_INVALID;
    *nmatched = 0;
    if (nl < 0 || nr < 0) return fail(c, ORBX_ERR_INVALID, "negative size");
    if (nl == 0) return ORBX_OK;
    if (!kl || !dl || !best_idx || !best_dist || !u_right || !depth || (nr > 0 && (!kr || !dr)) || (ex && !listed))
        return fail(c, ORBX_ERR_INVALID, "null buffer");
    for (int i = 0; i < nr; i++)
        if (kr[i].octave < 0 || kr[i].octave >= c->p.nlevels) return fail(c, ORBX_ERR_INVALID, "right keypoint octave out of range");
    const int stride = nl > nr ? nl : nr;
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, finish(c));
    const size_t S = (size_t)stride, NL = (size_t)nl;
    HIPCHECK(c, c->m0.ensure(S * sizeof(orbx_keypoint)));
    HIPCHECK(c, c->m1.ensure(S * 32));
    HIPCHECK(c, c->m2.ensure(S * sizeof(orbx_keypoint)));
    HIPCHECK(c, c->m3.ensure(S * 32));
    HIPCHECK(c, c->m4.ensure(16));
    HIPCHECK(c, c->m5.ensure(S * 4));
    HIPCHECK(c, c->m6.ensure(S * 4));
    HIPCHECK(c, c->m7.ensure(S * 4));
    HIPCHECK(c, c->m8.ensure(S * 4));
    HIPCHECK(c, c->m9.ensure(S * 4 * 3 + S));        // listed | min_d | max_d | active
    HIPCHECK(c, c->stereo_listed.ensure(S * 4));
    int32_t counts[3] = {nl, nr, 0};
    HIPCHECK(c, h2d(c, c->m0.p, kl, NL * sizeof(orbx_keypoint)));
    HIPCHECK(c, h2d(c, c->m1.p, dl, NL * 32));
    if (nr > 0) {
        HIPCHECK(c, h2d(c, c->m2.p, kr, (size_t)nr * sizeof(orbx_keypoint)));
        HIPCHECK(c, h2d(c, c->m3.p, dr, (size_t)nr * 32));
    }
    HIPCHECK(c, h2d(c, c->m4.p, counts, sizeof(counts)));
    int32_t* d_listed = c->m9.as<int32_t>();
    float* d_min = reinterpret_cast<float*>(d_listed + S);
    float* d_max = d_min + S;
    uint8_t* d_act = reinterpret_cast<uint8_t*>(d_max + S);
    int rc;
    if (ex) {
        // in/out state of the Frame: what the kernels only touch where the reference writes
        HIPCHECK(c, h2d(c, d_listed, listed, NL * 4));
        HIPCHECK(c, h2d(c, c->m7.p, u_right, NL * 4));
        HIPCHECK(c, h2d(c, c->m8.p, depth, NL * 4));
        HIPCHECK(c, h2d(c, c->m5.p, best_idx, NL * 4));
        HIPCHECK(c, h2d(c, c->m6.p, best_dist, NL * 4));
        if (active) HIPCHECK(c, h2d(c, d_act, active, NL));
        if (min_d) HIPCHECK(c, h2d(c, d_min, min_d, NL * 4));
        if (max_d) HIPCHECK(c, h2d(c, d_max, max_d, NL * 4));
        rc = orbx_stereo_match_ex_device(c, c->m0.as<orbx_keypoint>(), c->m1.as<uint32_t>(), c->m4.as<int32_t>(),
                                         c->m2.as<orbx_keypoint>(), c->m3.as<uint32_t>(), c->m4.as<int32_t>() + 1, 1,
                                         stride, nrows, mbf, mb, min_x, active ? d_act : nullptr, min_d ? d_min : nullptr,
                                         max_d ? d_max : nullptr, th_orb_dist, median_cut, d_listed, c->m5.as<int32_t>(),
                                         c->m6.as<int32_t>(), c->m7.as<float>(), c->m8.as<float>(), c->m4.as<int32_t>() + 2);
    } else {
        rc = orbx_stereo_match_device(c, c->m0.as<orbx_keypoint>(), c->m1.as<uint32_t>(), c->m4.as<int32_t>(),
                                      c->m2.as<orbx_keypoint>(), c->m3.as<uint32_t>(), c->m4.as<int32_t>() + 1, 1, stride,
                                      nrows, mbf, mb, min_x, c->m5.as<int32_t>(), c->m6.as<int32_t>(), c->m7.as<float>(),
                                      c->m8.as<float>(), c->m4.as<int32_t>() + 2);
    }
    if (rc != ORBX_OK) return rc;
    int32_t nm = 0;
    HIPCHECK(c, d2h(c, best_idx, c->m5.p, NL * 4));
    HIPCHECK(c, d2h(c, best_dist, c->m6.p, NL * 4));
    HIPCHECK(c, d2h(c, u_right, c->m7.p, NL * 4));
    HIPCHECK(c, d2h(c, depth, c->m8.p, NL * 4));
    if (ex) HIPCHECK(c, d2h(c, listed, d_listed, NL * 4));
    HIPCHECK(c, d2h(c, &nm, c->m4.as<int32_t>() + 2, 4));
    HIPCHECK(c, finish(c));
    *nmatched = nm;
    return ORBX_OK;
}

int orbx_stereo_match(orbx_ctx* c, const orbx_keypoint* kl, const uint32_t* dl, int nl, const orbx_keypoint* kr,
                      const uint32_t* dr, int nr, int nrows, float mbf, float mb, float min_x, int32_t* best_idx,
                      int32_t* best_dist, float* u_right, float* depth, int* nmatched)
{
    return stereo_host(c, kl, dl, nl, kr, dr, nr, nrows, mbf, mb, min_x, false, nullptr, nullptr, nullptr, 75, 1, nullptr,
                       best_idx, best_dist, u_right, depth, nmatched);
}

int orbx_stereo_match_ex(orbx_ctx* c, const orbx_keypoint* kl, const uint32_t* dl, int nl, const orbx_keypoint* kr,
                         const uint32_t* dr, int nr, int nrows, float mbf, float mb, float min_x, const uint8_t* active,
                         const float* min_d, const float* max_d, int th_orb_dist, int median_cut, int32_t* listed_dist,
                         int32_t* best_idx, int32_t* best_dist, float* u_right, float* depth, int* nmatched)
{
    return stereo_host(c, kl, dl, nl, kr, dr, nr, nrows, mbf, mb, min_x, true, active, min_d, max_d, th_orb_dist,
                       median_cut, listed_dist, best_idx, best_dist, u_right, depth, nmatched);
}

int orbx_stereo_match_sad_device(orbx_ctx* c, orbx_ctx* r, const orbx_keypoint* d_kl, const uint32_t* d_dl,
                                 const int32_t* d_nl, const orbx_keypoint* d_kr, const uint32_t* d_dr,
                                 const int32_t* d_nr, int nbatch, int stride, float mbf, float mb, int32_t* d_best_idx,
                                 int32_t* d_best_dist, float* d_u_right, float* d_depth, int32_t* d_nmatched)
{
    if (!c || !r) return ORBX_ERR_INVALID;
    if (!d_kl || !d_dl || !d_nl || !d_kr || !d_dr || !d_nr || !d_best_idx || !d_best_dist || !d_u_right || !d_depth || !d_nmatched)
        return fail(c, ORBX_ERR_INVALID, "null buffer");
    if (nbatch < 1 || stride < 1) return fail(c, ORBX_ERR_INVALID, "bad sizes");
    if (!(mb > 0.0f)) return fail(c, ORBX_ERR_INVALID, "baseline must be positive");
    if (c->device != r->device) return fail(c, ORBX_ERR_INVALID, "left and right contexts are on different devices");
    if (!c->have_pyramid || !r->have_pyramid) return fail(c, ORBX_ERR_STATE, "both contexts need a computed pyramid");
    if (c->gw != r->gw || c->gh != r->gh || c->p.nlevels != r->p.nlevels || c->p.scaleFactor != r->p.scaleFactor)
        return fail(c, ORBX_ERR_INVALID, "left and right contexts differ in frame size or pyramid parameters");
    if (nbatch > c->last_nframes || nbatch > r->last_nframes) return fail(c, ORBX_ERR_INVALID, "more pairs than frames in the pyramids");
    if (c->g.lv[0].h > kStereoMaxRows) return fail(c, ORBX_ERR_UNSUPPORTED, "image taller than 16000 rows");
    HIPCHECK(c, hipSetDevice(c->device));
    {   // the SAD windows read level-0 planes too
        int rc0 = materialise_level0(c);
        if (rc0 == ORBX_OK) rc0 = materialise_level0(r);
        if (rc0 != ORBX_OK) return rc0;
    }
    if (r->stream != c->stream) {   // the right pyramid must be complete before the left stream reads it
        if (!c->ev_pyr) return fail(c, ORBX_ERR_HIP, "no event available");
        HIPCHECK(c, hipEventRecord(c->ev_pyr, r->stream));
        HIPCHECK(c, hipStreamWaitEvent(c->stream, c->ev_pyr, 0));
    }
    {
        const size_t need = stereo_classic_ints(nbatch, stride, c->g.lv[0].h) * 4;
        if (need > c->stereo_rows.bytes) {
            HIPCHECK(c, finish(c));
            HIPCHECK(c, c->stereo_rows.ensure(need));
        }
    }
    launch_stereo_classic(d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, nbatch, stride, c->pyr.as<uint8_t>(), r->pyr.as<uint8_t>(),
                          c->g, c->p.scale, c->p.inv_scale, mbf, mb, d_best_idx, d_best_dist, d_u_right, d_depth,
                          d_nmatched, c->stereo_rows.as<int32_t>(), c->stream);
    HIPCHECK(c, hipGetLastError());
    return ORBX_OK;
}

int orbx_stereo_match_sad(orbx_ctx* c, orbx_ctx* r, const orbx_keypoint* kl, const uint32_t* dl, int nl,
                          const orbx_keypoint* kr, const uint32_t* dr, int nr, float mbf, float mb, int32_t* best_idx,
                          int32_t* best_dist, float* u_right, float* depth, int* nmatched)
{
    if (!c || !r || !nmatched) return ORBX_ERR_INVALID;
    *nmatched = 0;
    if (nl < 0 || nr < 0) return fail(c, ORBX_ERR_INVALID, "negative size");
    if (nl == 0) return ORBX_OK;
    if (!kl || !dl || !best_idx || !best_dist || !u_right || !depth || (nr > 0 && (!kr || !dr)))
        return fail(c, ORBX_ERR_INVALID, "null buffer");
    for (int i = 0; i < nr; i++)
        if (kr[i].octave < 0 || kr[i].octave >= c->p.nlevels) return fail(c, ORBX_ERR_INVALID, "right keypoint octave out of range");
    for (int i = 0; i < nl; i++)
        if (kl[i].octave < 0 || kl[i].octave >= c->p.nlevels) return fail(c, ORBX_ERR_INVALID, "left keypoint octave out of range");
    const int stride = nl > nr ? nl : nr;
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, finish(c));
    HIPCHECK(c, hipStreamSynchronize(r->stream));
    const size_t S = (size_t)stride;
    HIPCHECK(c, c->m0.ensure(S * sizeof(orbx_keypoint)));
    HIPCHECK(c, c->m1.ensure(S * 32));
    HIPCHECK(c, c->m2.ensure(S * sizeof(orbx_keypoint)));
    HIPCHECK(c, c->m3.ensure(S * 32));
    HIPCHECK(c, c->m4.ensure(16));
    HIPCHECK(c, c->m5.ensure(S * 4));
    HIPCHECK(c, c->m6.ensure(S * 4));
    HIPCHECK(c, c->m7.ensure(S * 4));
    HIPCHECK(c, c->m8.ensure(S * 4));
    int32_t counts[3] = {nl, nr, 0};
    HIPCHECK(c, h2d(c, c->m0.p, kl, (size_t)nl * sizeof(orbx_keypoint)));
    HIPCHECK(c, h2d(c, c->m1.p, dl, (size_t)nl * 32));
    if (nr > 0) {
        HIPCHECK(c, h2d(c, c->m2.p, kr, (size_t)nr * sizeof(orbx_keypoint)));
        HIPCHECK(c, h2d(c, c->m3.p, dr, (size_t)nr * 32));
    }
    HIPCHECK(c, h2d(c, c->m4.p, counts, sizeof(counts)));
    int rc = orbx_stereo_match_sad_device(c, r, c->m0.as<orbx_keypoint>(), c->m1.as<uint32_t>(), c->m4.as<int32_t>(),
                                          c->m2.as<orbx_keypoint>(), c->m3.as<uint32_t>(), c->m4.as<int32_t>() + 1, 1,
                                          stride, mbf, mb, c->m5.as<int32_t>(), c->m6.as<int32_t>(), c->m7.as<float>(),
                                          c->m8.as<float>(), c->m4.as<int32_t>() + 2);
    if (rc != ORBX_OK) return rc;
    int32_t nm = 0;
    HIPCHECK(c, d2h(c, best_idx, c->m5.p, (size_t)nl * 4));
    HIPCHECK(c, d2h(c, best_dist, c->m6.p, (size_t)nl * 4));
    HIPCHECK(c, d2h(c, u_right, c->m7.p, (size_t)nl * 4));
    HIPCHECK(c, d2h(c, depth, c->m8.p, (size_t)nl * 4));
    HIPCHECK(c, d2h(c, &nm, c->m4.as<int32_t>() + 2, 4));
    HIPCHECK(c, finish(c));
    *nmatched = nm;
    return ORBX_OK;
}

static int grid_prepare(orbx_ctx* c, const orbx_keypoint* d_kps, const int32_t* d_n, int N, int stride, int nbatch,
                        float min_x, float min_y, float inv_w, float inv_h)
{
    const size_t need_s = (size_t)nbatch * (kFrameGridCells + 1) * 4, need_l = (size_t)nbatch * stride * 4;
    if (need_s > c->grid_start.bytes || need_l > c->grid_list.bytes) {
        HIPCHECK(c, finish(c));
        HIPCHECK(c, c->grid_start.ensure(need_s));
        HIPCHECK(c, c->grid_list.ensure(need_l));
    }
    launch_grid_build(d_kps, d_n, N, stride, nbatch, min_x, min_y, inv_w, inv_h, c->grid_start.as<int>(),
                      c->grid_list.as<int>(), c->stream);
    return ORBX_OK;
}

int orbx_grid_search_device(orbx_ctx* c, const orbx_keypoint* d_kps, const uint32_t* d_desc, const int32_t* d_n, int N,
                            int stride, int nbatch, float min_x, float min_y, float inv_w, float inv_h, const float* d_qx,
                            const float* d_qy, const float* d_qr, const int32_t* d_qmin, const int32_t* d_qmax,
                            const uint32_t* d_qdesc, const int32_t* d_nq, int Q, int q_stride, int32_t* d_best_idx,
                            int32_t* d_best_dist, int32_t* d_second_dist, int32_t* d_best_lvl, int32_t* d_second_lvl,
                            int32_t* d_ncand)
{
    if (!c) return ORBX_ERR_INVALID;
    if (!d_kps || !d_desc || !d_qx || !d_qy || !d_qr || !d_qmin || !d_qmax || !d_qdesc || !d_best_idx || !d_best_dist || !d_second_dist)
        return fail(c, ORBX_ERR_INVALID, "null buffer");
    if (N < 0 || stride < 1 || N > stride || nbatch < 1 || nbatch > 65535 || Q < 0 || q_stride < Q)
        return fail(c, ORBX_ERR_INVALID, "bad sizes");
    if (q_stride == 0) return ORBX_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    int rc = grid_prepare(c, d_kps, d_n, N, stride, nbatch, min_x, min_y, inv_w, inv_h);
    if (rc != ORBX_OK) return rc;
    launch_grid_search(d_kps, d_desc, c->grid_start.as<int>(), c->grid_list.as<int>(), stride, nbatch, d_qx, d_qy, d_qr,
                       d_qmin, d_qmax, d_qdesc, d_nq, Q, q_stride, min_x, min_y, inv_w, inv_h, d_best_idx, d_best_dist,
                       d_second_dist, d_best_lvl, d_second_lvl, d_ncand, nullptr, nullptr, c->stream);
    HIPCHECK(c, hipGetLastError());
    return ORBX_OK;
}

// uploads one frame's keypoints and Q queries into the gq[] scratch buffers
static int grid_upload(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, int N, const float* qx, const float* qy,
                       const float* qr, const int32_t* qmin, const int32_t* qmax, const uint32_t* qdesc, int Q)
{
    // a frame without keypoints may come as a null pointer (the data() of an empty std::vector)
    if ((N > 0 && !kps) || !qx || !qy || !qr || !qmin || !qmax) return fail(c, ORBX_ERR_INVALID, "null buffer");
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, finish(c));
    const size_t n = (size_t)(N > 0 ? N : 1), q = (size_t)Q;
    HIPCHECK(c, c->gq[0].ensure(n * sizeof(orbx_keypoint)));
    HIPCHECK(c, c->gq[1].ensure(n * 32));
    for (int k = 2; k <= 6; k++) HIPCHECK(c, c->gq[k].ensure(q * 4));
    HIPCHECK(c, c->gq[7].ensure(q * 32));
    for (int k = 8; k <= 11; k++) HIPCHECK(c, c->gq[k].ensure((q + 1) * 4 * 2));
    if (N > 0) HIPCHECK(c, h2d(c, c->gq[0].p, kps, (size_t)N * sizeof(orbx_keypoint)));
    if (N > 0 && desc) HIPCHECK(c, h2d(c, c->gq[1].p, desc, (size_t)N * 32));
    HIPCHECK(c, h2d(c, c->gq[2].p, qx, q * 4));
    HIPCHECK(c, h2d(c, c->gq[3].p, qy, q * 4));
    HIPCHECK(c, h2d(c, c->gq[4].p, qr, q * 4));
    HIPCHECK(c, h2d(c, c->gq[5].p, qmin, q * 4));
    HIPCHECK(c, h2d(c, c->gq[6].p, qmax, q * 4));
    if (qdesc) HIPCHECK(c, h2d(c, c->gq[7].p, qdesc, q * 32));
    return ORBX_OK;
}

int orbx_grid_search(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, int N, float min_x, float min_y,
                     float inv_w, float inv_h, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                     const int32_t* qmax, const uint32_t* qdesc, int Q, int32_t* best_idx, int32_t* best_dist,
                     int32_t* second_dist, int32_t* best_lvl, int32_t* second_lvl, int32_t* ncand)
{
    if (!c) return ORBX_ERR_INVALID;
    if (N < 0 || Q < 0) return fail(c, ORBX_ERR_INVALID, "negative size");
    if (Q == 0) return ORBX_OK;
    if (!qdesc || !best_idx || !best_dist || !second_dist || (N > 0 && !desc)) return fail(c, ORBX_ERR_INVALID, "null buffer");
    int rc = grid_upload(c, kps, desc, N, qx, qy, qr, qmin, qmax, qdesc, Q);
    if (rc != ORBX_OK) return rc;
    const int stride = N > 0 ? N : 1;
    int32_t* o = c->gq[8].as<int32_t>();     // best_idx | best_dist
    int32_t* o2 = c->gq[9].as<int32_t>();    // second_dist | best_lvl
    int32_t* o3 = c->gq[10].as<int32_t>();   // second_lvl | ncand
    rc = orbx_grid_search_device(c, c->gq[0].as<orbx_keypoint>(), c->gq[1].as<uint32_t>(), nullptr, N, stride, 1, min_x,
                                 min_y, inv_w, inv_h, c->gq[2].as<float>(), c->gq[3].as<float>(), c->gq[4].as<float>(),
                                 c->gq[5].as<int32_t>(), c->gq[6].as<int32_t>(), c->gq[7].as<uint32_t>(), nullptr, Q, Q,
                                 o, o + Q + 1, o2, o2 + Q + 1, o3, o3 + Q + 1);
    if (rc != ORBX_OK) return rc;
    const size_t qb = (size_t)Q * 4;
    HIPCHECK(c, d2h(c, best_idx, o, qb));
    HIPCHECK(c, d2h(c, best_dist, o + Q + 1, qb));
    HIPCHECK(c, d2h(c, second_dist, o2, qb));
    if (best_lvl) HIPCHECK(c, d2h(c, best_lvl, o2 + Q + 1, qb));
    if (second_lvl) HIPCHECK(c, d2h(c, second_lvl, o3, qb));
    if (ncand) HIPCHECK(c, d2h(c, ncand, o3 + Q + 1, qb));
    HIPCHECK(c, finish(c));
    return ORBX_OK;
}

static int grid_lists(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, int N, const float* qx, const float* qy,
                      const float* qr, const int32_t* qmin, const int32_t* qmax, const uint32_t* qdesc, int Q, float min_x,
                      float min_y, float inv_w, float inv_h, std::vector<int32_t>& off, std::vector<int32_t>& idx,
                      std::vector<uint16_t>& dist);

int orbx_features_in_area(orbx_ctx* c, const orbx_keypoint* kps, int N, float min_x, float min_y, float inv_w, float inv_h,
                          const float* qx, const float* qy, const float* qr, const int32_t* qmin, const int32_t* qmax, int Q,
                          int32_t* cand_off, int32_t* cand_idx, int cap, int* total)
{
    if (!c || !total) return ORBX_ERR_INVALID;
    *total = 0;
    if (N < 0 || Q < 0 || !cand_off) return fail(c, ORBX_ERR_INVALID, "bad arguments");
    cand_off[0] = 0;
    if (Q == 0) return ORBX_OK;
    if (N > 0 && !getenv("ORBX_NO_ZERO_COPY")) {
        // the lists by 16 lanes per query, one upload block, small arrays read in place (as the whole-routine searches do)
        std::vector<int32_t> off, idx;
        std::vector<uint16_t> none;
        int rc = grid_lists(c, kps, nullptr, N, qx, qy, qr, qmin, qmax, nullptr, Q, min_x, min_y, inv_w, inv_h, off, idx, none);
        if (rc != ORBX_OK) return rc;
        std::memcpy(cand_off, off.data(), ((size_t)Q + 1) * 4);
        *total = off[Q];
        if (off[Q] == 0) return ORBX_OK;
        if (!cand_idx || cap < off[Q]) return fail(c, ORBX_ERR_CAPACITY, "candidate buffer too small");
        std::memcpy(cand_idx, idx.data(), (size_t)off[Q] * 4);
        return ORBX_OK;
    }
    int rc = grid_upload(c, kps, nullptr, N, qx, qy, qr, qmin, qmax, nullptr, Q);
    if (rc != ORBX_OK) return rc;
    const int stride = N > 0 ? N : 1;
    rc = grid_prepare(c, c->gq[0].as<orbx_keypoint>(), nullptr, N, stride, 1, min_x, min_y, inv_w, inv_h);
    if (rc != ORBX_OK) return rc;
    int32_t* d_cnt = c->gq[8].as<int32_t>();
    int32_t* d_off = c->gq[9].as<int32_t>();
    // pass 1: count
    launch_grid_search(c->gq[0].as<orbx_keypoint>(), nullptr, c->grid_start.as<int>(), c->grid_list.as<int>(), stride, 1,
                       c->gq[2].as<float>(), c->gq[3].as<float>(), c->gq[4].as<float>(), c->gq[5].as<int32_t>(),
                       c->gq[6].as<int32_t>(), nullptr, nullptr, Q, Q, min_x, min_y, inv_w, inv_h, nullptr, nullptr, nullptr,
                       nullptr, nullptr, d_cnt, nullptr, nullptr, c->stream);
    HIPCHECK(c, hipGetLastError());
    std::vector<int32_t> cnt(Q);
    HIPCHECK(c, d2h(c, cnt.data(), d_cnt, (size_t)Q * 4));
    HIPCHECK(c, finish(c));
    long long sum = 0;
    for (int q = 0; q < Q; q++) { cand_off[q] = (int32_t)sum; sum += cnt[q]; }
    if (sum > 0x7fffffffLL) return fail(c, ORBX_ERR_UNSUPPORTED, "more than 2^31 candidates");
    cand_off[Q] = (int32_t)sum;
    *total = (int)sum;
    if (sum == 0) return ORBX_OK;
    if (!cand_idx || cap < sum) return fail(c, ORBX_ERR_CAPACITY, "candidate buffer too small");
    // pass 2: fill
    HIPCHECK(c, c->gq[11].ensure((size_t)sum * 4));
    HIPCHECK(c, h2d(c, d_off, cand_off, (size_t)(Q + 1) * 4));
    launch_grid_search(c->gq[0].as<orbx_keypoint>(), nullptr, c->grid_start.as<int>(), c->grid_list.as<int>(), stride, 1,
                       c->gq[2].as<float>(), c->gq[3].as<float>(), c->gq[4].as<float>(), c->gq[5].as<int32_t>(),
                       c->gq[6].as<int32_t>(), nullptr, nullptr, Q, Q, min_x, min_y, inv_w, inv_h, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr, d_off, c->gq[11].as<int32_t>(), c->stream);
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, d2h(c, cand_idx, c->gq[11].p, (size_t)sum * 4));
    HIPCHECK(c, finish(c));
    return ORBX_OK;
}

// ---------------------------------------------------------------------------
// Searches with sequential exclusion (src/ORBmatcher.cc:155-241, :520-633): a query's outcome depends on what
// earlier queries of the same call claimed, so the device forms every candidate list of the call
// (Frame::GetFeaturesInArea) with the Hamming distance of every candidate, and the host walks the lists in query
// order exactly as the reference's loops do -- the walk reads distances, it computes none.
// ---------------------------------------------------------------------------
// after grid_upload(): CSR lists (off[Q+1], idx[]) in the reference's visiting order + dist[] per candidate
static int grid_candidates(orbx_ctx* c, int N, int Q, float min_x, float min_y, float inv_w, float inv_h,
                           std::vector<int32_t>& off, std::vector<int32_t>& idx, std::vector<uint16_t>& dist)
{
    const int stride = N > 0 ? N : 1;
    int rc = grid_prepare(c, c->gq[0].as<orbx_keypoint>(), nullptr, N, stride, 1, min_x, min_y, inv_w, inv_h);
    if (rc != ORBX_OK) return rc;
    int32_t* d_cnt = c->gq[8].as<int32_t>();
    int32_t* d_off = c->gq[9].as<int32_t>();
    launch_grid_search(c->gq[0].as<orbx_keypoint>(), nullptr, c->grid_start.as<int>(), c->grid_list.as<int>(), stride, 1,
                       c->gq[2].as<float>(), c->gq[3].as<float>(), c->gq[4].as<float>(), c->gq[5].as<int32_t>(),
                       c->gq[6].as<int32_t>(), nullptr, nullptr, Q, Q, min_x, min_y, inv_w, inv_h, nullptr, nullptr, nullptr,
                       nullptr, nullptr, d_cnt, nullptr, nullptr, c->stream);
    HIPCHECK(c, hipGetLastError());
    std::vector<int32_t> cnt(Q);
    HIPCHECK(c, d2h(c, cnt.data(), d_cnt, (size_t)Q * 4));
    HIPCHECK(c, finish(c));
    off.assign((size_t)Q + 1, 0);
    long long sum = 0;
    for (int q = 0; q < Q; q++) { off[q] = (int32_t)sum; sum += cnt[q]; }
    if (sum > 0x7fffffffLL) return fail(c, ORBX_ERR_UNSUPPORTED, "more than 2^31 candidates");
    off[Q] = (int32_t)sum;
    idx.assign((size_t)sum, 0);
    dist.assign((size_t)sum, 0);
    if (sum == 0) return ORBX_OK;
    HIPCHECK(c, c->gq[11].ensure((size_t)sum * 4));
    HIPCHECK(c, c->m2.ensure((size_t)sum * 2));
    HIPCHECK(c, h2d(c, d_off, off.data(), (size_t)(Q + 1) * 4));
    launch_grid_search(c->gq[0].as<orbx_keypoint>(), c->gq[1].as<uint32_t>(), c->grid_start.as<int>(), c->grid_list.as<int>(),
                       stride, 1, c->gq[2].as<float>(), c->gq[3].as<float>(), c->gq[4].as<float>(), c->gq[5].as<int32_t>(),
                       c->gq[6].as<int32_t>(), c->gq[7].as<uint32_t>(), nullptr, Q, Q, min_x, min_y, inv_w, inv_h, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr, d_off, c->gq[11].as<int32_t>(), c->stream,
                       c->m2.as<uint16_t>());
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, d2h(c, idx.data(), c->gq[11].p, (size_t)sum * 4));
    HIPCHECK(c, d2h(c, dist.data(), c->m2.p, (size_t)sum * 2));
    HIPCHECK(c, finish(c));
    return ORBX_OK;
}

// One frame's keypoints and Q window queries -> the queries' candidate lists with every distance (what grid_upload +
// grid_candidates do), with as few copy commands as a call of this size can have: a search of 1000 queries is a few dozen
// microseconds of kernels, and every copy command between host and device is a dependent step of 5-10 us.  The five
// small query arrays are read by the kernels where they lie in the page-locked staging area, the counts are written
// there by the count pass and the offsets read from there by the fill pass; only the three arrays the kernels gather
// from (keypoints, descriptors, query descriptors) and the two lists go through copies.
static int grid_lists(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, int N, const float* qx, const float* qy,
                      const float* qr, const int32_t* qmin, const int32_t* qmax, const uint32_t* qdesc, int Q, float min_x,
                      float min_y, float inv_w, float inv_h, std::vector<int32_t>& off, std::vector<int32_t>& idx,
                      std::vector<uint16_t>& dist)
{
    static const bool plain = getenv("ORBX_NO_ZERO_COPY") != nullptr;
    if (plain || Q <= 0 || N <= 0) {
        int rc = grid_upload(c, kps, desc, N, qx, qy, qr, qmin, qmax, qdesc, Q);
        if (rc != ORBX_OK) return rc;
        return grid_candidates(c, N, Q, min_x, min_y, inv_w, inv_h, off, idx, dist);
    }
    if (!kps || !qx || !qy || !qr || !qmin || !qmax || (desc != nullptr) != (qdesc != nullptr)) return fail(c, ORBX_ERR_INVALID, "null buffer");
    const bool with_dist = desc != nullptr;      // without descriptors: the lists alone (Frame::GetFeaturesInArea)
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, finish(c));
    const size_t q = (size_t)Q;
    // keypoints | descriptors | query descriptors: one block on the device, one copy
    const size_t kb = ((size_t)N * sizeof(orbx_keypoint) + 63) & ~(size_t)63, db = with_dist ? (size_t)N * 32 : 0, qdb = with_dist ? q * 32 : 0;
    HIPCHECK(c, c->gq[0].ensure(kb + db + qdb));
    {
        const size_t need_s = (size_t)(kFrameGridCells + 1) * 4, need_l = (size_t)N * 4;
        HIPCHECK(c, c->grid_start.ensure(need_s));
        HIPCHECK(c, c->grid_list.ensure(need_l));
    }
    // the staging block of this call: the device block's image, then the arrays the kernels read and write in place
    size_t o;
    const size_t qb = (q * 4 + 63) & ~(size_t)63, ob = ((q + 1) * 4 + 63) & ~(size_t)63;
    HIPCHECK(c, arena_reserve(c, kb + db + qdb + 6 * qb + ob, &o));
    std::memcpy(c->pin + o, kps, (size_t)N * sizeof(orbx_keypoint));
    if (with_dist) {
        std::memcpy(c->pin + o + kb, desc, db);
        std::memcpy(c->pin + o + kb + db, qdesc, qdb);
    }
    HIPCHECK(c, hipMemcpyAsync(c->gq[0].p, c->pin + o, kb + db + qdb, hipMemcpyHostToDevice, c->stream));
    uint8_t* const dblk = c->gq[0].as<uint8_t>();
    const uint32_t* const d_desc = with_dist ? reinterpret_cast<const uint32_t*>(dblk + kb) : nullptr;
    const uint32_t* const d_qdesc = with_dist ? reinterpret_cast<const uint32_t*>(dblk + kb + db) : nullptr;
    uint8_t* blk = c->pin + o + kb + db + qdb;
    float* z_qx = reinterpret_cast<float*>(blk);
    float* z_qy = reinterpret_cast<float*>(blk + qb);
    float* z_qr = reinterpret_cast<float*>(blk + 2 * qb);
    int32_t* z_qmin = reinterpret_cast<int32_t*>(blk + 3 * qb);
    int32_t* z_qmax = reinterpret_cast<int32_t*>(blk + 4 * qb);
    int32_t* z_cnt = reinterpret_cast<int32_t*>(blk + 5 * qb);
    int32_t* z_off = reinterpret_cast<int32_t*>(blk + 6 * qb);
    std::memcpy(z_qx, qx, q * 4);
    std::memcpy(z_qy, qy, q * 4);
    std::memcpy(z_qr, qr, q * 4);
    std::memcpy(z_qmin, qmin, q * 4);
    std::memcpy(z_qmax, qmax, q * 4);
    const orbx_keypoint* d_kps = c->gq[0].as<orbx_keypoint>();
    launch_grid_build(d_kps, nullptr, N, N, 1, min_x, min_y, inv_w, inv_h, c->grid_start.as<int>(), c->grid_list.as<int>(), c->stream);
    launch_grid_lists(d_kps, nullptr, c->grid_start.as<int>(), c->grid_list.as<int>(), N, 1, z_qx, z_qy, z_qr, z_qmin, z_qmax,
                      nullptr, nullptr, Q, Q, min_x, min_y, inv_w, inv_h, z_cnt, nullptr, nullptr, nullptr, c->stream);
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipStreamSynchronize(c->stream));      // not finish(): the staging block stays reserved
    off.assign(q + 1, 0);
    long long sum = 0;
    for (int k = 0; k < Q; k++) { off[k] = (int32_t)sum; sum += z_cnt[k]; }
    if (sum > 0x7fffffffLL) { (void)finish(c); return fail(c, ORBX_ERR_UNSUPPORTED, "more than 2^31 candidates"); }
    off[Q] = (int32_t)sum;
    idx.assign((size_t)sum, 0);
    dist.assign(with_dist ? (size_t)sum : 0, 0);
    if (sum == 0) { HIPCHECK(c, finish(c)); return ORBX_OK; }
    std::memcpy(z_off, off.data(), (q + 1) * 4);
    // the lists: indices | distances in one block, one copy back
    const size_t lb = (size_t)sum * (with_dist ? 6 : 4);
    HIPCHECK(c, c->gq[11].ensure(lb));
    int32_t* const d_idx = c->gq[11].as<int32_t>();
    uint16_t* const d_dist = reinterpret_cast<uint16_t*>(c->gq[11].as<uint8_t>() + (size_t)sum * 4);
    launch_grid_lists(d_kps, d_desc, c->grid_start.as<int>(), c->grid_list.as<int>(), N, 1, z_qx, z_qy, z_qr, z_qmin, z_qmax,
                      d_qdesc, nullptr, Q, Q, min_x, min_y, inv_w, inv_h, nullptr, z_off, d_idx, d_dist, c->stream);
    HIPCHECK(c, hipGetLastError());
    // (a reservation that has to enlarge the staging area drains the stream first: the fill pass is done with the block by then)
    size_t oo;
    HIPCHECK(c, arena_reserve(c, lb, &oo));
    HIPCHECK(c, hipMemcpyAsync(c->pin + oo, c->gq[11].p, lb, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    std::memcpy(idx.data(), c->pin + oo, (size_t)sum * 4);
    if (with_dist) std::memcpy(dist.data(), c->pin + oo + (size_t)sum * 4, (size_t)sum * 2);
    HIPCHECK(c, finish(c));
    return ORBX_OK;
}

// ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:1723-1764
// Bin of the rotation histogram (src/ORBmatcher.cc:590-594): round(rot * factor), bin HISTO_LENGTH wraps to 0; the reference
// asserts the range.  False for an angle difference that is not a number or far outside (a conversion of such a value
// to int is undefined), and for a bin outside the histogram.
static bool rotation_bin(float scaled, int nbins, int& bin)
{
    if (!(scaled > -1.0e6f && scaled < 1.0e6f)) return false;
    bin = (int)roundf(scaled);
    if (bin == nbins) bin = 0;
    return bin >= 0 && bin < nbins;
}

static void three_maxima(const std::vector<int32_t>* histo, int L, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) ind3 = -1;
}

int orbx_search_for_initialization(orbx_ctx* c, const orbx_keypoint* k1, const uint32_t* d1, int n1,
                                   const orbx_keypoint* k2, const uint32_t* d2, int n2, float min_x, float min_y,
                                   float inv_w, float inv_h, float* prev_matched_xy, int window_size, float nnratio,
                                   int check_orientation, int32_t* matches12, int* nmatches)
{
    if (!c || !nmatches) return ORBX_ERR_INVALID;
    *nmatches = 0;
    if (n1 < 0 || n2 < 0) return fail(c, ORBX_ERR_INVALID, "negative size");
    if (n1 == 0) return ORBX_OK;
    if (!k1 || !d1 || !prev_matched_xy || !matches12 || (n2 > 0 && (!k2 || !d2))) return fail(c, ORBX_ERR_INVALID, "null buffer");
    constexpr int kHisto = 30, kThLow = 50;   // HISTO_LENGTH, TH_LOW (src/ORBmatcher.cc:37-39)
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    // the queries: keypoints of F1 on level 0 (:536-538), window around vbPrevMatched[i1] on level 0 of F2 (:540)
    std::vector<int32_t> qi;
    qi.reserve(n1);
    for (int i = 0; i < n1; i++)
        if (k1[i].octave <= 0) qi.push_back(i);
    const int Q = (int)qi.size();
    if (Q == 0 || n2 == 0) return ORBX_OK;
    std::vector<float> qx(Q), qy(Q), qr(Q, (float)window_size);
    std::vector<int32_t> ql(Q);
    std::vector<uint32_t> qd((size_t)Q * 8);
    for (int q = 0; q < Q; q++) {
        const int i = qi[q];
        qx[q] = prev_matched_xy[2 * i];
        qy[q] = prev_matched_xy[2 * i + 1];
        ql[q] = k1[i].octave;
        std::memcpy(&qd[(size_t)q * 8], d1 + (size_t)i * 8, 32);
    }
    std::vector<int32_t> off, idx;
    std::vector<uint16_t> dist;
    int rc = grid_lists(c, k2, d2, n2, qx.data(), qy.data(), qr.data(), ql.data(), ql.data(), qd.data(), Q, min_x, min_y, inv_w,
                        inv_h, off, idx, dist);
    if (rc != ORBX_OK) return rc;

    int nm = 0;
    std::vector<int32_t> rotHist[kHisto];
    const float factor = 1.0f / kHisto;
    std::vector<int> vMatchedDistance(n2, 0x7fffffff), vnMatches21(n2, -1);
    for (int q = 0; q < Q; q++) {
        const int i1 = qi[q];
        if (off[q + 1] == off[q]) continue;                                   // :542-543
        int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx2 = -1;
        for (int k = off[q]; k < off[q + 1]; k++) {
            const int i2 = idx[k], d = dist[k];
            if (vMatchedDistance[i2] <= d) continue;                          // :559-560
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestIdx2 = i2; }
            else if (d < bestDist2) bestDist2 = d;
        }
        if (bestDist <= kThLow && bestDist < (float)bestDist2 * nnratio) {    // :574-576
            if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; nm--; }   // :578-582
            matches12[i1] = bestIdx2;
            vnMatches21[bestIdx2] = i1;
            vMatchedDistance[bestIdx2] = bestDist;
            nm++;
            if (check_orientation) {                                          // :588-598
                float rot = k1[i1].angle - k2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = 0;
                if (!rotation_bin(rot * factor, kHisto, bin)) return fail(c, ORBX_ERR_INVALID, "keypoint angles outside [0, 360)");
                rotHist[bin].push_back(i1);
            }
        }
    }
    if (check_orientation) {                                                  // :603-625
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, kHisto, ind1, ind2, ind3);
        for (int i = 0; i < kHisto; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int i1 : rotHist[i])
                if (matches12[i1] >= 0) { matches12[i1] = -1; nm--; }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)                                           // :628-630
        if (matches12[i1] >= 0) {
            prev_matched_xy[2 * i1] = k2[matches12[i1]].x;
            prev_matched_xy[2 * i1 + 1] = k2[matches12[i1]].y;
        }
    *nmatches = nm;
    return ORBX_OK;
}

// The queries of a projection search over map points (src/ORBmatcher.cc:161-179 = :57-75 = include/ORBmatcher.h:151-170):
// the points in view that are not bad, window r * mvScaleFactors[level] on levels [level - 1, level]; then every
// candidate list of the call with its distances.  qi = map point of each query, rr = its r.
static int projection_queries(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, int N, float min_x, float min_y,
                              float inv_w, float inv_h, const uint8_t* mp_track_in_view, const uint8_t* mp_is_bad,
                              const float* mp_proj_x, const float* mp_proj_y, const int32_t* mp_level,
                              const float* mp_view_cos, const uint32_t* mp_desc, int M, float th, std::vector<int32_t>& qi,
                              std::vector<float>& rr, std::vector<int32_t>& off, std::vector<int32_t>& idx,
                              std::vector<uint16_t>& dist)
{
    const bool bFactor = th != 1.0;                   // :159
    qi.clear();
    qi.reserve(M);
    for (int i = 0; i < M; i++) {
        if (!mp_track_in_view[i] || (mp_is_bad && mp_is_bad[i])) continue;
        if (mp_level[i] < 0 || mp_level[i] >= c->p.nlevels) return fail(c, ORBX_ERR_INVALID, "predicted level out of range");
        qi.push_back(i);
    }
    const int Q = (int)qi.size();
    off.assign(1, 0);
    idx.clear();
    dist.clear();
    rr.assign(Q, 0.0f);
    if (Q == 0) return ORBX_OK;
    std::vector<float> qx(Q), qy(Q), qr(Q);
    std::vector<int32_t> qmin(Q), qmax(Q);
    std::vector<uint32_t> qd((size_t)Q * 8);
    for (int q = 0; q < Q; q++) {
        const int i = qi[q];
        float r = mp_view_cos[i] > 0.998 ? 2.5f : 4.0f;       // RadiusByViewingCos, :243-249
        if (bFactor) r *= th;
        rr[q] = r;
        qx[q] = mp_proj_x[i];
        qy[q] = mp_proj_y[i];
        qr[q] = r * c->p.scale[mp_level[i]];
        qmin[q] = mp_level[i] - 1;
        qmax[q] = mp_level[i];
        std::memcpy(&qd[(size_t)q * 8], mp_desc + (size_t)i * 8, 32);
    }
    return grid_lists(c, kps, desc, N, qx.data(), qy.data(), qr.data(), qmin.data(), qmax.data(), qd.data(), Q, min_x, min_y, inv_w,
                      inv_h, off, idx, dist);
}

// One map point against its candidate list: the scan and the accept step shared by SearchByProjection
// (src/ORBmatcher.cc:186-237), SearchByProjection_Budget (:82-138), SearchByProjection_OnePoint and MatchCandidates
// (include/ORBmatcher.h:94-148, :196-251).  Returns bestIdx; -1 when the best distance is above TH_HIGH (or nothing
// was scanned), -2 when the ratio test rejected the best.
static int projection_accept(const orbx_keypoint* kps, const float* u_right, const int32_t* idx, const uint16_t* dist, int n,
                             float proj_xr, float r_scaled, float nnratio, int mp_index, int mp_observations,
                             int32_t* frame_obs, int32_t* frame_match, int32_t* match_score)
{
    constexpr int kThHigh = 100;                      // TH_HIGH, src/ORBmatcher.cc:38
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int k = 0; k < n; k++) {
        const int i = idx[k];
        if (frame_obs[i] > 0) continue;                                       // :197-199
        if (u_right && u_right[i] > 0) {                                      // :201-206
            const float er = fabsf(proj_xr - u_right[i]);
            if (er > r_scaled) continue;
        }
        const int d = dist[k];
        if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = kps[i].octave; bestIdx = i; }
        else if (d < bestDist2) { bestLevel2 = kps[i].octave; bestDist2 = d; }
    }
    if (bestDist <= kThHigh) {                                                // :228
        if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) return -2;   // :230-231
        frame_match[bestIdx] = mp_index;                                      // :233
        frame_obs[bestIdx] = mp_observations;
        match_score[bestIdx] = bestDist;                                      // :235
        return bestIdx;
    }
    return -1;
}

static int projection_search(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, const float* u_right, int N,
                             float min_x, float min_y, float inv_w, float inv_h, const uint8_t* mp_track_in_view,
                             const uint8_t* mp_is_bad, const float* mp_proj_x, const float* mp_proj_y,
                             const float* mp_proj_xr, const int32_t* mp_level, const float* mp_view_cos,
                             const uint32_t* mp_desc, const int32_t* mp_observations, int M, float th, float nnratio,
                             const double* time_constr_ms, int32_t* frame_obs, int32_t* frame_match, int32_t* match_score,
                             int* nmatches, int* stopped_at)
{
    if (!c || !nmatches) return ORBX_ERR_INVALID;
    const auto t0 = std::chrono::steady_clock::now();                         // timer.tic(), src/ORBmatcher.cc:54-55
    *nmatches = 0;
    if (stopped_at) *stopped_at = M;
    if (N < 0 || M < 0) return fail(c, ORBX_ERR_INVALID, "negative size");
    if (M == 0 || N == 0) return ORBX_OK;
    if (!kps || !desc || !mp_track_in_view || !mp_proj_x || !mp_proj_y || !mp_level || !mp_view_cos || !mp_desc ||
        !mp_observations || !frame_obs || !frame_match || !match_score || (u_right && !mp_proj_xr))
        return fail(c, ORBX_ERR_INVALID, "null buffer");
    std::vector<int32_t> qi, off, idx;
    std::vector<float> rr;
    std::vector<uint16_t> dist;
    int rc = projection_queries(c, kps, desc, N, min_x, min_y, inv_w, inv_h, mp_track_in_view, mp_is_bad, mp_proj_x,
                                mp_proj_y, mp_level, mp_view_cos, mp_desc, M, th, qi, rr, off, idx, dist);
    if (rc != ORBX_OK) return rc;
    const int Q = (int)qi.size();
    int nm = 0;
    for (int q = 0; q < Q; q++) {
        const int iMP = qi[q];
        if (off[q + 1] == off[q]) continue;                                   // :181-182
        const int got = projection_accept(kps, u_right, idx.data() + off[q], dist.data() + off[q], off[q + 1] - off[q],
                                          mp_proj_xr ? mp_proj_xr[iMP] : 0.0f, rr[q] * c->p.scale[mp_level[iMP]], nnratio, iMP,
                                          mp_observations[iMP], frame_obs, frame_match, match_score);
        if (got >= 0) nm++;
        else if (got == -2) continue;      // rejected by the ratio test: the reference's `continue` (:126-127) skips its clock
        if (time_constr_ms) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms >= *time_constr_ms) {
                if (stopped_at) *stopped_at = iMP;
                break;
            }
        }
    }
    *nmatches = nm;
    return ORBX_OK;
}

int orbx_search_by_projection(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, const float* u_right, int N,
                              float min_x, float min_y, float inv_w, float inv_h, const uint8_t* mp_track_in_view,
                              const uint8_t* mp_is_bad, const float* mp_proj_x, const float* mp_proj_y,
                              const float* mp_proj_xr, const int32_t* mp_level, const float* mp_view_cos,
                              const uint32_t* mp_desc, const int32_t* mp_observations, int M, float th, float nnratio,
                              int32_t* frame_obs, int32_t* frame_match, int32_t* match_score, int* nmatches)
{
    return projection_search(c, kps, desc, u_right, N, min_x, min_y, inv_w, inv_h, mp_track_in_view, mp_is_bad, mp_proj_x,
                             mp_proj_y, mp_proj_xr, mp_level, mp_view_cos, mp_desc, mp_observations, M, th, nnratio, nullptr,
                             frame_obs, frame_match, match_score, nmatches, nullptr);
}

int orbx_search_by_projection_budget(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, const float* u_right, int N,
                                     float min_x, float min_y, float inv_w, float inv_h, const uint8_t* mp_track_in_view,
                                     const uint8_t* mp_is_bad, const float* mp_proj_x, const float* mp_proj_y,
                                     const float* mp_proj_xr, const int32_t* mp_level, const float* mp_view_cos,
                                     const uint32_t* mp_desc, const int32_t* mp_observations, int M, float th, float nnratio,
                                     double time_constr_ms, int32_t* frame_obs, int32_t* frame_match, int32_t* match_score,
                                     int* nmatches, int* stopped_at)
{
    return projection_search(c, kps, desc, u_right, N, min_x, min_y, inv_w, inv_h, mp_track_in_view, mp_is_bad, mp_proj_x,
                             mp_proj_y, mp_proj_xr, mp_level, mp_view_cos, mp_desc, mp_observations, M, th, nnratio,
                             &time_constr_ms, frame_obs, frame_match, match_score, nmatches, stopped_at);
}

int orbx_projection_candidates(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, int N, float min_x, float min_y,
                               float inv_w, float inv_h, const uint8_t* mp_track_in_view, const uint8_t* mp_is_bad,
                               const float* mp_proj_x, const float* mp_proj_y, const int32_t* mp_level,
                               const float* mp_view_cos, const uint32_t* mp_desc, int M, float th, int32_t* cand_off,
                               int32_t* cand_idx, uint16_t* cand_dist, int cap, int* total)
{
    if (!c || !total) return ORBX_ERR_INVALID;
    *total = 0;
    if (N < 0 || M < 0 || !cand_off) return fail(c, ORBX_ERR_INVALID, "bad arguments");
    for (int i = 0; i <= M; i++) cand_off[i] = 0;
    if (M == 0 || N == 0) return ORBX_OK;
    if (!kps || !desc || !mp_track_in_view || !mp_proj_x || !mp_proj_y || !mp_level || !mp_view_cos || !mp_desc)
        return fail(c, ORBX_ERR_INVALID, "null buffer");
    std::vector<int32_t> qi, off, idx;
    std::vector<float> rr;
    std::vector<uint16_t> dist;
    int rc = projection_queries(c, kps, desc, N, min_x, min_y, inv_w, inv_h, mp_track_in_view, mp_is_bad, mp_proj_x,
                                mp_proj_y, mp_level, mp_view_cos, mp_desc, M, th, qi, rr, off, idx, dist);
    if (rc != ORBX_OK) return rc;
    // lists per map point: a point that asked for nothing has an empty one
    const int Q = (int)qi.size();
    int q = 0;
    for (int i = 0; i < M; i++) {
        cand_off[i] = off[q];
        if (q < Q && qi[q] == i) q++;
    }
    cand_off[M] = off[Q];
    *total = off[Q];
    if (off[Q] == 0) return ORBX_OK;
    if (!cand_idx || !cand_dist || cap < off[Q]) return fail(c, ORBX_ERR_CAPACITY, "candidate buffer too small");
    std::memcpy(cand_idx, idx.data(), (size_t)off[Q] * 4);
    std::memcpy(cand_dist, dist.data(), (size_t)off[Q] * 2);
    return ORBX_OK;
}

int orbx_match_candidates(orbx_ctx* c, const orbx_keypoint* kps, const float* u_right, int N, const int32_t* cand_idx,
                          const uint16_t* cand_dist, int ncand, int track_in_view, int is_bad, float proj_xr, int level,
                          float view_cos, float th, float nnratio, int mp_index, int mp_observations, int32_t* frame_obs,
                          int32_t* frame_match, int32_t* match_score, int* best_idx)
{
    if (!c || !best_idx) return ORBX_ERR_INVALID;
    *best_idx = -1;
    if (!track_in_view || is_bad || ncand == 0) return ORBX_OK;              // include/ORBmatcher.h:176-183
    if (ncand < 0 || !kps || !cand_idx || !cand_dist || !frame_obs || !frame_match || !match_score)
        return fail(c, ORBX_ERR_INVALID, "bad arguments");
    if (level < 0 || level >= c->p.nlevels) return fail(c, ORBX_ERR_INVALID, "predicted level out of range");
    for (int k = 0; k < ncand; k++)
        if (cand_idx[k] < 0 || cand_idx[k] >= N) return fail(c, ORBX_ERR_INVALID, "candidate index outside the frame");
    float r = view_cos > 0.998 ? 2.5f : 4.0f;                                 // :188
    if (th != 1.0) r *= th;                                                   // :190-192
    *best_idx = std::max(-1, projection_accept(kps, u_right, cand_idx, cand_dist, ncand, proj_xr, r * c->p.scale[level], nnratio,
                                               mp_index, mp_observations, frame_obs, frame_match, match_score));
    return ORBX_OK;
}

// ORBmatcher::SearchByBoW, both forms (src/ORBmatcher.cc:270-404 and :635-768): the queries are the keypoints of side 1
// that hold a good map point and lie in a vocabulary node both feature vectors have, a query's candidates are the
// keypoints of side 2 in that node.  The device computes every (query, candidate) distance in one call; the walk over
// the nodes, with what earlier queries matched taken out of the later ones' lists, runs on the host.
static int search_by_bow(orbx_ctx* c, const orbx_keypoint* k1, const uint32_t* d1, const uint8_t* has_point1, int n1,
                         const int32_t* node1, const int32_t* off1, const int32_t* idx1, int nodes1,
                         const orbx_keypoint* k2, const uint32_t* d2, const uint8_t* has_point2, int n2,
                         const int32_t* node2, const int32_t* off2, const int32_t* idx2, int nodes2, float nnratio,
                         int check_orientation, bool keyframes, int32_t* matches, int* nmatches)
{
    if (!c || !nmatches) return ORBX_ERR_INVALID;
    *nmatches = 0;
    if (n1 < 0 || n2 < 0 || nodes1 < 0 || nodes2 < 0) return fail(c, ORBX_ERR_INVALID, "negative size");
    const int nout = keyframes ? n1 : n2;
    if (nout > 0 && !matches) return fail(c, ORBX_ERR_INVALID, "null buffer");
    for (int i = 0; i < nout; i++) matches[i] = -1;                           // :274, :647
    if (n1 == 0 || n2 == 0 || nodes1 == 0 || nodes2 == 0) return ORBX_OK;
    if (!k1 || !d1 || !has_point1 || !node1 || !off1 || !idx1 || !k2 || !d2 || !node2 || !off2 || !idx2 ||
        (keyframes && !has_point2))
        return fail(c, ORBX_ERR_INVALID, "null buffer");
    for (int i = 1; i < nodes1; i++) if (node1[i] <= node1[i - 1]) return fail(c, ORBX_ERR_INVALID, "node ids must ascend");
    for (int i = 1; i < nodes2; i++) if (node2[i] <= node2[i - 1]) return fail(c, ORBX_ERR_INVALID, "node ids must ascend");
    for (int i = 0; i < off1[nodes1]; i++) if (idx1[i] < 0 || idx1[i] >= n1) return fail(c, ORBX_ERR_INVALID, "feature index outside side 1");
    for (int i = 0; i < off2[nodes2]; i++) if (idx2[i] < 0 || idx2[i] >= n2) return fail(c, ORBX_ERR_INVALID, "feature index outside side 2");
    constexpr int kHisto = 30, kThLow = 50;                                   // HISTO_LENGTH, TH_LOW
    // the common nodes (the iterator walk of :291-381 is a merge of two ascending sequences) and the queries
    struct Pair { int a, b; };
    std::vector<Pair> common;
    for (int a = 0, b = 0; a < nodes1 && b < nodes2;) {
        if (node1[a] == node2[b]) common.push_back(Pair{a++, b++});
        else if (node1[a] < node2[b]) a++;
        else b++;
    }
    std::vector<int32_t> qi, qnode, coff(1, 0), cidx;
    std::vector<uint32_t> qd;
    for (const Pair& pr : common)
        for (int i = off1[pr.a]; i < off1[pr.a + 1]; i++) {
            const int i1 = idx1[i];
            if (!has_point1[i1]) continue;                                    // :304-308, :671-675
            qi.push_back(i1);
            qnode.push_back(pr.b);
            cidx.insert(cidx.end(), idx2 + off2[pr.b], idx2 + off2[pr.b + 1]);
            coff.push_back((int32_t)cidx.size());
            if (cidx.size() > 0x7fffffffu) return fail(c, ORBX_ERR_UNSUPPORTED, "more than 2^31 candidates");
        }
    const int Q = (int)qi.size();
    if (Q == 0 || cidx.empty()) return ORBX_OK;
    qd.resize((size_t)Q * 8);
    for (int q = 0; q < Q; q++) std::memcpy(&qd[(size_t)q * 8], d1 + (size_t)qi[q] * 8, 32);
    std::vector<uint16_t> dist(cidx.size());
    int rc = orbx_hamming_csr(c, qd.data(), Q, d2, n2, coff.data(), cidx.data(), dist.data());
    if (rc != ORBX_OK) return rc;

    int nm = 0;
    std::vector<int32_t> rotHist[kHisto];
    std::vector<uint8_t> taken(n2, 0);                                        // vpMapPointMatches[idx] != NULL / vbMatched2
    for (int q = 0; q < Q; q++) {
        const int i1 = qi[q];
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (int k = coff[q]; k < coff[q + 1]; k++) {
            const int i2 = cidx[k];
            if (taken[i2]) continue;                                          // :320-321, :689
            if (keyframes && !has_point2[i2]) continue;                       // :689-693
            const int d = dist[k];
            if (d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdx2 = i2; }
            else if (d < bestDist2) bestDist2 = d;
        }
        const bool low = keyframes ? bestDist1 < kThLow : bestDist1 <= kThLow;   // :711 against :339
        if (low && (float)bestDist1 < nnratio * (float)bestDist2) {           // :341, :713
            taken[bestIdx2] = 1;
            if (keyframes) matches[i1] = bestIdx2; else matches[bestIdx2] = i1;
            if (check_orientation) {
                float rot = k1[i1].angle - k2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = 0;
                if (!rotation_bin(rot * (1.0f / kHisto), kHisto, bin)) return fail(c, ORBX_ERR_INVALID, "keypoint angles outside [0, 360)");
                rotHist[bin].push_back(keyframes ? i1 : bestIdx2);
            }
            nm++;
        }
    }
    if (check_orientation) {                                                  // :383-401, :747-765
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, kHisto, ind1, ind2, ind3);
        for (int i = 0; i < kHisto; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j : rotHist[i]) { matches[j] = -1; nm--; }
        }
    }
    *nmatches = nm;
    return ORBX_OK;
}

int orbx_search_by_bow_frame(orbx_ctx* c, const orbx_keypoint* kf_kps, const uint32_t* kf_desc, const uint8_t* kf_has_point,
                             int n_kf, const int32_t* kf_node, const int32_t* kf_off, const int32_t* kf_idx, int kf_nodes,
                             const orbx_keypoint* f_kps, const uint32_t* f_desc, int n_f, const int32_t* f_node,
                             const int32_t* f_off, const int32_t* f_idx, int f_nodes, float nnratio, int check_orientation,
                             int32_t* matches_f, int* nmatches)
{
    return search_by_bow(c, kf_kps, kf_desc, kf_has_point, n_kf, kf_node, kf_off, kf_idx, kf_nodes, f_kps, f_desc, nullptr, n_f,
                         f_node, f_off, f_idx, f_nodes, nnratio, check_orientation, false, matches_f, nmatches);
}

int orbx_search_by_bow_keyframes(orbx_ctx* c, const orbx_keypoint* k1, const uint32_t* d1, const uint8_t* has_point1, int n1,
                                 const int32_t* node1, const int32_t* off1, const int32_t* idx1, int nodes1,
                                 const orbx_keypoint* k2, const uint32_t* d2, const uint8_t* has_point2, int n2,
                                 const int32_t* node2, const int32_t* off2, const int32_t* idx2, int nodes2, float nnratio,
                                 int check_orientation, int32_t* matches12, int* nmatches)
{
    return search_by_bow(c, k1, d1, has_point1, n1, node1, off1, idx1, nodes1, k2, d2, has_point2, n2, node2, off2, idx2,
                         nodes2, nnratio, check_orientation, true, matches12, nmatches);
}

// The projection searches that keep only the best candidate: against the last frame (src/ORBmatcher.cc:1440-1593; a
// keypoint is out when its map point has observations, the accepted distance is TH_HIGH) and against a keyframe in
// relocalisation (:1595-1721; a keypoint is out when it holds any map point, the accepted distance is the caller's).
static int projected_points_search(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, const float* u_right, int N,
                                   float min_x, float min_y, float inv_w, float inv_h, const uint8_t* lp_visible,
                                   const float* lp_u, const float* lp_v, const float* lp_invzc, const int32_t* lp_octave,
                                   const float* lp_angle, const uint32_t* lp_desc, const int32_t* lp_observations, int M,
                                   float th, float mbf, int forward, int backward, int check_orientation, int max_dist,
                                   bool any_point_blocks, int32_t* frame_obs, int32_t* frame_match, int* nmatches)
{
    if (!c || !nmatches) return ORBX_ERR_INVALID;
    *nmatches = 0;
    if (N < 0 || M < 0) return fail(c, ORBX_ERR_INVALID, "negative size");
    if (M == 0 || N == 0) return ORBX_OK;
    if (!kps || !desc || !lp_visible || !lp_u || !lp_v || !lp_octave || !lp_desc || !lp_observations || !frame_obs ||
        !frame_match || (u_right && !lp_invzc) || (check_orientation && !lp_angle))
        return fail(c, ORBX_ERR_INVALID, "null buffer");
    constexpr int kHisto = 30;                         // HISTO_LENGTH (src/ORBmatcher.cc:39)
    // the queries: the visible map points of the last frame, window th * mvScaleFactors[octave] (:1499), the level range
    // by the direction of motion (:1503-1508)
    std::vector<int32_t> qi;
    qi.reserve(M);
    for (int i = 0; i < M; i++) {
        if (!lp_visible[i]) continue;
        if (lp_octave[i] < 0 || lp_octave[i] >= c->p.nlevels) return fail(c, ORBX_ERR_INVALID, "octave out of range");
        qi.push_back(i);
    }
    const int Q = (int)qi.size();
    if (Q == 0) return ORBX_OK;
    std::vector<float> qx(Q), qy(Q), qr(Q);
    std::vector<int32_t> qmin(Q), qmax(Q);
    std::vector<uint32_t> qd((size_t)Q * 8);
    for (int q = 0; q < Q; q++) {
        const int i = qi[q], o = lp_octave[i];
        qx[q] = lp_u[i];
        qy[q] = lp_v[i];
        qr[q] = th * c->p.scale[o];
        if (forward) { qmin[q] = o; qmax[q] = -1; }
        else if (backward) { qmin[q] = 0; qmax[q] = o; }
        else { qmin[q] = o - 1; qmax[q] = o + 1; }
        std::memcpy(&qd[(size_t)q * 8], lp_desc + (size_t)i * 8, 32);
    }
    std::vector<int32_t> off, idx;
    std::vector<uint16_t> dist;
    int rc = grid_lists(c, kps, desc, N, qx.data(), qy.data(), qr.data(), qmin.data(), qmax.data(), qd.data(), Q, min_x, min_y,
                        inv_w, inv_h, off, idx, dist);
    if (rc != ORBX_OK) return rc;

    int nm = 0;
    std::vector<int32_t> rotHist[kHisto];
    const float factor = 1.0f / kHisto;
    for (int q = 0; q < Q; q++) {
        const int i = qi[q];
        if (off[q + 1] == off[q]) continue;                                   // :1510-1511
        const float radius = qr[q];
        int bestDist = 256, bestIdx2 = -1;
        for (int k = off[q]; k < off[q + 1]; k++) {
            const int i2 = idx[k];
            if (any_point_blocks ? frame_obs[i2] >= 0 : frame_obs[i2] > 0) continue;   // :1664-1665, :1521-1523
            if (u_right && u_right[i2] > 0) {                                 // :1525-1531
                const float ur = lp_u[i] - mbf * lp_invzc[i];
                const float er = fabsf(ur - u_right[i2]);
                if (er > radius) continue;
            }
            const int d = dist[k];
            if (d < bestDist) { bestDist = d; bestIdx2 = i2; }
        }
        if (bestDist <= max_dist) {                                           // :1544, :1678
            frame_match[bestIdx2] = i;
            frame_obs[bestIdx2] = lp_observations[i];
            nm++;
            if (check_orientation) {                                          // :1554-1564
                float rot = lp_angle[i] - kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = 0;
                if (!rotation_bin(rot * factor, kHisto, bin)) return fail(c, ORBX_ERR_INVALID, "keypoint angles outside [0, 360)");
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {                                                  // :1570-1590
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, kHisto, ind1, ind2, ind3);
        for (int b = 0; b < kHisto; b++)
            if (b != ind1 && b != ind2 && b != ind3)
                for (int i2 : rotHist[b]) {
                    frame_match[i2] = -1;
                    frame_obs[i2] = -1;
                    nm--;
                }
    }
    *nmatches = nm;
    return ORBX_OK;
}

int orbx_search_by_projection_last(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, const float* u_right, int N,
                                   float min_x, float min_y, float inv_w, float inv_h, const uint8_t* lp_visible,
                                   const float* lp_u, const float* lp_v, const float* lp_invzc, const int32_t* lp_octave,
                                   const float* lp_angle, const uint32_t* lp_desc, const int32_t* lp_observations, int M,
                                   float th, float mbf, int forward, int backward, int check_orientation,
                                   int32_t* frame_obs, int32_t* frame_match, int* nmatches)
{
    return projected_points_search(c, kps, desc, u_right, N, min_x, min_y, inv_w, inv_h, lp_visible, lp_u, lp_v, lp_invzc,
                                   lp_octave, lp_angle, lp_desc, lp_observations, M, th, mbf, forward, backward,
                                   check_orientation, 100 /* TH_HIGH, src/ORBmatcher.cc:38 */, false, frame_obs, frame_match,
                                   nmatches);
}

int orbx_search_by_projection_keyframe(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, int N, float min_x,
                                       float min_y, float inv_w, float inv_h, const uint8_t* kp_visible, const float* kp_u,
                                       const float* kp_v, const int32_t* kp_level, const float* kp_angle,
                                       const uint32_t* kp_desc, const int32_t* kp_observations, int M, float th, int orb_dist,
                                       int check_orientation, int32_t* frame_obs, int32_t* frame_match, int* nmatches)
{
    if (c && (orb_dist < 0 || orb_dist > 256)) return fail(c, ORBX_ERR_INVALID, "orb_dist must be in [0, 256]");
    return projected_points_search(c, kps, desc, nullptr, N, min_x, min_y, inv_w, inv_h, kp_visible, kp_u, kp_v, nullptr,
                                   kp_level, kp_angle, kp_desc, kp_observations, M, th, 0.0f, 0, 0, check_orientation, orb_dist,
                                   true, frame_obs, frame_match, nmatches);
}

// ---------------------------------------------------------------------------
// The searches of the mapping and loop-closing threads.  Each map point looks for the most similar keypoint of levels
// [level - 1, level] inside its window; the device forms the lists (GetFeaturesInArea order) with every distance, the
// host applies what differs between the routines.
// ---------------------------------------------------------------------------
struct WindowLists {
    std::vector<int32_t> qi, off, idx;
    std::vector<uint16_t> dist;
};
static int window_lists(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, int N, float min_x, float min_y, float inv_w,
                        float inv_h, const uint8_t* visible, const float* u, const float* v, const int32_t* level,
                        const uint32_t* mp_desc, int M, float th, WindowLists& wl)
{
    wl.qi.clear();
    for (int i = 0; i < M; i++) {
        if (!visible[i]) continue;
        if (level[i] < 0 || level[i] >= c->p.nlevels) return fail(c, ORBX_ERR_INVALID, "predicted level out of range");
        wl.qi.push_back(i);
    }
    const int Q = (int)wl.qi.size();
    wl.off.assign(1, 0);
    wl.idx.clear();
    wl.dist.clear();
    if (Q == 0 || N == 0) { wl.off.assign((size_t)Q + 1, 0); return ORBX_OK; }
    std::vector<float> qx(Q), qy(Q), qr(Q);
    std::vector<int32_t> qmin(Q), qmax(Q);
    std::vector<uint32_t> qd((size_t)Q * 8);
    for (int q = 0; q < Q; q++) {
        const int i = wl.qi[q];
        qx[q] = u[i];
        qy[q] = v[i];
        qr[q] = th * c->p.scale[level[i]];
        qmin[q] = level[i] - 1;     // GetFeaturesInArea without levels + the loops' own level test = this level range
        qmax[q] = level[i];
        std::memcpy(&qd[(size_t)q * 8], mp_desc + (size_t)i * 8, 32);
    }
    return grid_lists(c, kps, desc, N, qx.data(), qy.data(), qr.data(), qmin.data(), qmax.data(), qd.data(), Q, min_x, min_y, inv_w,
                      inv_h, wl.off, wl.idx, wl.dist);
}

int orbx_search_by_projection_sim3(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, int N, float min_x, float min_y,
                                   float inv_w, float inv_h, const uint8_t* mp_visible, const float* mp_u, const float* mp_v,
                                   const int32_t* mp_level, const uint32_t* mp_desc, int M, int th, int32_t* kf_matched,
                                   int32_t* kf_match, int* nmatches)
{
    if (!c || !nmatches) return ORBX_ERR_INVALID;
    *nmatches = 0;
    if (N < 0 || M < 0) return fail(c, ORBX_ERR_INVALID, "negative size");
    if (N == 0 || M == 0) return ORBX_OK;
    if (!kps || !desc || !mp_visible || !mp_u || !mp_v || !mp_level || !mp_desc || !kf_matched || !kf_match)
        return fail(c, ORBX_ERR_INVALID, "null buffer");
    WindowLists wl;
    int rc = window_lists(c, kps, desc, N, min_x, min_y, inv_w, inv_h, mp_visible, mp_u, mp_v, mp_level, mp_desc, M, (float)th, wl);
    if (rc != ORBX_OK) return rc;
    int nm = 0;
    for (size_t q = 0; q < wl.qi.size(); q++) {
        int bestDist = 256, bestIdx = -1;
        for (int k = wl.off[q]; k < wl.off[q + 1]; k++) {
            const int idx = wl.idx[k];
            if (kf_matched[idx] >= 0) continue;                               // src/ORBmatcher.cc:491-492
            if ((int)wl.dist[k] < bestDist) { bestDist = wl.dist[k]; bestIdx = idx; }
        }
        if (bestDist <= 50) {                                                 // TH_LOW, :510-514
            kf_matched[bestIdx] = 1;
            kf_match[bestIdx] = wl.qi[q];
            nm++;
        }
    }
    *nmatches = nm;
    return ORBX_OK;
}

int orbx_fuse_search(orbx_ctx* c, const orbx_keypoint* kps, const uint32_t* desc, const float* u_right, int N, float min_x,
                     float min_y, float inv_w, float inv_h, const uint8_t* mp_visible, const float* mp_u, const float* mp_v,
                     const float* mp_ur, const int32_t* mp_level, const uint32_t* mp_desc, int M, float th, int gate,
                     int32_t* best_idx, int32_t* best_dist, int* nfused)
{
    if (!c || !nfused) return ORBX_ERR_INVALID;
    *nfused = 0;
    if (N < 0 || M < 0) return fail(c, ORBX_ERR_INVALID, "negative size");
    if (M == 0) return ORBX_OK;
    if (!best_idx || !best_dist) return fail(c, ORBX_ERR_INVALID, "null buffer");
    for (int i = 0; i < M; i++) { best_idx[i] = -1; best_dist[i] = 256; }
    if (N == 0) return ORBX_OK;
    if (!kps || !desc || !mp_visible || !mp_u || !mp_v || !mp_level || !mp_desc || (gate && u_right && !mp_ur))
        return fail(c, ORBX_ERR_INVALID, "null buffer");
    WindowLists wl;
    int rc = window_lists(c, kps, desc, N, min_x, min_y, inv_w, inv_h, mp_visible, mp_u, mp_v, mp_level, mp_desc, M, th, wl);
    if (rc != ORBX_OK) return rc;
    int nf = 0;
    for (size_t q = 0; q < wl.qi.size(); q++) {
        const int i = wl.qi[q];
        const float u = mp_u[i], v = mp_v[i];
        int bestDist = 256, bestIdx = -1;
        for (int k = wl.off[q]; k < wl.off[q + 1]; k++) {
            const int idx = wl.idx[k];
            if (gate) {                                                       // src/ORBmatcher.cc:1026-1050
                const int kpLevel = kps[idx].octave;
                if (kpLevel < 0 || kpLevel >= c->p.nlevels) return fail(c, ORBX_ERR_INVALID, "keypoint octave out of range");
                if (u_right && u_right[idx] >= 0) {
                    const float ex = u - kps[idx].x, ey = v - kps[idx].y, er = mp_ur[i] - u_right[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * c->p.inv_sigma2[kpLevel] > 7.8) continue;
                } else {
                    const float ex = u - kps[idx].x, ey = v - kps[idx].y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * c->p.inv_sigma2[kpLevel] > 5.99) continue;
                }
            }
            if ((int)wl.dist[k] < bestDist) { bestDist = wl.dist[k]; bestIdx = idx; }
        }
        best_dist[i] = bestDist;
        if (bestDist <= 50) { best_idx[i] = bestIdx; nf++; }                 // TH_LOW, :1064, :1194
    }
    *nfused = nf;
    return ORBX_OK;
}

int orbx_search_by_sim3(orbx_ctx* c, const orbx_keypoint* k1, const uint32_t* d1, int n1, const orbx_keypoint* k2,
                        const uint32_t* d2, int n2, float min_x, float min_y, float inv_w, float inv_h,
                        const uint8_t* p1_visible, const float* p1_u, const float* p1_v, const int32_t* p1_level,
                        const uint32_t* p1_desc, const uint8_t* p2_visible, const float* p2_u, const float* p2_v,
                        const int32_t* p2_level, const uint32_t* p2_desc, float th, int32_t* matches12, int* nfound)
{
    if (!c || !nfound) return ORBX_ERR_INVALID;
    *nfound = 0;
    if (n1 < 0 || n2 < 0) return fail(c, ORBX_ERR_INVALID, "negative size");
    if (n1 > 0 && !matches12) return fail(c, ORBX_ERR_INVALID, "null buffer");
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return ORBX_OK;
    if (!k1 || !d1 || !k2 || !d2 || !p1_visible || !p1_u || !p1_v || !p1_level || !p1_desc || !p2_visible || !p2_u || !p2_v ||
        !p2_level || !p2_desc)
        return fail(c, ORBX_ERR_INVALID, "null buffer");
    std::vector<int32_t> vnMatch1(n1, -1), vnMatch2(n2, -1);
    auto half = [&](const orbx_keypoint* kps, const uint32_t* desc, int N, const uint8_t* vis, const float* u, const float* v,
                    const int32_t* lvl, const uint32_t* pd, int M, std::vector<int32_t>& out) -> int {
        WindowLists wl;
        int rc = window_lists(c, kps, desc, N, min_x, min_y, inv_w, inv_h, vis, u, v, lvl, pd, M, th, wl);
        if (rc != ORBX_OK) return rc;
        for (size_t q = 0; q < wl.qi.size(); q++) {
            int bestDist = 0x7fffffff, bestIdx = -1;
            for (int k = wl.off[q]; k < wl.off[q + 1]; k++)
                if ((int)wl.dist[k] < bestDist) { bestDist = wl.dist[k]; bestIdx = wl.idx[k]; }
            if (bestDist <= 100) out[wl.qi[q]] = bestIdx;                     // TH_HIGH, src/ORBmatcher.cc:1333, :1413
        }
        return ORBX_OK;
    };
    int rc = half(k2, d2, n2, p1_visible, p1_u, p1_v, p1_level, p1_desc, n1, vnMatch1);
    if (rc != ORBX_OK) return rc;
    rc = half(k1, d1, n1, p2_visible, p2_u, p2_v, p2_level, p2_desc, n2, vnMatch2);
    if (rc != ORBX_OK) return rc;
    int nf = 0;
    for (int i1 = 0; i1 < n1; i1++) {                                         // :1419-1435
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0 && vnMatch2[idx2] == i1) { matches12[i1] = idx2; nf++; }
    }
    *nfound = nf;
    return ORBX_OK;
}

int orbx_search_for_triangulation(orbx_ctx* c, const orbx_keypoint* k1, const uint32_t* d1, const uint8_t* has_point1,
                                  const uint8_t* stereo1, int n1, const int32_t* node1, const int32_t* off1, const int32_t* idx1,
                                  int nodes1, const orbx_keypoint* k2, const uint32_t* d2, const uint8_t* has_point2,
                                  const uint8_t* stereo2, int n2, const int32_t* node2, const int32_t* off2, const int32_t* idx2,
                                  int nodes2, const float* F12, float ex, float ey, int only_stereo, int check_orientation,
                                  int32_t* matches12, int* nmatches)
{
    if (!c || !nmatches) return ORBX_ERR_INVALID;
    *nmatches = 0;
    if (n1 < 0 || n2 < 0 || nodes1 < 0 || nodes2 < 0) return fail(c, ORBX_ERR_INVALID, "negative size");
    if (n1 > 0 && !matches12) return fail(c, ORBX_ERR_INVALID, "null buffer");
    for (int i = 0; i < n1; i++) matches12[i] = -1;                           // src/ORBmatcher.cc:791
    if (n1 == 0 || n2 == 0 || nodes1 == 0 || nodes2 == 0) return ORBX_OK;
    if (!k1 || !d1 || !has_point1 || !stereo1 || !node1 || !off1 || !idx1 || !k2 || !d2 || !has_point2 || !stereo2 || !node2 ||
        !off2 || !idx2 || !F12)
        return fail(c, ORBX_ERR_INVALID, "null buffer");
    for (int i = 1; i < nodes1; i++) if (node1[i] <= node1[i - 1]) return fail(c, ORBX_ERR_INVALID, "node ids must ascend");
    for (int i = 1; i < nodes2; i++) if (node2[i] <= node2[i - 1]) return fail(c, ORBX_ERR_INVALID, "node ids must ascend");
    for (int i = 0; i < off1[nodes1]; i++) if (idx1[i] < 0 || idx1[i] >= n1) return fail(c, ORBX_ERR_INVALID, "feature index outside side 1");
    for (int i = 0; i < off2[nodes2]; i++) if (idx2[i] < 0 || idx2[i] >= n2) return fail(c, ORBX_ERR_INVALID, "feature index outside side 2");
    for (int i = 0; i < n2; i++)
        if (k2[i].octave < 0 || k2[i].octave >= c->p.nlevels) return fail(c, ORBX_ERR_INVALID, "keypoint octave out of range");
    constexpr int kHisto = 30, kThLow = 50;
    // queries: the keypoints of the first keyframe without a map point (stereo ones only, if asked) in a node both have
    std::vector<int32_t> qi, coff(1, 0), cidx;
    for (int a = 0, b = 0; a < nodes1 && b < nodes2;) {
        if (node1[a] < node2[b]) { a++; continue; }
        if (node1[a] > node2[b]) { b++; continue; }
        for (int i = off1[a]; i < off1[a + 1]; i++) {
            const int i1 = idx1[i];
            if (has_point1[i1] || (only_stereo && !stereo1[i1])) continue;    // :812-822
            qi.push_back(i1);
            cidx.insert(cidx.end(), idx2 + off2[b], idx2 + off2[b + 1]);
            if (cidx.size() > 0x7fffffffu) return fail(c, ORBX_ERR_UNSUPPORTED, "more than 2^31 candidates");
            coff.push_back((int32_t)cidx.size());
        }
        a++;
        b++;
    }
    const int Q = (int)qi.size();
    if (Q == 0 || cidx.empty()) return ORBX_OK;
    std::vector<uint32_t> qd((size_t)Q * 8);
    for (int q = 0; q < Q; q++) std::memcpy(&qd[(size_t)q * 8], d1 + (size_t)qi[q] * 8, 32);
    std::vector<uint16_t> dist(cidx.size());
    int rc = orbx_hamming_csr(c, qd.data(), Q, d2, n2, coff.data(), cidx.data(), dist.data());
    if (rc != ORBX_OK) return rc;
    // ORBmatcher::CheckDistEpipolarLine, :251-268
    auto epipolar_ok = [&](const orbx_keypoint& kp1, const orbx_keypoint& kp2) -> bool {
        const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
        const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
        const float cc = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
        const float num = a * kp2.x + b * kp2.y + cc;
        const float den = a * a + b * b;
        if (den == 0) return false;
        const float dsqr = num * num / den;
        return dsqr < 3.84 * c->p.sigma2[kp2.octave];
    };
    int nm = 0;
    std::vector<int32_t> rotHist[kHisto];
    for (int q = 0; q < Q; q++) {
        const int i1 = qi[q];
        const bool bStereo1 = stereo1[i1] != 0;
        int bestDist = kThLow, bestIdx2 = -1;                                 // :828-829
        for (int k = coff[q]; k < coff[q + 1]; k++) {
            const int i2 = cidx[k];
            if (has_point2[i2]) continue;                                     // :838 (vbMatched2 is never set in the reference)
            const bool bStereo2 = stereo2[i2] != 0;
            if (only_stereo && !bStereo2) continue;                           // :843-845
            const int d = dist[k];
            if (d > kThLow || d > bestDist) continue;                         // :851-852
            if (!bStereo1 && !bStereo2) {                                     // :856-862
                const float distex = ex - k2[i2].x, distey = ey - k2[i2].y;
                if (distex * distex + distey * distey < 100 * c->p.scale[k2[i2].octave]) continue;
            }
            if (epipolar_ok(k1[i1], k2[i2])) { bestIdx2 = i2; bestDist = d; }
        }
        if (bestIdx2 >= 0) {                                                  // :871-888
            matches12[i1] = bestIdx2;
            nm++;
            if (check_orientation) {
                float rot = k1[i1].angle - k2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = 0;
                if (!rotation_bin(rot * (1.0f / kHisto), kHisto, bin)) return fail(c, ORBX_ERR_INVALID, "keypoint angles outside [0, 360)");
                rotHist[bin].push_back(i1);
            }
        }
    }
    if (check_orientation) {                                                  // :904-922
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, kHisto, ind1, ind2, ind3);
        for (int i = 0; i < kHisto; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j : rotHist[i]) { matches12[j] = -1; nm--; }
        }
    }
    *nmatches = nm;
    return ORBX_OK;
}

// K, dist, R, P -> the model the kernel evaluates; RR = P(:, 0:3) * R as cv::undistortPoints forms it
static int undistort_model(orbx_ctx* c, const double* K, const double* dist, int ndist, const double* R, const double* P,
                           int p_cols, int fisheye, UndistortModel& m)
{
    if (!K) return fail(c, ORBX_ERR_INVALID, "camera matrix is required");
    if (ndist < 0 || (ndist > 0 && !dist)) return fail(c, ORBX_ERR_INVALID, "bad distortion coefficients");
    if (fisheye ? (ndist != 4) : !(ndist == 0 || ndist == 4 || ndist == 5 || ndist == 8 || ndist == 12))
        return fail(c, ORBX_ERR_UNSUPPORTED, "distortion vector must have 4, 5, 8 or 12 entries (fisheye: 4); the tilt model is not supported");
    if (P && p_cols != 3 && p_cols != 4) return fail(c, ORBX_ERR_INVALID, "P must be 3x3 or 3x4");
    m.fx = K[0]; m.fy = K[4]; m.cx = K[2]; m.cy = K[5];
    if (m.fx == 0 || m.fy == 0) return fail(c, ORBX_ERR_INVALID, "zero focal length");
    m.ifx = 1. / m.fx;
    m.ify = 1. / m.fy;
    for (int i = 0; i < 12; i++) m.k[i] = i < ndist ? dist[i] : 0.0;
    double r[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (R) std::memcpy(r, R, sizeof(r));
    if (!P) std::memcpy(m.RR, r, sizeof(r));
    else
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += P[i * p_cols + k] * r[k * 3 + j];
                m.RR[i * 3 + j] = s;
            }
    m.ndist = ndist;
    m.fisheye = fisheye != 0;
    return ORBX_OK;
}

int orbx_undistort_keypoints_device(orbx_ctx* c, const orbx_keypoint* d_in, const int32_t* d_n, int n, int cap, int nframes,
                                    const double* K, const double* dist, int ndist, const double* R, const double* P,
                                    int p_cols, int fisheye, orbx_keypoint* d_out)
{
    if (!c) return ORBX_ERR_INVALID;
    if (nframes < 1 || nframes > 65535 || cap < 0 || n < 0 || n > cap) return fail(c, ORBX_ERR_INVALID, "bad sizes");
    if (cap == 0) return ORBX_OK;
    if (!d_in || !d_out) return fail(c, ORBX_ERR_INVALID, "null buffer");
    UndistortModel m;
    int rc = undistort_model(c, K, dist, ndist, R, P, p_cols, fisheye, m);
    if (rc != ORBX_OK) return rc;
    HIPCHECK(c, hipSetDevice(c->device));
    launch_undistort(d_in, d_n, n, cap, nframes, m, d_out, c->stream);
    HIPCHECK(c, hipGetLastError());
    return ORBX_OK;
}

int orbx_undistort_keypoints(orbx_ctx* c, const orbx_keypoint* kps_in, int n, const double* K, const double* dist, int ndist,
                             const double* R, const double* P, int p_cols, int fisheye, orbx_keypoint* kps_out)
{
    if (!c) return ORBX_ERR_INVALID;
    if (n < 0) return fail(c, ORBX_ERR_INVALID, "bad sizes");
    UndistortModel m;
    int rc = undistort_model(c, K, dist, ndist, R, P, p_cols, fisheye, m);   // argument errors even for n == 0
    if (rc != ORBX_OK) return rc;
    if (n == 0) return ORBX_OK;
    if (!kps_in || !kps_out) return fail(c, ORBX_ERR_INVALID, "null buffer");
    HIPCHECK(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)n * sizeof(orbx_keypoint);
    HIPCHECK(c, c->gq[0].ensure(bytes));
    orbx_keypoint* d = c->gq[0].as<orbx_keypoint>();
    HIPCHECK(c, h2d(c, d, kps_in, bytes));
    launch_undistort(d, nullptr, n, n, 1, m, d, c->stream);
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, d2h(c, kps_out, d, bytes));
    HIPCHECK(c, finish(c));
    return ORBX_OK;
}

} // extern "C"
