"""ctypes binding of liborbx.so (declarations follow include/orbx.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# layout of cv::KeyPoint / orbx_keypoint
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

ERR_NAMES = {0: "ORBX_OK", -1: "ORBX_ERR_INVALID", -2: "ORBX_ERR_NO_DEVICE", -3: "ORBX_ERR_HIP",
             -4: "ORBX_ERR_CAPACITY", -5: "ORBX_ERR_UNSUPPORTED", -6: "ORBX_ERR_STATE"}


class OrbxError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__("%s (%d)%s" % (ERR_NAMES.get(code, "ORBX_ERR"), code, (": " + msg) if msg else ""))


def library_path():
    # ORBX_LIBRARY: another build of the same library (diagnostic builds, A/B comparisons)
    return os.environ.get("ORBX_LIBRARY") or os.path.join(_HERE, "lib", "liborbx.so")


_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
_ip = C.POINTER(C.c_int32)
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes); every symbol include/orbx.h declares
SIGNATURES = {
    "orbx_version": (C.c_char_p, []),
    "orbx_create": (_i, [C.POINTER(_vp), _i, _f, _i, _i, _i, _i]),
    "orbx_destroy": (None, [_vp]),
    "orbx_last_error": (C.c_char_p, [_vp]),
    "orbx_set_stream": (_i, [_vp, _vp]),
    "orbx_get_params": (_i, [_vp, _ip, _fp, _ip, _ip, _ip]),
    "orbx_get_tables": (_i, [_vp, _fp, _fp, _fp, _fp, _ip]),
    "orbx_max_keypoints": (_i, [_vp, _i, _i, _ip]),
    "orbx_level_size": (_i, [_vp, _i, _i, _i, _ip, _ip]),
    "orbx_extract": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _ip]),
    "orbx_extract_batch": (_i, [_vp, C.POINTER(_vp), _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "orbx_extract_batch_device": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_size_t, _vp, _vp, _i, _vp]),
    "orbx_compute_pyramid": (_i, [_vp, _vp, _i, _i, _i]),
    "orbx_pyramid_level": (_i, [_vp, _i, _i, C.POINTER(_vp), _ip, _ip, _ip]),
    "orbx_debug_candidates": (_i, [_vp, _i, _i, _vp, _i, _ip]),
    "orbx_debug_level_counts": (_i, [_vp, _i, _vp]),
    "orbx_debug_blurred_level": (_i, [_vp, _i, _i, _vp, C.c_size_t]),
    "orbx_stage_times": (_i, [_vp, C.POINTER(C.c_char_p), _fp, _ip, _i]),
    "orbx_reset_stage_times": (_i, [_vp]),
    "orbx_set_stage_timing": (_i, [_vp, _i]),
    "orbx_hamming": (_i, [_vp, _vp]),
    "orbx_match_bruteforce": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "orbx_match_bruteforce_device": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "orbx_match_bruteforce_device_static": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "orbx_match_windowed": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "orbx_hamming_csr": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "orbx_stereo_match": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _ip]),
    "orbx_stereo_match_sad": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _ip]),
    "orbx_stereo_match_sad_device": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f,
                                          _vp, _vp, _vp, _vp, _vp]),
    "orbx_features_in_area": (_i, [_vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _ip]),
    "orbx_grid_search": (_i, [_vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                              _vp, _vp, _vp, _vp, _vp, _vp]),
    "orbx_grid_search_device": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "orbx_search_for_initialization": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _f, _f, _f, _f, _vp, _i, _f, _i, _vp, _ip]),
    "orbx_search_by_projection": (_i, [_vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _i, _f, _f, _vp, _vp, _vp, _ip]),
    "orbx_search_by_projection_budget": (_i, [_vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                              _vp, _i, _f, _f, C.c_double, _vp, _vp, _vp, _ip, _ip]),
    "orbx_projection_candidates": (_i, [_vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f,
                                        _vp, _vp, _vp, _i, _ip]),
    "orbx_match_candidates": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _f, _i, _f, _f, _f, _i, _i, _vp, _vp, _vp, _ip]),
    "orbx_search_by_bow_frame": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _f, _i, _vp, _ip]),
    "orbx_search_by_bow_keyframes": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _f, _i,
                                          _vp, _ip]),
    "orbx_search_by_projection_keyframe": (_i, [_vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _i,
                                                _vp, _vp, _ip]),
    "orbx_search_by_projection_sim3": (_i, [_vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _ip]),
    "orbx_fuse_search": (_i, [_vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _vp, _vp, _ip]),
    "orbx_search_by_sim3": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _f, _vp, _ip]),
    "orbx_search_for_triangulation": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp,
                                           _i, _vp, _f, _f, _i, _i, _vp, _ip]),
    "orbx_search_by_projection_last": (_i, [_vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                                            _f, _f, _i, _i, _i, _vp, _vp, _ip]),
    "orbx_stereo_match_device": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _f,
                                      _vp, _vp, _vp, _vp, _vp]),
    "orbx_stereo_match_ex": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp, _i, _i, _vp,
                                  _vp, _vp, _vp, _vp, _ip]),
    "orbx_stereo_match_ex_device": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp,
                                         _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "orbx_detection_write_yaml": (_i, [C.c_char_p, C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, _i, _vp, _vp, _i, _i]),
    "orbx_detection_read_yaml": (_i, [C.c_char_p, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "orbx_io_last_error": (C.c_char_p, []),
    "orbx_undistort_keypoints": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "orbx_undistort_keypoints_device": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
}

_lib = None


def _share_torch_hip_runtime():
    """PyTorch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, no
    SONAME), distinct from the system one liborbx.so names.  Two HIP runtimes in
    one process cannot both own the GPU, so when torch is installed its runtime
    is put into the global symbol scope first and liborbx binds to it; without
    torch, liborbx uses the system runtime it was linked against."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return None
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            return C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        return None
    return None


def load_library():
    """Load liborbx.so; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise OrbxError(-2, "liborbx.so is not built: run `make -C orb_slam_amd/csrc` (or __graft_entry__.build())")
    _share_torch_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(lib, ctx, rc):
    if rc != 0:
        msg = lib.orbx_last_error(ctx).decode("utf-8", "replace") if ctx else ""
        raise OrbxError(rc, msg)


def ptr(a):
    """device/host pointer of a numpy array, a torch tensor, an int or None"""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    raise TypeError("cannot take the address of %r" % type(a))
