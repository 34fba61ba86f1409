"""MI355X (gfx950) ORB front-end: host-side mirror of the reference's
ORBextractor / ORBmatcher interface on top of the C ABI in include/orbx.h.

The package holds only what the hot path needs: csrc/ (HIP kernels + C ABI),
lib/ (the built liborbx.so) and these thin ctypes classes.  There is no CPU
fallback: importing works anywhere, constructing an extractor needs the built
library and a gfx950 device.
"""
from ._lib import OrbxError, KP_DTYPE, load_library, library_path  # noqa: F401
from .extractor import ORBextractor  # noqa: F401
from .matcher import ORBmatcher  # noqa: F401
from .frame import Frame, CameraModel, StereoFrame  # noqa: F401
from .detection_io import ExportToYML, KeyPointLog  # noqa: F401

__all__ = ["ORBextractor", "ORBmatcher", "Frame", "CameraModel", "StereoFrame", "ExportToYML", "KeyPointLog", "OrbxError", "KP_DTYPE", "load_library", "library_path"]
