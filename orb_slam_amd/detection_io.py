"""Detection I/O in the reference's wire format: Frame::ExportToYML (src/Frame.cc:590-609) and
KeyPointLog::ReadFromYAML (include/Util.hpp:400-449).  Host only: no device is needed, so a whole
sequence extracted in batches on the GPU can be written out and replayed by the unmodified tracker
(its ENABLE_DETECTION_IO path)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, OrbxError


def _check(lib, rc):
    if rc != 0:
        raise OrbxError(rc, lib.orbx_io_last_error().decode("utf-8", "replace"))


def _kd(keys, desc):
    keys = np.zeros(0, KP_DTYPE) if keys is None else np.ascontiguousarray(keys, KP_DTYPE)
    desc = np.zeros((0, 32), np.uint8) if desc is None else np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    if len(keys) != len(desc):
        raise ValueError("keypoints and descriptors differ in length")
    return keys, desc


def ExportToYML(path, mTimeStamp, mvKeys, mDescriptors, mvKeysRight=None, mDescriptorsRight=None, mK=None,
                nNextId=0, mnId=0, legacy_flat=False):
    """Frame::ExportToYML: one YAML document per frame."""
    lib = _lib.load_library()
    kl, dl = _kd(mvKeys, mDescriptors)
    kr, dr = _kd(mvKeysRight, mDescriptorsRight)
    k = None if mK is None else np.ascontiguousarray(mK, np.float32).reshape(3, 3)
    _check(lib, lib.orbx_detection_write_yaml(
        str(path).encode(), float(nNextId), float(mnId), float(mTimeStamp), None if k is None else k.ctypes.data,
        kl.ctypes.data if len(kl) else None, dl.ctypes.data if len(kl) else None, len(kl),
        kr.ctypes.data if len(kr) else None, dr.ctypes.data if len(kr) else None, len(kr), 1 if legacy_flat else 0))


class KeyPointLog:
    """KeyPointLog (include/Util.hpp:397-457): mTimeStamp, mvKeysLeft, mvKeysRight, mDiscLeft, mDiscRight."""

    def __init__(self):
        self.mTimeStamp = -1.0
        self.mvKeysLeft = np.zeros(0, KP_DTYPE)
        self.mvKeysRight = np.zeros(0, KP_DTYPE)
        self.mDiscLeft = np.zeros((0, 32), np.uint8)
        self.mDiscRight = np.zeros((0, 32), np.uint8)

    def ReadFromYAML(self, kpt_path):
        """Returns False (timestamp -1, empty lists) when the file cannot be opened, like the reference;
        a file that opens but does not hold the expected keys raises."""
        lib = _lib.load_library()
        self.__init__()
        ts, n, nr = C.c_double(), C.c_int(), C.c_int()
        path = str(kpt_path).encode()
        rc = lib.orbx_detection_read_yaml(path, C.byref(ts), None, None, 0, C.byref(n), None, None, 0, C.byref(nr))
        if rc == -7:
            return False
        _check(lib, rc)
        kl, kr = np.zeros(n.value, KP_DTYPE), np.zeros(nr.value, KP_DTYPE)
        dl, dr = np.zeros((n.value, 32), np.uint8), np.zeros((nr.value, 32), np.uint8)
        _check(lib, lib.orbx_detection_read_yaml(path, C.byref(ts), kl.ctypes.data, dl.ctypes.data, n.value, C.byref(n),
                                                 kr.ctypes.data, dr.ctypes.data, nr.value, C.byref(nr)))
        self.mTimeStamp = ts.value
        self.mvKeysLeft, self.mvKeysRight, self.mDiscLeft, self.mDiscRight = kl, kr, dl, dr
        return True
