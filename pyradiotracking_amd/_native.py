"""ctypes binding of the C-ABI in include/rt_analyze.h (librt_analyze.so).

The library is the product: if it cannot be loaded, or no GPU is usable, every
entry point here raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "librt_analyze.so")

RT_OK = 0
RT_E_INVALID = -1
RT_E_UNSUPPORTED = -2
RT_E_NO_DEVICE = -3
RT_E_HIP = -4
RT_E_CAPACITY = -5
RT_E_ONE_SEGMENT = -6
RT_E_NOMEM = -7
RT_E_HOT_OVERFLOW = -8  # RT_MODE_SPARSE: a candidate list overflowed, the call has no result and is consumed

RT_MODE_AUTO, RT_MODE_DENSE, RT_MODE_SPARSE, RT_MODE_PREFILTER, RT_MODE_RUNFILTER = 0, 1, 2, 3, 4
RT_FLAG_TIMING = 1
RT_FLAG_NO_LIN_DETREND = 2  # subtract the segment mean before windowing even for hamming / hann / boxcar windows
RT_FLAG_GROUP_DETECT = 4  # sparse detection by groups of candidate lists at any number of streams (default: from 1 024 streams per handle)
RT_FLAG_NO_GROUP_DETECT = 8  # ... never

SUPPORTED_NPERSEG = tuple(range(8, 8193)) + (16384,)  # 8 .. 8192 and 16384 (32 .. 4096 powers of two: the fused scans; everything else: general transforms, dense path)
FUSED_NPERSEG = (256, 512, 1024, 2048, 4096)


class RtConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("n_streams", C.c_int32),
        ("nperseg", C.c_int32),
        ("mode", C.c_int32),
        ("max_samples", C.c_int64),
        ("sample_rate", C.c_double),
        ("window", C.POINTER(C.c_float)),
        ("scale", C.c_float),
        ("threshold", C.c_float),
        ("snr_threshold", C.c_float),
        ("calibration_db", C.c_float),
        ("min_duration_s", C.c_double),
        ("max_duration_s", C.c_double),
        ("hot_capacity", C.c_int32),
        ("record_capacity", C.c_int32),
        ("segs_per_chunk", C.c_int32),
        ("flags", C.c_int32),
        ("hip_stream", C.c_void_p),
        ("lanes", C.c_int32),
        ("record_pool", C.c_int32),
    ]


class RtCallInfo(C.Structure):
    _fields_ = [
        ("n_seg", C.c_int32),
        ("mode_used", C.c_int32),
        ("fell_back", C.c_int32),
        ("n_dense_streams", C.c_int32),
        ("n_hot", C.c_int64),
        ("n_records", C.c_int64),
        ("ms_stft", C.c_float),
        ("ms_detect", C.c_float),
        ("ms_total", C.c_float),
        ("segs_per_chunk", C.c_int32),
    ]


#: numpy view of ``rt_record`` (40 bytes)
RECORD_DTYPE = np.dtype(
    [
        ("stream", "<i4"),
        ("fi", "<i4"),
        ("start", "<i4"),
        ("end", "<i4"),
        ("max_p", "<f4"),
        ("mean_p", "<f4"),
        ("std_db", "<f4"),
        ("row_mean", "<f4"),
        ("shadowed", "<i4"),
        ("reserved", "<i4"),
    ]
)

#: every symbol include/rt_analyze.h declares
ABI_SYMBOLS = (
    "rt_abi_version",
    "rt_create",
    "rt_destroy",
    "rt_reset",
    "rt_reset_stream",
    "rt_set_stream_params",
    "rt_process",
    "rt_process_u8",
    "rt_process_host",
    "rt_process_u8_host",
    "rt_fetch",
    "rt_extract",
    "rt_spectrogram",
    "rt_calibrate_read",
    "rt_get_call_info",
    "rt_last_error",
    "rt_dev_alloc",
    "rt_dev_free",
    "rt_dev_upload",
    "rt_dev_download",
    "rt_device_count",
)

_lib = None


class NativeError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rt_analyze error {code}: {message}")
        self.code = code


def load_library(path: Optional[str] = None):
    """dlopen librt_analyze.so (built in-tree by pyradiotracking_amd.build).

    If torch is importable it is imported first, so both share one HIP
    runtime (torch ships its own libamdhip64 with the same soname)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("RT_ANALYZE_LIB") or LIB_PATH  # env override: A/B builds of the kernels
    if not os.path.exists(p):
        raise ImportError(
            f"{p} is missing: build it with `python -m pyradiotracking_amd.build` "
            "(hipcc --offload-arch=gfx950); this package has no CPU fallback"
        )
    if not os.environ.get("RT_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the product
            pass
    lib = C.CDLL(p, mode=getattr(os, "RTLD_NOW", 2) | getattr(os, "RTLD_GLOBAL", 0x100))
    vp = C.c_void_p
    lib.rt_abi_version.restype = C.c_int
    lib.rt_create.argtypes = [C.POINTER(RtConfig), C.POINTER(vp)]
    lib.rt_destroy.argtypes = [vp]
    lib.rt_destroy.restype = None
    lib.rt_reset.argtypes = [vp]
    lib.rt_reset_stream.argtypes = [vp, C.c_int32]
    lib.rt_set_stream_params.argtypes = [vp, vp, vp]
    lib.rt_process.argtypes = [vp, vp, C.c_int64, C.c_int64]
    lib.rt_process_u8.argtypes = [vp, vp, C.c_int64, C.c_int64]
    lib.rt_process_host.argtypes = [vp, vp, C.c_int64, C.c_int64]
    lib.rt_process_u8_host.argtypes = [vp, vp, C.c_int64, C.c_int64]
    lib.rt_fetch.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.rt_extract.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, C.c_int32]
    lib.rt_spectrogram.argtypes = [vp, vp, C.c_int64, C.c_int64, vp]
    lib.rt_calibrate_read.argtypes = [vp, vp, C.c_int64, C.c_int64]
    lib.rt_get_call_info.argtypes = [vp, C.POINTER(RtCallInfo)]
    lib.rt_last_error.argtypes = [vp]
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_dev_alloc.argtypes = [C.c_int32, C.c_size_t, C.POINTER(vp)]
    lib.rt_dev_free.argtypes = [C.c_int32, vp]
    lib.rt_dev_upload.argtypes = [C.c_int32, vp, vp, C.c_size_t]
    lib.rt_dev_download.argtypes = [C.c_int32, vp, vp, C.c_size_t]
    lib.rt_device_count.argtypes = [C.POINTER(C.c_int)]
    for name in ABI_SYMBOLS:
        getattr(lib, name)  # AttributeError if the build lost a symbol
    if path is None:
        _lib = lib
    return lib


def device_count() -> int:
    lib = load_library()
    n = C.c_int(0)
    lib.rt_device_count(C.byref(n))
    return n.value


def _raise(lib, handle, code):
    msg = lib.rt_last_error(handle)
    text = msg.decode("utf-8", "replace") if msg else ""
    if code == RT_E_ONE_SEGMENT:
        # the reference indexes times[1] and raises IndexError (analyze.py:354)
        raise IndexError("index 1 is out of bounds for axis 0 with size 1")
    raise NativeError(code, text)


class DeviceBuffer:
    """A raw hipMalloc allocation owned through the C-ABI (torch-free staging)."""

    def __init__(self, device: int, nbytes: int):
        self._lib = load_library()
        self.device = device
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        rc = self._lib.rt_dev_alloc(device, self.nbytes, C.byref(p))
        if rc != RT_OK:
            _raise(self._lib, None, rc)
        self.ptr = p.value

    def upload(self, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        rc = self._lib.rt_dev_upload(self.device, self.ptr, a.ctypes.data, a.nbytes)
        if rc != RT_OK:
            _raise(self._lib, None, rc)

    def download(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        rc = self._lib.rt_dev_download(self.device, out.ctypes.data, self.ptr, out.nbytes)
        if rc != RT_OK:
            _raise(self._lib, None, rc)
        return out

    def free(self):
        if self.ptr:
            self._lib.rt_dev_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class NativeAnalyzer:
    """Owns one ``rt_handle``."""

    def __init__(
        self,
        *,
        n_streams: int,
        nperseg: int,
        max_samples: int,
        sample_rate: float,
        window_f32: np.ndarray,
        scale: float,
        threshold: float,
        snr_threshold: float,
        calibration_db: float,
        min_duration_s: float,
        max_duration_s: float,
        device: int = 0,
        mode: int = RT_MODE_AUTO,
        hot_capacity: int = 0,
        record_capacity: int = 0,
        segs_per_chunk: int = 0,
        timing: bool = False,
        hip_stream: Optional[int] = None,
        lanes: int = 1,
        subtract_first: bool = False,
        record_pool: int = 0,
        group_detect: Optional[bool] = None,
    ):
        self._lib = load_library()
        self._handle = C.c_void_p()
        w = np.ascontiguousarray(window_f32, dtype=np.float32)
        if w.shape != (nperseg,):
            raise ValueError("window must have nperseg coefficients")
        cfg = RtConfig()
        cfg.device = device
        cfg.n_streams = n_streams
        cfg.nperseg = nperseg
        cfg.mode = mode
        cfg.max_samples = max_samples
        cfg.sample_rate = float(sample_rate)
        cfg.window = w.ctypes.data_as(C.POINTER(C.c_float))
        cfg.scale = float(scale)
        cfg.threshold = float(threshold)
        cfg.snr_threshold = float(snr_threshold)
        cfg.calibration_db = float(calibration_db)
        cfg.min_duration_s = float(min_duration_s)
        cfg.max_duration_s = float(max_duration_s)
        cfg.hot_capacity = hot_capacity
        cfg.record_capacity = record_capacity
        cfg.segs_per_chunk = segs_per_chunk
        cfg.flags = ((RT_FLAG_TIMING if timing else 0) | (RT_FLAG_NO_LIN_DETREND if subtract_first else 0)
                     | (0 if group_detect is None else RT_FLAG_GROUP_DETECT if group_detect else RT_FLAG_NO_GROUP_DETECT))
        cfg.hip_stream = hip_stream
        cfg.lanes = int(lanes)
        cfg.record_pool = int(record_pool)
        rc = self._lib.rt_create(C.byref(cfg), C.byref(self._handle))
        if rc != RT_OK:
            self._handle = C.c_void_p()
            _raise(self._lib, None, rc)
        self.n_streams = n_streams
        self.nperseg = nperseg
        self.device = device
        self.max_samples = max_samples

    # -- lifecycle --------------------------------------------------------
    def close(self):
        if getattr(self, "_handle", None) and self._handle.value:
            self._lib.rt_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != RT_OK:
            _raise(self._lib, self._handle, rc)

    def reset(self):
        self._check(self._lib.rt_reset(self._handle))

    def reset_stream(self, stream: int):
        self._check(self._lib.rt_reset_stream(self._handle, int(stream)))

    def set_stream_params(self, threshold: Optional[np.ndarray], calibration_db: Optional[np.ndarray]):
        """Per-stream linear thresholds / calibration (float32 ``[S]`` each, or None = the handle's value)."""

        def arr(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.shape != (self.n_streams,):
                raise ValueError(f"expected {self.n_streams} values")
            return a

        t, c = arr(threshold), arr(calibration_db)
        self._check(
            self._lib.rt_set_stream_params(
                self._handle, t.ctypes.data if t is not None else None, c.ctypes.data if c is not None else None
            )
        )

    # -- analysis ---------------------------------------------------------
    def process_device(self, iq_ptr: int, n_samples: int, stream_stride: Optional[int] = None):
        self._check(self._lib.rt_process(self._handle, iq_ptr, n_samples, stream_stride or n_samples))

    def process_device_u8(self, iq_ptr: int, n_samples: int, stream_stride: Optional[int] = None):
        self._check(self._lib.rt_process_u8(self._handle, iq_ptr, n_samples, stream_stride or n_samples))

    def process_host(self, iq: np.ndarray):
        a = np.ascontiguousarray(iq, dtype=np.complex64)
        if a.ndim == 1:
            a = a[None, :]
        if a.shape[0] != self.n_streams:
            raise ValueError(f"expected {self.n_streams} streams, got {a.shape[0]}")
        self._keep = a  # async H2D copy reads it until the fetch
        self._check(self._lib.rt_process_host(self._handle, a.ctypes.data, a.shape[1], a.shape[1]))

    def process_host_u8(self, raw: np.ndarray):
        """``[S, 2*B]`` uint8 (interleaved I, Q) in host memory; staged by the library (one buffer per call in flight)."""
        a = np.ascontiguousarray(raw, dtype=np.uint8)
        if a.ndim == 1:
            a = a[None, :]
        if a.shape[0] != self.n_streams or a.shape[1] % 2:
            raise ValueError(f"expected uint8 [{self.n_streams}, 2*B]")
        self._check(self._lib.rt_process_u8_host(self._handle, a.ctypes.data, a.shape[1] // 2, a.shape[1] // 2))

    def fetch(self, allow_truncated: bool = False) -> np.ndarray:
        """Records of the oldest enqueued call.  A call whose records were truncated (``RT_E_CAPACITY``: an ``rt_extract``
        call with more records in a stream than ``record_capacity``, or no memory to grow) raises unless ``allow_truncated`` -- then the truncated list is
        returned and ``last_truncated`` is set; either way it is consumed, so the next fetch belongs to the next
        call.  A call without any result (``RT_E_HOT_OVERFLOW``: sparse mode, candidate lists overflowed) always
        raises: an empty array would read as "no signals"."""
        n = C.c_size_t(0)
        self.last_truncated = False
        rc = self._lib.rt_fetch(self._handle, None, 0, C.byref(n))  # size query: the call stays pending
        if rc != RT_OK and rc != RT_E_CAPACITY:
            self._check(rc)  # incl. RT_E_HOT_OVERFLOW: the library has dropped the call
        out = np.zeros(n.value, dtype=RECORD_DTYPE)
        if n.value:
            rc = self._lib.rt_fetch(self._handle, out.ctypes.data, n.value, C.byref(n))
        if rc == RT_E_CAPACITY:
            self.last_truncated = True
        if rc != RT_OK and not (allow_truncated and rc == RT_E_CAPACITY):
            self._check(rc)
        return out

    def extract_device(self, spec_ptr: int, n_seg: int, n_bins: int, last_ptr: Optional[int], n_seg_last: int):
        self._check(self._lib.rt_extract(self._handle, spec_ptr, n_seg, n_bins, last_ptr, n_seg_last))

    def spectrogram_device(self, iq_ptr: int, n_samples: int, stream_stride: int, out_ptr: int):
        self._check(self._lib.rt_spectrogram(self._handle, iq_ptr, n_samples, stream_stride, out_ptr))

    def calibrate_read(self, iq_ptr: int, n_samples: int, stream_stride: int):
        self._check(self._lib.rt_calibrate_read(self._handle, iq_ptr, n_samples, stream_stride))

    def call_info(self) -> RtCallInfo:
        info = RtCallInfo()
        self._check(self._lib.rt_get_call_info(self._handle, C.byref(info)))
        return info
