"""ctypes binding of include/irmv_hip.h (libirmv_hip.so).

Thin by design: every function here is one C-ABI call.  There is no fallback of
any kind -- if the HIP library is missing or a call fails, IrmvError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _build

OK, ERR_ARG, ERR_HIP, ERR_MODEL = 0, -1, -2, -3
RESIZE_STRETCH, RESIZE_LETTERBOX = 0, 1
ARMOR_SMALL, ARMOR_LARGE = 0, 1
SUBMIT_H2D = 1
SUBMIT_ASYNC_UPLOAD = 2
POINTS_AUTO, POINTS_KEYPOINT_HEAD, POINTS_CLASSICAL = 0, 1, 2
NUM_CLASSES = 14
MAX_DET_CAP = 256
CAND_CAP = 8192


class IrmvError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"irmv_hip error {code}: {msg}")
        self.code = code


class EngineCfg(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32),
        ("src_width", C.c_int32), ("src_height", C.c_int32), ("net_size", C.c_int32),
        ("resize_mode", C.c_int32), ("rotate180", C.c_int32), ("swap_rb", C.c_int32),
        ("score_thr", C.c_float), ("iou_thr", C.c_float),
        ("max_det", C.c_int32), ("pre_nms_cap", C.c_int32), ("num_slots", C.c_int32),
        ("armor_size", C.c_int32),
        ("camera_matrix", C.c_double * 9), ("dist_coeffs", C.c_double * 5),
        ("weights_path", C.c_char_p), ("weights_blob", C.c_void_p), ("weights_bytes", C.c_uint64),
        ("weights_on_device", C.c_int32), ("num_streams", C.c_int32),
        ("point_source", C.c_int32), ("binary_threshold", C.c_int32),
        ("light_min_ratio", C.c_float), ("light_max_ratio", C.c_float), ("light_max_angle", C.c_float), ("reserved0", C.c_float),
        ("armor_min_small_center_distance", C.c_double), ("armor_max_small_center_distance", C.c_double),
        ("armor_min_large_center_distance", C.c_double), ("armor_max_large_center_distance", C.c_double),
    ]


class Det(C.Structure):
    _fields_ = [
        ("xyxy", C.c_float * 4), ("score", C.c_float), ("class_id", C.c_int32),
        ("anchor", C.c_int32), ("pnp_ok", C.c_int32), ("kpts", C.c_float * 8),
        ("rvec", C.c_double * 3), ("tvec", C.c_double * 3), ("quat", C.c_double * 4),
        ("armor_valid", C.c_int32), ("armor_size", C.c_int32), ("n_lights", C.c_int32), ("reserved", C.c_int32),
    ]


class RawDets(C.Structure):
    _fields_ = [
        ("num_dets", C.c_int32), ("n_candidates", C.c_int32),
        ("det_boxes", C.POINTER(C.c_float)), ("det_scores", C.POINTER(C.c_float)),
        ("det_classes", C.POINTER(C.c_int32)), ("det_anchors", C.POINTER(C.c_int32)),
        ("det_kpts", C.POINTER(C.c_float)),
    ]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("layer", C.c_char * 32), ("flops", C.c_double),
                ("bytes", C.c_double), ("ms", C.c_float), ("reserved", C.c_int32)]


# every symbol include/irmv_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("irmv_last_error", C.c_char_p, []),
    ("irmv_version", C.c_char_p, []),
    ("irmv_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("irmv_device_synchronize", C.c_int, [C.c_int]),
    ("irmv_engine_cfg_default", None, [C.POINTER(EngineCfg)]),
    ("irmv_engine_create", C.c_int, [C.POINTER(EngineCfg), C.POINTER(_P)]),
    ("irmv_engine_destroy", None, [_P]),
    ("irmv_engine_num_slots", C.c_int, [_P]),
    ("irmv_engine_max_det", C.c_int, [_P]),
    ("irmv_engine_num_streams", C.c_int, [_P]),
    ("irmv_engine_sync_launch", C.c_int, [_P]),
    ("irmv_engine_numa_node", C.c_int, [_P]),
    ("irmv_engine_numa_placed", C.c_int, [_P]),
    ("irmv_numa_device_node", C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    ("irmv_numa_bind_thread", C.c_int, [C.c_int]),
    ("irmv_numa_page_node", C.c_int, [C.c_void_p]),
    ("irmv_numa_parse_cpulist", C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.c_int]),
    ("irmv_engine_src_buffer", C.POINTER(C.c_uint8), [_P, C.c_int]),
    ("irmv_engine_src_device_buffer", C.c_void_p, [_P, C.c_int]),
    ("irmv_engine_submit", C.c_int, [_P, C.c_int, C.c_int, C.c_uint32]),
    ("irmv_engine_wait", C.c_int, [_P]),
    ("irmv_engine_wait_slots", C.c_int, [_P, C.c_int, C.c_int]),
    ("irmv_engine_wait_upload", C.c_int, [_P, C.c_int, C.c_int]),
    ("irmv_engine_set_extract_params", C.c_int, [_P, C.c_int, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_double)]),
    ("irmv_engine_point_source", C.c_int, [_P]),
    ("irmv_engine_results", C.c_int, [_P, C.c_int, C.POINTER(Det), C.c_int, C.POINTER(C.c_int)]),
    ("irmv_engine_detect", C.c_int, [_P, C.c_int, C.POINTER(Det), C.c_int, C.POINTER(C.c_int)]),
    ("irmv_engine_last_detect_ms", C.c_double, [_P]),
    ("irmv_engine_rotated_image", C.c_int, [_P, C.c_int, C.POINTER(C.c_uint8)]),
    ("irmv_engine_extract_armors", C.c_int, [_P, C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(Det)]),
    ("irmv_engine_read_input", C.c_int, [_P, C.c_int, C.POINTER(C.c_float)]),
    ("irmv_engine_read_head", C.c_int, [_P, C.c_int, C.POINTER(C.c_float)]),
    ("irmv_engine_write_head", C.c_int, [_P, C.c_int, C.POINTER(C.c_float)]),
    ("irmv_engine_run_post", C.c_int, [_P, C.c_int, C.c_int]),
    ("irmv_engine_debug_poke_candidate_counts", C.c_int, [_P, C.c_int]),
    ("irmv_engine_read_tap", C.c_int, [_P, C.c_int, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    ("irmv_engine_read_raw", C.c_int, [_P, C.c_int, C.POINTER(RawDets)]),
    ("irmv_engine_num_anchors", C.c_int, [_P]),
    ("irmv_engine_head_channels", C.c_int, [_P]),
    ("irmv_engine_profile", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(KernelStat), C.c_int, C.POINTER(C.c_int)]),
    ("irmv_pnp_create", C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_P)]),
    ("irmv_pnp_destroy", None, [_P]),
    ("irmv_pnp_solve", C.c_int, [_P, C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
]

_lib = None


def lib_path() -> str:
    """The in-tree library; IRMV_LIB_PATH names another build of it (A/B runs of two kernel versions on one box)."""
    return os.environ.get("IRMV_LIB_PATH") or _build.LIB_PATH


def load():
    """Load libirmv_hip.so (must have been built in-tree; see __graft_entry__.build)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise IrmvError(ERR_HIP, f"{path} is missing: build the HIP extension first "
                                     "(python -m irmv_detection_amd._build); there is no CPU fallback")
        L = C.CDLL(path)
        for name, res, args in SYMBOLS:
            if os.environ.get("IRMV_LIB_PATH") and not hasattr(L, name):
                continue                    # A/B runs against an OLDER build of the library: entries added since are absent there
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, allow=()):
    if rc != OK and rc not in allow:
        raise IrmvError(rc, load().irmv_last_error().decode(errors="replace"))
    return rc


def device_count() -> int:
    """HIP devices visible to this process (0 without a GPU)."""
    n = C.c_int(0)
    return n.value if load().irmv_device_count(C.byref(n)) == OK else 0


def device_synchronize(device: int = 0) -> None:
    check(load().irmv_device_synchronize(device))


def numa_parse_cpulist(text: str):
    """The library's own parser of a sysfs cpulist ("0-3,8,10-11"), as used for thread placement."""
    buf = (C.c_int * 4096)()
    n = load().irmv_numa_parse_cpulist(text.encode(), buf, 4096)
    return [buf[i] for i in range(min(n, 4096))]


def numa_bind_to_device(device: int = 0) -> int:
    """Bind the calling thread to the CPUs of the host NUMA node closest to `device` (multi-GPU runners call this per rank /
    per worker thread BEFORE creating the engine and filling its slots).  Returns the node, or -1 if nothing was bound."""
    node = C.c_int(-1)
    if not hasattr(load(), "irmv_numa_device_node"):
        return -1
    if load().irmv_numa_device_node(device, C.byref(node)) != OK or node.value < 0:
        return -1
    return node.value if load().irmv_numa_bind_thread(node.value) == OK else -1
