"""ctypes wrapper around oracle/liboracle.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.  See
oracle/irmv_oracle.h for what each entry restates and why parity is unpinned.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

RESIZE_STRETCH, RESIZE_LETTERBOX = 0, 1


class LightParams(C.Structure):
    _fields_ = [("binary_threshold", C.c_int), ("light_min_ratio", C.c_float), ("light_max_ratio", C.c_float),
                ("light_max_angle", C.c_float), ("armor_min_small_center_distance", C.c_double),
                ("armor_max_small_center_distance", C.c_double), ("armor_min_large_center_distance", C.c_double),
                ("armor_max_large_center_distance", C.c_double)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("orc_net.c", "orc_post.c", "orc_light.c", "irmv_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u8p, f32p, f64p, i32p = (C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_double),
                                 C.POINTER(C.c_int))
        L.orc_preprocess.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, u8p]
        L.orc_preprocess.restype = None
        L.orc_rotate180.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.orc_net_load.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_net_load.restype = C.c_void_p
        L.orc_net_free.argtypes = [C.c_void_p]
        L.orc_net_nc.argtypes = [C.c_void_p]
        L.orc_net_nk.argtypes = [C.c_void_p]
        L.orc_head_channels.argtypes = [C.c_void_p]
        L.orc_num_anchors.argtypes = [C.c_int]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads.restype = None
        L.orc_net_forward.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, f32p, C.c_char_p, f32p, i32p]
        L.orc_conv_layer.argtypes = [C.c_void_p, C.c_char_p, f32p, C.c_int, C.c_int, f32p]
        L.orc_expf.argtypes = [C.c_float]
        L.orc_expf.restype = C.c_float
        L.orc_round_f16.argtypes = [C.c_float]
        L.orc_round_f16.restype = C.c_float
        L.orc_logit_threshold.argtypes = [C.c_float]
        L.orc_logit_threshold.restype = C.c_float
        L.orc_decode_nms.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                     C.c_int, f32p, f32p, i32p, i32p, f32p, i32p]
        L.orc_decode_candidates.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_float, f32p,
                                            C.POINTER(C.c_uint64), C.c_int]
        L.orc_nms_sorted.argtypes = [f32p, i32p, C.c_int, C.c_float, C.c_int, i32p]
        L.orc_parse_output.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
        L.orc_parse_output.restype = None
        L.orc_solve_pnp_ippe.argtypes = [f64p, f64p, f32p, C.c_int, f64p, f64p, f64p, f64p, f64p]
        L.orc_undistort_points.argtypes = [f64p, f64p, f32p, C.c_int, f64p]
        L.orc_project_points.argtypes = [f64p, f64p, f64p, f64p, f64p, C.c_int, f64p]
        L.orc_armor_object_points.argtypes = [C.c_int, f64p]
        L.orc_rodrigues.argtypes = [f64p, f64p]
        L.orc_rvec_to_quat.argtypes = [f64p, f64p]
        L.orc_light_params_default.argtypes = [C.POINTER(LightParams)]
        L.orc_find_external_contours.argtypes = [u8p, C.c_int, C.c_int, C.POINTER(C.c_short), C.c_int, i32p, C.c_int]
        L.orc_min_area_rect.argtypes = [C.POINTER(C.c_short), C.c_int, f32p]
        L.orc_extract_armor.argtypes = [u8p, C.c_int, C.c_int, f32p, C.POINTER(LightParams), i32p, f32p, f32p, i32p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ---- preprocess ----------------------------------------------------------
def preprocess(src: np.ndarray, net: int = 640, mode: int = RESIZE_STRETCH, rotate180: bool = True,
               swap_rb: bool = False, want_u8: bool = False):
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw, _ = src.shape
    out = np.empty((3, net, net), np.float32)
    u8 = np.empty((net, net, 3), np.uint8) if want_u8 else None
    lib().orc_preprocess(_p(src, C.c_uint8), sw, sh, net, mode, int(rotate180), int(swap_rb),
                         _p(out, C.c_float), _p(u8, C.c_uint8) if want_u8 else None)
    return (out, u8) if want_u8 else out


def rotate180(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.empty_like(src)
    lib().orc_rotate180(_p(src, C.c_uint8), src.shape[1], src.shape[0], _p(dst, C.c_uint8))
    return dst


# ---- network -------------------------------------------------------------
class Net:
    def __init__(self, blob: bytes):
        self._h = lib().orc_net_load(blob, len(blob))
        if not self._h:
            raise ValueError("oracle: bad weight blob")
        self.nc = lib().orc_net_nc(self._h)
        self.nk = lib().orc_net_nk(self._h)
        self.no = lib().orc_head_channels(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_net_free(self._h)
            self._h = None

    def forward(self, in_chw: np.ndarray, emulate_fp16: bool = False, tap: str | None = None):
        in_chw = np.ascontiguousarray(in_chw, np.float32)
        net = in_chw.shape[-1]
        A = lib().orc_num_anchors(net)
        head = np.empty((A, self.no), np.float32)
        tap_buf, shape = None, (C.c_int * 3)()
        if tap is not None:
            tap_buf = np.empty(net * net * 16, np.float32)   # largest module output (layer 0)
        rc = lib().orc_net_forward(self._h, _p(in_chw, C.c_float), net, int(emulate_fp16),
                                   _p(head, C.c_float), tap.encode() if tap else None,
                                   _p(tap_buf, C.c_float) if tap else None, shape)
        if rc != 0:
            raise RuntimeError("oracle forward failed")
        if tap is not None:
            H, W, Cc = shape[0], shape[1], shape[2]
            if H == 0:
                raise KeyError(tap)
            return head, tap_buf[:H * W * Cc].reshape(H, W, Cc).copy()
        return head

    def conv_layer(self, name: str, x_nhwc: np.ndarray, cout: int, stride: int) -> np.ndarray:
        x = np.ascontiguousarray(x_nhwc, np.float32)
        H, W, _ = x.shape
        y = np.empty((H // stride, W // stride, cout), np.float32)
        rc = lib().orc_conv_layer(self._h, name.encode(), _p(x, C.c_float), H, W, _p(y, C.c_float))
        if rc != 0:
            raise RuntimeError(f"oracle conv {name} failed")
        return y


def expf(x: float) -> float:
    return float(lib().orc_expf(C.c_float(x)))


def round_f16(x: float) -> float:
    return float(lib().orc_round_f16(C.c_float(x)))


# ---- decode + NMS ----------------------------------------------------------
def decode_nms(head: np.ndarray, net: int, nc: int, nk: int, score_thr: float = 0.25,
               iou_thr: float = 0.45, max_det: int = 100, pre_nms_cap: int = 4096):
    head = np.ascontiguousarray(head, np.float32)
    boxes = np.zeros((max_det, 4), np.float32)
    scores = np.zeros(max_det, np.float32)
    classes = np.zeros(max_det, np.int32)
    anchors = np.zeros(max_det, np.int32)
    kpts = np.zeros((max_det, max(nk, 1)), np.float32)
    ncand = C.c_int(0)
    n = lib().orc_decode_nms(_p(head, C.c_float), net, nc, nk, score_thr, iou_thr, max_det, pre_nms_cap,
                             _p(boxes, C.c_float), _p(scores, C.c_float), _p(classes, C.c_int),
                             _p(anchors, C.c_int), _p(kpts, C.c_float), C.byref(ncand))
    return dict(num_dets=n, boxes=boxes[:n], scores=scores[:n], classes=classes[:n],
                anchors=anchors[:n], kpts=kpts[:n, :nk], n_candidates=ncand.value)


def decode_candidates(head: np.ndarray, net: int, nc: int, nk: int, score_thr: float = 0.25):
    head = np.ascontiguousarray(head, np.float32)
    A = lib().orc_num_anchors(net)
    boxes = np.empty((A, 4), np.float32)
    keys = np.empty(A * nc, np.uint64)
    n = lib().orc_decode_candidates(_p(head, C.c_float), net, nc, nk, score_thr, _p(boxes, C.c_float),
                                    _p(keys, C.c_uint64), A * nc)
    return boxes, keys[:n]


def nms_sorted(boxes: np.ndarray, classes: np.ndarray, iou_thr: float, max_det: int) -> np.ndarray:
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
    classes = np.ascontiguousarray(classes, np.int32)
    keep = np.zeros(max(max_det, 1), np.int32)
    n = lib().orc_nms_sorted(_p(boxes, C.c_float), _p(classes, C.c_int), len(classes), iou_thr, max_det,
                             _p(keep, C.c_int))
    return keep[:n].copy()


def parse_output(det_boxes: np.ndarray, src_w: int, src_h: int, net: int = 640,
                 mode: int = RESIZE_STRETCH) -> np.ndarray:
    b = np.ascontiguousarray(det_boxes, np.float32).reshape(-1, 4)
    out = np.empty_like(b)
    lib().orc_parse_output(_p(b, C.c_float), len(b), src_w, src_h, net, mode, _p(out, C.c_float))
    return out


# ---- PnP -------------------------------------------------------------------
def solve_pnp_ippe(K, D, img_pts, armor_size: int = 0):
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    D = np.ascontiguousarray(D, np.float64).reshape(5)
    pts = np.ascontiguousarray(img_pts, np.float32).reshape(8)
    r1, t1, r2, t2 = (np.zeros(3) for _ in range(4))
    err = np.zeros(2)
    ok = lib().orc_solve_pnp_ippe(_p(K, C.c_double), _p(D, C.c_double), _p(pts, C.c_float), armor_size,
                                  _p(r1, C.c_double), _p(t1, C.c_double), _p(r2, C.c_double),
                                  _p(t2, C.c_double), _p(err, C.c_double))
    return dict(ok=bool(ok), rvec=r1, tvec=t1, rvec2=r2, tvec2=t2, err=err)


def undistort_points(K, D, pts):
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    D = np.ascontiguousarray(D, np.float64).reshape(5)
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    out = np.empty(pts.shape, np.float64)
    lib().orc_undistort_points(_p(K, C.c_double), _p(D, C.c_double), _p(pts, C.c_float), len(pts),
                               _p(out, C.c_double))
    return out


def project_points(K, D, rvec, tvec, obj):
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    D = np.ascontiguousarray(D, np.float64).reshape(5)
    rvec = np.ascontiguousarray(rvec, np.float64).reshape(3)
    tvec = np.ascontiguousarray(tvec, np.float64).reshape(3)
    obj = np.ascontiguousarray(obj, np.float64).reshape(-1, 3)
    out = np.empty((len(obj), 2), np.float64)
    lib().orc_project_points(_p(K, C.c_double), _p(D, C.c_double), _p(rvec, C.c_double),
                             _p(tvec, C.c_double), _p(obj, C.c_double), len(obj), _p(out, C.c_double))
    return out


def armor_object_points(armor_size: int = 0) -> np.ndarray:
    obj = np.zeros(12)
    lib().orc_armor_object_points(armor_size, _p(obj, C.c_double))
    return obj.reshape(4, 3)


def rodrigues(rvec) -> np.ndarray:
    rvec = np.ascontiguousarray(rvec, np.float64).reshape(3)
    R = np.zeros(9)
    lib().orc_rodrigues(_p(rvec, C.c_double), _p(R, C.c_double))
    return R.reshape(3, 3)


def rvec_to_quat(rvec) -> np.ndarray:
    rvec = np.ascontiguousarray(rvec, np.float64).reshape(3)
    q = np.zeros(4)
    lib().orc_rvec_to_quat(_p(rvec, C.c_double), _p(q, C.c_double))
    return q


# ---- classical light extraction (row f1) ----------------------------------------
def light_params(**kw) -> "LightParams":
    P = LightParams()
    lib().orc_light_params_default(C.byref(P))
    for k, v in kw.items():
        setattr(P, k, v)
    return P


def find_external_contours(binary: np.ndarray):
    """-> list of int16 [n, 2] (x, y) arrays, OpenCV order (last found first)"""
    b = np.ascontiguousarray(binary, np.uint8)
    h, w = b.shape
    cap = 8 * (w + h) * 8 + 4096
    pts = np.zeros((cap, 2), np.int16)
    off = np.zeros(4097, np.int32)
    n = lib().orc_find_external_contours(_p(b, C.c_uint8), w, h, _p(pts, C.c_short), cap, _p(off, C.c_int), 4096)
    return [pts[off[i]:off[i + 1]].copy() for i in range(n)]


def min_area_rect(pts: np.ndarray) -> np.ndarray:
    p = np.ascontiguousarray(pts, np.int16).reshape(-1, 2)
    out = np.zeros(8, np.float32)
    lib().orc_min_area_rect(_p(p, C.c_short), len(p), _p(out, C.c_float))
    return out.reshape(4, 2)


def extract_armor(img: np.ndarray, xyxy, params=None):
    img = np.ascontiguousarray(img, np.uint8)
    rows, cols, _ = img.shape
    P = params or light_params()
    box = np.ascontiguousarray(xyxy, np.float32).reshape(4)
    size, nl = C.c_int(0), C.c_int(0)
    pts, center = np.zeros(8, np.float32), np.zeros(2, np.float32)
    ok = lib().orc_extract_armor(_p(img, C.c_uint8), cols, rows, _p(box, C.c_float), C.byref(P), C.byref(size),
                                 _p(pts, C.c_float), _p(center, C.c_float), C.byref(nl))
    return dict(ok=bool(ok), size=size.value, pts=pts.reshape(4, 2), center=center, n_lights=nl.value)
