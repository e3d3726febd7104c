"""Python mirror of the reference's operator interface for the hot path.

Same names, argument meaning and error behaviour as
`irmv_detection::YoloEngine` (reference include/irmv_detection/yolo_engine.hpp:16-73)
and `irmv_detection::PnPSolver` (include/irmv_detection/pnp_solver.hpp:12-38), so
the parity tests read like the reference's own tests (test/yolo_test.cpp).
Everything here is plumbing above the C ABI of libirmv_hip.so; all arithmetic
runs in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import enum
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import capi
from .capi import IrmvError  # noqa: F401  (re-export)


class ArmorClass(enum.IntEnum):
    """reference include/irmv_detection/armor.hpp:7"""
    B1 = 0; B2 = 1; B3 = 2; B4 = 3; B5 = 4; BO = 5; BS = 6
    R1 = 7; R2 = 8; R3 = 9; R4 = 10; R5 = 11; RO = 12; RS = 13
    UNKNOWN = 14


class ArmorSize(enum.IntEnum):
    """reference include/irmv_detection/armor.hpp:9"""
    SMALL = 0; LARGE = 1; UNKNOWN = 2


@dataclass
class bbox:
    """YoloEngine::bbox, reference include/irmv_detection/yolo_engine.hpp:19-26"""
    xyxy: Tuple[float, float, float, float]
    score: float
    class_id: ArmorClass


@dataclass
class Light:
    """The two members of reference armor.hpp:11-53 that PnP consumes."""
    top: Tuple[float, float] = (0.0, 0.0)
    bottom: Tuple[float, float] = (0.0, 0.0)


@dataclass
class Armor:
    """reference include/irmv_detection/armor.hpp:55-77, plus the pose the node
    derives per armor (src/irm_detector.cpp:204-230)."""
    left_light: Light = field(default_factory=Light)
    right_light: Light = field(default_factory=Light)
    size: ArmorSize = ArmorSize.SMALL
    armor_class: ArmorClass = ArmorClass.UNKNOWN
    confidence: float = 0.0
    center: Tuple[float, float] = (0.0, 0.0)
    bbox_xyxy: Tuple[float, float, float, float] = (0.0, 0.0, 0.0, 0.0)
    pnp_ok: bool = False
    valid: bool = True      # the four points exist (classical extraction found two gated lights / keypoint head)
    n_lights: int = 0
    no_answer: bool = False  # classical extraction ran out of scratch for this bbox (irmv_det.armor_valid == -1)
    rvec: Optional[np.ndarray] = None
    tvec: Optional[np.ndarray] = None
    quat_xyzw: Optional[np.ndarray] = None

    def image_points(self) -> np.ndarray:
        """left.bottom, left.top, right.top, right.bottom (src/pnp_solver.cpp:41-44)"""
        return np.array([self.left_light.bottom, self.left_light.top,
                         self.right_light.top, self.right_light.bottom], np.float32)


# config/camera_info.yaml:7,12
DEFAULT_CAMERA_MATRIX = (957.669211, 0.0, 345.943891, 0.0, 969.127115, 284.057302, 0.0, 0.0, 1.0)
DEFAULT_DIST_COEFFS = (-0.405274, 0.126058, -0.026939, -0.006503, 0.0)


class YoloEngine:
    """YoloEngine(onnx_file_path, src_image_size, enable_profiling=False).

    `onnx_file_path`: like the reference (src/yolo_engine.cpp:28-40) the sibling
    compiled model is loaded -- here `<stem>.irmw`.  Keyword-only extras select
    what the reference bakes in: weights from memory (`weights_blob` = bytes, or
    `weights_device_ptr` + `weights_bytes` after an RCCL broadcast), `num_slots`
    (the reference node builds 3 engines, one per TripleBuffer slot), `slot` (the
    slot `detect()` works on), NMS thresholds, resize mode, PnP constants.
    """

    def __init__(self, onnx_file_path: Optional[str], src_image_size: Tuple[int, int] = (1280, 1024),
                 enable_profiling: bool = False, *, device: int = 0, net_size: int = 640,
                 weights_blob: Optional[bytes] = None, weights_device_ptr: int = 0, weights_bytes: int = 0,
                 num_slots: int = 1, slot: int = 0, resize_mode: int = capi.RESIZE_STRETCH,
                 rotate180: bool = True, swap_rb: bool = False, score_thr: float = 0.25,
                 iou_thr: float = 0.45, max_det: int = 100, pre_nms_cap: int = 4096,
                 camera_matrix: Sequence[float] = DEFAULT_CAMERA_MATRIX,
                 dist_coeffs: Sequence[float] = DEFAULT_DIST_COEFFS, armor_size: int = capi.ARMOR_SMALL,
                 num_streams: int = 0, point_source: int = capi.POINTS_AUTO, binary_threshold: int = 150,
                 light_min_ratio: float = 0.1, light_max_ratio: float = 0.4, light_max_angle: float = 40.0,
                 armor_center_distances: Sequence[float] = (0.8, 3.2, 3.2, 5.5), warmup: int = 0):
        L = capi.load()
        cfg = capi.EngineCfg()
        L.irmv_engine_cfg_default(C.byref(cfg))
        cfg.device = device
        cfg.src_width, cfg.src_height = int(src_image_size[0]), int(src_image_size[1])
        cfg.net_size = net_size
        cfg.resize_mode, cfg.rotate180, cfg.swap_rb = resize_mode, int(rotate180), int(swap_rb)
        cfg.score_thr, cfg.iou_thr, cfg.max_det, cfg.pre_nms_cap = score_thr, iou_thr, max_det, pre_nms_cap
        cfg.num_slots, cfg.armor_size = num_slots, armor_size
        cfg.num_streams = num_streams
        cfg.point_source, cfg.binary_threshold = point_source, binary_threshold
        cfg.light_min_ratio, cfg.light_max_ratio, cfg.light_max_angle = light_min_ratio, light_max_ratio, light_max_angle
        (cfg.armor_min_small_center_distance, cfg.armor_max_small_center_distance,
         cfg.armor_min_large_center_distance, cfg.armor_max_large_center_distance) = armor_center_distances
        cfg.camera_matrix = (C.c_double * 9)(*camera_matrix)
        cfg.dist_coeffs = (C.c_double * 5)(*(list(dist_coeffs) + [0.0] * 5)[:5])
        self._blob_keepalive = None
        if weights_device_ptr:
            cfg.weights_blob, cfg.weights_bytes, cfg.weights_on_device = weights_device_ptr, weights_bytes, 1
        elif weights_blob is not None:
            self._blob_keepalive = C.create_string_buffer(weights_blob, len(weights_blob))
            cfg.weights_blob = C.cast(self._blob_keepalive, C.c_void_p)
            cfg.weights_bytes = len(weights_blob)
        elif onnx_file_path is not None:
            cfg.weights_path = os.fsencode(onnx_file_path)
        self._h = C.c_void_p()
        self._L = L
        rc = L.irmv_engine_create(C.byref(cfg), C.byref(self._h))
        self._blob_keepalive = None
        if rc != capi.OK:
            self._h = None
            capi.check(rc)
        self.src_image_size = (cfg.src_width, cfg.src_height)
        self.net_size = net_size
        self.num_slots = num_slots
        self.slot = slot
        self.max_det = max_det
        self.enable_profiling = enable_profiling
        self.num_streams = L.irmv_engine_num_streams(self._h)
        self.sync_launch = "eager" if (hasattr(L, "irmv_engine_sync_launch") and L.irmv_engine_sync_launch(self._h) == 1) else "graph"   # how detect() reaches the GPU on this box (include/irmv_hip.h)
        has_numa = hasattr(L, "irmv_engine_numa_node")               # (absent only in an older build loaded through IRMV_LIB_PATH)
        self.numa_node = L.irmv_engine_numa_node(self._h) if has_numa else -1            # host NUMA node closest to the device (-1 unknown)
        self.numa_placed = bool(L.irmv_engine_numa_placed(self._h)) if has_numa else False  # the pinned frame slots were allocated / first touched there
        self.num_anchors = L.irmv_engine_num_anchors(self._h)
        self.head_channels = L.irmv_engine_head_channels(self._h)
        self._dets = (capi.Det * max_det)()
        # the reference warms up with 50 detect() calls in its constructor (:114-116)
        for _ in range(warmup):
            self.detect()

    # ---- lifetime -------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.irmv_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- reference API -----------------------------------------------------------
    def get_src_image_buffer(self, slot: Optional[int] = None) -> np.ndarray:
        """uint8 [H, W, 3] view of the pinned frame slot (yolo_engine.hpp:35)."""
        slot = self.slot if slot is None else slot
        p = self._L.irmv_engine_src_buffer(self._h, slot)
        if not p:
            raise IrmvError(capi.ERR_ARG, "bad slot")
        w, h = self.src_image_size
        return np.ctypeslib.as_array(p, shape=(h, w, 3))

    def src_page_node(self, slot: int = 0) -> int:
        """NUMA node that holds the first page of the pinned frame slot (move_pages query); < 0: unknown."""
        p = self._L.irmv_engine_src_buffer(self._h, slot)
        return int(self._L.irmv_numa_page_node(C.cast(p, C.c_void_p)))

    def detect(self, slot: Optional[int] = None) -> List[bbox]:
        """std::vector<bbox> detect() (src/yolo_engine.cpp:153-177)."""
        return [bbox(tuple(d.xyxy), d.score, ArmorClass(d.class_id)) for d in self._detect_raw(slot)]

    def detect_armors(self, slot: Optional[int] = None) -> List[Armor]:
        """detect() plus what the node derives per armor: the four points and the
        pose (src/irm_detector.cpp:181-230), all computed on the GPU."""
        return [self._to_armor(d) for d in self._detect_raw(slot)]

    def get_profiling_time(self) -> float:
        """ms of the last detect() (yolo_engine.hpp:33)."""
        return float(self._L.irmv_engine_last_detect_ms(self._h))

    def get_rotated_image(self, slot: Optional[int] = None) -> np.ndarray:
        """The 180-degree rotated frame (yolo_engine.hpp:34)."""
        slot = self.slot if slot is None else slot
        w, h = self.src_image_size
        out = np.empty((h, w, 3), np.uint8)
        capi.check(self._L.irmv_engine_rotated_image(self._h, slot, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def extract_armors(self, bboxes, slot: Optional[int] = None) -> List[Armor]:
        """IrmDetector::extract_armors(get_rotated_image(), bboxes) (src/irm_detector.cpp:292-355) on the GPU,
        on the slot's current frame; bboxes: `bbox` objects or xyxy rows in rotated-frame pixels.  Returns one
        Armor per input box (check `.valid`)."""
        slot = self.slot if slot is None else slot
        rows = [b.xyxy if isinstance(b, bbox) else b for b in bboxes]
        xy = np.ascontiguousarray(rows, np.float32).reshape(-1, 4)
        out = (capi.Det * max(len(xy), 1))()
        capi.check(self._L.irmv_engine_extract_armors(self._h, slot, xy.ctypes.data_as(C.POINTER(C.c_float)), len(xy), out))
        res = []
        for i in range(len(xy)):
            a = self._to_armor(out[i])
            if isinstance(bboxes[i], bbox):
                a.armor_class, a.confidence = bboxes[i].class_id, bboxes[i].score
            res.append(a)
        return res

    # 5 x 7 bitmap glyphs of the ArmorClass names (B1..B5, BO, BS, R1..R5, RO, RS, UNKNOWN): the same table as the C++ facade
    # (include/irmv_detection/yolo_engine.hpp), drawn at scale 3 = the size of FONT_HERSHEY_SIMPLEX at scale 1
    _GLYPHS = {"B": (0x1e, 0x11, 0x11, 0x1e, 0x11, 0x11, 0x1e), "R": (0x1e, 0x11, 0x11, 0x1e, 0x14, 0x12, 0x11),
               "O": (0x0e, 0x11, 0x11, 0x11, 0x11, 0x11, 0x0e), "S": (0x0f, 0x10, 0x10, 0x0e, 0x01, 0x01, 0x1e),
               "U": (0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x0e), "N": (0x11, 0x19, 0x15, 0x13, 0x11, 0x11, 0x11),
               "K": (0x11, 0x12, 0x14, 0x18, 0x14, 0x12, 0x11), "W": (0x11, 0x11, 0x11, 0x15, 0x15, 0x1b, 0x11),
               "1": (0x04, 0x0c, 0x04, 0x04, 0x04, 0x04, 0x0e), "2": (0x0e, 0x11, 0x01, 0x02, 0x04, 0x08, 0x1f),
               "3": (0x1e, 0x01, 0x01, 0x0e, 0x01, 0x01, 0x1e), "4": (0x02, 0x06, 0x0a, 0x12, 0x1f, 0x02, 0x02),
               "5": (0x1f, 0x10, 0x1e, 0x01, 0x01, 0x11, 0x0e)}

    @classmethod
    def _draw_label(cls, image: np.ndarray, text: str, x: int, y: int, color) -> None:
        """cv::putText(image, text, (x, y), FONT_HERSHEY_SIMPLEX, 1, color, 2) without OpenCV: (x, y) is the text's bottom-left corner."""
        S = 3
        h, w = image.shape[:2]
        for k, ch in enumerate(text):
            g = cls._GLYPHS.get(ch)
            if g is None:
                continue
            for r in range(7):
                for c in range(5):
                    if (g[r] >> (4 - c)) & 1:
                        x0, y0 = x + k * 6 * S + c * S, y - 7 * S + r * S
                        image[max(y0, 0):max(min(y0 + S, h), 0), max(x0, 0):max(min(x0 + S, w), 0)] = color

    def visualize_bboxes(self, image: np.ndarray, bboxes: Sequence[bbox]) -> None:
        """Draw 2-px rectangles and the class name in place; like the reference, print and return on
        a size mismatch (src/yolo_engine.cpp:222-243)."""
        w, h = self.src_image_size
        if image.shape[1] != w or image.shape[0] != h:
            print("[YoloEngine::visualize_bboxes] Image size mismatch")
            return
        for b in bboxes:
            color = (0, 0, 255) if b.class_id.name[0] == "B" else (255, 0, 0)
            x1, y1, x2, y2 = (int(v) for v in b.xyxy)
            x1c, x2c = max(min(x1, w - 1), 0), max(min(x2, w - 1), 0)
            y1c, y2c = max(min(y1, h - 1), 0), max(min(y2, h - 1), 0)
            for t in range(2):
                for y in (y1 + t, y2 - t):
                    if 0 <= y < h:
                        image[y, x1c:x2c + 1] = color
                for x in (x1 + t, x2 - t):
                    if 0 <= x < w:
                        image[y1c:y2c + 1, x] = color
            self._draw_label(image, b.class_id.name, x1, y1, color)

    # ---- batched / asynchronous extension (MI355X-first surface) ------------------
    def submit(self, first_slot: int = 0, count: Optional[int] = None, h2d: bool = True, async_upload: bool = False) -> None:
        """Enqueue slots [first_slot, first_slot + count) and return.  `async_upload`: the frames' H2D copy rides the
        engine's upload stream, event-chained to the kernels, so it overlaps other slot groups' compute."""
        count = self.num_slots - first_slot if count is None else count
        flags = (capi.SUBMIT_H2D if h2d else 0) | (capi.SUBMIT_ASYNC_UPLOAD if (h2d and async_upload) else 0)
        capi.check(self._L.irmv_engine_submit(self._h, first_slot, count, flags))

    def wait(self) -> None:
        capi.check(self._L.irmv_engine_wait(self._h))

    def wait_upload(self, first_slot: int, count: int = 1) -> None:
        """Block until these pinned slots have been uploaded (a producer may overwrite them; the kernels still run)."""
        capi.check(self._L.irmv_engine_wait_upload(self._h, first_slot, count))

    def wait_slots(self, first_slot: int, count: int = 1) -> None:
        """Block until these slots' results are host-visible; other slots stay in flight."""
        capi.check(self._L.irmv_engine_wait_slots(self._h, first_slot, count))

    def results(self, slot: int) -> List[Armor]:
        n = C.c_int(0)
        capi.check(self._L.irmv_engine_results(self._h, slot, self._dets, self.max_det, C.byref(n)))
        return [self._to_armor(self._dets[i]) for i in range(n.value)]

    def src_device_ptr(self, slot: int = 0) -> int:
        return int(self._L.irmv_engine_src_device_buffer(self._h, slot))

    # ---- stage read-backs for parity tests -----------------------------------------
    def read_input(self, slot: int = 0) -> np.ndarray:
        out = np.empty((3, self.net_size, self.net_size), np.float32)
        capi.check(self._L.irmv_engine_read_input(self._h, slot, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def read_head(self, slot: int = 0) -> np.ndarray:
        out = np.empty((self.num_anchors, self.head_channels), np.float32)
        capi.check(self._L.irmv_engine_read_head(self._h, slot, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def write_head(self, head: np.ndarray, slot: int = 0) -> None:
        head = np.ascontiguousarray(head, np.float32)
        assert head.shape == (self.num_anchors, self.head_channels)
        capi.check(self._L.irmv_engine_write_head(self._h, slot, head.ctypes.data_as(C.POINTER(C.c_float))))

    def run_post(self, first_slot: int = 0, count: int = 1) -> None:
        capi.check(self._L.irmv_engine_run_post(self._h, first_slot, count))

    def read_tap(self, name: str, slot: int = 0) -> np.ndarray:
        shape = (C.c_int * 3)()
        capi.check(self._L.irmv_engine_read_tap(self._h, slot, name.encode(), None, shape))
        out = np.empty(tuple(shape), np.float32)
        capi.check(self._L.irmv_engine_read_tap(self._h, slot, name.encode(),
                                                out.ctypes.data_as(C.POINTER(C.c_float)), shape))
        return out

    def read_raw(self, slot: int = 0) -> dict:
        """EfficientNMS-layout outputs in net-input coordinates (what the reference
        binds at src/yolo_engine.cpp:82-85)."""
        md = self.max_det
        boxes = np.zeros((md, 4), np.float32); scores = np.zeros(md, np.float32)
        classes = np.zeros(md, np.int32); anchors = np.zeros(md, np.int32); kpts = np.zeros((md, 8), np.float32)
        raw = capi.RawDets(0, 0, boxes.ctypes.data_as(C.POINTER(C.c_float)), scores.ctypes.data_as(C.POINTER(C.c_float)),
                           classes.ctypes.data_as(C.POINTER(C.c_int32)), anchors.ctypes.data_as(C.POINTER(C.c_int32)),
                           kpts.ctypes.data_as(C.POINTER(C.c_float)))
        capi.check(self._L.irmv_engine_read_raw(self._h, slot, C.byref(raw)))
        n = raw.num_dets
        return dict(num_dets=n, n_candidates=raw.n_candidates, boxes=boxes[:n], scores=scores[:n],
                    classes=classes[:n], anchors=anchors[:n], kpts=kpts[:n],
                    boxes_padded=boxes, scores_padded=scores)

    def debug_poke_candidate_counts(self, value: int) -> None:
        """Fault injection (tests): overwrite every slot's candidate counter."""
        capi.check(self._L.irmv_engine_debug_poke_candidate_counts(self._h, int(value)))

    def profile(self, first_slot: int = 0, count: Optional[int] = None) -> List[dict]:
        count = self.num_slots - first_slot if count is None else count
        stats = (capi.KernelStat * 256)()
        n = C.c_int(0)
        capi.check(self._L.irmv_engine_profile(self._h, first_slot, count, stats, 256, C.byref(n)))
        return [dict(name=stats[i].name.decode(), layer=stats[i].layer.decode(), flops=stats[i].flops,
                     bytes=stats[i].bytes, ms=stats[i].ms) for i in range(min(n.value, 256))]

    # ---- internals -------------------------------------------------------------------
    def _detect_raw(self, slot):
        slot = self.slot if slot is None else slot
        n = C.c_int(0)
        capi.check(self._L.irmv_engine_detect(self._h, slot, self._dets, self.max_det, C.byref(n)))
        return [self._dets[i] for i in range(n.value)]

    @staticmethod
    def _to_armor(d) -> Armor:
        k = list(d.kpts)
        a = Armor(left_light=Light(top=(k[2], k[3]), bottom=(k[0], k[1])),
                  right_light=Light(top=(k[4], k[5]), bottom=(k[6], k[7])),
                  size=ArmorSize(d.armor_size) if d.armor_size in (0, 1) else ArmorSize.UNKNOWN,
                  armor_class=ArmorClass(d.class_id), confidence=d.score, valid=d.armor_valid == 1, n_lights=d.n_lights, no_answer=d.armor_valid < 0,
                  bbox_xyxy=tuple(d.xyxy), pnp_ok=bool(d.pnp_ok),
                  rvec=np.array(d.rvec), tvec=np.array(d.tvec), quat_xyzw=np.array(d.quat))
        a.center = ((k[0] + k[2] + k[4] + k[6]) / 4.0, (k[1] + k[3] + k[5] + k[7]) / 4.0)
        return a


class PnPSolver:
    """PnPSolver(camera_matrix[9], distortion_coefficients) -- reference
    include/irmv_detection/pnp_solver.hpp:15-23, src/pnp_solver.cpp."""

    def __init__(self, camera_matrix: Sequence[float], distortion_coefficients: Sequence[float], *, device: int = 0):
        self._L = capi.load()
        self.camera_matrix = np.asarray(camera_matrix, np.float64).reshape(9).copy()
        d = list(distortion_coefficients) + [0.0] * 5
        self.dist_coeffs = np.asarray(d[:5], np.float64)
        self._h = C.c_void_p()
        capi.check(self._L.irmv_pnp_create(device, self.camera_matrix.ctypes.data_as(C.POINTER(C.c_double)),
                                           self.dist_coeffs.ctypes.data_as(C.POINTER(C.c_double)), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self._L.irmv_pnp_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def solvePnP(self, armor: Armor):
        """-> (ok, rvec[3], tvec[3]).  Like the reference, always solves with the
        SMALL armor model (src/pnp_solver.cpp:47-48)."""
        ok, r, t = self.solve_batch(armor.image_points().reshape(1, 8), capi.ARMOR_SMALL)
        return bool(ok[0]), r[0], t[0]

    def solve_batch(self, img_pts: np.ndarray, armor_size: int = capi.ARMOR_SMALL):
        pts = np.ascontiguousarray(img_pts, np.float32).reshape(-1, 8)
        n = len(pts)
        rvec = np.zeros((n, 3)); tvec = np.zeros((n, 3)); ok = np.zeros(n, np.int32)
        capi.check(self._L.irmv_pnp_solve(self._h, pts.ctypes.data_as(C.POINTER(C.c_float)), n, armor_size,
                                          rvec.ctypes.data_as(C.POINTER(C.c_double)),
                                          tvec.ctypes.data_as(C.POINTER(C.c_double)),
                                          ok.ctypes.data_as(C.POINTER(C.c_int32))))
        return ok, rvec, tvec

    def calculateDistanceToCenter(self, image_point: Tuple[float, float]) -> float:
        """|p - (cx, cy)|.  The reference reads cx, cy with at<float>() from a
        CV_64F matrix (src/pnp_solver.cpp:56-57) and so uses garbage; this uses
        the true principal point (documented deviation)."""
        cx, cy = float(self.camera_matrix[2]), float(self.camera_matrix[5])
        return float(np.hypot(np.float32(image_point[0]) - np.float32(cx), np.float32(image_point[1]) - np.float32(cy)))
