/*
 * irmv_hip.h -- C ABI of libirmv_hip.so: the MI355X (gfx950) armor-detection hot path.
 *
 * This is the boundary a maintainer of illini-robomaster/irmv_detection binds
 * against to replace the TensorRT/NPP/CUDA-graph implementation of
 * irmv_detection::YoloEngine and the OpenCV implementation of
 * irmv_detection::PnPSolver (reference include/irmv_detection/yolo_engine.hpp:28-35,
 * include/irmv_detection/pnp_solver.hpp:15-23).  Plain C types only: pointers,
 * sizes, POD structs.  The header-only C++ facade in include/irmv_detection/
 * wraps these entry points 1:1 behind the reference's own class names.
 *
 * Every entry returns IRMV_OK (0) or a negative error code; irmv_last_error()
 * returns a thread-local message.  (The reference checks no CUDA/NPP/TensorRT
 * return code at all -- src/yolo_engine.cpp passim.)
 *
 * There is no CPU fallback anywhere behind this ABI: without a HIP device every
 * compute entry fails with IRMV_ERR_HIP.
 */
#ifndef IRMV_HIP_H
#define IRMV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRMV_OK 0
#define IRMV_ERR_ARG (-1)      /* bad argument / configuration            */
#define IRMV_ERR_HIP (-2)      /* HIP runtime failure (message has detail) */
#define IRMV_ERR_MODEL (-3)    /* weight blob missing or not matching      */

#define IRMV_RESIZE_STRETCH 0   /* reference behaviour: src/yolo_engine.cpp:186-190 */
#define IRMV_RESIZE_LETTERBOX 1 /* north-star variant */

#define IRMV_ARMOR_SMALL 0 /* 135 x 55 mm; the reference always solves with this one (src/pnp_solver.cpp:47) */
#define IRMV_ARMOR_LARGE 1 /* 225 x 55 mm */

#define IRMV_POINTS_AUTO 0
#define IRMV_POINTS_KEYPOINT_HEAD 1
#define IRMV_POINTS_CLASSICAL 2 /* gray -> threshold -> contours -> minAreaRect -> lights, on the GPU */

#define IRMV_NUM_CLASSES 14   /* ArmorClass B1..RS; 14 = UNKNOWN (include/irmv_detection/armor.hpp:7) */
#define IRMV_MAX_DET_CAP 256
#define IRMV_CAND_CAP 8192    /* most candidates the NMS walk can take (upper bound of pre_nms_cap) */

typedef struct irmv_engine irmv_engine;

/* Replaces the arguments of YoloEngine::YoloEngine (src/yolo_engine.cpp:24-26)
 * plus everything that constructor bakes in (640, EfficientNMS thresholds inside
 * the absent TensorRT plan, PnP constants of src/pnp_solver.cpp:7-34). */
typedef struct irmv_engine_cfg {
    uint32_t struct_size;      /* = sizeof(irmv_engine_cfg) */
    int32_t device;            /* HIP device ordinal */
    int32_t src_width;         /* camera frame, e.g. 1280 (cv::Size src_image_size) */
    int32_t src_height;        /* e.g. 1024 */
    int32_t net_size;          /* 640 (src/yolo_engine.cpp:98-99,189) */
    int32_t resize_mode;       /* IRMV_RESIZE_* */
    int32_t rotate180;         /* 1 = reference (nppiMirror both axes, :182-184) */
    int32_t swap_rb;           /* 0 = reference: producer deposits model channel order */
    float score_thr;           /* EfficientNMS score_threshold, default 0.25 */
    float iou_thr;             /* EfficientNMS iou_threshold, default 0.45 */
    int32_t max_det;           /* EfficientNMS max_output_boxes, default 100, <= IRMV_MAX_DET_CAP */
    int32_t pre_nms_cap;       /* candidates entering the NMS walk, default 4096, <= IRMV_CAND_CAP */
    int32_t num_slots;         /* frames in flight; the reference node uses 3 (src/irm_detector.cpp:35-38) */
    int32_t armor_size;        /* IRMV_ARMOR_*: object points of the fused PnP */
    double camera_matrix[9];   /* row-major K (config/camera_info.yaml:7) */
    double dist_coeffs[5];     /* k1 k2 p1 p2 k3 (config/camera_info.yaml:12) */
    const char *weights_path;  /* "<stem>.onnx" (sibling "<stem>.irmw" is loaded, like :28-31) or a ".irmw" path; NULL -> weights_blob */
    const void *weights_blob;  /* .irmw image in host memory, or in device memory if weights_on_device */
    uint64_t weights_bytes;
    int32_t weights_on_device; /* 1: weights_blob is a device pointer (e.g. filled by an RCCL broadcast) */
    int32_t num_streams;       /* compute streams (0 = default: one per 64 slots, at least 2 and at most 4; one per slot for engines
                                  of <= 4 slots).  A multi-slot submit() is cut into that many sub-batches replayed as concurrent
                                  graphs, whose launch gaps and tails fill each other (measured: 128 frames as 2 x 64 +9 % over 1 x 128,
                                  as 3 or 4 graphs -10 %; 192 frames as 3 x 64 +5 % over 128 as 2 x 64); a single-slot submit rides
                                  stream (slot mod num_streams), so the steps of different slots overlap (three single frames in
                                  flight: 6.1 k FPS against 2.7 k one at a time) */
    /* Source of the four armor points PnP consumes.  The reference obtains them by classical CV inside
     * each bbox (IrmDetector::extract_armors, src/irm_detector.cpp:292-355); a pose-style model carries them
     * in a keypoint head.  IRMV_POINTS_AUTO picks the keypoint head when the model has one. */
    int32_t point_source;      /* IRMV_POINTS_* */
    int32_t binary_threshold;  /* 150 (src/irm_detector.cpp:152) */
    float light_min_ratio;     /* 0.1 */
    float light_max_ratio;     /* 0.4 */
    float light_max_angle;     /* 40 degrees */
    float reserved0;
    double armor_min_small_center_distance; /* 0.8 */
    double armor_max_small_center_distance; /* 3.2 */
    double armor_min_large_center_distance; /* 3.2 */
    double armor_max_large_center_distance; /* 5.5 */
} irmv_engine_cfg;

/* One detection: YoloEngine::bbox (yolo_engine.hpp:19-26) in source-frame
 * pixels as produced by parse_output (src/yolo_engine.cpp:202-220), plus what
 * the node derives per armor downstream: the four points PnP consumes
 * (src/pnp_solver.cpp:41-44), rvec/tvec (:49-51) and the pose quaternion
 * (src/irm_detector.cpp:218-226). */
typedef struct irmv_det {
    float xyxy[4];
    float score;
    int32_t class_id;  /* 0..13, 14 = UNKNOWN */
    int32_t anchor;    /* index of the originating anchor (debug / parity) */
    int32_t pnp_ok;    /* 1 if rvec/tvec are valid */
    float kpts[8];     /* left-bottom, left-top, right-top, right-bottom; source-frame pixels */
    double rvec[3];
    double tvec[3];
    double quat[4];    /* x, y, z, w */
    int32_t armor_valid; /* 1: kpts are an armor's points (keypoint head: always; classical: two gated lights found);
                            0: no armor in this bbox; -1: no answer, the extraction scratch was exhausted (the frame's
                            bboxes cover more than 8 frame areas in total, or one bbox holds more than 1024 contours /
                            4096 contour points) -- reported, never replaced by a truncated result */
    int32_t armor_size;  /* IRMV_ARMOR_*: classical path derives it from the light-centre distance (src/irm_detector.cpp:340-341) */
    int32_t n_lights;    /* classical path: lights that passed is_light() in this bbox */
    int32_t reserved;
} irmv_det;

/* EfficientNMS-layout view of one frame's result in net-input coordinates
 * (what the reference binds as num_dets / det_boxes / det_scores / det_classes,
 * src/yolo_engine.cpp:53-57,82-85).  Arrays hold max_det entries. */
typedef struct irmv_raw_dets {
    int32_t num_dets;
    int32_t n_candidates;  /* (anchor, class) pairs above score_thr before any cap */
    float *det_boxes;      /* [max_det][4] xyxy, net-input pixels */
    float *det_scores;     /* [max_det] */
    int32_t *det_classes;  /* [max_det] */
    int32_t *det_anchors;  /* [max_det] */
    float *det_kpts;       /* [max_det][8] net-input pixels */
} irmv_raw_dets;

typedef struct irmv_kernel_stat {
    char name[48];      /* kernel family, e.g. "conv3x3s1_mt2_nt4" */
    char layer[32];     /* graph node, e.g. "model.22.cv2.0.0" */
    double flops;       /* algorithmic FLOPs of this launch (2*MAC) */
    double bytes;       /* algorithmic bytes: inputs once + outputs once + weights once */
    float ms;           /* HIP-event duration on the engine's compute stream */
    int32_t reserved;
} irmv_kernel_stat;

const char *irmv_last_error(void);
const char *irmv_version(void);
int irmv_device_count(int *count);
int irmv_device_synchronize(int device);   /* hipDeviceSynchronize on that device */

void irmv_engine_cfg_default(irmv_engine_cfg *cfg);
int irmv_engine_create(const irmv_engine_cfg *cfg, irmv_engine **out);
void irmv_engine_destroy(irmv_engine *e);
int irmv_engine_num_slots(const irmv_engine *e);
int irmv_engine_max_det(const irmv_engine *e);
int irmv_engine_num_streams(const irmv_engine *e);
int irmv_engine_sync_launch(const irmv_engine *e);   /* how a synchronous single-frame step is launched on this box: 0 = one hipGraph replay, 1 = kernel by kernel
                                                        behind the upload (same kernels, same bits; timed at creation, IRMV_SYNC_LAUNCH=graph|eager forces) */

/* ---- NUMA placement of the frame hand-off (multi-GPU nodes; the reference is single-device, test/yolo_test.cpp:16) ----
 * An engine allocates and first-touches its pinned frame slots on the host NUMA node closest to its device
 * (hipDeviceAttributeHostNumaId; IRMV_NUMA=0 switches that off).  The threads that FILL the slots and submit belong on the
 * same node: a runner binds each of them with irmv_numa_bind_thread(node of its device) before it creates the engine. */
int irmv_engine_numa_node(const irmv_engine *e);     /* host NUMA node of the engine's device; -1 unknown */
int irmv_engine_numa_placed(const irmv_engine *e);   /* 1: the frame slots were allocated under that node's CPU set and memory policy */
int irmv_numa_device_node(int device, int *node);    /* the same attribute without an engine (before irmv_engine_create) */
int irmv_numa_bind_thread(int node);                 /* sched_setaffinity(calling thread, CPUs of /sys/devices/system/node/nodeN/cpulist within the process's cpuset) */
int irmv_numa_page_node(const void *p);              /* node holding the page of p (move_pages query); < 0 unknown */
int irmv_numa_parse_cpulist(const char *s, int *cpus, int cap);   /* "0-3,8" -> {0,1,2,3,8}; returns the count (test aid) */

/* Pinned host frame slot (src_height*src_width*3 bytes, HWC u8), valid for the
 * engine's lifetime; producer threads write straight into it -- the counterpart
 * of YoloEngine::get_src_image_buffer() (yolo_engine.hpp:35) and of the
 * TripleBuffer hand-off (include/irmv_detection/triple_buffer.hpp:24-40). */
uint8_t *irmv_engine_src_buffer(irmv_engine *e, int slot);
/* Device-side staging of the same slot (for producers that already hold the
 * frame in HBM, and for HBM-resident benchmarking). */
void *irmv_engine_src_device_buffer(irmv_engine *e, int slot);

#define IRMV_SUBMIT_H2D 1u          /* copy the pinned slots -> HBM first */
#define IRMV_SUBMIT_ASYNC_UPLOAD 2u /* ... on the engine's upload stream, event-chained to the compute stream: this
                                       group's frames cross PCIe while other groups' kernels run (one cross-stream hop) */

/* Enqueue one step for slots [first, first+count) and return immediately:
 *   [async H2D of the frames] -> ONE hipGraph {preprocess -> network -> decode -> NMS -> keypoints -> PnP} ->
 *   async D2H of the results.
 * With IRMV_SUBMIT_ASYNC_UPLOAD the upload rides a side stream -- the dGPU form of the reference's TripleBuffer
 * hand-off (triple_buffer.hpp:24-40, src/camera.cpp:40-61): submit slot n+1 while slot n is in flight, collect each
 * with irmv_engine_wait_slots().  count == 1 is the reference's per-slot detect(); count > 1 batches independent
 * frames through every kernel.  One thread submits.
 * A multi-slot step is cut into one sub-batch (one captured graph) per compute stream of the engine; a submit of exactly
 * one stream's share of the engine's slots, aligned to it (e.g. slots [128, 256) of a 256-slot, two-stream engine), IS
 * that sub-batch -- the same graph on the same stream -- so a producer may feed the shares separately. */
int irmv_engine_submit(irmv_engine *e, int first_slot, int count, uint32_t flags);
/* Block until everything submitted so far is done and host-visible. */
int irmv_engine_wait(irmv_engine *e);
/* Block until the pinned slots [first, first+count) have been uploaded by their last submit: a producer may overwrite them
 * from then on (the TripleBuffer's consumer can give the buffer back) while the kernels still run. */
int irmv_engine_wait_upload(irmv_engine *e, int first_slot, int count);
/* Block until the results of slots [first, first+count) are host-visible; other slots stay in flight. */
int irmv_engine_wait_slots(irmv_engine *e, int first_slot, int count);
/* Results of one slot after wait(): up to cap detections, score-descending. */
int irmv_engine_results(irmv_engine *e, int slot, irmv_det *out, int cap, int *n);
/* submit(slot, 1, H2D) + wait + results == YoloEngine::detect() (src/yolo_engine.cpp:153-177) */
int irmv_engine_detect(irmv_engine *e, int slot, irmv_det *out, int cap, int *n);
/* Wall-clock ms of the last detect() (get_profiling_time(), yolo_engine.hpp:33) */
double irmv_engine_last_detect_ms(const irmv_engine *e);

/* 180-degree rotated frame of a slot (what get_rotated_image() aliases after the
 * in-place mirror, src/yolo_engine.cpp:77-78,182-184), rotated on the GPU. */
int irmv_engine_rotated_image(irmv_engine *e, int slot, uint8_t *dst_hwc);

/* IrmDetector::extract_armors(get_rotated_image(), bboxes) (src/irm_detector.cpp:183,292-355) on the GPU:
 * for each of the n boxes (xyxy, rotated-frame pixels) on the slot's current frame -> out[i].kpts (LB, LT, RT,
 * RB), armor_valid, armor_size, n_lights, and the PnP pose.  Works for any model / point_source. */
int irmv_engine_extract_armors(irmv_engine *e, int slot, const float *xyxy, int n, irmv_det *out);
/* The node's live parameters of that extraction (IrmDetector::param_event_callback, src/irm_detector.cpp:372-403):
 * binary_threshold, light.{min_ratio,max_ratio,max_angle}, armor.{min_small,max_small,min_large,max_large}_center_distance.
 * Takes effect from the next submit / extract call. */
int irmv_engine_set_extract_params(irmv_engine *e, int binary_threshold, float light_min_ratio, float light_max_ratio,
                                   float light_max_angle, const double center_distances[4]);
/* IRMV_POINTS_KEYPOINT_HEAD or IRMV_POINTS_CLASSICAL: where this engine's four armor points come from (AUTO resolved). */
int irmv_engine_point_source(const irmv_engine *e);

/* ---- stage-wise read-backs used by the parity tests -------------------- */
int irmv_engine_read_input(irmv_engine *e, int slot, float *chw);             /* [3][net][net], as the reference's input_buffer_ */
int irmv_engine_read_head(irmv_engine *e, int slot, float *head);             /* [anchors][64+nc+nk] */
int irmv_engine_write_head(irmv_engine *e, int slot, const float *head);      /* inject a head tensor ... */
int irmv_engine_run_post(irmv_engine *e, int first_slot, int count);          /* ... and run decode->NMS->PnP only */
/* fault injection for the robustness test: overwrite every slot's candidate counter (the one piece of state a step leaves for
 * the next kernel of the same step) with `value`.  The next step must stay inside its buffers and reset the counter; the
 * step after it must be correct again.  IRMV_ERR_ARG for an engine that keeps no counters (IRMV_SPLIT_SCAN=0). */
int irmv_engine_debug_poke_candidate_counts(irmv_engine *e, int value);
int irmv_engine_read_tap(irmv_engine *e, int slot, const char *name, float *nhwc, int shape[3]);
int irmv_engine_read_raw(irmv_engine *e, int slot, irmv_raw_dets *out);
int irmv_engine_num_anchors(const irmv_engine *e);
int irmv_engine_head_channels(const irmv_engine *e);

/* Run one step eagerly with a HIP event pair around every kernel launch. */
int irmv_engine_profile(irmv_engine *e, int first_slot, int count, irmv_kernel_stat *stats, int cap, int *n);

/* ---- PnPSolver (include/irmv_detection/pnp_solver.hpp:15-23) ------------ */
typedef struct irmv_pnp irmv_pnp;
int irmv_pnp_create(int device, const double camera_matrix[9], const double dist_coeffs[5], irmv_pnp **out);
void irmv_pnp_destroy(irmv_pnp *p);
/* img_pts [n][8] (LB, LT, RT, RB; pixels) -> rvec [n][3], tvec [n][3], ok [n];
 * IPPE on the GPU, one lane per armor (cv::solvePnP(..., SOLVEPNP_IPPE), src/pnp_solver.cpp:49-51). */
int irmv_pnp_solve(irmv_pnp *p, const float *img_pts, int n, int armor_size, double *rvec, double *tvec, int32_t *ok);

#ifdef __cplusplus
}
#endif
#endif /* IRMV_HIP_H */
