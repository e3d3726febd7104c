/*
 * irmv_oracle.h -- CPU oracle for the armor-detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library.  The HIP engine never calls into it.
 *
 * PARITY UNPINNED: the reference (illini-robomaster/irmv_detection) asserts no
 * numerical output anywhere in its tests (test/yolo_test.cpp:36,106 check only a
 * box count and a latency), ships no model file, and delegates all arithmetic to
 * TensorRT / NPP / OpenCV, none of which is present in /root/reference or in this
 * image.  The reference cannot be compiled here (needs NvInfer.h, npp.h, OpenCV,
 * ROS2 ament).  This oracle is therefore a restatement of the *published*
 * algorithms at the reference's call sites, cross-checked against torch-CPU,
 * numpy and scipy in tests/ ("Oracle").
 *
 * Reference call sites restated here:
 *   orc_preprocess      src/yolo_engine.cpp:179-200 (mirror, resize, /255, HWC->CHW)
 *   orc_net_forward     src/yolo_engine.cpp:105 (TensorRT enqueueV3: YOLOv8n body,
 *                       architecture per Ultralytics yolov8.yaml scale n)
 *   orc_decode_nms      EfficientNMS_TRT plugin bound at src/yolo_engine.cpp:53-57,
 *                       :82-85; TensorRT OSS efficientNMSPlugin semantics
 *   orc_parse_output    src/yolo_engine.cpp:202-220
 *   orc_solve_pnp_ippe  src/pnp_solver.cpp:18-51 (cv::solvePnP, SOLVEPNP_IPPE;
 *                       OpenCV calib3d ippe.cpp; Collins & Bartoli IJCV 2014)
 *   orc_rvec_to_quat    src/irm_detector.cpp:218-226 (cv::Rodrigues + tf2 getRotation)
 *   orc_extract_armor   src/irm_detector.cpp:292-355 + include/irmv_detection/armor.hpp:11-77
 *                       (cvtColor / threshold / findContours / minAreaRect restated)
 */
#ifndef IRMV_ORACLE_H
#define IRMV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_RESIZE_STRETCH 0
#define ORC_RESIZE_LETTERBOX 1

/* ---- preprocess ------------------------------------------------------- */
/* src: u8 HWC [sh][sw][3].  out_chw: f32 [3][net][net] in [0,1].
 * out_u8 (optional): the resized u8 HWC image [net][net][3] (the reference's
 * resized_image_buffer_, src/yolo_engine.cpp:186-190). */
void orc_preprocess(const uint8_t *src, int sw, int sh, int net, int resize_mode,
                    int rotate180, int swap_rb, float *out_chw, uint8_t *out_u8);

/* 180-degree rotation of a u8 HWC frame (nppiMirror both axes, :182-184). */
void orc_rotate180(const uint8_t *src, int sw, int sh, uint8_t *dst);

/* ---- network ---------------------------------------------------------- */
typedef struct orc_net orc_net;
orc_net *orc_net_load(const uint8_t *blob, size_t bytes);
void orc_net_free(orc_net *);
int orc_net_nc(const orc_net *);
int orc_net_nk(const orc_net *);
int orc_head_channels(const orc_net *); /* 64 + nc + nk */
int orc_num_anchors(int net);
void orc_set_threads(int n);   /* OpenMP threads of orc_net_forward */

/* in_chw f32 [3][net][net] -> head f32 [anchors][64+nc+nk] (levels P3,P4,P5,
 * each row-major).  emulate_fp16 != 0 rounds the input and every stored
 * activation to IEEE fp16 (the HIP engine's storage type); accumulation is fp32
 * either way.  tap (optional): name of a module output ("0".."21", or
 * "22.cv2.0.1" style head intermediates) copied NHWC into tap_out; tap_shape
 * receives {H, W, C}.  Returns 0 on success. */
int orc_net_forward(const orc_net *, const float *in_chw, int net, int emulate_fp16,
                    float *head, const char *tap, float *tap_out, int *tap_shape);

/* one conv on an NHWC f32 tensor (unit-test entry) */
int orc_conv_layer(const orc_net *, const char *layer_name, const float *x_nhwc, int H, int W,
                   float *y_nhwc);

/* ---- decode + NMS (strict fp32, no contraction) ------------------------ */
float orc_expf(float x);
float orc_logit_threshold(float score_thr);

/* head [anchors][64+nc+nk] -> EfficientNMS-style outputs, score-descending.
 * det_boxes [max_det][4] xyxy in net-input pixels; det_kpts [max_det][nk]
 * (net-input pixels, may be NULL); det_anchor [max_det] (may be NULL).
 * n_candidates: number of (anchor, class) pairs above threshold before the
 * pre_nms_cap cut.  Returns num_dets. */
int orc_decode_nms(const float *head, int net, int nc, int nk, float score_thr, float iou_thr,
                   int max_det, int pre_nms_cap, float *det_boxes, float *det_scores,
                   int *det_classes, int *det_anchor, float *det_kpts, int *n_candidates);

/* decode only: boxes [anchors][4], optional sorted candidate keys (u64) */
int orc_decode_candidates(const float *head, int net, int nc, int nk, float score_thr,
                          float *boxes, uint64_t *keys, int keys_cap);

/* greedy class-aware NMS on an explicit, already score-sorted candidate list */
int orc_nms_sorted(const float *boxes /*[n][4]*/, const int *classes, int n, float iou_thr,
                   int max_det, int *keep_idx);

/* parse_output scaling (src/yolo_engine.cpp:202-220) */
void orc_parse_output(const float *det_boxes, int n, int src_w, int src_h, int net,
                      int resize_mode, float *out_xyxy);

/* ---- PnP (fp64) --------------------------------------------------------- */
/* img_pts: 4 points, order left-bottom, left-top, right-top, right-bottom
 * (src/pnp_solver.cpp:41-44), source-frame pixels.  armor_size 0 = small
 * (135x55 mm), 1 = large (225x55 mm) (include/irmv_detection/pnp_solver.hpp:29-32).
 * Outputs the lower-reprojection-error solution first.  Returns 1 on success. */
int orc_solve_pnp_ippe(const double K[9], const double D[5], const float img_pts[8],
                       int armor_size, double rvec[3], double tvec[3], double rvec2[3],
                       double tvec2[3], double err[2]);
void orc_undistort_points(const double K[9], const double D[5], const float *pts, int n,
                          double *out_xy);
void orc_project_points(const double K[9], const double D[5], const double rvec[3],
                        const double tvec[3], const double *obj /*[n][3]*/, int n,
                        double *out_uv);
void orc_armor_object_points(int armor_size, double obj[12]);
void orc_rodrigues(const double rvec[3], double R[9]);
void orc_rvec_to_quat(const double rvec[3], double quat_xyzw[4]);

/* ---- classical armor-point extraction (row f1; orc_light.c) ---------------- */
typedef struct orc_light_params {
    int binary_threshold;                       /* 150  (src/irm_detector.cpp:152) */
    float light_min_ratio, light_max_ratio;     /* 0.1, 0.4 */
    float light_max_angle;                      /* 40 degrees */
    double armor_min_small_center_distance, armor_max_small_center_distance;   /* 0.8, 3.2 */
    double armor_min_large_center_distance, armor_max_large_center_distance;   /* 3.2, 5.5 */
} orc_light_params;
void orc_light_params_default(orc_light_params *P);
/* external contours (RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) of a binary w x h image, OpenCV order */
int orc_find_external_contours(const uint8_t *bin, int w, int h, short *pts, int pts_cap, int *offsets, int max_contours);
void orc_min_area_rect(const short *pts, int n, float corners[8]);
/* IrmDetector::extract_armors for ONE bbox on the (rotated) u8 HWC frame; pts = LB, LT, RT, RB */
int orc_extract_armor(const uint8_t *img, int cols, int rows, const float xyxy[4], const orc_light_params *P,
                      int *size, float pts[8], float center[2], int *n_lights_out);

#ifdef __cplusplus
}
#endif
#endif
