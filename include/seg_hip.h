/*
 * seg_hip.h -- C ABI of the person-segmentation operators in libcoma_hip.so (MI355X / gfx950, fp32 arithmetic, NHWC fp32 storage).
 *
 * The reference segments the person with detectron2's PointRend (Mask R-CNN R50-FPN + point head):
 *   utils/adaptive_mask_inpainting.py:1182-1236  PointRendPredictor: DefaultPredictor(cfg)(image) on the decoded x0 of 21 of the 49
 *                                                denoising steps and on every final image, person masks merged by np.any
 *   src/generation/segment_human.py:24-169       the post-inpaint stage: the same predictor over every inpainted image
 *   imports/pointrend/config/ (yaml)           the architecture constants
 * detectron2 / torchvision are third-party and not under the reference tree; these entry points are the operators that network
 * decomposes into at inference.  coma_amd/seg/model.py records them into a launch plan (sd_model_* of sd_hip.h: one hipGraph per
 * forward, no host synchronisation inside -- every data-dependent size (proposals kept, detections) stays on the device as a count that
 * the next operator reads).  Same conventions as coma_hip.h: int return codes, coma_last_error(), caller-owned device buffers,
 * explicit hipStream_t.  All index outputs follow ONE tie rule: equal scores are taken in ascending index order.
 */
#ifndef SEG_HIP_H
#define SEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* operator codes of a recorded launch (sd_plan.h PK_SEG: PlanRec.i[0]) */
enum {
  SEG_OP_CONV = 1, SEG_OP_RESIZE, SEG_OP_MAXPOOL, SEG_OP_SUBSAMPLE, SEG_OP_MEMSET, SEG_OP_RPN_SELECT, SEG_OP_SORT, SEG_OP_NMS, SEG_OP_ROI_ALIGN,
  SEG_OP_BOX_PREDICT, SEG_OP_FINALIZE, SEG_OP_POINT_SAMPLE, SEG_OP_UPSAMPLE2X, SEG_OP_TOPK_POINTS, SEG_OP_POINT_LOGIT, SEG_OP_PASTE, SEG_OP_RPN_SELECT_LEVELS
};

/* out[m, n] = act(sum_k A[m, k] W[n, k] + bias[n] (+ res)),  m = (b, oy, ox), k = (ky, kx, c): Conv2d / Linear + folded FrozenBatchNorm
 * + ReLU + residual of detectron2's ResNet / FPN / RPN head / box head / mask heads.  fp32 in, fp32 MFMA, fp32 out. */
typedef struct seg_conv_desc {
  const void* x;        /* f32 [batch, in_h, in_w, ldx], c channels used; c % 4 == 0 */
  int batch, in_h, in_w, c, ldx;     /* ldx = 0: c */
  const void* w;        /* f32 [n][kpad], k = (ky * kw + kx) * c + ch, zero beyond kh * kw * c */
  int n, kpad;          /* kpad % 32 == 0 */
  int kh, kw, stride, pad;
  int out_h, out_w;
  const void* bias;     /* f32 [n] or NULL */
  const void* res;      /* f32 residual or NULL */
  int ldr;              /* 0: n */
  int res_mode;         /* 0 none; 1: [batch * out_h * out_w][ldr]; 2: [batch, out_h / 2, out_w / 2][ldr] read at (oy >> 1, ox >> 1) = FPN's
                           `lateral + F.interpolate(top_down, scale_factor=2, mode="nearest")` */
  void* out;            /* f32 [batch * out_h * out_w][ldo] */
  int ldo;              /* 0: n */
  int relu;
  const void* m_dev;    /* i32 [M / unit_rows] on the device or NULL: of every unit of unit_rows consecutive rows (one image's ROIs) only the
                           first m_dev[u] * rows_per_item rows are computed */
  int rows_per_item, unit_rows;
  int tile;             /* 0 = by rule (n <= 32: 128x32, n <= 64: 128x64, else 128x128 -- or 128x64 where the 128x128 grid leaves the last
                           round of the CUs mostly empty); 1 / 2 / 3 = 128x128 / 128x64 / 128x32 (tests, tuning) */
  void* workspace;      /* f32 scratch for split-K partial tiles or NULL (then K is never split) */
  size_t workspace_bytes;
  int split_k;          /* 0 = by rule (fewer than 1024 tiles and K >= 256: the slice count a per-CU cost model picks, slices of >= 128 k values,
                           as many as the workspace holds), -1 = never, S > 1 = exactly S slices (tests); the slices are summed in order:
                           deterministic */
} seg_conv_desc;
int seg_conv_gemm_f32(const seg_conv_desc* d, void* stream);

/* DefaultPredictor's ResizeShortestEdge (PIL bilinear on uint8: horizontal pass, uint8 in between, vertical pass; 22-bit fixed-point
 * weights computed by the host exactly as Pillow does) + GeneralizedRCNN.preprocess_image (- pixel_mean, zero pad to [pad_h, pad_w]).
 * src u8 [batch, h, w, 3] -> out f32 [batch, pad_h, pad_w, 4] (channel 3 zero); resized (optional) u8 [batch, new_h, new_w, 3].
 * bounds i32 [new][2] = (first source index, taps), kk i32 [new][ksize]. */
int seg_resize_normalize_u8(const void* src, int batch, int h, int w, int new_h, int new_w, int pad_h, int pad_w, const void* bounds_x,
                            const void* kk_x, int ksize_x, const void* bounds_y, const void* kk_y, int ksize_y, float mean0, float mean1,
                            float mean2, void* tmp /* u8 [batch, h, new_w, 3] */, void* resized, void* out, void* stream);

/* F.max_pool2d(x, 3, stride 2, padding 1) (ResNet stem) and F.max_pool2d(x, 1, stride 2) (FPN LastLevelMaxPool); NHWC f32, c % 4 == 0 */
int seg_maxpool3x3s2_f32(const void* x, int batch, int h, int w, int c, void* out, void* stream);
int seg_subsample2_f32(const void* x, int batch, int h, int w, int c, void* out, void* stream);
int seg_memset(void* dst, int byte, size_t bytes, void* stream);

/* RPN.predict_proposals + the per-level half of find_top_rpn_proposals for one FPN level: the pre_topk best of the fh * fw * 3 anchors by
 * objectness logit (ties: ascending anchor index), decoded (Box2BoxTransform, weights 1) and clipped to the image.
 * pred f32 [batch][fh * fw][ld]: columns 0..2 logits of the 3 anchors, 3 + 4a .. 6 + 4a the deltas of anchor a.
 * Candidate slot s in [cand_offset, cand_offset + k): key u64 = (descending-order bits of the logit) << 32 | (anchor_base + anchor index);
 * key = ~0 marks a slot that takes no part (non-finite or empty box); boxes f32 [.][4]; group i32 = level. */
int seg_rpn_select(const void* pred, int ld, int batch, int fh, int fw, int stride, const void* cell_anchors /* f32 [3][4], DefaultAnchorGenerator's
                   cell anchors of this level, computed by the host as detectron2 does (f64 -> f32) */, int level, int anchor_base, int pre_topk,
                   float img_h, float img_w, int cand_offset, int cap, void* cand_keys, void* cand_boxes, void* cand_group, void* stream);

/* seg_rpn_select for n_levels (<= 6) consecutive FPN levels in ONE launch (a level is one workgroup per image: side by side the levels take
 * as long as the largest).  Level l: stride first_stride << l, group id l, anchor_base = anchors of the levels before it, candidate slots
 * [sum of min(anchors, pre_topk) of the levels before it, + min(anchors, pre_topk)) -- the layout RPN.predict_proposals' per-level loop
 * produces (utils/adaptive_mask_inpainting.py:1225-1236 -> detectron2 find_top_rpn_proposals).  preds / cell_anchors / fh / fw: HOST arrays. */
int seg_rpn_select_levels(const void* const* preds, const void* const* cell_anchors, const int* fh, const int* fw, int n_levels, int first_stride,
                          int ld, int batch, int pre_topk, float img_h, float img_w, int cap, void* cand_keys, void* cand_boxes, void* cand_group,
                          void* key_scratch /* optional u32 [batch][anchors of all levels]: the objectness keys are first written densely (one more
                          launch over the whole chip) and the selection's five passes read them coalesced; NULL: read in place */, void* stream);

/* Sort the cap (a power of two <= 8192) candidate slots of each image by key, ascending (= score descending, ties ascending index).
 * -> sorted boxes / scores / group / source (low 32 bits of the key), n_valid i32 [batch] = slots with key != ~0. */
int seg_sort_candidates(const void* keys, const void* boxes, const void* group, int batch, int cap, void* s_boxes, void* s_scores,
                        void* s_group, void* s_src, void* n_valid, void* stream);

/* torchvision batched_nms ("vanilla": IoU on raw coordinates, same group only; IoU = inter / (area_i + area_j - inter) > thresh
 * suppresses) over the sorted candidates; the first max_keep survivors in visiting order.
 * mask_ws u64 [batch][cap][cap / 64].  keep_pos i32 [batch][max_keep] = sorted positions; out_* gathered rows (rows >= count zero). */
int seg_nms(const void* s_boxes, const void* s_scores, const void* s_group, const void* s_src, const void* n_valid, int batch, int cap,
            float thresh, int max_keep, void* mask_ws, void* keep_pos, void* out_boxes, void* out_scores, void* out_group, void* out_src,
            void* out_count, void* stream);

/* ROIPooler(7, scales 1/4..1/32, sampling_ratio 0, ROIAlignV2 = aligned): level = clamp(floor(4 + log2(sqrt(area) / 224 + 1e-8)), 2, 5);
 * torchvision roi_align with ceil(roi / out) samples per bin and axis.  boxes f32 [batch][R][4], count i32 [batch]
 * -> out f32 [batch * R][out_size * out_size * c] (NHWC per ROI), level i32 [batch * R] (optional). */
int seg_roi_align_f32(const void* p2, const void* p3, const void* p4, const void* p5, int h2, int w2, int c, const void* boxes,
                      const void* count, int batch, int R, int out_size, void* out, void* level, void* stream);

/* FastRCNNOutputLayers.inference up to the candidate list: softmax over 81 logits, per-class Box2BoxTransform (weights 10, 10, 5, 5),
 * clip, score > thresh.  pred f32 [batch * R][ld] = 81 logits | 320 deltas.  Candidates as in seg_rpn_select: key low word =
 * roi * 80 + class, group = class.  cand_count i32 [batch] counts every candidate that passed (may exceed cap: overflow is detectable). */
int seg_box_predict(const void* pred, int ld, const void* proposals, const void* count, int batch, int R, float img_h, float img_w,
                    float score_thresh, int cap, void* cand_keys, void* cand_boxes, void* cand_group, void* cand_count, void* probs,
                    void* stream);

/* detector_postprocess on the boxes: scale by (out_w / img_w, out_h / img_h), clip to the output image, valid = nonempty.
 * det_boxes f32 [batch][R][4] -> out_boxes f32 [batch][R][4], valid i32 [batch][R] (0 beyond count). */
int seg_finalize_detections(const void* det_boxes, const void* count, int batch, int R, float img_h, float img_w, int out_h, int out_w,
                            void* out_boxes, void* valid, void* stream);

/* point_rend point_sample / point_sample_fine_grained_features (F.grid_sample bilinear, zero padding, align_corners False), NHWC.
 * per_roi = 0: feat f32 [batch, fh, fw, c], point = box-relative coords mapped through the ROI's box to image pixels, x feat_scale;
 * per_roi = 1: feat f32 [batch * R, fh, fw, c] (the coarse mask of each ROI), coords relative to the map.
 * coords f32 [batch * R][P][2] (x, y) or NULL = the regular grid_side x grid_side grid ((i + 0.5) / side, x fastest), P = side^2.
 * -> out[copy][(roi * P + p) * ldo + col0 + ch], copies spaced copy_stride floats (the coarse features feed every point-head layer). */
int seg_point_sample_f32(const void* feat, int fh, int fw, int c, int per_roi, float feat_scale, const void* boxes, const void* count,
                         int batch, int R, const void* coords, int P, int grid_side, void* out, int ldo, int col0, int n_copies,
                         long long copy_stride, void* stream);

/* F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) of n maps [s, s] -> [2s, 2s]; rows limited by *count * R-units */
int seg_upsample2x_f32(const void* x, const void* count, int batch, int R, int s, void* out, void* stream);

/* get_uncertain_point_coords_on_grid on the predicted-class logits: the k positions of smallest |logit| of each [s, s] map (ties:
 * ascending index), ascending index order -> idx i32 [batch * R][k], coords f32 [batch * R][k][2] = ((idx % s + 0.5) / s, (idx / s + 0.5) / s) */
int seg_topk_points(const void* logits, const void* count, int batch, int R, int s, int k, void* idx, void* coords, void* stream);

/* StandardPointHead.predictor restricted to each instance's own class (the only channel mask_rcnn_inference reads) + the scatter into
 * the [s, s] logit map: map[roi][idx[roi][p]] = dot(x[roi * P + p][0:kdim], w[class[roi]]) + bias[class[roi]]; idx NULL = p. */
int seg_point_logit_scatter(const void* x, int ldx, int kdim, const void* w, const void* bias, const void* classes, const void* count,
                            int batch, int R, int P, const void* idx, void* map, int s, void* stream);

/* mask_rcnn_inference (sigmoid) + paste_masks_in_image (grid_sample over the whole image, >= 0.5) + the plug-in's merge
 * (utils/adaptive_mask_inpainting.py:1230-1234: np.any over the masks of cat_id).  logits f32 [batch * R][s][s], out_boxes / valid of
 * seg_finalize_detections.  masks u8 [batch][R][out_h][out_w] (optional), merged u8 [batch][out_h][out_w]. */
int seg_paste_masks(const void* logits, int s, const void* out_boxes, const void* valid, const void* classes, const void* count, int batch,
                    int R, int out_h, int out_w, int cat_id, void* masks, void* merged, void* stream);

#ifdef __cplusplus
}
#endif
#endif
