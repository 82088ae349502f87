/*
 * coma_hip.h -- C ABI of libcoma_hip.so, the MI355X (gfx950) implementation of ComA's dense hot path.
 *
 * The reference (snuvclab/coma) has no native layer: its boundary for this path is a set of Python
 * methods on torch tensors.  Each entry point below replaces the tensor-op chain of one such method;
 * the citation after "replaces:" is the reference file:line (relative to the reference repo root).
 * The Python host mirror in coma_amd/ binds these with ctypes (see INTEGRATION.md for the stub a
 * reference maintainer would add to utils/coma.py).
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (COMA_E_*); coma_last_error() gives the text
 *     (thread-local).  Nothing throws, nothing allocates device memory: the caller owns every
 *     buffer (device pointers unless stated) and passes the hipStream_t to launch on (NULL = default).
 *   - row-major, innermost index last.  H = #human vertices, O = #object points, N = #orientation
 *     bins, S = #samples in this call, R = voxels per axis.
 *   - "in/out" accumulators are added to, never overwritten, so calls compose over sample batches
 *     and an all-reduce(SUM) over ranks gives the single-process result.
 */
#ifndef COMA_HIP_H
#define COMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever ANY exported signature of coma_hip.h / sd_hip.h / seg_hip.h changes; coma_amd/_lib.py refuses a library that
 * reports another value (a stale build loaded through COMA_HIP_LIB would otherwise be called with mismatched argument lists) */
#define COMA_ABI_VERSION 8

#define COMA_OK 0
#define COMA_E_INVALID (-1) /* bad argument (null pointer, non-positive size, unsupported shape) */
#define COMA_E_LAUNCH (-2)  /* HIP reported an error at launch */
#define COMA_E_DEVICE (-3)  /* no usable gfx950 device / runtime failure */

int coma_abi_version(void);
const char* coma_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * K1+K2+K3  fused contact / relative-orientation accumulator.
 * replaces: utils/coma.py:279-323 (ComA.aggregate_single_sample_for_contact) looped over samples as
 *           in utils/coma.py:257-268, with canonicalize_a_wrt_b_to_p (utils/coma.py:123-172),
 *           geodesic_gaussian_scores (:102-112) and negative_exp (:116-119) fused in.
 * human_verts/human_normals : f32 [S,H,3]
 * obj_verts/obj_normals     : f32 [S,O,3], or [O,3] shared by all samples when obj_sample_stride == 0
 *                             (obj_sample_stride is in floats: O*3 for per-sample objects)
 * sphere_grid               : f32 [N,3]   (the f32 rounding of get_uniform_points_on_sphere, :18-26)
 * principle_vec/sub_vec     : HOST pointers, 3 floats each
 * prob_h_wrt_o, prob_o_wrt_h: f32 [H,O,N] in/out ;  nom, den, cnt : f32 [H,O] in/out
 * Numerics: distance/threshold test bit-exact w.r.t. the reference's f32 sequence; the histogram is
 * evaluated in f32 (reference: f64 intermediates rounded into f32 sums), parity <= 1e-3 relative.
 * ------------------------------------------------------------------------------------------- */
int coma_contact_accumulate_f32(const float* human_verts, const float* human_normals,
                                const float* obj_verts, const float* obj_normals,
                                int64_t obj_sample_stride, const float* sphere_grid,
                                int S, int H, int O, int N,
                                const float* principle_vec, const float* sub_principle_vec,
                                float spatial_grid_size, float spatial_grid_thres,
                                float normal_gaussian_sigma, float eps,
                                float* prob_h_wrt_o, float* prob_o_wrt_h,
                                float* nom, float* den, float* cnt, void* stream);

/* K4a+K4b  normalise a histogram in place and reduce it to the per-pair contact map.
 * replaces: utils/coma.py:328-330 (normalize_prob_grid_for_normals, one grid) fused with
 *           utils/coma.py:342-356 (compute_contact_map, one of "human"/"obj").
 * prob [M,N] in/out (M = H*O) becomes prob/(sum_k prob + eps);
 * contact[m] = (sum_k prob[m,k] * (1 - p.n_k)/2) * nom[m]/den[m].   contact may be NULL
 * (normalise only). */
int coma_contact_map_f32(float* prob, const float* sphere_grid, const float* principle_vec /*host*/,
                         const float* nom, const float* den, int64_t M, int N, float eps,
                         float* contact, void* stream);

/* K4b  significant pairs + masked max.
 * replaces: utils/coma.py:376-377 (significant_contact_pairs) and :402-427
 *           (aggregate_contact_for_significant_pairs).
 * pairs u8 [H,O] = cnt >= threshold (threshold = f32(ratio * used_count), formed by the caller);
 * col_any u8 [O], row_any u8 [H] = any over the other axis;
 * which = 0 ("human"): out f32 [H] = max over {o : col_any[o]} contact[h,o], zeros if none
 * which = 1 ("obj")  : out f32 [O] = max over {h : row_any[h]} contact[h,o], zeros if none.
 * contact/out may be NULL (pairs + any vectors only). */
int coma_significant_pairs_u8(const float* cnt, float threshold, int H, int O, uint8_t* pairs,
                              uint8_t* col_any, uint8_t* row_any, void* stream);
int coma_masked_max_f32(const float* contact, const uint8_t* col_any, const uint8_t* row_any, int H,
                        int O, int which, float* out, void* stream);

/* K4c  negated normalised Shannon entropy of the (normalised) histogram.
 * replaces: utils/coma.py:328-330 + :455-463 / :467-475 (compute_nonphysical_response_sphere).
 * prob [M,N] in/out is normalised in place first; score[m] = 1 + sum_k plogp(round(p*n_bin)/n_bin)/ln(n_bin). */
int coma_entropy_f32(float* prob, int64_t M, int N, float eps, float n_bin, float* score, void* stream);

/* Consumer of the accumulator state: the optimisation app's target selection.
 * replaces: src/application/optimize.py:190-192 (`np.argmax(prob_grid_canon_human_wrt_obj[:, o_ref, :], axis=1)` -> the
 *           bin whose direction becomes the per-vertex orientation target) and :195-196 (`np.nonzero(np.max(nom / denom,
 *           axis=1) > contact_threshold)`, `np.argmax(nom[selected], axis=1)`).
 * coma_row_argmax_i64: idx[m] = argmax_k x[m*row_stride + col_offset + k] (k < n) with NumPy's rules -- the FIRST
 *   maximum, NaN counts as the maximum; val (optional) = that element (= np.max of the row).  idx or val may be NULL.
 * coma_contact_select_u8: selected[h] = (max_o nom[h,o]/den[h,o]) > threshold, NaN-propagating max (false). */
int coma_row_argmax_i64(const float* x, int64_t rows, int n, int64_t row_stride, int64_t col_offset, int64_t* idx,
                        float* val, void* stream);
int coma_contact_select_u8(const float* nom, const float* den, int H, int O, float threshold, uint8_t* selected,
                           void* stream);

/* K5  occupancy splat.
 * replaces: utils/coma_occupancy.py:287-295 (dense [H,R,R,R] f64 distance test) by an equivalent
 *           sparse test over the voxels whose centres can lie inside the threshold sphere.
 * q       : f32 [S,H,3] = f32(human_verts - obj_vert0), subtraction in f64 by the caller (:288-289)
 * centers : f64 [3,R] device; centers[c][i] = spatial_grid[c] at index i along axis c, i.e. the
 *           per-axis voxel centres exactly as load_voxelgrid builds them (:171 -- note the product
 *           voxel_size * f32(index) is rounded to f32 there, so the table is NOT start+voxel*(i+.5))
 * voxel   : centre spacing (2.4/R), used only to bound the candidate box
 * thres   : voxel*scale_tolerance (:242), the f64 threshold of the test d < thres
 * counts  : f32 [H,R,R,R] in/out; integer-valued, bit-exact (f64 distance, (x+y)+z order). */
int coma_occupancy_splat(const float* q, int S, int H, int R, const double* centers, double voxel,
                         double thres, float* counts, void* stream);

/* K6  occupancy reducer.
 * replaces: utils/coma_occupancy.py:297-312 (normalize_prob_grid_for_spatials + max over humans).
 * counts [H,R3] in/out -> counts/rowsum (no eps: an empty row becomes NaN, as in the reference);
 * select u8 [H] or NULL (= all) picks the rows the max runs over; out f32 [R3];
 * rowsum f32 [H] is scratch/output.  NaN propagates like torch.max. */
int coma_occupancy_reduce(float* counts, const uint8_t* select, int H, int64_t R3, float* rowsum,
                          float* out, void* stream);

/* K5 + K6 fused (SURVEY.md 8d structure B): zero grid -> splat all S samples -> normalise -> max over humans, the per-vertex
 * grid written once.  replaces: utils/coma_occupancy.py:272-312 for the usual "register every sample, aggregate, reduce" order;
 * afterwards counts holds the normalised grid exactly as return_aggregated_spatial_grids leaves it (write_raw = 0) or the raw
 * counts the reference exports before reducing (write_raw = 1; the max is over the normalised values either way), rowsum the
 * hit totals.
 * thres_sq_cut : the smallest double x with sqrt(x) >= thres (host: step ulps from thres*thres), so that the kernel's
 *                (dx^2+dy^2)+dz^2 < thres_sq_cut is bit-for-bit the reference's sqrt(...) < thres;
 * window       : candidate cells per axis, >= the number of voxel centres an open interval of length 2*thres (+ the 0.01-voxel
 *                margins) can contain: ceil(2*scale_tolerance) + 2;
 * workspace    : coma_occupancy_fused_workspace_bytes(S, H, R, window) bytes of device scratch (16 bytes per (vertex, sample) + 4 bytes per
 *                (vertex, sample, window plane) + the per-group maxima).  R*R % 4 == 0, R*R <= 20480. */
size_t coma_occupancy_fused_workspace_bytes(int S, int H, int R, int window);
int coma_occupancy_fused(const float* q, int S, int H, int R, const double* centers, double voxel, double thres,
                         double thres_sq_cut, int window, const uint8_t* select, int write_raw, float* counts, float* rowsum,
                         float* out, void* workspace, size_t workspace_bytes, void* stream);

/* K7  nearest-vertex index map (first minimum wins ties).
 * replaces: utils/coma.py:87-91 (argmin over f64 squared distances).
 * points f64 [P,3], verts f64 [V,3] -> idx i64 [P]; bit-exact. */
int coma_nearest_vertex_i64(const double* points, const double* verts, int P, int V, int64_t* idx,
                            void* stream);

/* Two-view DLT triangulation + candidate scoring.
 * replaces: src/generation/optimize_depth.py:202-237 (solve_DLT) and :291-295 (reprojection MSE in both views).
 * views f64 [n_views][28]: per camera {rot[9], trans[3]} of get_projection_matrix (:164-183), {mr[9] = R C, tmr[3] = t R C} of
 * get_view2joints_render (:185-200), scale, max(resolution), resolution/2 (x, y) -- built on the host with the reference's own
 * expressions.  ref_xy f64 [J,2] pixel joints of the reference view; cand_view i32 [P], cand_xy f64 [P,J,2] ->
 * tri f64 [P,J,3], ref_mse / other_mse f64 [P].  <= 1e-9 relative against np.linalg.pinv on well-conditioned pairs. */
int coma_dlt_score_f64(const double* views, int n_views, int ref_view, const double* ref_xy, const int* cand_view,
                       const double* cand_xy, int P, int J, double* tri, double* ref_mse, double* other_mse, void* stream);

/* RANSAC reprojection matrix over the selected candidates sel i32 [C] (indices into the P candidates above):
 * mse[a][b] = mean_j |xy_b[j] - render_{view(b)}(tri_a[j])|^2, counts[a] = #{b : mse[a][b] < threshold}.
 * replaces: src/generation/optimize_depth.py:329-350 (the candidates^2 loop). */
int coma_ransac_mse_f64(const double* views, const double* tri, const int* cand_view, const double* cand_xy, const int* sel,
                        int C, int J, double threshold, double* mse, int* counts, void* stream);

/* Sample ingestion: area-weighted vertex normals of S posed meshes with one shared topology.
 * replaces: open3d TriangleMesh.compute_vertex_normals() + normalize_vectors_np in
 *           prepare_affordance_extraction_inputs (utils/coma.py:672-686).
 * verts f64 [S,V,3]; faces i32 [F,3]; vf_offsets i32 [V+1] / vf_faces i32 [3F]: vertex -> incident faces (CSR, ascending
 * face index: the order in which open3d accumulates); eps >= 0 applies the reference's second v/(|v|+eps); normals f64 [S,V,3]. */
int coma_vertex_normals_f64(const double* verts, const int32_t* faces, const int32_t* vf_offsets, const int32_t* vf_faces,
                            int S, int V, int F, double eps, double* normals, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COMA_HIP_H */
