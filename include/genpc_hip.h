/*
 * genpc_hip.h -- C ABI of libgenpc_hip.so, the MI355X (gfx950) implementation of
 * GenPC's geometric hot path.
 *
 * Every entry point takes raw DEVICE pointers, plain sizes and a HIP stream
 * (hipStream_t passed as void*; NULL = the legacy default stream, which is what
 * the reference launches on).  No torch / ATen types cross this boundary.
 *
 * Contract (same as the reference's pybind modules):
 *   - the CALLER allocates every buffer, outputs and scratch included;
 *   - fp32 clouds are row-major contiguous [B,N,3]; indices are int32;
 *   - launches are asynchronous on `stream`; nothing is retained after return;
 *   - return 1 = ok, 0 = HIP error (message on stderr, genpc_last_error()),
 *     -1 = invalid shape (EMD: n != m, B > 512, n % 256 != 0).
 *
 * Arithmetic mode: genpc_set_arith sets the process-wide default, genpc_set_arith_thread
 * an override for the calling host thread (< 0 removes it); genpc_get_arith returns what a
 * call made by this thread would use.  An entry point reads the mode once, at entry.
 *   GENPC_ARITH_FMA (default)  d = fma(dz,dz, fma(dx,dx, dy*dy))  -- the
 *       contraction nvcc's default -fmad=true applies to the reference's
 *       `x2*x2+y2*y2+z2*z2`;
 *   GENPC_ARITH_STRICT         d = (dx*dx + dy*dy) + dz*dz, no contraction.
 *   Both are bit-reproduced by oracle/genpc_oracle.c (fma_mode 1 / 0).
 */
#ifndef GENPC_HIP_H
#define GENPC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define GENPC_ARITH_STRICT 0
#define GENPC_ARITH_FMA 1

/* Library / device ------------------------------------------------------- */
int genpc_abi_version(void);              /* bumps when a signature or a documented behaviour changes (16: genpc_hpr_* asynchronous,
                                           * counts -1 on an internal error; genpc_fps*: out_idx[0] -2 = failed the check;
                                           * genpc_fps_tune takes bits) */
const char *genpc_last_error(void);       /* last HIP error string, "" if none */
int genpc_set_arith(int mode);            /* process default; returns the previous one */
int genpc_set_arith_thread(int mode);     /* calling thread only, < 0: follow the default; returns the previous override */
int genpc_get_arith(void);
/* The calling host thread's modes -- genpc_set_arith_thread, genpc_nn_tune, genpc_emd_tune, genpc_pose_tune, genpc_fps_tune --
 * as eight ints, and their restoration in another thread: a caller that hands work to threads of its own (the lanes of
 * genpc_amd/pipeline.py) exports before it starts them and imports at the top of each.  Return the number of ints used. */
int genpc_thread_state_export(int out[8]);
int genpc_thread_state_import(const int in[8]);
/* The table of tuning / A-B switches: every switch is an environment variable GENPC_<NAME>, read once per process
 * through one function (csrc/common.hip: tune_env) that records it with its default and a line of documentation; none
 * changes a result.  Writes "NAME=value (default d) -- what" lines for the switches consulted so far into buf (at most
 * len bytes, NUL-terminated); returns the size the whole table needs (call with len 0 to size the buffer). */
int genpc_tune_table(char *buf, int len);
/* Frees the per-device scratch pool (split-target partials, EMD lists). */
int genpc_release_workspace(void);
/* Nearest-neighbour kernel selection, for tests and experiments: every path returns
 * the same bits.  path: 4 cell-sorted pruned exact search (two launches, O(N + M) work when most
 * queries have a target nearby; slower than the filters when many do not -- opt-in), 3 one-f16-MFMA
 * filter (default), 1 fp32-MFMA filter (default below ~6 M pairs), 0 VALU brute force, < 0 keep
 * (2 was the split-bf16 filter, removed in round 3: rejected).  hooks: bit mask of test hooks
 * (8: every query takes the exhaustive pass, 16: every listed tile is evaluated
 * exactly, 512: count what the filtered paths do, see genpc_nn_stats, 2048 / 4096: duplicate pre-pass of the f16
 * filter off / on whatever the policy says), < 0 keep.  Applies to calls made by the CALLING
 * host thread only (thread-local; other threads keep the defaults).  Returns the previous
 * path.  Environment (read once, at first use): GENPC_NN_PATH (valu | mfma32 | f16 |
 * grid), GENPC_NN_DEBUG; A/B switches of the f16 filter: GENPC_NN_HT=1024, GENPC_NN_NOWIDE.                                                          */
int genpc_nn_tune(int path, int hooks);
/* Counters of the filtered nearest-neighbour paths, accumulated on the current device
 * while hook 512 is set: out[0] queries answered, out[1] queries re-done by the exhaustive
 * pass (filter proof failed: exact ties, non-finite input), out[2] exact re-evaluations of
 * 16-/32-target pieces.  Synchronises `stream`; reset != 0 zeroes them afterwards.      */
int genpc_nn_stats(unsigned long long out[3], int reset, void *stream);
/* Exact-duplicate pre-pass of the filtered nearest-neighbour path (csrc/nn_dedupe.hip): mask[e][k / 32] bit k % 32 is set
 * iff point k of cloud e has bit-identical coordinates to a point of lower index in the same cloud -- such a target can
 * never be reported by the reference's first-index-wins scan (chamfer3D.cu:30-71).  xyz [b, n, 3] float, mask
 * [b, ceil(n / 32)] uint32 (device, every word written).  The nearest-neighbour entry points run it by themselves where
 * it pays (hooks 2048 / 4096 of genpc_nn_tune force it off / on); exposed for tests.  Returns 1 / 0 / -1.            */
int genpc_nn_duplicate_mask(int b, int n, const float *xyz, unsigned *mask, void *stream);
/* Kernel-level timing for bench.py: while enabled, HIP events bracket the filter kernel
 * (nn_f16_kernel) of every nearest-neighbour call THE CALLING HOST THREAD makes, on the call's stream
 * (the switch and the events are per thread: other threads' launches are untouched).  Returns the
 * duration in ms of the last bracketed launch (-1 if none), then sets the switch to `enable`. */
float genpc_nn_profile(int enable);

/* Chamfer3D -------------------------------------------------------------- *
 * Replaces chamfer_cuda_forward (loss_functions/Chamfer3D/chamfer3D.cu:136-154,
 * bound as chamfer_3D.forward in chamfer_cuda.cpp:17-19,31): for every point
 * of xyz1[B,N,3] the squared distance to, and index of, its nearest point in
 * xyz2[B,M,3] (dist1/idx1), and the converse (dist2/idx2).  Lowest index wins
 * ties.  The argument order is the one the reference keeps in its commented-out
 * raw-pointer prototype (chamfer3D.cu:135).                                  */
int genpc_chamfer_forward(int b, int n, const float *xyz1, int m,
                          const float *xyz2, float *dist1, int *idx1,
                          float *dist2, int *idx2, void *stream);

/* One direction only: result[B,N], result_i[B,N] for queries xyz[B,N,3] against
 * targets xyz2[B,M,3].  This is NmDistanceKernel itself (chamfer3D.cu:12-134);
 * the partial-matching losses (utils/loss_util.py:35-43) only consume this half. */
int genpc_nm_distance(int b, int n, const float *xyz, int m, const float *xyz2,
                      float *result, int *result_i, void *stream);

/* Nearest neighbour with a search limit -- what the reference asks of open3d's KD-tree in
 * remove_close_points (reg_xyz.py:41-52: search_knn_vector_3d(point, 1), keep the point unless the
 * returned SQUARED distance is below the threshold): queries whose nearest target lies within
 * radius2 (squared distance, <=) get exactly what genpc_nm_distance returns; the others get
 * result = +inf, result_i = -1.  Runs on the cell-sorted search, which then never looks farther
 * than the limit: O(N + M) whatever the overlap of the two clouds.  Returns -1 for radius2 < 0. */
int genpc_nm_distance_within(int b, int n, const float *xyz, int m, const float *xyz2,
                             float radius2, float *result, int *result_i, void *stream);

/* Replaces chamfer_cuda_backward (chamfer3D.cu:176-195, chamfer_3D.backward in
 * chamfer_cuda.cpp:22-26,32).  gradxyz1[B,N,3] / gradxyz2[B,M,3] must be zeroed
 * by the caller (dist_chamfer_3D.py:56-57); the kernel accumulates into them. */
int genpc_chamfer_backward(int b, int n, const float *xyz1, int m,
                           const float *xyz2, const float *graddist1,
                           const int *idx1, const float *graddist2,
                           const int *idx2, float *gradxyz1, float *gradxyz2,
                           void *stream);

/* EMD (auction) ---------------------------------------------------------- *
 * Replaces emd_cuda_forward (loss_functions/emd/emd_cuda.cu:228-282, bound as
 * emd.forward in emd.cpp:12-17,27).  Same 14 caller-allocated buffers with the
 * same initial state as emd_module.py:43-54 (assignment/assignment_inv = -1,
 * the rest 0; unass_cnt/unass_cnt_sum/cnt_tmp are int32[512]).  On return:
 * dist[B,n] squared distance to the assigned point, assignment[B,n],
 * assignment_inv, price, bid, bid_increments, max_increments hold the
 * final auction state; max_idx is election scratch (the reference never resets it, emd_cuda.cu:181-194, and
 * no caller reads it: here every object that was bid for ends at -1); unass_idx/unass_cnt/unass_cnt_sum/cnt_tmp hold the
 * last round's compaction (order within unass_idx is unspecified, as in the
 * reference).                                                               */
int genpc_emd_forward(int b, int n, int m, const float *xyz1, const float *xyz2,
                      float *dist, int *assignment, float *price,
                      int *assignment_inv, int *bid, float *bid_increments,
                      float *max_increments, int *unass_idx, int *unass_cnt,
                      int *unass_cnt_sum, int *cnt_tmp, int *max_idx, float eps,
                      int iters, void *stream);

/* Implementation of genpc_emd_forward, for tests and A/B (applies to the calling host thread): 2 all rounds in ONE
 * launch whose threads own the points (csrc/emd_auction.hip; needs eps >= 0 and the whole launch resident: B n <= 3.5 x 256
 * x CUs, else the call falls back to 1), 1 a launch per round step with the cell-sorted culled bid (csrc/emd_grid.hip: a
 * bidder visits only the grid rows that can hold an object worth more than its current second-best), 0 the same with the
 * tiled bid over all objects, < 0 the default (2 when admitted; else 1 when eps >= 0 and n >= 4096 or B n >= 65536; else 0).
 * All give the same bits.  hooks (>= 0 to set; < 0 keep): 1 = count what the culled bid does (genpc_emd_stats; implies
 * the launch-per-round path).  Returns the previous setting. */
int genpc_emd_tune(int grid, int hooks);
/* Synchronises `stream` and returns 1 if a one-launch EMD call on it was abandoned since the last reset (its workgroups
 * did not all become resident within the spin bound; that call's dist is NaN), 0 if none, -1 on error. */
int genpc_emd_status(int reset, void *stream);
/* 1 = the calling host thread's last one-launch EMD call ran beside other streams' persistent launches -- the only situation in
 * which it can be abandoned (then genpc_emd_status() == 1 and dist is NaN); reading clears the flag; no device access.
 * genpc_amd/emd.py checks it behind every forward and repeats an abandoned call on the launch-per-round path. */
int genpc_emd_contended(void);
/* Counters of the culled bid, accumulated on the current device while hook 1 is set: out[0] bidder-rounds, out[1] rows
 * of their search boxes, out[2] rows kept by the bound, out[3] objects tested, out[4] exact (fp64) evaluations, out[5]
 * exact first-place ties (full re-scan), out[6] bidders without seeds (probe).  Synchronises `stream`; reset != 0 zeroes. */
int genpc_emd_stats(unsigned long long out[8], int reset, void *stream);

/* The CalcDist step alone (emd_cuda.cu:217-226): dist[B,n] = |xyz1[j] - xyz2[assignment[j]]|^2 (0 where assignment < 0). */
int genpc_emd_calc_dist(int b, int n, const float *xyz1, const float *xyz2, const int *assignment, float *dist, void *stream);

/* Replaces emd_cuda_backward (emd_cuda.cu:302-316, emd.backward in
 * emd.cpp:19-23,28): gradxyz[B,n,3] (caller-zeroed) += 2*graddist*(xyz1-xyz2[idx]). */
int genpc_emd_backward(int b, int n, const float *xyz1, const float *xyz2,
                       float *gradxyz, const float *graddist, const int *idx,
                       void *stream);

/* Depth prompting: projection, splat, colour gather ------------------------ *
 * Replaces DepthPrompting.getUvs (DepthPrompting.py:239-271) for the cameras the
 * caller selects: view[C,12] are 3x4 row-major world->camera matrices (look-at,
 * camera looks down -Z: kaolin Camera.from_args, utils/camera_utils.py:143-147),
 * focal = 1/tan(fovy/2), near/far as kaolin's pinhole defaults (1e-2, 1e2).
 * Outputs uv[C,N,2], depth[C,N] (NDC z), optional transformed[C,N,3] (NULL to
 * skip) and optional bbox[C,4] = min_x, min_y, max_x, max_y of the NDC xy.
 * rescale != 0: uv = (xy - bbox centre) / max extent * padmul + 0.5 with
 * padmul = float(1 - 2*padding); else uv = (xy + 1) / 2.                       */
int genpc_get_uvs(int c, int n, const float *view, float focal, float znear,
                  float zfar, const float *xyz, float *transformed, float *uv,
                  float *depth, int rescale, float padmul, float *bbox,
                  void *stream);

/* (uv * res).long(), swapped to (row, col), clipped to [0, clip_max]
 * (DepthPrompting.py:179-184, ScaleAdapter.py:59-62).  pix[N,2] int32.          */
int genpc_uv_to_pixels(int n, const float *uv, float res, int clip_max, int *pix,
                       void *stream);

/* Replaces DepthPrompting.paintPixels (DepthPrompting.py:292-339): paints
 * colors[N,ch] into img[ch,res,res] IN PLACE with a (2*point_size-1)^2 square
 * stamp, no z-test; on collisions the highest point index wins (the reference's
 * CPU index_put order; undefined on its GPU path).  out[ch,res,res] receives the
 * vertically flipped image the reference returns.  owner[res*res] is int32
 * scratch.  Returns -1 on a non-positive res/ch/point_size.                     */
int genpc_paint_pixels(int res, int n, const int *pix, const float *colors, int ch,
                       int point_size, float *img, float *out, int *owner,
                       void *stream);

/* Replaces the Python loop of ScaleAdapter.colorPoint (ScaleAdapter.py:57-66):
 * out[N,ch] = img[:, h-1-row, col] (the image is read flipped top-bottom).      */
int genpc_gather_colors(int n, const int *pix, const float *img, int ch, int h,
                        int w, float *out, void *stream);

/* Replaces DepthPrompting.getVisiblePoints (DepthPrompting.py:273-290): open3d's
 * PointCloud.hidden_point_removal(camera, radius) for every viewpoint -- Katz' operator:
 * spherical flipping p' = v + 2 (radius - |v|) v / |v|, v = p - eye, then the vertices of
 * the convex hull of the flipped points and the origin.  Computed exactly, without a hull:
 * a flipped point is a vertex iff the polygon of normals (tilts of its own direction) that
 * keep every other flipped point below it is not empty (genpc_amd/csrc/hpr.hip; double
 * arithmetic; at most 4096 viewpoints and 2^31 - 1 (viewpoint, point) pairs per call).  points[N,3] float, eyes[C,3] DOUBLE (both
 * device), radius > 0; visible[C,N] bytes, counts[C].  second_pass_points (HOST int, may
 * be NULL): how many points the wave-per-point passes took over (lattice-like inputs: many).
 * Asynchronous: every pass reads its item count from device memory, nothing is fetched -- unless
 * second_pass_points is given, which costs the call's only stream synchronisation.  An internal error (none known)
 * cannot be returned without one either: it turns every count into -1.  Polygons of any size: up to 128 and
 * 1024 vertices in LDS, beyond that in global memory.  Differences from qhull: normals tilted more than atan(1e4) from
 * the point's direction are not considered; of exact duplicates (-0 == +0) only the copy with the
 * lowest index takes part -- qhull reports one copy of a coincident group too, so counts agree.     */
int genpc_hpr_visibility(int c, int n, const float *points, const double *eyes,
                         double radius, unsigned char *visible, int *counts,
                         int *second_pass_points, void *stream);

/* The same pass for a caller that only needs THE BEST view (DepthPrompting.viewpoint_select,
 * DepthPrompting.py:87-98: argmax over the viewpoints of the visible count): after the first polygon
 * kernel a view's count is a lower bound and the number of its still undecided points is known, so a
 * view whose upper bound stays below the best lower bound cannot win (nor tie) and its remaining
 * points are skipped.  counts[c]: exact for the views with exact[v] = 1 (exact may be NULL), a lower
 * bound below the maximum for the others -- argmax(counts) (first maximum) is the reference's choice.
 * visible[c,n] is complete only for the exact views.  Other arguments as genpc_hpr_visibility.       */
int genpc_hpr_best_view_counts(int c, int n, const float *points, const double *eyes, double radius,
                               unsigned char *visible, int *counts, unsigned char *exact,
                               int *second_pass_points, void *stream);

/* A cheaper visibility for viewpoint ranking.  NOT the reference's operator (that is
 * genpc_hpr_visibility above): a z-buffer test -- a point is visible from camera c
 * when no point whose (2*point_size-1)^2 pixel stamp covers its pixel of a res x res
 * image is nearer by more than tol (NDC depth).  uv[C,N,2] / depth[C,N] as produced
 * by genpc_get_uvs; visible[C,N] bytes, counts[C] = visible points per camera.   */
int genpc_zbuffer_visibility(int c, int n, const float *uv, const float *depth,
                             int res, int point_size, float tol,
                             unsigned char *visible, int *counts, void *stream);

/* SE(3)+scale alignment ---------------------------------------------------- *
 * Pose model of ObjectPoseOptim.forward (optim_registration/diff_obj_pose.py:
 * 408-423): params[10] = rot_6d[6], trans[3], log_scale[1] (device memory),
 * pts = (R ((v - center) s)^T)^T + center + trans.                             */
int genpc_pose_transform(int n, const float *v, const float *center,
                         const float *params, float *pts, void *stream);

/* Chamfer half of compute_loss_function (diff_obj_pose.py:326-334) plus the
 * orthogonality term (:543-545) and its analytic gradient:
 *   loss = cd_weight * (mean sqrt(d1) + 0.5 mean sqrt(d2)) + reg_weight * |RR^T-I|_F
 * with d1/i1 = NN of the transformed complete cloud in `partial` and d2/i2 the
 * converse (one genpc_chamfer_forward call).  loss_out[3] = loss, cd, |RR^T-I|_F;
 * grad[10] in parameter order.  All pointers are device memory.                 */
int genpc_pose_cd_grad(int nc, const float *v, const float *center,
                       const float *params, int np, const float *partial,
                       const float *d1, const int *i1, const float *d2,
                       const int *i2, float cd_weight, float reg_weight,
                       float *loss_out, float *grad, void *stream);

/* The multi-start Adam loop of object_pose_optimization (diff_obj_pose.py:516-594)
 * with the Chamfer half of its loss: `starts` initial rotations R_y(90 deg * k),
 * iters+1 Adam steps each (lr, 0.2 lr, 0.1 lr for rot/trans/log-scale), best start
 * by lowest loss seen keeps its FINAL parameters.  Writes transform[16] =
 * [[sR, t],[0,1]] row-major, history[starts*(iters+1)] (optional) and
 * best_params[10] (optional); device memory, no host synchronisation.           */
int genpc_pose_optimize_cd(int nc, const float *complete, int np,
                           const float *partial, float lr, int iters, int starts,
                           float *transform, float *history, float *best_params,
                           void *stream);

/* The same loop for B scans in lock-step: complete[B,nc,3], partial[B,np,3],
 * transform[B,16], history[B,starts*(iters+1)] (optional), best_params[B,10]
 * (optional).  One batched NN launch per step serves all B scans.               */
int genpc_pose_optimize_cd_batch(int b, int nc, const float *complete, int np,
                                 const float *partial, float lr, int iters,
                                 int starts, float *transform, float *history,
                                 float *best_params, void *stream);

/* Silhouette ("mask") half of the pose loss -------------------------------------- *
 * The reference compares PulsarPointsRenderer images (pytorch3d, CUDA only -- absent here and
 * unpinned) of the partial cloud with its colours (diff_obj_pose.py:108-134; load_point_cloud
 * returns vert_col for every input the pipeline produces, :136-164) and of the posed complete cloud
 * with its colours (:426-433).  This library draws both with its OWN differentiable colour splat, same
 * camera (eye (0,0,3) looking at the origin, focal 4 NDC) and radii (world units):
 *   Zv = 3 - z;  u = S/2 (1 + 4 x / Zv);  v = S/2 (1 - 4 y / Zv);  rho = S/2 * 4 * radius / Zv
 *   a_i = min(0.999, max(0, 1 - |pixel centre - (u, v)|^2 / rho^2))
 *   O = 1 - prod_i (1 - a_i);  A_ch = sum_i a_i c_i,ch / sum_i a_i;  img_ch = O * A_ch
 * (coverage-weighted colour, order-independent; Pulsar's softmax in depth is NOT reproduced).
 * genpc_splat_image writes img[size, size, 3] (H, W, C like the reference's renders, row 0 at the
 * top) for pts[n,3] with colours col[n,3] in [0,1]; col == NULL draws white (what load_point_cloud
 * substitutes when a PLY has no colours, :157-158).                                              */
int genpc_splat_image(int n, const float *pts, const float *col, float radius, int size,
                      float *img, void *stream);

/* compute_loss_function's mask terms on given images (diff_obj_pose.py:286-311 with
 * normalize_images :204-217, compute_soft_mask :261-278, dice_loss :238-259): img, ref [size,size,3]
 * device float32 -> loss_out[1] = 30 MSE + BCE + 10 Dice of the luminance soft masks after the
 * per-channel statistical normalisation of img towards ref; grad (NULL to skip) [size,size,3] =
 * d loss / d img.  The same device code as the alignment loop's; pinned to the reference's own
 * Python through tests/golden/ref_py_mask_loss.npz.  Returns 1 / 0 / -1 (size < 2).               */
int genpc_mask_loss(int size, const float *img, const float *ref, float *loss_out, float *grad,
                    void *stream);

/* compute_loss_function as a whole (diff_obj_pose.py:286-336) + the orthogonality term, and its
 * analytic gradient:
 *   loss = mask_weight * mask_loss + cd_weight * cd + reg_weight * |RR^T - I|_F
 *   mask_loss = 30 MSE(m, m_ref) + BCE(m, m_ref) + 10 Dice(m, m_ref)  on the soft masks
 *   m = sigmoid((luminance of the normalised image - 0.1) / 0.05) of the posed cloud (colours
 *   vert_col[nc,3], splat radius 1.1 * radius, :385) and of `partial` (colours partial_col[np,3],
 *   radius, :118); per-channel statistical normalisation as at :204-217.  NULL colours = white.
 * d1/i1/d2/i2 as in genpc_pose_cd_grad.  mask_weight = 0 skips the mask term (render_size, radius
 * and the colours are then ignored).  loss_out[4] = loss, cd, |RR^T-I|_F, mask_loss; grad[10].   */
int genpc_pose_loss_grad(int nc, const float *v, const float *vert_col, const float *center,
                         const float *params, int np, const float *partial,
                         const float *partial_col, const float *d1, const int *i1,
                         const float *d2, const int *i2, float cd_weight, float reg_weight,
                         float mask_weight, float radius, int render_size, float *loss_out,
                         float *grad, void *stream);

/* The renderer of the mask term (genpc_splat_image, genpc_pose_loss_grad, genpc_pose_optimize_batch), per calling host thread:
 * 1 = Pulsar's published blending function (Lassner & Zollhoefer, CVPR 2021, eq. 1-2) with the reference's arguments
 *     (diff_obj_pose.py:126-131,428-433: gamma 1e-2, znear 1e-4, zfar 5, bg 0): over the discs covering a pixel
 *       I_ch = sum_i a_i e_i c_i,ch / (B + sum_i a_i e_i),  e_i = exp(z_i / gamma),  z_i = (zfar - Zv_i) / (zfar - znear),  B = exp(1e-10 / gamma)
 *     with the coverage a_i and the camera of the splat above -- a near surface hides a far one.  THE DEFAULT.  What of it is
 *     from memory (pytorch3d is absent and unpinned) is listed in oracle/genpc_oracle_geom.c;
 * 0 = the coverage splat described above (order-independent: front and back surfaces are averaged; rounds 2-4);
 * < 0 = back to the default (environment GENPC_RENDER_BLEND).  Returns the previous setting (-1 = default). */
int genpc_render_tune(int blend);

/* Nearest-neighbour path of genpc_pose_optimize_batch, for tests and A/B (applies to the calling host thread): 1 the
 * seeded cell search from the second Adam step on (csrc/nn_seeded.hip: every query starts from last step's answer and
 * searches only the ball it leaves; grids built once per call, the moving cloud's in its rest frame), 0 the brute-force
 * filter at every step (on real shapes three of the four starts are misaligned, most of their queries have no near
 * target and the seeded search loses), 2 measure both during the call and take the faster one (the default: a few
 * hipEventSynchronize per call on the call's stream), < 0 the default / environment GENPC_POSE_SEEDED.  Same bits in
 * every mode.  Returns the previous setting. */
int genpc_pose_tune(int seeded);
/* The calling host thread's alignment loops (full objective, small clouds): 1 = the Chamfer half of an Adam step -- nearest
 * neighbours + gradient -- on a side stream of the highest priority class beside the silhouette half, 0 = one stream,
 * < 0 = default (GENPC_POSE_DUAL, on).
 * Same results either way.  Returns the previous setting. */
int genpc_pose_dual(int on);

/* object_pose_optimization's loop with the FULL objective for B scans in lock-step: as
 * genpc_pose_optimize_cd_batch plus, per Adam step, the splat of the posed cloud, the mask loss
 * and its gradient (four more launches; the reference image of `partial` is drawn once).
 * complete_col[B,nc,3] / partial_col[B,np,3]: the clouds' colours in [0,1] (vert_col of
 * load_point_cloud, :136-164), NULL = white.  radius / render_size: diff_obj_pose.py:496;
 * mask_weight 1 is the reference's weight (:331).                                                 */
int genpc_pose_optimize_batch(int b, int nc, const float *complete, const float *complete_col,
                              int np, const float *partial, const float *partial_col, float lr,
                              int iters, int starts, float radius, int render_size,
                              float mask_weight, float *transform, float *history,
                              float *best_params, void *stream);

/* Voxel-grid down-sampling ---------------------------------------------------- *
 * Counterpart of open3d's PointCloud.voxel_down_sample, which reg() applies to both clouds
 * before every ICP / scale search (reg_xyz.py:154-155,178-183; open3d absent, unpinned -- its
 * published definition): grid anchored at min_bound - voxel_size / 2, index =
 * floor((p - anchor) / voxel_size) in double, one output point per occupied voxel = the mean
 * of its points accumulated in point order (double).  Output order: ascending (i, j, k).
 * colors[n,3] (NULL = none): per-point colours, averaged per voxel into out_colors[n,3] the same
 * way (what open3d does for the coloured clouds of load_xyz / glb2point, utils/dataUtils.py:174-189,
 * 217-250).  voxel_size is a double, as in open3d (0.03 != 0.03f moves points near a cell face).
 * out[n,3] must hold up to n points; *out_count (device int) receives the number written, or
 * -1 when a coordinate is not finite or the grid would exceed 2^21 cells along an axis.
 * Returns -1 for voxel_size <= 0.                                                        */
int genpc_voxel_down_sample(int n, const float *xyz, const float *colors, double voxel_size,
                            float *out, float *out_colors, int *out_count, void *stream);

/* ICP + scale search --------------------------------------------------------- *
 * Batched point-to-point ICP with the semantics of open3d's registration_icp as
 * reg_xyz.py calls it (:18-20,28-37: TransformationEstimationPointToPoint, default
 * criteria 30 iterations / 1e-6 / 1e-6): K candidates share source[ns,3] and
 * target[nt,3] and differ in their initial transform init[K,16] (row-major 4x4,
 * DOUBLE, device memory).  out_T[K,16] double, stats[K,3] double = fitness,
 * inlier_rmse, iterations.  Correspondence: fp32 nearest neighbour with
 * d2 <= max_dist^2.                                                            */
int genpc_icp_batch(int k, int ns, const float *source, int nt, const float *target,
                    double max_dist, const double *init, int max_iter,
                    double rel_fitness, double rel_rmse, double *out_T,
                    double *stats, void *stream);

/* Scores of iterative_scale_search (reg_xyz.py:60-96) for K anisotropic scale
 * candidates scales[K,3] in one batched NN launch:
 *   scores[k] = mean sqrt(NN(source*scales[k] -> target))
 *             + cd_inv_weight * mean sqrt(NN(target -> source*scales[k])).    */
int genpc_scale_search_scores(int k, int ns, const float *source, int nt,
                              const float *target, const float *scales,
                              float cd_inv_weight, float *scores, void *stream);

/* Hardware-premise probe ------------------------------------------------------ *
 * D[p] = A[p] B[p] + C[p] for `problems` independent 32x16 * 16x32 + 32x32 products computed by ONE
 * v_mfma_f32_32x32x16_f16 each (A, B: f16 bit patterns, row-major; C, D float32; device memory).
 * The default nearest-neighbour filter's proof leans on this instruction's internal summation error
 * being <= 6.5 * 2^-24 * sum|terms| (csrc/nn_f16.hip); tests/test_gpu_mfma_premise.py measures it
 * with this entry on the box the suite runs on.                                                  */
int genpc_mfma_f16_probe(int problems, const unsigned short *a, const unsigned short *b,
                         const float *c, float *d, void *stream);

/* out[i] = decode(encode(base[i], v[i])) of the 16-bit lower-bound code the nearest-neighbour filter hands its second and
 * third list minima to the finish step in (csrc/nn.h: list_enc / list_dec).  For v >= base the result must never exceed
 * v (tests/test_gpu_fastdiv.py::test_list_codes_are_lower_bounds).  Device memory, n elements.               */
int genpc_list_code_probe(long long n, const float *base, const float *v, float *out, void *stream);

/* fast[i] = csrc/fastdiv.h's shared-reciprocal division num[i] / den[i] (scalar form), fast_packed[i] = its packed
 * form, ieee[i] = the compiler's correctly rounded division, in_range[i] = 1 where the callers' range test
 * (both magnitudes in [2^-50, 2^50]) lets the fast form be used: there the three must agree bit for bit
 * (tests/test_gpu_fastdiv.py).  All arrays device memory, n elements.                              */
int genpc_fastdiv_probe(long long n, const float *num, const float *den, float *fast,
                        float *fast_packed, float *ieee, unsigned char *in_range, void *stream);

/* Farthest point sampling --------------------------------------------------- *
 * Deterministic counterpart of fpsample.fps_sampling as the reference uses it
 * (main.py:21-24, reg_xyz.py:215, DepthPrompting.py:88; third-party, random start):
 * start index 0, fp32 squared distances in the library's arithmetic mode, first
 * arg-max.  xyz[C,N,3] -> out_idx[C,k] int32, C clouds side by side.  Returns -1
 * unless 0 < k <= n <= 262144.  out_idx[c][0] is 0 for a good sequence; -1: the
 * hand-off among the cloud's workgroups timed out (something kept them from running
 * together); -2: the finished sequence failed the device-side check of every step
 * against the definition (GENPC_FPS_VERIFY, on by default) -- sample that cloud again
 * (genpc_amd/fps.py does).                                                       */
int genpc_fps(int c, int n, const float *xyz, int k, int *out_idx, void *stream);

/* The same for C clouds of DIFFERENT sizes in one pass (the metric's two subsamplings and the fused
 * cloud's run side by side: a step costs its latency, not its work): host arrays n[C], k[C],
 * xyz[C] (device pointers to [n_j,3]), out_idx[C] (device pointers to [k_j]).                    */
int genpc_fps_multi(int c, const int *n, const int *k, const float *const *xyz,
                    int *const *out_idx, void *stream);
/* The device-side check of finished sequences OFF the caller's critical path (calling host thread; returns the previous
 * setting).  1: samplings of clouds the one-workgroup kernel takes (<= 24576 points) return without waiting for their check --
 * the cloud, the indices and the recorded minima are copied (stream-ordered) and checked on a side stream of the library's own;
 * a failed check is counted instead of poisoning out_idx[c][0].  genpc_fps_deferred_check(stream) waits for the checks of the
 * samplings enqueued on `stream` by this device so far and returns how many failed since the last call (-1: error): when it is
 * not 0 the caller samples again with the check in line (genpc_fps_defer(0)).  genpc_amd/pipeline.py does so once per
 * completed scan.  0 (default): out_idx is final when the call's work on the stream is.  (2: test hook -- deferred, and the
 * check's copy of one recorded minimum is zeroed: the check must fail, tests/test_gpu_fps.py.)                                */
int genpc_fps_defer(int on);
/* Makes the side streams the library pairs with `stream` (alignment loop, sampling check) and gives each a first command: a
 * stream's hardware queue is decided when it is first used, and the pairing overlaps best when these come before other
 * streams of the process (csrc/pose.hip).  Optional; pipeline.run_in_lanes calls it before making its lanes.  Returns 1. */
int genpc_streams_prepare(void *stream);
int genpc_fps_deferred_check(void *stream);
/* Test hook (calling host thread; returns the previous setting): 1 = the pre-fix form of the sampling's workers -- pivots read
 * as per-lane LDS broadcasts and running minima lowered with PACKED fp32 instructions on register pairs, which is what drew
 * wrong samples next to other streams' matrix instructions (csrc/fps.hip; kept reachable so that
 * tests/test_gpu_concurrency.py can show the trigger); 0 = one register at a time (shipped).  Bits, for bisecting the
 * trigger (tools/fps_reject_probe.py): 1 per-lane LDS pivot reads, 2 packed update, 4 sixteen wait states in front of it,
 * 8 its operands copied through fresh registers, 16 its results leave their pair through 32-bit registers, 32 (with 2) the
 * six packed instructions written out on {c, c} pairs, four wait states behind each (64: one), 128 (with 2 | 32) written out
 * with half selection (op_sel) on (x, y) / (y, z) pairs -- the form that fails beside other streams' kernels; 1 alone means 3. */
int genpc_fps_tune(int legacy_pivot);
/* Diagnostics: rounds[j] (host, c <= 32) = inter-workgroup exchanges cloud j of the last
 * genpc_fps_multi call on this stream took (one exchange yields several samples).  Synchronises.  */
int genpc_fps_stats(int c, int *rounds, void *stream);

/* Statistical outlier filter ------------------------------------------------ *
 * mean_out[N] = mean Euclidean distance of every point to its k nearest points of
 * the same cloud, itself included (the per-point statistic of open3d's
 * remove_statistical_outlier, utils/dataUtils.py:648-662 -> reg_xyz.py:134,217).
 * k in {8, 16, 20, 32}; -1 otherwise.                                           */
int genpc_knn_mean_distance(int n, const float *xyz, int k, float *mean_out,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GENPC_HIP_H */
