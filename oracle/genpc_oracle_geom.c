/*
 * genpc_oracle_geom.c -- CPU restatement of the projection / splat / colour-gather
 * / pose-optimisation half of GenPC's geometric hot path (SURVEY.md 8a rows
 * a13-a16).  TEST INFRASTRUCTURE ONLY, same rules as genpc_oracle.c.
 *
 * Parity status of these rows: "PARITY UNPINNED" for everything that lives in a
 * third-party library (below); PINNED TO THE REFERENCE'S OWN PYTHON, executed in the
 * build container, for the reference's own arithmetic around those calls
 * (tests/golden/make_reference_vectors.py -> ref_py_*.npz, checked by
 * tests/test_reference_vectors.py): compute_loss_function + normalize_images +
 * compute_soft_mask + dice_loss and their autograd gradient, paintPixels / getRawDepth,
 * getUvs' rescale, build_transform.  The reference implements the rest
 * on top of third-party libraries that are neither in /root/reference nor in
 * this image and for which the reference pins no version except kaolin 0.18.0
 * (README.md:27): kaolin cameras (DepthPrompting.py:245, utils/camera_utils.py:
 * 143-147), pytorch3d transforms and the Pulsar renderer (optim_registration/
 * diff_obj_pose.py:396,419,426-433), torch autograd + torch.optim.Adam.  What is
 * restated here is the reference's OWN arithmetic around those calls, plus the
 * published algorithms of the small library functions on the path (look-at view
 * matrix, OpenGL pinhole projection, 6D -> rotation Gram-Schmidt, Adam).  The
 * analytic gradients are pinned against torch autograd (tests/test_oracle_pose.py).
 *
 * All arithmetic is fp32 with the operation order spelled out, so that the HIP
 * kernels (genpc_amd/csrc/project.hip, pose.hip) can be compared bit for bit
 * where no reduction order is involved.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* from genpc_oracle.c */
void oracle_nm_distance(int b, int n, const float *xyz, int m, const float *xyz2,
                        float *result, int *result_i, int fma_mode);

/* ------------------------------------------------------------------------
 * Cameras.  kaolin 0.18 Camera.from_args(eye, at, up, fov, width, height)
 * (utils/camera_utils.py:143-147, DepthPrompting.py:123-131): look-at extrinsics
 * for a right-handed camera looking down -Z, pinhole intrinsics from the vertical
 * field of view with near = 1e-2, far = 1e2.  view[12] is the 3x4 row-major
 * world -> camera matrix.
 * ---------------------------------------------------------------------- */
static void normalize3(float *v)
{
    float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] /= n; v[1] /= n; v[2] /= n;
}

static void cross3(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

ORACLE_API void oracle_look_at(const float *eye, const float *at, const float *up, float *view)
{
    float back[3] = {eye[0] - at[0], eye[1] - at[1], eye[2] - at[2]};
    normalize3(back);
    float right[3];
    cross3(up, back, right);
    normalize3(right);
    float upv[3];
    cross3(back, right, upv);
    const float *rows[3] = {right, upv, back};
    for (int r = 0; r < 3; r++) {
        view[r * 4 + 0] = rows[r][0];
        view[r * 4 + 1] = rows[r][1];
        view[r * 4 + 2] = rows[r][2];
        view[r * 4 + 3] = -(rows[r][0] * eye[0] + rows[r][1] * eye[1] + rows[r][2] * eye[2]);
    }
}

/* utils/camera_utils.py:104-113 */
ORACLE_API void oracle_calculate_up_vector(const double *eye, const double *target, double *up)
{
    double g[3] = {target[0] - eye[0], target[1] - eye[1], target[2] - eye[2]};
    double wu[3] = {0, 1, 0};
    double s[3] = {g[1] * wu[2] - g[2] * wu[1], g[2] * wu[0] - g[0] * wu[2], g[0] * wu[1] - g[1] * wu[0]};
    /* np.allclose(cross, 0): |x| <= atol(1e-8) */
    if (fabs(s[0]) <= 1e-8 && fabs(s[1]) <= 1e-8 && fabs(s[2]) <= 1e-8) {
        up[0] = 0; up[1] = 0; up[2] = 1;
        return;
    }
    up[0] = s[1] * g[2] - s[2] * g[1];
    up[1] = s[2] * g[0] - s[0] * g[2];
    up[2] = s[0] * g[1] - s[1] * g[0];
    double n = sqrt(up[0] * up[0] + up[1] * up[1] + up[2] * up[2]);
    up[0] /= n; up[1] /= n; up[2] /= n;
}

/* One point through one camera: view transform, then OpenGL-style projection and
 * perspective divide -> NDC (x, y, z).  focal = 1/tan(fovy/2) (square image). */
static inline void project_one(const float *V, float focal, float zn, float zf, const float *p, float *o)
{
    float xc = fmaf(V[2], p[2], fmaf(V[1], p[1], V[0] * p[0])) + V[3];
    float yc = fmaf(V[6], p[2], fmaf(V[5], p[1], V[4] * p[0])) + V[7];
    float zc = fmaf(V[10], p[2], fmaf(V[9], p[1], V[8] * p[0])) + V[11];
    float w = -zc;
    float A = (zf + zn) / (zn - zf);
    float B = (2.0f * zf * zn) / (zn - zf);
    o[0] = (focal * xc) / w;
    o[1] = (focal * yc) / w;
    o[2] = fmaf(A, zc, B) / w;
}

/* The part of DepthPrompting.getUvs that is the reference's own torch code (DepthPrompting.py:246-268),
 * for ONE camera: t[n,3] = cam.transform(points) -> uv[n,2], depth[n].  Pinned to that code itself
 * by tests/golden/ref_py_uvs.npz.  bbox (optional) receives min_x, min_y, max_x, max_y. */
static void rescale_row(int n, const float *t, int rescale, float padmul, float *uv, float *depth, float *bbox)
{
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int j = 0; j < n; j++) {
        const float *o = t + (size_t)j * 3;
        mnx = o[0] < mnx ? o[0] : mnx;
        mny = o[1] < mny ? o[1] : mny;
        mxx = o[0] > mxx ? o[0] : mxx;
        mxy = o[1] > mxy ? o[1] : mxy;
    }
    if (bbox) { bbox[0] = mnx; bbox[1] = mny; bbox[2] = mxx; bbox[3] = mxy; }
    float cx = (mnx + mxx) / 2.0f, cy = (mny + mxy) / 2.0f;
    float sx = mxx - mnx, sy = mxy - mny;
    float sc = sx > sy ? sx : sy;
    for (int j = 0; j < n; j++) {
        const float *o = t + (size_t)j * 3;
        if (rescale) {
            uv[j * 2 + 0] = ((o[0] - cx) / sc) * padmul + 0.5f;
            uv[j * 2 + 1] = ((o[1] - cy) / sc) * padmul + 0.5f;
        } else {
            uv[j * 2 + 0] = (o[0] + 1.0f) * 0.5f;
            uv[j * 2 + 1] = (o[1] + 1.0f) * 0.5f;
        }
        depth[j] = o[2];
    }
}

/* getUvs on given transformed points [C,N,3] (what kaolin's Camera.transform returned) */
ORACLE_API void oracle_rescale_uvs(int c, int n, const float *transformed, int rescale, float padmul, float *uv,
                                   float *depth)
{
    for (int i = 0; i < c; i++)
        rescale_row(n, transformed + (size_t)i * n * 3, rescale, padmul, uv + (size_t)i * n * 2, depth + (size_t)i * n,
                    NULL);
}

/* DepthPrompting.getUvs, DepthPrompting.py:239-271.  transformed may be NULL.
 * padmul = float(1 - 2*padding).  bbox (optional) receives [C,4] = min_x, min_y,
 * max_x, max_y of the NDC xy. */
ORACLE_API void oracle_get_uvs(int c, int n, const float *view, float focal, float zn, float zf,
                               const float *xyz, float *transformed, float *uv, float *depth,
                               int rescale, float padmul, float *bbox)
{
    float *tmp = (float *)malloc(sizeof(float) * (size_t)n * 3);
    for (int i = 0; i < c; i++) {
        const float *V = view + (size_t)i * 12;
        for (int j = 0; j < n; j++) project_one(V, focal, zn, zf, xyz + (size_t)j * 3, tmp + (size_t)j * 3);
        if (transformed) memcpy(transformed + (size_t)i * n * 3, tmp, sizeof(float) * (size_t)n * 3);
        rescale_row(n, tmp, rescale, padmul, uv + (size_t)i * n * 2, depth + (size_t)i * n, bbox ? bbox + i * 4 : NULL);
    }
    free(tmp);
}

/* uv -> pixel (row, col), DepthPrompting.py:179-184 / ScaleAdapter.py:59-62:
 * (uv * res).long() truncates toward zero, columns swapped to (row = v, col = u),
 * clipped to [0, res-1]. */
ORACLE_API void oracle_uv_to_pixels(int n, const float *uv, float res, int clip_max, int *pix)
{
    for (int j = 0; j < n; j++) {
        long pu = (long)(uv[j * 2 + 0] * res);
        long pv = (long)(uv[j * 2 + 1] * res);
        long r = pv < 0 ? 0 : (pv > clip_max ? clip_max : pv);
        long cc = pu < 0 ? 0 : (pu > clip_max ? clip_max : pu);
        pix[j * 2 + 0] = (int)r;
        pix[j * 2 + 1] = (int)cc;
    }
}

/* DepthPrompting.paintPixels, DepthPrompting.py:292-339.  img[C,res,res] is
 * painted in place (no z-test); on collisions the write that comes LAST in the
 * reference's index order wins -- point-major, so the highest point index (on the
 * GPU torch's index_put leaves the winner undefined; this is the CPU order).  The
 * square stamp covers offsets -(p-1)..(p-1).  out[C,res,res] = vertical flip. */
ORACLE_API void oracle_paint_pixels(int res, int n, const int *pix, const float *colors, int ch,
                                    int point_size, float *img, float *out)
{
    for (int j = 0; j < n; j++) {
        int r0 = pix[j * 2 + 0], c0 = pix[j * 2 + 1];
        for (int dx = -point_size + 1; dx < point_size; dx++)
            for (int dy = -point_size + 1; dy < point_size; dy++) {
                int r = r0 + dx, cc = c0 + dy;
                if (r < 0 || r >= res || cc < 0 || cc >= res) continue;
                for (int k = 0; k < ch; k++) img[((size_t)k * res + r) * res + cc] = colors[(size_t)j * ch + k];
            }
    }
    for (int k = 0; k < ch; k++)
        for (int r = 0; r < res; r++)
            memcpy(out + ((size_t)k * res + r) * res, img + ((size_t)k * res + (res - 1 - r)) * res,
                   sizeof(float) * (size_t)res);
}

/* ScaleAdapter.colorPoint, ScaleAdapter.py:58-66: colours[i] = flipped_img[:, row, col]
 * with the image flipped top-bottom first (:57). */
ORACLE_API void oracle_gather_colors(int n, const int *pix, const float *img, int ch, int h, int w, float *out)
{
    for (int j = 0; j < n; j++) {
        int r = pix[j * 2 + 0], cc = pix[j * 2 + 1];
        for (int k = 0; k < ch; k++) out[(size_t)j * ch + k] = img[((size_t)k * h + (h - 1 - r)) * w + cc];
    }
}

/* ------------------------------------------------------------------------
 * Pose model, optim_registration/diff_obj_pose.py:408-423.
 * params = rot_6d[6], trans[3], log_scale[1].
 * pytorch3d rotation_6d_to_matrix: a1 = d6[:3], a2 = d6[3:]; b1 = a1/|a1|;
 * b2 = a2 - (b1.a2) b1; b2 /= |b2|; b3 = b1 x b2; R = rows (b1, b2, b3).
 * F.normalize divides by max(|v|, 1e-12).
 * ---------------------------------------------------------------------- */
ORACLE_API void oracle_rot6d_to_matrix(const float *d6, float *R)
{
    float a1[3] = {d6[0], d6[1], d6[2]}, a2[3] = {d6[3], d6[4], d6[5]};
    float n1 = sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
    n1 = n1 > 1e-12f ? n1 : 1e-12f;
    float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    float dt = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    float b2[3] = {a2[0] - dt * b1[0], a2[1] - dt * b1[1], a2[2] - dt * b1[2]};
    float n2 = sqrtf(b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2]);
    n2 = n2 > 1e-12f ? n2 : 1e-12f;
    b2[0] /= n2; b2[1] /= n2; b2[2] /= n2;
    float b3[3];
    cross3(b1, b2, b3);
    for (int k = 0; k < 3; k++) { R[k] = b1[k]; R[3 + k] = b2[k]; R[6 + k] = b3[k]; }
}

/* pts = (R @ ((v - c) * s).T).T + c + t, diff_obj_pose.py:419-423 */
static inline void pose_point(const float *R, float s, const float *c, const float *t, const float *v, float *o)
{
    float lx = (v[0] - c[0]) * s, ly = (v[1] - c[1]) * s, lz = (v[2] - c[2]) * s;
    o[0] = fmaf(R[2], lz, fmaf(R[1], ly, R[0] * lx)) + c[0] + t[0];
    o[1] = fmaf(R[5], lz, fmaf(R[4], ly, R[3] * lx)) + c[1] + t[1];
    o[2] = fmaf(R[8], lz, fmaf(R[7], ly, R[6] * lx)) + c[2] + t[2];
}

ORACLE_API void oracle_pose_transform(int n, const float *v, const float *center, const float *params, float *pts)
{
    float R[9];
    oracle_rot6d_to_matrix(params, R);
    float s = expf(params[9]);
    for (int j = 0; j < n; j++) pose_point(R, s, center, params + 6, v + (size_t)j * 3, pts + (size_t)j * 3);
}

/* Loss (CD half of compute_loss_function, diff_obj_pose.py:326-334, plus the
 * orthogonality term :543-545) and its gradient with respect to the 10 parameters.
 *   cd   = mean_j sqrt(d1[j]) + 0.5 * mean_k sqrt(d2[k])
 *          d1: pts -> partial nearest neighbour, d2: partial -> pts
 *   loss = cd_weight * cd + reg_weight * ||R R^T - I||_F
 * Accumulation is in double here (the GPU reduces fp32 partial sums in an
 * unspecified order; compare with a tolerance).  Points with d == 0 contribute no
 * gradient (torch would produce inf * 0 = NaN there).
 * out: loss_out[0] = loss, [1] = cd, [2] = ortho_err; grad[10]. */
static void rot6d_backward(const float *d6, const double *gR, double *g6)
{
    /* forward in double */
    double a1[3] = {d6[0], d6[1], d6[2]}, a2[3] = {d6[3], d6[4], d6[5]};
    double n1 = sqrt(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
    double b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    double dt = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    double u[3] = {a2[0] - dt * b1[0], a2[1] - dt * b1[1], a2[2] - dt * b1[2]};
    double n2 = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    double b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    const double *g1 = gR, *g2 = gR + 3, *g3 = gR + 6;
    /* b3 = b1 x b2 */
    double gb1[3], gb2[3];
    /* d(b1 x b2)/db1 : g_b1 += b2 x g3 ; g_b2 += g3 x b1 */
    gb1[0] = g1[0] + (b2[1] * g3[2] - b2[2] * g3[1]);
    gb1[1] = g1[1] + (b2[2] * g3[0] - b2[0] * g3[2]);
    gb1[2] = g1[2] + (b2[0] * g3[1] - b2[1] * g3[0]);
    gb2[0] = g2[0] + (g3[1] * b1[2] - g3[2] * b1[1]);
    gb2[1] = g2[1] + (g3[2] * b1[0] - g3[0] * b1[2]);
    gb2[2] = g2[2] + (g3[0] * b1[1] - g3[1] * b1[0]);
    /* b2 = u/|u| */
    double dot2 = gb2[0] * b2[0] + gb2[1] * b2[1] + gb2[2] * b2[2];
    double gu[3] = {(gb2[0] - dot2 * b2[0]) / n2, (gb2[1] - dot2 * b2[1]) / n2, (gb2[2] - dot2 * b2[2]) / n2};
    /* u = a2 - (b1.a2) b1 */
    double gub1 = gu[0] * b1[0] + gu[1] * b1[1] + gu[2] * b1[2];
    double ga2[3] = {gu[0] - gub1 * b1[0], gu[1] - gub1 * b1[1], gu[2] - gub1 * b1[2]};
    for (int k = 0; k < 3; k++) gb1[k] += -dt * gu[k] - gub1 * a2[k];
    /* b1 = a1/|a1| */
    double dot1 = gb1[0] * b1[0] + gb1[1] * b1[1] + gb1[2] * b1[2];
    for (int k = 0; k < 3; k++) g6[k] = (gb1[k] - dot1 * b1[k]) / n1;
    for (int k = 0; k < 3; k++) g6[3 + k] = ga2[k];
}

ORACLE_API void oracle_pose_loss_grad(int nc, const float *v, const float *center, const float *params,
                                      int np_, const float *partial, const float *d1, const int *i1,
                                      const float *d2, const int *i2, float cd_weight, float reg_weight,
                                      float *loss_out, float *grad)
{
    float R[9];
    oracle_rot6d_to_matrix(params, R);
    float s = expf(params[9]);
    const float *c = center, *t = params + 6;
    double gt[3] = {0, 0, 0}, gs = 0, gR[9] = {0};
    double sum1 = 0, sum2 = 0;
    float p[3];
    for (int j = 0; j < nc; j++) {
        sum1 += sqrtf(d1[j]);
        if (d1[j] == 0.0f) continue;
        const float *vj = v + (size_t)j * 3;
        pose_point(R, s, c, t, vj, p);
        const float *q = partial + (size_t)i1[j] * 3;
        double w = (double)cd_weight / nc * 0.5 / sqrt((double)d1[j]) * 2.0;
        double g[3] = {w * (p[0] - q[0]), w * (p[1] - q[1]), w * (p[2] - q[2])};
        double l[3] = {vj[0] - c[0], vj[1] - c[1], vj[2] - c[2]};
        for (int a = 0; a < 3; a++) {
            gt[a] += g[a];
            for (int b = 0; b < 3; b++) gR[a * 3 + b] += g[a] * s * l[b];
            gs += g[a] * (R[a * 3 + 0] * l[0] + R[a * 3 + 1] * l[1] + R[a * 3 + 2] * l[2]);
        }
    }
    for (int k = 0; k < np_; k++) {
        sum2 += sqrtf(d2[k]);
        if (d2[k] == 0.0f) continue;
        int j = i2[k];
        const float *vj = v + (size_t)j * 3;
        pose_point(R, s, c, t, vj, p);
        const float *q = partial + (size_t)k * 3;
        double w = (double)cd_weight * 0.5 / np_ * 0.5 / sqrt((double)d2[k]) * 2.0;
        double g[3] = {-w * (q[0] - p[0]), -w * (q[1] - p[1]), -w * (q[2] - p[2])};
        double l[3] = {vj[0] - c[0], vj[1] - c[1], vj[2] - c[2]};
        for (int a = 0; a < 3; a++) {
            gt[a] += g[a];
            for (int b = 0; b < 3; b++) gR[a * 3 + b] += g[a] * s * l[b];
            gs += g[a] * (R[a * 3 + 0] * l[0] + R[a * 3 + 1] * l[1] + R[a * 3 + 2] * l[2]);
        }
    }
    double cd = sum1 / nc + 0.5 * sum2 / np_;
    /* ortho_err = ||R R^T - I||_F ; d/dR = (2 (RR^T - I) R) / err  (E symmetric) */
    double E[9], err2 = 0;
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            double e = 0;
            for (int k = 0; k < 3; k++) e += (double)R[a * 3 + k] * R[b * 3 + k];
            e -= (a == b);
            E[a * 3 + b] = e;
            err2 += e * e;
        }
    double err = sqrt(err2);
    if (err > 0)
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                double acc = 0;
                for (int k = 0; k < 3; k++) acc += E[a * 3 + k] * R[k * 3 + b];
                gR[a * 3 + b] += reg_weight * 2.0 * acc / err;
            }
    double g6[6];
    rot6d_backward(params, gR, g6);
    for (int k = 0; k < 6; k++) grad[k] = (float)g6[k];
    for (int k = 0; k < 3; k++) grad[6 + k] = (float)gt[k];
    grad[9] = (float)(gs * s);
    loss_out[0] = (float)(cd_weight * cd + reg_weight * err);
    loss_out[1] = (float)cd;
    loss_out[2] = (float)err;
}

/* torch.optim.Adam (betas 0.9/0.999, eps 1e-8, no weight decay, no amsgrad), one
 * step for the three parameter groups of diff_obj_pose.py:524-528 (lr, 0.2 lr,
 * 0.1 lr).  step is 1-based. */
ORACLE_API void oracle_adam_step(float *params, const float *grad, float *m, float *v, int step, float lr)
{
    const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
    double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    for (int k = 0; k < 10; k++) {
        double l = k < 6 ? lr : (k < 9 ? (double)lr * 0.2 : (double)lr * 0.1);
        m[k] = (float)(b1 * m[k] + (1.0 - b1) * grad[k]);
        v[k] = (float)(b2 * v[k] + (1.0 - b2) * (double)grad[k] * grad[k]);
        double denom = sqrt((double)v[k]) / sqrt(bc2) + eps;
        params[k] = (float)(params[k] - (l / bc1) * (m[k] / denom));
    }
}

/* The multi-start loop of object_pose_optimization, diff_obj_pose.py:516-588, CD
 * half only (no renderer): starts x (iters+1) Adam steps from R_y(90 deg * start),
 * trans = 0, log_scale = log(0.75); the start with the lowest loss seen keeps its
 * FINAL parameters (:570-576 -- not the parameters at that lowest loss).  Returns
 * T = [[s R, t],[0,1]] row-major in transform[16] (:464-468) and the loss history
 * [starts, iters+1]. */
ORACLE_API void oracle_pose_optimize_cd(int nc, const float *complete, int np_, const float *partial, float lr,
                                        int iters, int starts, int fma_mode, float *transform, float *history,
                                        float *best_params)
{
    float center[3] = {0, 0, 0};
    {
        double acc[3] = {0, 0, 0};
        for (int j = 0; j < nc; j++)
            for (int k = 0; k < 3; k++) acc[k] += complete[(size_t)j * 3 + k];
        for (int k = 0; k < 3; k++) center[k] = (float)(acc[k] / nc);
    }
    float *pts = (float *)malloc(sizeof(float) * (size_t)nc * 3);
    float *d1 = (float *)malloc(sizeof(float) * (size_t)nc);
    int *i1 = (int *)malloc(sizeof(int) * (size_t)nc);
    float *d2 = (float *)malloc(sizeof(float) * (size_t)np_);
    int *i2 = (int *)malloc(sizeof(int) * (size_t)np_);
    float best_loss = INFINITY;
    for (int st = 0; st < starts; st++) {
        /* get_init_rot('y', 90*start): R_y(theta), 6D = first two ROWS (matrix_to_rotation_6d) */
        double th = st * 90.0 * M_PI / 180.0;
        float params[10] = {(float)cos(th), 0.0f, (float)sin(th), 0.0f, 1.0f, 0.0f, 0, 0, 0, logf(0.75f)};
        float m[10] = {0}, vv[10] = {0};
        float local_best = INFINITY;
        int patience_counter = 0;
        for (int it = 0; it <= iters; it++) {
            oracle_pose_transform(nc, complete, center, params, pts);
            oracle_nm_distance(1, nc, pts, np_, partial, d1, i1, fma_mode);
            oracle_nm_distance(1, np_, partial, nc, pts, d2, i2, fma_mode);
            float lo[3], grad[10];
            oracle_pose_loss_grad(nc, complete, center, params, np_, partial, d1, i1, d2, i2, 3.0f, 0.001f, lo, grad);
            if (history) history[(size_t)st * (iters + 1) + it] = lo[0];
            oracle_adam_step(params, grad, m, vv, it + 1, lr);
            /* early stop, diff_obj_pose.py:529-556: patience 300, counted after the optimizer step; the iterations a start
             * does not run are NaN in the history */
            if (lo[0] < local_best) { local_best = lo[0]; patience_counter = 0; }
            else if (++patience_counter > 300) {
                if (history) for (int q = it + 1; q <= iters; q++) history[(size_t)st * (iters + 1) + q] = NAN;
                break;
            }
        }
        if (local_best < best_loss) {
            best_loss = local_best;
            memcpy(best_params, params, sizeof(params));
        }
    }
    float R[9];
    oracle_rot6d_to_matrix(best_params, R);
    float s = expf(best_params[9]);
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) transform[a * 4 + b] = R[a * 3 + b] * s;
        transform[a * 4 + 3] = best_params[6 + a];
    }
    transform[12] = transform[13] = transform[14] = 0.0f;
    transform[15] = 1.0f;
    free(pts); free(d1); free(i1); free(d2); free(i2);
}

/* ------------------------------------------------------------------------
 * Silhouette ("mask") half of compute_loss_function, diff_obj_pose.py:286-336, and the
 * renders it compares (:108-134 reference image of the partial cloud with ITS colours, :426-433
 * image of the posed complete cloud with its colours; load_point_cloud always returns colours for the
 * reference's inputs, :136-164 with utils/dataUtils.py:182-187,229-246 -- ones only for a colourless
 * PLY).
 *
 * WHAT IS RESTATED FROM THE REFERENCE (its own torch code; pinned to that code itself, executed in the
 * build container, by tests/golden/ref_py_mask_loss.npz -- see tests/golden/make_reference_vectors.py):
 *   normalize_images 'statistical' (:204-217): per CHANNEL, result' = clamp((result - mean) / (std + 1e-6)
 *     * (std_ref + 1e-6) + mean_ref, 0, 1), torch.std = unbiased;
 *   compute_soft_mask (:261-278): m = sigmoid((0.299 R + 0.587 G + 0.114 B - 0.1) / 0.05);
 *   mask_loss = 30 MSE(m, m_ref) + BCE(m, m_ref) + 10 Dice(m, m_ref)   (:304-311,238-259),
 *     F.binary_cross_entropy clamps its logs at -100; Dice smooth 1e-6;
 *   total = mask_loss * 1 + cd * 3 (+ 1e-3 |RR^T - I|_F)   (:329-333,543-546).
 * WHAT IS NOT: the renderer.  The reference renders with pytorch3d's PulsarPointsRenderer
 * (CUDA only, absent here, unpinned): sphere splatting blended by a softmax in depth (gamma 1e-2).
 * This build defines its OWN differentiable colour splat with the same camera and radii (PARITY
 * UNPINNED against Pulsar -- wherever a mask_loss value is quoted it is this splat's):
 *   camera (pytorch3d look_at_view_transform(eye=(0,0,3)) + PerspectiveCameras(focal 4, NDC),
 *     from memory):  Zv = 3 - z;  u = S/2 (1 + 4 x / Zv);  v = S/2 (1 - 4 y / Zv);
 *     rho = S/2 * 4 * radius / Zv (radius_world=True); points with Zv outside (1e-4, 5) are skipped;
 *   coverage of pixel (r, c) by point i:  a_i = min(0.999, max(0, 1 - ((c + .5 - u)^2 + (r + .5 - v)^2) / rho^2));
 *   occupancy  O = 1 - prod_i (1 - a_i);  colour  A_ch = sum_i a_i c_i,ch / sum_i a_i  (coverage-weighted mean,
 *   order-independent: NO depth ordering, unlike Pulsar);  image  I_ch = O * A_ch  (background 0, as bg_col).
 * The posed cloud is splatted with 1.1 * radius (:385), the reference cloud with radius (:118).
 * Images are [S, S, 3] (H, W, C like the reference's).
 * ---------------------------------------------------------------------- */
#define MASK_AMAX 0.999
/* ---- blend 1: Pulsar's blending function (PulsarPointsRenderer, diff_obj_pose.py:110-132,374-385,426-433) -------------
 * pytorch3d is absent and unpinned; what follows restates the PUBLISHED blending function of the renderer -- Lassner &
 * Zollhoefer, "Pulsar: Efficient Sphere-based Neural Rendering", CVPR 2021, section 3.2, eq. (1)-(2):
 *     I = sum_i w_i c_i,   w_i = o_i d_i exp(o_i z_i / gamma) / ( exp(eps / gamma) + sum_k o_k d_k exp(o_k z_k / gamma) )
 * over the spheres the pixel's ray meets; z_i the normalised depth of sphere i, 1 at the near plane and 0 at the far
 * plane; d_i the weight from the normalised orthogonal distance of the ray to the sphere centre (1 at the centre, 0 at the
 * rim); o_i the opacity; the first term of the denominator weighs the background colour.  With the reference's arguments
 * (:126-131,428-433): gamma = 1e-2, znear = 1e-4, zfar = 5.0, bg_col = 0, radius_world = True, opacity 1 (the unified
 * renderer's default), radii `radius` / 1.1 `radius`.  Restated here as
 *     z_i = (zfar - Zv_i) / (zfar - znear),  Zv_i = 3 - z  (the camera of the coverage splat above)
 *     d_i = a_i = min(0.999, max(0, 1 - r^2 / rho^2))   (the coverage of the splat above: same footprint, same camera)
 *     I_ch = sum_i a_i e_i c_i,ch / (B + sum_i a_i e_i),  e_i = exp(z_i / gamma),  B = exp(eps / gamma), eps = 1e-10
 * A front surface hides a back surface (0.2 world units in depth = a factor e^4 in weight), which the coverage splat
 * (blend 0) averages.  [FROM MEMORY, not checkable here]: the exact fall-off of d_i inside the disc (taken quadratic in
 * the normalised distance, as blend 0's coverage), the depth that enters z_i (taken at the sphere's CENTRE; Pulsar
 * intersects the ray with the sphere: a difference of at most radius / (zfar - znear) / gamma = 0.44 in the exponent
 * between centre and rim), eps, the perspective footprint (a disc of radius focal R / Zv; a sphere projects to an ellipse
 * off-axis).  The STRUCTURE -- softmax in depth with the reference's gamma and planes, background term, radii in world
 * units -- is the paper's.  Pinned to a dense torch-autograd evaluation of these formulas (tests/test_oracle_pose.py). */
#define PULSAR_GAMMA 1e-2
#define PULSAR_ZNEAR 1e-4
#define PULSAR_ZFAR 5.0
#define PULSAR_EPS 1e-10
static int g_blend = 0;          /* 0: the coverage splat, 1: Pulsar's blending function */
ORACLE_API int oracle_set_blend(int blend)
{
    const int prev = g_blend;
    g_blend = blend ? 1 : 0;
    return prev;
}
static double pulsar_z(double zv) { return (PULSAR_ZFAR - zv) / (PULSAR_ZFAR - PULSAR_ZNEAR); }

/* What of Pulsar's renderer is restated FROM MEMORY can be switched here, in the oracle only, so that the
 * registration's sensitivity to those choices can be measured (tools/renderer_sensitivity.py; the kernels implement the
 * defaults):  falloff 0: a = 1 - r^2 / rho^2 (default), 1: a = 1 - r / rho;   depth 0: the sphere's centre enters the exponent
 * (default), 1: the ray-sphere hit, z = Zv - R sqrt(1 - r^2 / rho^2) (orthographic within the disc). */
static int g_var_falloff = 0, g_var_depth = 0;
ORACLE_API void oracle_set_render_variant(int falloff_linear, int depth_hit)
{
    g_var_falloff = falloff_linear ? 1 : 0;
    g_var_depth = depth_hit ? 1 : 0;
}
/* coverage a and its derivative with respect to s = r^2 / rho^2 */
static inline double var_cover(double s, double *da_ds)
{
    if (!g_var_falloff) { *da_ds = -1.0; return 1.0 - s; }
    const double r = sqrt(s);
    *da_ds = r > 1e-12 ? -0.5 / r : 0.0;
    return 1.0 - r;
}
/* exponent z / gamma of the disc at s = r^2 / rho^2 and its derivative with respect to s (0 for the centre's depth) */
static inline double var_ze(double zv, double radius, double s, double *dze_ds)
{
    if (!g_var_depth) { *dze_ds = 0.0; return pulsar_z(zv) / PULSAR_GAMMA; }
    const double h = sqrt(s < 1.0 ? 1.0 - s : 0.0);
    /* z_hit = zv - R h;  d z_hit / d s = R / (2 h);  z / gamma = (zfar - z_hit) / ((zfar - znear) gamma) */
    *dze_ds = h > 1e-9 ? -(radius / (2.0 * h)) / ((PULSAR_ZFAR - PULSAR_ZNEAR) * PULSAR_GAMMA) : 0.0;
    return pulsar_z(zv - radius * h) / PULSAR_GAMMA;
}

/* blend 1: logt[P] = m = max(eps, max_i z_i) / gamma (the exponent every weight is taken relative to), den[P] = B' + sum a e',
 * num[P*3] = sum a e' c with e' = exp(z_i / gamma - m), B' = exp(eps / gamma - m) */
static void splat_accumulate_pulsar(int n, const float *pts, const float *col, double radius, int S, double *mexp, double *den,
                                    double *num)
{
    const double hs = 0.5 * S;
    for (int q = 0; q < S * S; q++) { mexp[q] = PULSAR_EPS / PULSAR_GAMMA; den[q] = 0.0; num[3 * q] = num[3 * q + 1] = num[3 * q + 2] = 0.0; }
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) for (int q = 0; q < S * S; q++) den[q] = exp(PULSAR_EPS / PULSAR_GAMMA - mexp[q]);
        for (int i = 0; i < n; i++) {
            const double x = pts[(size_t)i * 3 + 0], y = pts[(size_t)i * 3 + 1], z = pts[(size_t)i * 3 + 2];
            const double zv = 3.0 - z;
            if (!(zv > 1e-4) || !(zv < 5.0)) continue;
            const double u = hs * (1.0 + 4.0 * x / zv), v = hs * (1.0 - 4.0 * y / zv), rho = hs * 4.0 * radius / zv;
            const double cr = col ? col[(size_t)i * 3 + 0] : 1.0, cg = col ? col[(size_t)i * 3 + 1] : 1.0,
                         cb = col ? col[(size_t)i * 3 + 2] : 1.0;
            int c0 = (int)floor(u - rho - 0.5), c1 = (int)ceil(u + rho - 0.5);
            int r0 = (int)floor(v - rho - 0.5), r1 = (int)ceil(v + rho - 0.5);
            if (c0 < 0) c0 = 0;
            if (r0 < 0) r0 = 0;
            if (c1 > S - 1) c1 = S - 1;
            if (r1 > S - 1) r1 = S - 1;
            for (int r = r0; r <= r1; r++)
                for (int c = c0; c <= c1; c++) {
                    const double dx = c + 0.5 - u, dy = r + 0.5 - v;
                    const double sq = (dx * dx + dy * dy) / (rho * rho);
                    double dads, dzds;
                    double a = var_cover(sq, &dads);
                    if (a <= 0.0) continue;
                    if (a > MASK_AMAX) a = MASK_AMAX;
                    const double ze = var_ze(zv, radius, sq, &dzds);
                    const size_t q = (size_t)r * S + c;
                    if (pass == 0) { if (ze > mexp[q]) mexp[q] = ze; continue; }
                    const double w = a * exp(ze - mexp[q]);
                    den[q] += w;
                    num[3 * q + 0] += w * cr;
                    num[3 * q + 1] += w * cg;
                    num[3 * q + 2] += w * cb;
                }
        }
    }
}

/* logt[P] = sum log(1 - a), den[P] = sum a, num[P*3] = sum a c  (col NULL: white) */
static void splat_accumulate(int n, const float *pts, const float *col, double radius, int S, double *logt, double *den,
                             double *num)
{
    if (g_blend) { splat_accumulate_pulsar(n, pts, col, radius, S, logt, den, num); return; }
    for (int q = 0; q < S * S; q++) { logt[q] = 0.0; den[q] = 0.0; num[3 * q] = num[3 * q + 1] = num[3 * q + 2] = 0.0; }
    const double hs = 0.5 * S;
    for (int i = 0; i < n; i++) {
        const double x = pts[(size_t)i * 3 + 0], y = pts[(size_t)i * 3 + 1], z = pts[(size_t)i * 3 + 2];
        const double zv = 3.0 - z;
        if (!(zv > 1e-4) || !(zv < 5.0)) continue;
        const double u = hs * (1.0 + 4.0 * x / zv), v = hs * (1.0 - 4.0 * y / zv), rho = hs * 4.0 * radius / zv;
        const double cr = col ? col[(size_t)i * 3 + 0] : 1.0, cg = col ? col[(size_t)i * 3 + 1] : 1.0,
                     cb = col ? col[(size_t)i * 3 + 2] : 1.0;
        int c0 = (int)floor(u - rho - 0.5), c1 = (int)ceil(u + rho - 0.5);
        int r0 = (int)floor(v - rho - 0.5), r1 = (int)ceil(v + rho - 0.5);
        if (c0 < 0) c0 = 0;
        if (r0 < 0) r0 = 0;
        if (c1 > S - 1) c1 = S - 1;
        if (r1 > S - 1) r1 = S - 1;
        for (int r = r0; r <= r1; r++)
            for (int c = c0; c <= c1; c++) {
                const double dx = c + 0.5 - u, dy = r + 0.5 - v;
                double a = 1.0 - (dx * dx + dy * dy) / (rho * rho);
                if (a <= 0.0) continue;
                if (a > MASK_AMAX) a = MASK_AMAX;
                const size_t q = (size_t)r * S + c;
                logt[q] += log(1.0 - a);
                den[q] += a;
                num[3 * q + 0] += a * cr;
                num[3 * q + 1] += a * cg;
                num[3 * q + 2] += a * cb;
            }
    }
}

static void splat_compose(int P, const double *logt, const double *den, const double *num, double *I)
{
    if (g_blend) {      /* (the background term keeps the denominator positive) */
        for (int q = 0; q < P; q++)
            for (int ch = 0; ch < 3; ch++) I[3 * q + ch] = num[3 * q + ch] / den[q];
        return;
    }
    for (int q = 0; q < P; q++) {
        const double O = 1.0 - exp(logt[q]);
        for (int ch = 0; ch < 3; ch++) I[3 * q + ch] = den[q] > 0.0 ? O * num[3 * q + ch] / den[q] : 0.0;
    }
}

/* image of a coloured cloud: img[S, S, 3]; col NULL = white */
ORACLE_API void oracle_splat_image(int n, const float *pts, const float *col, float radius, int S, float *img)
{
    const int P = S * S;
    double *w = (double *)malloc(sizeof(double) * (size_t)P * 8);
    splat_accumulate(n, pts, col, radius, S, w, w + P, w + 2 * (size_t)P);
    splat_compose(P, w, w + P, w + 2 * (size_t)P, w + 5 * (size_t)P);
    for (int q = 0; q < 3 * P; q++) img[q] = (float)w[5 * (size_t)P + q];
    free(w);
}

/* torch.sigmoid on a float32 tensor: the soft masks are fp32 in the reference, and that shows:
 * for (x - 0.1) / 0.05 > ~16.6 the result is exactly 1.0f, log(1 - m) is -inf and
 * F.binary_cross_entropy's clamp at -100 decides the loss of that pixel (a pixel the posed cloud
 * covers and the reference image does not costs 100 (1 - m_ref) / P, not 18 / P), while its
 * gradient m (1 - m) is exactly 0.  The restatement keeps the sigmoid's argument, m and its logs in fp32
 * for that reason; everything around them accumulates in double. */
static double sigm(double x) { return (double)(1.0f / (1.0f + expf(-(float)x))); }
static double log_as_f32(double m) { return (double)logf((float)m); }
static const double kLum[3] = {0.299, 0.587, 0.114};
static double soft_mask(const double *rgb)
{
    const float lum = (float)(kLum[0] * rgb[0] + kLum[1] * rgb[1] + kLum[2] * rgb[2]);
    return sigm((double)((lum - 0.1f) / 0.05f));
}

/* mask_loss(result image I[P*3], reference image Iref[P*3]) and d mask_loss / d I (NULL to skip). */
static double mask_loss_images(int P, const double *I, const double *Iref, double *dLdI)
{
    double mu[3] = {0, 0, 0}, mur[3] = {0, 0, 0}, sd[3], sdr[3], k[3];
    for (int q = 0; q < P; q++)
        for (int ch = 0; ch < 3; ch++) { mu[ch] += I[3 * q + ch]; mur[ch] += Iref[3 * q + ch]; }
    for (int ch = 0; ch < 3; ch++) { mu[ch] /= P; mur[ch] /= P; }
    double var[3] = {0, 0, 0}, varr[3] = {0, 0, 0};
    for (int q = 0; q < P; q++)
        for (int ch = 0; ch < 3; ch++) {
            var[ch] += (I[3 * q + ch] - mu[ch]) * (I[3 * q + ch] - mu[ch]);
            varr[ch] += (Iref[3 * q + ch] - mur[ch]) * (Iref[3 * q + ch] - mur[ch]);
        }
    for (int ch = 0; ch < 3; ch++) {
        sd[ch] = sqrt(var[ch] / (P - 1));
        sdr[ch] = sqrt(varr[ch] / (P - 1));
        k[ch] = (sdr[ch] + 1e-6) / (sd[ch] + 1e-6);
    }
    double s_mse = 0, s_bce = 0, s_int = 0, s_m = 0, s_r = 0;
    for (int q = 0; q < P; q++) {
        double xn[3];
        for (int ch = 0; ch < 3; ch++) {
            const double x0 = (I[3 * q + ch] - mu[ch]) * k[ch] + mur[ch];
            xn[ch] = x0 < 0 ? 0 : (x0 > 1 ? 1 : x0);
        }
        const double m = soft_mask(xn), mr = soft_mask(Iref + 3 * q);
        double lm = log_as_f32(m), l1m = log_as_f32(1.0 - m);
        if (lm < -100) lm = -100;
        if (l1m < -100) l1m = -100;
        s_mse += (m - mr) * (m - mr);
        s_bce += -(mr * lm + (1.0 - mr) * l1m);
        s_int += m * mr;
        s_m += m;
        s_r += mr;
    }
    const double den = s_m + s_r + 1e-6, num = 2.0 * s_int + 1e-6;
    const double loss = 30.0 * s_mse / P + s_bce / P + 10.0 * (1.0 - num / den);
    if (!dLdI) return loss;
    /* G_ch = d loss / d xn_ch (through the sigmoid, the luminance and the clamp), then through the statistics */
    double *G = dLdI;
    double sG[3] = {0, 0, 0}, sGd[3] = {0, 0, 0};
    for (int q = 0; q < P; q++) {
        double xn[3];
        int inside[3];
        for (int ch = 0; ch < 3; ch++) {
            const double x0 = (I[3 * q + ch] - mu[ch]) * k[ch] + mur[ch];
            inside[ch] = x0 > 0 && x0 < 1;
            xn[ch] = x0 < 0 ? 0 : (x0 > 1 ? 1 : x0);
        }
        const double m = soft_mask(xn), mr = soft_mask(Iref + 3 * q);
        double dm = 30.0 * 2.0 * (m - mr) / P;
        /* BCE: -(mr/m - (1-mr)/(1-m)) / P where the logs are not clamped (torch's backward divides
         * (m - mr) by max(m (1 - m), 1e-12): the same wherever m (1 - m) > 0, and where it is 0 the
         * factor m (1 - m) of the sigmoid below makes the product 0 either way) */
        double db = 0;
        if (log_as_f32(m) > -100) db -= mr / m;
        if (log_as_f32(1.0 - m) > -100) db += (1.0 - mr) / (1.0 - m);
        dm += db / P;
        dm += 10.0 * (-(2.0 * mr * den - num) / (den * den));
        for (int ch = 0; ch < 3; ch++) {
            const double g = inside[ch] ? dm * m * (1.0 - m) / 0.05 * kLum[ch] : 0.0;
            G[3 * q + ch] = g;
            sG[ch] += g;
            sGd[ch] += g * (I[3 * q + ch] - mu[ch]);
        }
    }
    for (int q = 0; q < P; q++)
        for (int ch = 0; ch < 3; ch++) {
            double v = k[ch] * (G[3 * q + ch] - sG[ch] / P);
            if (sd[ch] > 0)
                v -= (sdr[ch] + 1e-6) / ((sd[ch] + 1e-6) * (sd[ch] + 1e-6)) * (I[3 * q + ch] - mu[ch]) / ((P - 1) * sd[ch]) * sGd[ch];
            dLdI[3 * q + ch] = v;
        }
    return loss;
}

/* img, ref: [S, S, 3]; grad (NULL to skip): d mask_loss / d img, [S, S, 3] */
ORACLE_API float oracle_mask_loss(int S, const float *img, const float *ref, float *grad)
{
    const int P = S * S;
    double *a = (double *)calloc(9 * (size_t)P, sizeof(double));
    for (int q = 0; q < 3 * P; q++) { a[q] = img[q]; a[3 * (size_t)P + q] = ref[q]; }
    const double l = mask_loss_images(P, a, a + 3 * (size_t)P, grad ? a + 6 * (size_t)P : NULL);
    if (grad)
        for (int q = 0; q < 3 * P; q++) grad[q] = (float)a[6 * (size_t)P + q];
    free(a);
    return (float)l;
}

/* Full loss and gradient: mask_weight * mask_loss + cd_weight * cd + reg_weight * |RR^T - I|_F.
 * vert_col[nc,3] (NULL = white): the complete cloud's colours; ref_img[S*S*3]: oracle_splat_image(partial,
 * partial_col, radius).  loss_out[4] = total, cd, ortho, mask. */
ORACLE_API void oracle_pose_full_loss_grad(int nc, const float *v, const float *vert_col, const float *center,
                                           const float *params, int np_, const float *partial, const float *d1,
                                           const int *i1, const float *d2, const int *i2, float cd_weight,
                                           float reg_weight, float mask_weight, float radius, int S,
                                           const float *ref_img, float *loss_out, float *grad)
{
    float lo3[3], g_cd[10];
    oracle_pose_loss_grad(nc, v, center, params, np_, partial, d1, i1, d2, i2, cd_weight, reg_weight, lo3, g_cd);
    const int P = S * S;
    float *pts = (float *)malloc(sizeof(float) * (size_t)nc * 3);
    oracle_pose_transform(nc, v, center, params, pts);
    double *logt = (double *)malloc(sizeof(double) * (size_t)P * 14);
    double *den = logt + P, *num = den + P, *I = num + 3 * (size_t)P, *Ir = I + 3 * (size_t)P, *dLdI = Ir + 3 * (size_t)P;
    const double rad = 1.1 * (double)radius;
    splat_accumulate(nc, pts, vert_col, rad, S, logt, den, num);
    splat_compose(P, logt, den, num, I);
    for (int q = 0; q < 3 * P; q++) Ir[q] = ref_img[q];
    const double ml = mask_loss_images(P, I, Ir, dLdI);
    /* back through the splat to the points, then to (R, s, t) like the CD term.
     * I_ch = O A_ch:  d I_ch / d a_i = T / (1 - a_i) A_ch + O (c_i,ch - A_ch) / D   (T = exp(logt), D = den) */
    float R[9];
    oracle_rot6d_to_matrix(params, R);
    const float s = expf(params[9]);
    double gt[3] = {0, 0, 0}, gs = 0, gR[9] = {0};
    const double hs = 0.5 * S;
    for (int i = 0; i < nc; i++) {
        const double x = pts[(size_t)i * 3 + 0], y = pts[(size_t)i * 3 + 1], z = pts[(size_t)i * 3 + 2];
        const double zv = 3.0 - z;
        if (!(zv > 1e-4) || !(zv < 5.0)) continue;
        const double u = hs * (1.0 + 4.0 * x / zv), vv = hs * (1.0 - 4.0 * y / zv), rho = hs * 4.0 * rad / zv;
        const double ci[3] = {vert_col ? vert_col[(size_t)i * 3 + 0] : 1.0, vert_col ? vert_col[(size_t)i * 3 + 1] : 1.0,
                              vert_col ? vert_col[(size_t)i * 3 + 2] : 1.0};
        int c0 = (int)floor(u - rho - 0.5), c1 = (int)ceil(u + rho - 0.5);
        int r0 = (int)floor(vv - rho - 0.5), r1 = (int)ceil(vv + rho - 0.5);
        if (c0 < 0) c0 = 0;
        if (r0 < 0) r0 = 0;
        if (c1 > S - 1) c1 = S - 1;
        if (r1 > S - 1) r1 = S - 1;
        double gu = 0, gv = 0, grho = 0, gze = 0;
        for (int r = r0; r <= r1; r++)
            for (int c = c0; c <= c1; c++) {
                const double dx = c + 0.5 - u, dy = r + 0.5 - vv;
                double a = 1.0 - (dx * dx + dy * dy) / (rho * rho);
                if (g_blend) {
                    /* I_ch = N_ch / D:  d I_ch / d w_i = (c_i,ch - I_ch) / D,  w_i = a_i e_i;  d w / d a = e_i (unclamped a
                     * only), d w / d (z / gamma) = w_i (also where a is clamped: the depth still moves the weight).
                     * With s = r^2 / rho^2:  d s / d u = -2 dx / rho^2, d s / d v = -2 dy / rho^2, d s / d rho = -2 s / rho;
                     * the default variants have d a / d s = -1 and an exponent that does not depend on s. */
                    const double sq = (dx * dx + dy * dy) / (rho * rho);
                    double dads, dzds;
                    a = var_cover(sq, &dads);
                    if (a <= 0.0) continue;
                    const size_t q = (size_t)r * S + c;
                    const double ze = var_ze(zv, rad, sq, &dzds);
                    const double e = exp(ze - logt[q]);
                    double w = 0.0;
                    for (int ch = 0; ch < 3; ch++) w += dLdI[3 * q + ch] * (ci[ch] - I[3 * q + ch]) / den[q];
                    const double ac = a > MASK_AMAX ? MASK_AMAX : a;
                    gze += w * ac * e;
                    /* d w_i / d s = e (d a / d s, unclamped only) + a e (d ze / d s) */
                    const double dwds = w * e * ((a >= MASK_AMAX ? 0.0 : dads) + ac * dzds);
                    gu += dwds * (-2.0 * dx / (rho * rho));
                    gv += dwds * (-2.0 * dy / (rho * rho));
                    grho += dwds * (-2.0 * sq / rho);
                    continue;
                }
                if (a <= 0.0 || a >= MASK_AMAX) continue;        /* clamped: no gradient */
                const size_t q = (size_t)r * S + c;
                const double T = exp(logt[q]), O = 1.0 - T;
                double w = 0.0;                                   /* dL/da_i */
                for (int ch = 0; ch < 3; ch++) {
                    const double A = num[3 * q + ch] / den[q];
                    w += dLdI[3 * q + ch] * (T / (1.0 - a) * A + O * (ci[ch] - A) / den[q]);
                }
                gu += w * 2.0 * dx / (rho * rho);
                gv += w * 2.0 * dy / (rho * rho);
                grho += w * 2.0 * (dx * dx + dy * dy) / (rho * rho * rho);
            }
        /* u = hs (1 + 4x/zv), v = hs (1 - 4y/zv), rho = hs 4 rad / zv, zv = 3 - z */
        /* (blend 1: z / gamma = (zfar - zv) / (zfar - znear) / gamma) */
        const double gzv = gu * (-hs * 4.0 * x / (zv * zv)) + gv * (hs * 4.0 * y / (zv * zv)) + grho * (-rho / zv) +
                           gze * (-1.0 / ((PULSAR_ZFAR - PULSAR_ZNEAR) * PULSAR_GAMMA));
        const double g[3] = {mask_weight * gu * hs * 4.0 / zv, mask_weight * gv * (-hs * 4.0 / zv), mask_weight * (-gzv)};
        const float *vj = v + (size_t)i * 3;
        const double l[3] = {vj[0] - center[0], vj[1] - center[1], vj[2] - center[2]};
        for (int a = 0; a < 3; a++) {
            gt[a] += g[a];
            for (int b = 0; b < 3; b++) gR[a * 3 + b] += g[a] * s * l[b];
            gs += g[a] * (R[a * 3 + 0] * l[0] + R[a * 3 + 1] * l[1] + R[a * 3 + 2] * l[2]);
        }
    }
    double g6[6];
    rot6d_backward(params, gR, g6);
    for (int k = 0; k < 6; k++) grad[k] = g_cd[k] + (float)g6[k];
    for (int k = 0; k < 3; k++) grad[6 + k] = g_cd[6 + k] + (float)gt[k];
    grad[9] = g_cd[9] + (float)(gs * s);
    loss_out[0] = lo3[0] + (float)(mask_weight * ml);
    loss_out[1] = lo3[1];
    loss_out[2] = lo3[2];
    loss_out[3] = (float)ml;
    free(pts); free(logt);
}

/* object_pose_optimization with the full objective (diff_obj_pose.py:496-594):
 * loss = mask_loss + 3 cd + 1e-3 |RR^T - I|_F, multi-start, best-of-starts as in
 * oracle_pose_optimize_cd. */
ORACLE_API void oracle_pose_optimize(int nc, const float *complete, const float *complete_col, int np_,
                                     const float *partial, const float *partial_col, float lr, int iters, int starts,
                                     int fma_mode, float radius, int S, float mask_weight, float *transform,
                                     float *history, float *best_params)
{
    float center[3] = {0, 0, 0};
    {
        double acc[3] = {0, 0, 0};
        for (int j = 0; j < nc; j++)
            for (int k = 0; k < 3; k++) acc[k] += complete[(size_t)j * 3 + k];
        for (int k = 0; k < 3; k++) center[k] = (float)(acc[k] / nc);
    }
    float *pts = (float *)malloc(sizeof(float) * (size_t)nc * 3);
    float *d1 = (float *)malloc(sizeof(float) * (size_t)nc);
    int *i1 = (int *)malloc(sizeof(int) * (size_t)nc);
    float *d2 = (float *)malloc(sizeof(float) * (size_t)np_);
    int *i2 = (int *)malloc(sizeof(int) * (size_t)np_);
    float *ref = (float *)malloc(sizeof(float) * (size_t)S * S * 3);
    oracle_splat_image(np_, partial, partial_col, radius, S, ref);
    float best_loss = INFINITY;
    for (int st = 0; st < starts; st++) {
        double th = st * 90.0 * M_PI / 180.0;
        float params[10] = {(float)cos(th), 0.0f, (float)sin(th), 0.0f, 1.0f, 0.0f, 0, 0, 0, logf(0.75f)};
        float m[10] = {0}, vv[10] = {0};
        float local_best = INFINITY;
        int patience_counter = 0;
        for (int it = 0; it <= iters; it++) {
            oracle_pose_transform(nc, complete, center, params, pts);
            oracle_nm_distance(1, nc, pts, np_, partial, d1, i1, fma_mode);
            oracle_nm_distance(1, np_, partial, nc, pts, d2, i2, fma_mode);
            float lo[4], grad[10];
            oracle_pose_full_loss_grad(nc, complete, complete_col, center, params, np_, partial, d1, i1, d2, i2, 3.0f, 0.001f,
                                       mask_weight, radius, S, ref, lo, grad);
            if (history) history[(size_t)st * (iters + 1) + it] = lo[0];
            oracle_adam_step(params, grad, m, vv, it + 1, lr);
            /* early stop, diff_obj_pose.py:529-556: patience 300, counted after the optimizer step; the iterations a start
             * does not run are NaN in the history */
            if (lo[0] < local_best) { local_best = lo[0]; patience_counter = 0; }
            else if (++patience_counter > 300) {
                if (history) for (int q = it + 1; q <= iters; q++) history[(size_t)st * (iters + 1) + q] = NAN;
                break;
            }
        }
        if (local_best < best_loss) {
            best_loss = local_best;
            memcpy(best_params, params, sizeof(params));
        }
    }
    float R[9];
    oracle_rot6d_to_matrix(best_params, R);
    float s = expf(best_params[9]);
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) transform[a * 4 + b] = R[a * 3 + b] * s;
        transform[a * 4 + 3] = best_params[6 + a];
    }
    transform[12] = transform[13] = transform[14] = 0.0f;
    transform[15] = 1.0f;
    free(pts); free(d1); free(i1); free(d2); free(i2); free(ref);
}

/* ------------------------------------------------------------------------
 * Point-to-point ICP, the algorithm behind open3d.pipelines.registration.
 * registration_icp as the reference calls it (reg_xyz.py:18-20,28-37): default
 * ICPConvergenceCriteria (relative_fitness 1e-6, relative_rmse 1e-6, 30
 * iterations), TransformationEstimationPointToPoint without scaling.  open3d is a
 * third-party dependency that is neither in /root/reference nor in this image and
 * that the reference does not pin: PARITY UNPINNED; this restates the published
 * algorithm:
 *   evaluate(T): p' = T p; 1-NN of p' in target; correspondence if d2 <= r^2;
 *                fitness = #corr / #source, rmse = sqrt(mean d2 over corr)
 *   loop: update = argmin_R,t sum |q - (R p' + t)|^2 (Kabsch / Umeyama, no
 *         scale) ; T = update T ; evaluate ; stop when both |delta fitness| and
 *         |delta rmse| < 1e-6.
 * Numeric conventions of this build (shared with the HIP path): T is kept in
 * double; p' is rounded to fp32; the NN is the fp32 NmDistance of genpc_oracle.c;
 * sums are double.  The optimal rotation is obtained with Horn's quaternion
 * method (largest eigenvector of the 4x4 N matrix, Jacobi sweeps), which equals
 * the SVD solution with the reflection fix that Eigen::umeyama applies.
 * ---------------------------------------------------------------------- */
static void jacobi4(double A[4][4], double V[4][4])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0;
        for (int i = 0; i < 4; i++)
            for (int j = i + 1; j < 4; j++) off += A[i][j] * A[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                if (fabs(A[p][q]) < 1e-300) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; k++) {
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; k++) {
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 4; k++) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

/* sums[17]: n, sum p[3], sum q[3], sum p q^T [9] (row-major, p index first), sum d2.
 * Writes the 4x4 row-major update that maps p onto q in the least-squares sense. */
ORACLE_API void oracle_kabsch_from_sums(const double *sums, double *update)
{
    double n = sums[0];
    double mp[3] = {sums[1] / n, sums[2] / n, sums[3] / n};
    double mq[3] = {sums[4] / n, sums[5] / n, sums[6] / n};
    double S[3][3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) S[a][b] = sums[7 + a * 3 + b] - n * mp[a] * mq[b];
    double N[4][4] = {
        {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
        {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
        {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
        {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
    double V[4][4];
    jacobi4(N, V);
    int best = 0;
    for (int i = 1; i < 4; i++)
        if (N[i][i] > N[best][best]) best = i;
    double w = V[0][best], x = V[1][best], y = V[2][best], z = V[3][best];
    double nn = sqrt(w * w + x * x + y * y + z * z);
    w /= nn; x /= nn; y /= nn; z /= nn;
    double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                      {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                      {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) update[a * 4 + b] = R[a][b];
        update[a * 4 + 3] = mq[a] - (R[a][0] * mp[0] + R[a][1] * mp[1] + R[a][2] * mp[2]);
    }
    update[12] = update[13] = update[14] = 0;
    update[15] = 1;
}

static void icp_evaluate(int ns, const float *source, int nt, const float *target, const double *T, float md2,
                         int fma_mode, float *pts, float *d, int *idx, double *sums)
{
    for (int j = 0; j < ns; j++)
        for (int a = 0; a < 3; a++)
            pts[(size_t)j * 3 + a] = (float)(T[a * 4 + 0] * (double)source[(size_t)j * 3 + 0] +
                                             T[a * 4 + 1] * (double)source[(size_t)j * 3 + 1] +
                                             T[a * 4 + 2] * (double)source[(size_t)j * 3 + 2] + T[a * 4 + 3]);
    oracle_nm_distance(1, ns, pts, nt, target, d, idx, fma_mode);
    for (int k = 0; k < 17; k++) sums[k] = 0;
    for (int j = 0; j < ns; j++) {
        if (!(d[j] <= md2)) continue;
        const float *p = pts + (size_t)j * 3, *q = target + (size_t)idx[j] * 3;
        sums[0] += 1;
        for (int a = 0; a < 3; a++) {
            sums[1 + a] += p[a];
            sums[4 + a] += q[a];
            for (int b = 0; b < 3; b++) sums[7 + a * 3 + b] += (double)p[a] * (double)q[b];
        }
        sums[16] += d[j];
    }
}

/* init / out_T: 4x4 row-major double.  stats[3] = fitness, inlier_rmse, iterations done. */
ORACLE_API void oracle_icp(int ns, const float *source, int nt, const float *target, double max_dist,
                           const double *init, int max_iter, double rel_fitness, double rel_rmse, int fma_mode,
                           double *out_T, double *stats)
{
    float *pts = (float *)malloc(sizeof(float) * (size_t)ns * 3);
    float *d = (float *)malloc(sizeof(float) * (size_t)ns);
    int *idx = (int *)malloc(sizeof(int) * (size_t)ns);
    double T[16], sums[17];
    memcpy(T, init, sizeof T);
    float md2 = (float)(max_dist * max_dist);
    icp_evaluate(ns, source, nt, target, T, md2, fma_mode, pts, d, idx, sums);
    double fitness = sums[0] / ns, rmse = sums[0] > 0 ? sqrt(sums[16] / sums[0]) : 0.0;
    int it = 0;
    for (; it < max_iter; it++) {
        if (sums[0] < 1) break;
        double U[16], Tn[16];
        oracle_kabsch_from_sums(sums, U);
        for (int a = 0; a < 4; a++)
            for (int b = 0; b < 4; b++) {
                double acc = 0;
                for (int k = 0; k < 4; k++) acc += U[a * 4 + k] * T[k * 4 + b];
                Tn[a * 4 + b] = acc;
            }
        memcpy(T, Tn, sizeof T);
        icp_evaluate(ns, source, nt, target, T, md2, fma_mode, pts, d, idx, sums);
        double f2 = sums[0] / ns, r2 = sums[0] > 0 ? sqrt(sums[16] / sums[0]) : 0.0;
        int conv = fabs(fitness - f2) < rel_fitness && fabs(rmse - r2) < rel_rmse;
        fitness = f2;
        rmse = r2;
        if (conv) {
            it++;
            break;
        }
    }
    memcpy(out_T, T, sizeof T);
    stats[0] = fitness;
    stats[1] = rmse;
    stats[2] = it;
    free(pts); free(d); free(idx);
}

/* ------------------------------------------------------------------------
 * Per-point statistic of open3d's remove_statistical_outlier (utils/dataUtils.py:
 * 648-662; open3d absent, unpinned -- published algorithm restated): mean Euclidean
 * distance to the k nearest points of the same cloud, the point itself included.
 * fp32 squared distances in `fma_mode`, square roots and the sum (ascending) in
 * double, result rounded to fp32.
 * ---------------------------------------------------------------------- */
static inline float sqd(float dx, float dy, float dz, int fma_mode)
{
    if (fma_mode) {
        float t = dy * dy;
        t = fmaf(dx, dx, t);
        return fmaf(dz, dz, t);
    }
    float a = dx * dx, b = dy * dy, c = dz * dz;
    float s = a + b;
    return s + c;
}

ORACLE_API void oracle_knn_mean_distance(int n, const float *xyz, int k, int fma_mode, float *mean_out)
{
#pragma omp parallel
    {
        float *best = (float *)malloc(sizeof(float) * (size_t)k);
#pragma omp for schedule(static)
        for (int j = 0; j < n; j++) {
            for (int i = 0; i < k; i++) best[i] = INFINITY;
            float qx = xyz[(size_t)j * 3], qy = xyz[(size_t)j * 3 + 1], qz = xyz[(size_t)j * 3 + 2];
            for (int u = 0; u < n; u++) {
                float d = sqd(xyz[(size_t)u * 3] - qx, xyz[(size_t)u * 3 + 1] - qy, xyz[(size_t)u * 3 + 2] - qz, fma_mode);
                if (d < best[k - 1]) {
                    int i = k - 1;
                    while (i > 0 && best[i - 1] > d) {
                        best[i] = best[i - 1];
                        i--;
                    }
                    best[i] = d;
                }
            }
            double acc = 0;
            int cnt = 0;
            for (int i = 0; i < k; i++)
                if (best[i] < INFINITY) {
                    acc += sqrt((double)best[i]);
                    cnt++;
                }
            mean_out[j] = (float)(acc / cnt);
        }
        free(best);
    }
}

/* Z-buffer visibility (the build's counterpart of DepthPrompting.getVisiblePoints;
 * NOT Katz' operator, see include/genpc_hip.h): per camera, per pixel minimum depth,
 * visible = depth <= zmin + tol. */
ORACLE_API void oracle_zbuffer_visibility(int c, int n, const float *uv, const float *depth, int res, int point_size,
                                          float tol, unsigned char *visible, int *counts)
{
    float *zb = (float *)malloc(sizeof(float) * (size_t)res * res);
    int *px = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < c; i++) {
        for (int k = 0; k < res * res; k++) zb[k] = INFINITY;
        for (int j = 0; j < n; j++) {
            size_t q = (size_t)i * n + j;
            long pu = (long)(uv[q * 2 + 0] * (float)res), pv = (long)(uv[q * 2 + 1] * (float)res);
            pu = pu < 0 ? 0 : (pu > res - 1 ? res - 1 : pu);
            pv = pv < 0 ? 0 : (pv > res - 1 ? res - 1 : pv);
            px[j] = (int)(pv * res + pu);
            for (int dy = -point_size + 1; dy < point_size; dy++)
                for (int dx = -point_size + 1; dx < point_size; dx++) {
                    long r = pv + dy, cc = pu + dx;
                    if (r < 0 || r >= res || cc < 0 || cc >= res) continue;
                    if (depth[q] < zb[r * res + cc]) zb[r * res + cc] = depth[q];
                }
        }
        int cnt = 0;
        for (int j = 0; j < n; j++) {
            size_t q = (size_t)i * n + j;
            int v = depth[q] <= zb[px[j]] + tol;
            visible[q] = (unsigned char)v;
            cnt += v;
        }
        counts[i] = cnt;
    }
    free(zb); free(px);
}


/* ------------------------------------------------------------------------
 * Voxel-grid down-sampling -- open3d's PointCloud.voxel_down_sample as reg() uses it
 * (reg_xyz.py:154-155,178-183; open3d absent and unpinned: published definition).
 *   anchor = min_bound - voxel / 2;  index = floor((p - anchor) / voxel)  (double);
 *   one output point per occupied voxel: the mean of its points, summed in point order (double).
 * Output order: ascending (i, j, k).  Returns the number of output points, -1 on a non-finite
 * coordinate / an axis with more than 2^21 voxels.
 * ---------------------------------------------------------------------- */
typedef struct { unsigned long long key; int idx; } vox_t;
static int vox_cmp(const void *a, const void *b)
{
    const vox_t *x = (const vox_t *)a, *y = (const vox_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

/* colors / out_colors: optional [n,3] attributes averaged per voxel like the points (open3d averages the
 * colours of a coloured cloud); voxel_size is a DOUBLE like open3d's (0.03 is not 0.03f). */
ORACLE_API int oracle_voxel_down_sample(int n, const float *xyz, const float *colors, double voxel_size, float *out,
                                        float *out_colors)
{
    if (n <= 0) return 0;
    const double voxel = voxel_size;
    float mn[3] = {xyz[0], xyz[1], xyz[2]};
    for (int i = 1; i < n; i++)
        for (int k = 0; k < 3; k++)
            if (xyz[(size_t)i * 3 + k] < mn[k]) mn[k] = xyz[(size_t)i * 3 + k];
    vox_t *v = (vox_t *)malloc(sizeof(vox_t) * (size_t)n);
    for (int i = 0; i < n; i++) {
        unsigned long long key = 0;
        for (int k = 0; k < 3; k++) {
            const double c = floor(((double)xyz[(size_t)i * 3 + k] - ((double)mn[k] - voxel * 0.5)) / voxel);
            if (!(c >= 0.0 && c < 2097152.0)) { free(v); return -1; }
            key = (key << 21) | (unsigned long long)c;
        }
        v[i].key = key;
        v[i].idx = i;
    }
    qsort(v, (size_t)n, sizeof(vox_t), vox_cmp);
    int m = 0;
    for (int i = 0; i < n;) {
        double s[3] = {0, 0, 0}, c[3] = {0, 0, 0};
        int j = i;
        for (; j < n && v[j].key == v[i].key; j++)
            for (int k = 0; k < 3; k++) {
                s[k] += (double)xyz[(size_t)v[j].idx * 3 + k];
                if (colors) c[k] += (double)colors[(size_t)v[j].idx * 3 + k];
            }
        for (int k = 0; k < 3; k++) out[(size_t)m * 3 + k] = (float)(s[k] / (j - i));
        if (colors && out_colors)
            for (int k = 0; k < 3; k++) out_colors[(size_t)m * 3 + k] = (float)(c[k] / (j - i));
        m++;
        i = j;
    }
    free(v);
    return m;
}
