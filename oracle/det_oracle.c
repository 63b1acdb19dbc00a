/*
 * det_oracle.c -- CPU ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Plain-C restatement of the reference's detection-op arithmetic, written so that every
 * float operation happens in the same order as in the reference (no FMA contraction:
 * build with -ffp-contract=off).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load the library built from this file.
 *
 * What each function follows (paths relative to /root/reference/minddet/models):
 *   orc_rot_overlap        centerpoint/det3d_ms/ops/iou-bev-nms-org.cpp:123-215 (box_overlap)
 *                          == centerpoint/det3d_ms/ops/iou3d_nms/src/iou3d_nms_kernel.cu:104-225
 *   orc_iou_bev            iou-bev-nms-org.cpp:217-224 / iou3d_nms_kernel.cu:227-234
 *   orc_boxes_iou_bev      iou-bev-nms-org.cpp:227-234 ; iou3d_nms_kernel.cu:251-265
 *   orc_boxes_overlap_bev  iou3d_nms_kernel.cu:236-249
 *   orc_nms_rot_aot        iou-bev-nms-org.cpp:237-283 (boxes_iou_nms_cpu: >= thr, no eps,
 *                          zero-area boxes dropped) -- N taken from the caller, not 1000
 *   orc_nms_rot_mask       iou3d_nms_kernel.cu:267-311 + iou3d_nms.cpp:90-136 (> thr, fmaxf eps)
 *   orc_nms_normal_mask    iou3d_nms_kernel.cu:314-372 + iou3d_nms.cpp:139-186
 *   orc_iou_aligned        pointpillars/src/core/box_np_ops.py:639-679 (iou_jit)
 *   orc_nms_aligned_jit    pointpillars/src/core/nms.py:85-112 (nms_jit: >= thr, eps)
 *   orc_nms_aligned_plus1  pointpillars/src/core/nms.py:7-41 (apply_nms: +1 areas, keep <= thr)
 *   orc_circle_nms         centerpoint/det3d_ms/core/utils/circle_nms_jit.py:6-36
 *
 * One deliberate, documented deviation: the reference calls the float overloads
 * cos/sin/atan2 of the host libm, whose last-bit behaviour is libm-build dependent.
 * The oracle defines them as the double-precision libm function rounded once to float
 * (orc_cosf/orc_sinf/orc_atan2f below).  oracle/_ref (the reference source itself,
 * compiled by oracle/Makefile) produced tests/golden/reference_vectors.npz, which
 * tests/test_oracle_golden.py uses to pin this restatement: identical keep lists, IoU within 2e-6.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float orc_cosf(float a) { return (float)cos((double)a); }
static float orc_sinf(float a) { return (float)sin((double)a); }
static float orc_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }

/* reference's own min/max (ternaries, not fminf/fmaxf): iou-bev-nms-org.cpp:26-32 */
static float lo2(float a, float b) { return a > b ? b : a; }
static float hi2(float a, float b) { return a > b ? a : b; }

#define ORC_EPS 1e-8f
#define ORC_MARGIN 1e-2f

/* cross of (p1-p0) x (p2-p0): iou-bev-nms-org.cpp:59-61 */
static float cross3(float p1x, float p1y, float p2x, float p2y, float p0x, float p0y) {
    return (p1x - p0x) * (p2y - p0y) - (p2x - p0x) * (p1y - p0y);
}

/* segment p0->p1 against q0->q1; iou-bev-nms-org.cpp:63-111 (check_rect_cross + intersection) */
static int seg_hit(float p1x, float p1y, float p0x, float p0y, float q1x, float q1y, float q0x,
                   float q0y, float *ox, float *oy) {
    int bb = lo2(p0x, p1x) <= hi2(q0x, q1x) && lo2(q0x, q1x) <= hi2(p0x, p1x) &&
             lo2(p0y, p1y) <= hi2(q0y, q1y) && lo2(q0y, q1y) <= hi2(p0y, p1y);
    if (!bb) return 0;
    float s1 = cross3(q0x, q0y, p1x, p1y, p0x, p0y);
    float s2 = cross3(p1x, p1y, q1x, q1y, p0x, p0y);
    float s3 = cross3(p0x, p0y, q1x, q1y, q0x, q0y);
    float s4 = cross3(q1x, q1y, p1x, p1y, q0x, q0y);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = cross3(q1x, q1y, p1x, p1y, p0x, p0y);
    if (fabsf(s5 - s1) > ORC_EPS) {
        *ox = (s5 * q0x - s1 * q1x) / (s5 - s1);
        *oy = (s5 * q0y - s1 * q1y) / (s5 - s1);
    } else {
        float a0 = p0y - p1y, b0 = p1x - p0x, c0 = p0x * p1y - p1x * p0y;
        float a1 = q0y - q1y, b1 = q1x - q0x, c1 = q0x * q1y - q1x * q0y;
        float D = a0 * b1 - a1 * b0;
        *ox = (b0 * c1 - b1 * c0) / D;
        *oy = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

/* iou-bev-nms-org.cpp:72-82 */
static int inside_box(const float *box, float px, float py) {
    float cx = box[0], cy = box[1];
    float c = orc_cosf(-box[6]), s = orc_sinf(-box[6]);
    float rx = (px - cx) * c + (py - cy) * (-s);
    float ry = (px - cx) * s + (py - cy) * c;
    return fabsf(rx) < box[3] / 2 + ORC_MARGIN && fabsf(ry) < box[4] / 2 + ORC_MARGIN;
}

static void corners_of(const float *b, float *X, float *Y) {
    float hx = b[3] / 2, hy = b[4] / 2;
    float x1 = b[0] - hx, y1 = b[1] - hy, x2 = b[0] + hx, y2 = b[1] + hy;
    float px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
    float c = orc_cosf(b[6]), s = orc_sinf(b[6]);
    for (int k = 0; k < 4; ++k) { /* rotate_around_center, iou-bev-nms-org.cpp:113-117 */
        X[k] = (px[k] - b[0]) * c + (py[k] - b[1]) * (-s) + b[0];
        Y[k] = (px[k] - b[0]) * s + (py[k] - b[1]) * c + b[1];
    }
    X[4] = X[0];
    Y[4] = Y[0];
}

float orc_rot_overlap(const float *A, const float *B) {
    float ax[5], ay[5], bx[5], by[5];
    corners_of(A, ax, ay);
    corners_of(B, bx, by);
    float qx[24], qy[24]; /* reference uses 16; 24 can never overflow (16 hits + 8 corners) */
    float sx = 0.f, sy = 0.f;
    int n = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float ox, oy;
            if (seg_hit(ax[i + 1], ay[i + 1], ax[i], ay[i], bx[j + 1], by[j + 1], bx[j], by[j], &ox,
                        &oy)) {
                qx[n] = ox; qy[n] = oy;
                sx = sx + ox; sy = sy + oy;
                ++n;
            }
        }
    for (int k = 0; k < 4; ++k) {
        if (inside_box(A, bx[k], by[k])) {
            sx = sx + bx[k]; sy = sy + by[k];
            qx[n] = bx[k]; qy[n] = by[k];
            ++n;
        }
        if (inside_box(B, ax[k], ay[k])) {
            sx = sx + ax[k]; sy = sy + ay[k];
            qx[n] = ax[k]; qy[n] = ay[k];
            ++n;
        }
    }
    if (n == 0) return 0.f; /* reference divides 0/0 here and then sums nothing: result 0 */
    sx /= (float)n;
    sy /= (float)n;
    float ang[24];
    for (int k = 0; k < n; ++k) ang[k] = orc_atan2f(qy[k] - sy, qx[k] - sx);
    /* bubble sort ascending by angle, swap iff ang[i] > ang[i+1]; iou-bev-nms-org.cpp:198-208 */
    for (int j = 0; j < n - 1; ++j)
        for (int i = 0; i < n - j - 1; ++i)
            if (ang[i] > ang[i + 1]) {
                float t;
                t = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = t;
                t = qx[i]; qx[i] = qx[i + 1]; qx[i + 1] = t;
                t = qy[i]; qy[i] = qy[i + 1]; qy[i + 1] = t;
            }
    float area = 0.f;
    for (int k = 0; k < n - 1; ++k) {
        float ux = qx[k] - qx[0], uy = qy[k] - qy[0];
        float vx = qx[k + 1] - qx[0], vy = qy[k + 1] - qy[0];
        area += ux * vy - uy * vx;
    }
    return fabsf(area) / 2.0f;
}

float orc_iou_bev(const float *A, const float *B) {
    float sa = A[3] * A[4], sb = B[3] * B[4];
    float so = orc_rot_overlap(A, B);
    return so / fmaxf(sa + sb - so, ORC_EPS);
}

void orc_boxes_iou_bev(const float *A, int64_t na, const float *B, int64_t nb, float *out) {
    for (int64_t i = 0; i < na; ++i)
        for (int64_t j = 0; j < nb; ++j) out[i * nb + j] = orc_iou_bev(A + i * 7, B + j * 7);
}

void orc_boxes_overlap_bev(const float *A, int64_t na, const float *B, int64_t nb, float *out) {
    for (int64_t i = 0; i < na; ++i)
        for (int64_t j = 0; j < nb; ++j) out[i * nb + j] = orc_rot_overlap(A + i * 7, B + j * 7);
}

/* boxes_iou_nms_cpu semantics. keep[N] i32 (leading *num valid, rest 0). */
int orc_nms_rot_aot(const float *boxes, int64_t n, float thr, int32_t *keep, int32_t *num) {
    float *area = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    uint8_t *dead = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int64_t i = 0; i < n; ++i) {
        keep[i] = 0;
        area[i] = boxes[i * 7 + 3] * boxes[i * 7 + 4];
        if (area[i] == 0) dead[i] = 1;
    }
    for (int64_t i = 0; i < n; ++i) {
        if (dead[i]) continue;
        for (int64_t j = i + 1; j < n; ++j) {
            if (dead[j]) continue;
            float so = orc_rot_overlap(boxes + i * 7, boxes + j * 7);
            float ovr = so / (area[i] + area[j] - so);
            if (ovr >= thr) dead[j] = 1;
        }
    }
    int32_t m = 0;
    for (int64_t i = 0; i < n; ++i)
        if (!dead[i]) keep[m++] = (int32_t)i;
    *num = m;
    free(area);
    free(dead);
    return 0;
}

/* iou_normal, iou3d_nms_kernel.cu:314-325 */
static float iou_normal7(const float *a, const float *b) {
    float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    float inter = w * h;
    float sa = a[3] * a[4], sb = b[3] * b[4];
    return inter / fmaxf(sa + sb - inter, ORC_EPS);
}

/* CUDA-path semantics: strict '>' and greedy over pre-sorted boxes (mask + host scan is
 * equivalent to the sequential loop below). keep is int64 as in NmsGpu. */
static int nms_mask_generic(const float *boxes, int64_t n, float thr, int64_t *keep, int32_t *num,
                            int rotated) {
    uint8_t *dead = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    int32_t m = 0;
    for (int64_t i = 0; i < n; ++i) keep[i] = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (dead[i]) continue;
        keep[m++] = i;
        for (int64_t j = i + 1; j < n; ++j) {
            if (dead[j]) continue;
            float v = rotated ? orc_iou_bev(boxes + i * 7, boxes + j * 7)
                              : iou_normal7(boxes + i * 7, boxes + j * 7);
            if (v > thr) dead[j] = 1;
        }
    }
    *num = m;
    free(dead);
    return 0;
}
int orc_nms_rot_mask(const float *boxes, int64_t n, float thr, int64_t *keep, int32_t *num) {
    return nms_mask_generic(boxes, n, thr, keep, num, 1);
}
int orc_nms_normal_mask(const float *boxes, int64_t n, float thr, int64_t *keep, int32_t *num) {
    return nms_mask_generic(boxes, n, thr, keep, num, 0);
}

/* iou_jit: boxes [N,4], query [K,4] -> [N,K] */
void orc_iou_aligned(const float *b, int64_t n, const float *q, int64_t k, float eps, float *out) {
    for (int64_t kk = 0; kk < k; ++kk) {
        float qa = (q[kk * 4 + 2] - q[kk * 4 + 0] + eps) * (q[kk * 4 + 3] - q[kk * 4 + 1] + eps);
        for (int64_t i = 0; i < n; ++i) {
            float v = 0.f;
            float iw = lo2(b[i * 4 + 2], q[kk * 4 + 2]) - hi2(b[i * 4 + 0], q[kk * 4 + 0]) + eps;
            if (iw > 0) {
                float ih = lo2(b[i * 4 + 3], q[kk * 4 + 3]) - hi2(b[i * 4 + 1], q[kk * 4 + 1]) + eps;
                if (ih > 0) {
                    float ua = (b[i * 4 + 2] - b[i * 4 + 0] + eps) * (b[i * 4 + 3] - b[i * 4 + 1] + eps) +
                               qa - iw * ih;
                    v = iw * ih / ua;
                }
            }
            out[i * k + kk] = v;
        }
    }
}

/* Greedy NMS over boxes ALREADY sorted by descending score (x1,y1,x2,y2 rows, stride 4),
 * optional group ids (pairs in different groups never interact: the per-class loop of
 * centernet/src/post_process.py:41-52 and of the two-stage heads).
 * mode 0: nms_jit      (w,h = max(min-max+eps,0); area with eps; suppress if ovr >= thr)
 * mode 1: apply_nms    (+1 pixel convention; suppress if ovr >  thr, i.e. keep <= thr)
 * mode 2: iou_normal-like on corner boxes (strict >, fmaxf(union,1e-8)) -- torchvision-style
 * keepmask[n] u8. */
int orc_nms_aligned(const float *b, const int32_t *grp, int64_t n, float thr, float eps, int mode,
                    uint8_t *keepmask) {
    for (int64_t i = 0; i < n; ++i) keepmask[i] = 1;
    float off = mode == 1 ? 1.0f : eps;
    for (int64_t i = 0; i < n; ++i) {
        if (!keepmask[i]) continue;
        const float *a = b + i * 4;
        float area_i = (a[2] - a[0] + off) * (a[3] - a[1] + off);
        for (int64_t j = i + 1; j < n; ++j) {
            if (!keepmask[j]) continue;
            if (grp && grp[i] != grp[j]) continue;
            const float *c = b + j * 4;
            float area_j = (c[2] - c[0] + off) * (c[3] - c[1] + off);
            float w = hi2(lo2(a[2], c[2]) - hi2(a[0], c[0]) + off, 0.0f);
            float h = hi2(lo2(a[3], c[3]) - hi2(a[1], c[1]) + off, 0.0f);
            float inter = w * h;
            float ovr;
            if (mode == 2)
                ovr = inter / fmaxf(area_i + area_j - inter, ORC_EPS);
            else
                ovr = inter / (area_i + area_j - inter);
            int sup = (mode == 0) ? (ovr >= thr) : (ovr > thr);
            if (sup) keepmask[j] = 0;
        }
    }
    return 0;
}

/* circle_nms: dets [n,3] = x,y,score, pre-sorted by the caller exactly as the reference
 * does (argsort descending of scores), suppress if squared centre distance <= thresh. */
int orc_circle_nms(const float *d, int64_t n, float thresh, uint8_t *keepmask) {
    for (int64_t i = 0; i < n; ++i) keepmask[i] = 1;
    for (int64_t i = 0; i < n; ++i) {
        if (!keepmask[i]) continue;
        for (int64_t j = i + 1; j < n; ++j) {
            if (!keepmask[j]) continue;
            float dx = d[i * 3] - d[j * 3], dy = d[i * 3 + 1] - d[j * 3 + 1];
            float dist = dx * dx + dy * dy;
            if (dist <= thresh) keepmask[j] = 0;
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------
 * rotate_iou_kernel_eval restated: pointpillars/eval_gpu/rotate_iou.py:167-262 (rbbox_to_corners,
 * point_in_quadrilateral, line_segment_intersection, quadrilateral_intersection, sort_vertex_in_convex_polygon,
 * area, devRotateIoUEval) and the N x K driver :264-340.  Boxes are 5 floats (cx, cy, dx, dy, angle).
 * criterion -1: IoU, 0: inter/area1, 1: inter/area2, 2: raw intersection.  The reference runs under numba.cuda
 * (absent here) and ships no fixture: parity unpinned; cross-checked in tests against orc_rot_overlap.
 * --------------------------------------------------------------------------------------------------------- */
static void rie_corners(const float *rb, float *c) {
    const float a_cos = cosf(rb[4]), a_sin = sinf(rb[4]);
    const float cx[4] = {-rb[2] / 2, -rb[2] / 2, rb[2] / 2, rb[2] / 2};
    const float cy[4] = {-rb[3] / 2, rb[3] / 2, rb[3] / 2, -rb[3] / 2};
    for (int i = 0; i < 4; ++i) {
        c[2 * i] = a_cos * cx[i] + a_sin * cy[i] + rb[0];
        c[2 * i + 1] = -a_sin * cx[i] + a_cos * cy[i] + rb[1];
    }
}
static int rie_in_quad(float px, float py, const float *c) {
    const float ab0 = c[2] - c[0], ab1 = c[3] - c[1], ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    const float ap0 = px - c[0], ap1 = py - c[1];
    const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}
static int rie_seg(const float *p1, const float *p2, int i, int j, float *t) {
    const float A0 = p1[2 * i], A1 = p1[2 * i + 1], B0 = p1[2 * ((i + 1) % 4)], B1 = p1[2 * ((i + 1) % 4) + 1];
    const float C0 = p2[2 * j], C1 = p2[2 * j + 1], D0 = p2[2 * ((j + 1) % 4)], D1 = p2[2 * ((j + 1) % 4) + 1];
    const float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    const int acd = DA1 * CA0 > CA1 * DA0;
    const int bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd != bcd) {
        const int abc = CA1 * BA0 > BA1 * CA0, abd = DA1 * BA0 > BA1 * DA0;
        if (abc != abd) {
            const float DC0 = D0 - C0, DC1 = D1 - C1;
            const float ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
            const float DH = BA1 * DC0 - BA0 * DC1, Dx = ABBA * DC0 - BA0 * CDDC, Dy = ABBA * DC1 - BA1 * CDDC;
            t[0] = Dx / DH;
            t[1] = Dy / DH;
            return 1;
        }
    }
    return 0;
}
float orc_rotate_iou_eval_pair(const float *r1, const float *r2, int criterion) {
    float c1[8], c2[8], pts[48];
    rie_corners(r1, c1);
    rie_corners(r2, c2);
    int n = 0;
    for (int i = 0; i < 4; ++i) {
        if (rie_in_quad(c1[2 * i], c1[2 * i + 1], c2)) { pts[2 * n] = c1[2 * i]; pts[2 * n + 1] = c1[2 * i + 1]; ++n; }
        if (rie_in_quad(c2[2 * i], c2[2 * i + 1], c1)) { pts[2 * n] = c2[2 * i]; pts[2 * n + 1] = c2[2 * i + 1]; ++n; }
    }
    float t[2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (rie_seg(c1, c2, i, j, t) && n < 24) { pts[2 * n] = t[0]; pts[2 * n + 1] = t[1]; ++n; }
    float ai = 0.f;
    if (n > 0) {
        float cx = 0.f, cy = 0.f, vs[24];
        for (int i = 0; i < n; ++i) { cx += pts[2 * i]; cy += pts[2 * i + 1]; }
        cx /= (float)n;
        cy /= (float)n;
        for (int i = 0; i < n; ++i) {
            float v0 = pts[2 * i] - cx, v1 = pts[2 * i + 1] - cy;
            const float d = sqrtf(v0 * v0 + v1 * v1);
            v0 = v0 / d;
            v1 = v1 / d;
            if (v1 < 0) v0 = -2 - v0;
            vs[i] = v0;
        }
        for (int i = 1; i < n; ++i) {
            if (vs[i - 1] > vs[i]) {
                const float temp = vs[i], tx = pts[2 * i], ty = pts[2 * i + 1];
                int j = i;
                while (j > 0 && vs[j - 1] > temp) {
                    vs[j] = vs[j - 1];
                    pts[2 * j] = pts[2 * j - 2];
                    pts[2 * j + 1] = pts[2 * j - 1];
                    --j;
                }
                vs[j] = temp;
                pts[2 * j] = tx;
                pts[2 * j + 1] = ty;
            }
        }
        for (int i = 0; i < n - 2; ++i) {
            const float *a = pts, *b = pts + 2 * i + 2, *c = pts + 2 * i + 4;
            ai += fabsf(((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0f);
        }
    }
    const float a1 = r1[2] * r1[3], a2 = r2[2] * r2[3];
    if (criterion == -1) return ai / (a1 + a2 - ai);
    if (criterion == 0) return ai / a1;
    if (criterion == 1) return ai / a2;
    return ai;
}
void orc_rotate_iou_eval(const float *boxes, int64_t n, const float *query, int64_t k, int criterion, float *out) {
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < k; ++j) out[i * k + j] = orc_rotate_iou_eval_pair(boxes + i * 5, query + j * 5, criterion);
}
