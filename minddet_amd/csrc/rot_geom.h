// rot_geom.h -- rotated-rectangle (BEV) overlap for gfx950, one pair per lane.
//
// Computes exactly what the reference's box_overlap computes
// (minddet/models/centerpoint/det3d_ms/ops/iou3d_nms/src/iou3d_nms_kernel.cu:104-225 ==
//  minddet/models/centerpoint/det3d_ms/ops/iou-bev-nms-org.cpp:123-215) but laid out for
// CDNA4 instead of translated:
//   * everything that depends on ONE box (trig, the four rotated corners, the inverse-
//     rotation used by the corner-in-box test, area, a conservative AABB) is computed once
//     per box into a 20-float record (rot_prep_kernel), not once per pair;
//   * a pair whose inflated AABBs are disjoint returns 0 without touching the clipper -- this
//     is result-preserving (no edge can cross and no corner can pass the MARGIN test), and it
//     is what keeps a 64-lane wave from idling behind one overlapping lane;
//   * the polygon scratch (<= 24 points + angles) lives in LDS, slot-major / lane-minor, so
//     dynamic indexing never goes to scratch memory.
// Float ops follow the reference's order with FMA contraction OFF, trig is evaluated in
// double and rounded once to float (bit-identical to oracle/det_oracle.c).
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace md {

constexpr int ROT_REC = 20;   // floats per box record
constexpr int ROT_PTS = 24;   // polygon scratch slots per lane
constexpr float ROT_EPS = 1e-8f;
constexpr float ROT_MARGIN = 1e-2f;

// record layout: [0..3] X corners, [4..7] Y corners, 8 cx, 9 cy, 10 cos(-a), 11 sin(-a),
// 12 dx/2+MARGIN, 13 dy/2+MARGIN, 14 area, 15 aabb x0, 16 aabb x1, 17 aabb y0, 18 aabb y1, 19 pad
__device__ __forceinline__ void rot_make_record(const float *__restrict__ b, float *__restrict__ r) {
    const float hx = b[3] / 2, hy = b[4] / 2;
    const float x1 = b[0] - hx, y1 = b[1] - hy, x2 = b[0] + hx, y2 = b[1] + hy;
    const float px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
    const float c = (float)cos((double)b[6]), s = (float)sin((double)b[6]);
    float mnx = 0, mxx = 0, mny = 0, mxy = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float X = (px[k] - b[0]) * c + (py[k] - b[1]) * (-s) + b[0];
        const float Y = (px[k] - b[0]) * s + (py[k] - b[1]) * c + b[1];
        r[k] = X;
        r[4 + k] = Y;
        if (k == 0) { mnx = mxx = X; mny = mxy = Y; }
        else {
            mnx = fminf(mnx, X); mxx = fmaxf(mxx, X);
            mny = fminf(mny, Y); mxy = fmaxf(mxy, Y);
        }
    }
    r[8] = b[0];
    r[9] = b[1];
    r[10] = (float)cos((double)(-b[6]));
    r[11] = (float)sin((double)(-b[6]));
    r[12] = b[3] / 2 + ROT_MARGIN;
    r[13] = b[4] / 2 + ROT_MARGIN;
    r[14] = b[3] * b[4];
    // inflate by the MARGIN reach (0.01*sqrt2) plus rounding slack proportional to magnitude
    r[15] = mnx - (0.03f + 2e-5f * fabsf(mnx));
    r[16] = mxx + (0.03f + 2e-5f * fabsf(mxx));
    r[17] = mny - (0.03f + 2e-5f * fabsf(mny));
    r[18] = mxy + (0.03f + 2e-5f * fabsf(mxy));
    r[19] = 0.f;
}

__device__ __forceinline__ float rlo(float a, float b) { return a > b ? b : a; }
__device__ __forceinline__ float rhi(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float rcross3(float p1x, float p1y, float p2x, float p2y, float p0x, float p0y) {
    return (p1x - p0x) * (p2y - p0y) - (p2x - p0x) * (p1y - p0y);
}

// segment p0->p1 vs q0->q1 (iou3d_nms_kernel.cu:43-49,63-92)
__device__ __forceinline__ bool rot_seg_hit(float p1x, float p1y, float p0x, float p0y, float q1x, float q1y,
                                            float q0x, float q0y, float &ox, float &oy) {
    const bool bb = rlo(p0x, p1x) <= rhi(q0x, q1x) && rlo(q0x, q1x) <= rhi(p0x, p1x) &&
                    rlo(p0y, p1y) <= rhi(q0y, q1y) && rlo(q0y, q1y) <= rhi(p0y, p1y);
    if (!bb) return false;
    const float s1 = rcross3(q0x, q0y, p1x, p1y, p0x, p0y);
    const float s2 = rcross3(p1x, p1y, q1x, q1y, p0x, p0y);
    const float s3 = rcross3(p0x, p0y, q1x, q1y, q0x, q0y);
    const float s4 = rcross3(q1x, q1y, p1x, p1y, q0x, q0y);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = rcross3(q1x, q1y, p1x, p1y, p0x, p0y);
    if (fabsf(s5 - s1) > ROT_EPS) {
        ox = (s5 * q0x - s1 * q1x) / (s5 - s1);
        oy = (s5 * q0y - s1 * q1y) / (s5 - s1);
    } else {
        const float a0 = p0y - p1y, b0 = p1x - p0x, c0 = p0x * p1y - p1x * p0y;
        const float a1 = q0y - q1y, b1 = q1x - q0x, c1 = q0x * q1y - q1x * q0y;
        const float D = a0 * b1 - a1 * b0;
        ox = (b0 * c1 - b1 * c0) / D;
        oy = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

// corner (px,py) inside the MARGIN-inflated box whose record is r (iou3d_nms_kernel.cu:51-61)
__device__ __forceinline__ bool rot_inside(const float *r, float px, float py) {
    const float rx = (px - r[8]) * r[10] + (py - r[9]) * (-r[11]);
    const float ry = (px - r[8]) * r[11] + (py - r[9]) * r[10];
    return fabsf(rx) < r[12] && fabsf(ry) < r[13];
}

// A = "box_a" (row), B = "box_b" (column).  scratch: LDS base of this lane's polygon arrays,
// element k of array t at scratch[(t*ROT_PTS + k) * stride] (stride = lanes sharing the region).
__device__ __forceinline__ float rot_overlap(const float *A, const float *B, float *scratch, int stride) {
    if (A[16] < B[15] || B[16] < A[15] || A[18] < B[17] || B[18] < A[17]) return 0.f;
    float *qx = scratch, *qy = scratch + ROT_PTS * stride, *qa = scratch + 2 * ROT_PTS * stride;
    int n = 0;
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a1x = A[(i + 1) & 3], a1y = A[4 + ((i + 1) & 3)], a0x = A[i], a0y = A[4 + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float ox, oy;
            if (rot_seg_hit(a1x, a1y, a0x, a0y, B[(j + 1) & 3], B[4 + ((j + 1) & 3)], B[j], B[4 + j], ox, oy)) {
                if (n < ROT_PTS) { qx[n * stride] = ox; qy[n * stride] = oy; }
                sx = sx + ox; sy = sy + oy;
                ++n;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (rot_inside(A, B[k], B[4 + k])) {
            sx = sx + B[k]; sy = sy + B[4 + k];
            if (n < ROT_PTS) { qx[n * stride] = B[k]; qy[n * stride] = B[4 + k]; }
            ++n;
        }
        if (rot_inside(B, A[k], A[4 + k])) {
            sx = sx + A[k]; sy = sy + A[4 + k];
            if (n < ROT_PTS) { qx[n * stride] = A[k]; qy[n * stride] = A[4 + k]; }
            ++n;
        }
    }
    if (n == 0) return 0.f;
    if (n > ROT_PTS) n = ROT_PTS;  // cannot happen for real boxes (<= 16 hits + 8 corners)
    sx /= (float)n;
    sy /= (float)n;
    for (int k = 0; k < n; ++k)
        qa[k * stride] = (float)atan2((double)(qy[k * stride] - sy), (double)(qx[k * stride] - sx));
    for (int j = 0; j < n - 1; ++j)
        for (int i = 0; i < n - j - 1; ++i) {
            const float a = qa[i * stride], b = qa[(i + 1) * stride];
            if (a > b) {
                qa[i * stride] = b; qa[(i + 1) * stride] = a;
                float t = qx[i * stride]; qx[i * stride] = qx[(i + 1) * stride]; qx[(i + 1) * stride] = t;
                t = qy[i * stride]; qy[i * stride] = qy[(i + 1) * stride]; qy[(i + 1) * stride] = t;
            }
        }
    float area = 0.f;
    const float x0 = qx[0], y0 = qy[0];
    for (int k = 0; k < n - 1; ++k) {
        const float ux = qx[k * stride] - x0, uy = qy[k * stride] - y0;
        const float vx = qx[(k + 1) * stride] - x0, vy = qy[(k + 1) * stride] - y0;
        area += ux * vy - uy * vx;
    }
    return fabsf(area) / 2.0f;
}

}  // namespace md
