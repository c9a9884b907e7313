// tile_math.h -- oriented bounding box of a projected Gaussian and its candidate tile window
// (tile_culling.cu:69-122,138-156), shared by the binning kernels and the band pre-cull of the fused
// per-Gaussian stage.  fp32, no contraction, same operation order as the reference; cos/sin of the
// OBB angle are formed algebraically (see binning.hip).
#pragma once
#include "gs_common.h"

namespace gs {

struct Obb {
    float p[8];   // tl, tr, bl, br  (x, y)
    int radius_tiles;
};

// tile_culling.cu:69-122
__device__ inline Obb compute_obb(float u, float v, float a, float b, float c, float mh) {
    Obb o;
    const float left = (a + c) / 2;
    const float right = __builtin_sqrtf((a - c) * (a - c) / 4.0f + b * b);
    const float l1 = left + right;
    const float l2 = left - right;
    const float r_major = mh * __builtin_sqrtf(l1);
    const float r_minor = mh * __builtin_sqrtf(l2);
    float ct, st;
    if (__builtin_fabsf(b) < bits_f(0x24e69595u)) {   // fabsf(b) < 1e-16 (double compare)
        if (a >= c) { ct = 1.0f; st = 0.0f; }
        else { ct = -4.37113883e-8f; st = 1.0f; }     // cos/sin of float(pi/2)
    } else {
        const float y = l1 - a;
        const float h = __builtin_sqrtf(b * b + y * y);
        ct = b / h;
        st = y / h;
    }
    o.p[0] = -1 * r_major * ct + r_minor * st + u;
    o.p[1] = -1 * r_major * st - r_minor * ct + v;
    o.p[2] = r_major * ct + r_minor * st + u;
    o.p[3] = r_major * st - r_minor * ct + v;
    o.p[4] = -1 * r_major * ct - r_minor * st + u;
    o.p[5] = -1 * r_major * st + r_minor * ct + v;
    o.p[6] = r_major * ct - r_minor * st + u;
    o.p[7] = r_major * st + r_minor * ct + v;
    o.radius_tiles = f2i(__builtin_ceilf(r_major / 16.0f) + 1);
    return o;
}

struct Window {
    int sx, ex, sy, ey;
};

// tile_culling.cu:138-156 (+ the tile-row restriction used for multi-GPU sharding)
__device__ inline Window candidate_window(float u, float v, int r, int ntx, int nty, int row0,
                                          int row1) {
    Window w;
    const int px = f2i(__builtin_floorf(u / 16.0f));
    w.sx = f2i(fmaxf(0.0f, (float)(int)((unsigned)px - (unsigned)r)));
    w.ex = f2i(fminf((float)ntx, (float)(int)((unsigned)px + (unsigned)r)));
    const int py = f2i(__builtin_floorf(v / 16.0f));
    w.sy = f2i(fmaxf(0.0f, (float)(int)((unsigned)py - (unsigned)r)));
    w.ey = f2i(fminf((float)nty, (float)(int)((unsigned)py + (unsigned)r)));
    w.sy = max(w.sy, row0);
    w.ey = min(w.ey, row1);
    return w;
}

}  // namespace gs
