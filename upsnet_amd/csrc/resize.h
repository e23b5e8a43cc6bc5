// resize.h -- cv2.resize INTER_LINEAR restated for device code (shared by panoptic.hip and postprocess.hip).
#pragma once

// ---- cv2.resize INTER_LINEAR restated (see oracle/c/upsnet_oracle.c: orc_lin_coef)
__device__ static inline void pan_lin_coef(int d, int dsize, int ssize, int &s0, int &s1, float &f)
{
    const double scale = (double)ssize / (double)dsize;
    float fx = (float)(((double)d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= ssize - 1) { fx = 0.f; sx = ssize - 1; }
    s0 = sx;
    s1 = sx + 1 < ssize ? sx + 1 : ssize - 1;
    f = fx;
}

__device__ static inline float pan_resize_coef(const float *__restrict__ src, int ssize, int x0, int x1, float fx, int y0,
                                               int y1, float fy)
{
    const float a0 = 1.0f - fx, a1 = fx, b0 = 1.0f - fy, b1 = fy;
    const float r0 = src[y0 * ssize + x0] * a0 + src[y0 * ssize + x1] * a1;
    const float r1 = src[y1 * ssize + x0] * a0 + src[y1 * ssize + x1] * a1;
    return r0 * b0 + r1 * b1;
}

__device__ static inline float pan_resize_at(const float *__restrict__ src, int ssize, int dw, int dh, int dx, int dy)
{
    int x0, x1, y0, y1;
    float fx, fy;
    pan_lin_coef(dx, dw, ssize, x0, x1, fx);
    pan_lin_coef(dy, dh, ssize, y0, y1, fy);
    return pan_resize_coef(src, ssize, x0, x1, fx, y0, y1, fy);
}

