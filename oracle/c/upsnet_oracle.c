/*
 * upsnet_oracle.c -- TEST INFRASTRUCTURE ONLY (the "oracle").
 *
 * A plain-C, single-thread CPU restatement of the reference's per-image inference hot path
 * (uber-research/UPSNet). It is the checker for the HIP kernels in upsnet_amd/csrc; nothing in
 * the product path may link, import or call it. Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference/).
 * All arithmetic is IEEE fp32 evaluated in source order with NO fused multiply-add
 * (build with -ffp-contract=off), which is what "bit-exact" means for the HIP kernels that are
 * compared against it (they are built with -ffp-contract=off too).
 *
 * Parity status: the reference ships no tests/golden vectors for this path.
 * The restatement is pinned two ways instead:
 *   (1) against the reference's own .cu kernels hipified+compiled from /root/reference into
 *       oracle/_ref/ (see oracle/Makefile) and run on the MI355X (tests/test_ref_kernels_gpu.py);
 *   (2) against golden vectors produced by importing the reference's Python modules
 *       (tests/golden/make_golden.py) for the host-side glue.
 * cv2.resize (mask_removal.py:68) is third-party arithmetic that is absent here: its
 * INTER_LINEAR formula is restated from OpenCV's published algorithm -> "parity unpinned" for
 * that one function.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* The two loops that dominate the CPU baseline of bench.py (deformable im2col: one thread per channel; ROIAlign: one thread per ROI)
 * are OpenMP-parallel: every output element is still computed by ONE thread with the same expressions in the same order, so results do
 * not depend on the thread count. orc_set_threads(n): threads those loops use (default: 1 -- the tests run single-threaded). */
static int orc_threads = 1;
void orc_set_threads(int n) { orc_threads = n > 0 ? n : 1; }


/* ------------------------------------------------------------------------------------------
 * ROIAlign forward, NCHW.  upsnet/operators/src/roi_align_kernel.cu:43-95 (bilinear_interpolate)
 * and :163-235 (RoIAlignForward).  rois are [N,5] = (batch, x1, y1, x2, y2).
 * ---------------------------------------------------------------------------------------- */
static float orc_roi_bilinear(const float *data, int height, int width, float y, float x)
{
    /* roi_align_kernel.cu:51-54 */
    if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) return 0.0f;
    if (y <= 0) y = 0; /* :56-61 */
    if (x <= 0) x = 0;
    int y_low = (int)y, x_low = (int)x, y_high, x_high; /* :63-66 */
    if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
    if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
    float ly = y - (float)y_low, lx = x - (float)x_low; /* :82-84 */
    float hy = 1.0f - ly, hx = 1.0f - lx;
    float v1 = data[y_low * width + x_low], v2 = data[y_low * width + x_high];
    float v3 = data[y_high * width + x_low], v4 = data[y_high * width + x_high];
    float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    float val = w1 * v1;
    val = val + w2 * v2;
    val = val + w3 * v3;
    val = val + w4 * v4; /* :92 left-to-right */
    return val;
}

void orc_roi_align_forward(const float *feat, int channels, int height, int width,
                           const float *rois, int num_rois, int pooled_h, int pooled_w,
                           int sampling_ratio, float spatial_scale, float *out)
{
#pragma omp parallel for schedule(dynamic, 4) num_threads(orc_threads)
    for (int n = 0; n < num_rois; ++n) {
        const float *r = rois + n * 5;
        int roi_batch_ind = (int)roundf(r[0]);           /* :181 */
        float roi_start_w = r[1] * spatial_scale;         /* :185-188 no rounding */
        float roi_start_h = r[2] * spatial_scale;
        float roi_end_w = r[3] * spatial_scale;
        float roi_end_h = r[4] * spatial_scale;
        float roi_width = fmaxf(roi_end_w - roi_start_w, 1.0f);   /* :195-196 */
        float roi_height = fmaxf(roi_end_h - roi_start_h, 1.0f);
        float bin_size_h = roi_height / (float)pooled_h;
        float bin_size_w = roi_width / (float)pooled_w;
        int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)pooled_h);
        int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)pooled_w);
        const float count = (float)(grid_h * grid_w);     /* :212 */
        for (int c = 0; c < channels; ++c) {
            const float *plane = feat + ((size_t)roi_batch_ind * channels + c) * height * width;
            for (int ph = 0; ph < pooled_h; ++ph)
                for (int pw = 0; pw < pooled_w; ++pw) {
                    float acc = 0.0f;
                    for (int iy = 0; iy < grid_h; ++iy) {
                        /* :217-219 */
                        const float y = roi_start_h + (float)ph * bin_size_h +
                                        ((float)iy + .5f) * bin_size_h / (float)grid_h;
                        for (int ix = 0; ix < grid_w; ++ix) {
                            const float x = roi_start_w + (float)pw * bin_size_w +
                                            ((float)ix + .5f) * bin_size_w / (float)grid_w;
                            acc += orc_roi_bilinear(plane, height, width, y, x);
                        }
                    }
                    acc /= count;                          /* :231 */
                    out[(((size_t)n * channels + c) * pooled_h + ph) * pooled_w + pw] = acc;
                }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Deformable im2col v1 / v2, NCHW, one image (parallel_imgs = 1 as functions/deform_conv.py:50).
 * upsnet/operators/src/deform_conv_kernel.cu:88-118 (bilinear) and :194-242 (im2col kernel);
 * v2: mod_deform_conv_kernel.cu:187-249 (adds "* mask", :243).
 * col layout: [(c*kh*kw + i*kw + j), h_col, w_col].
 * ---------------------------------------------------------------------------------------- */
static float orc_dcn_bilinear(const float *plane, int height, int width, float h, float w)
{
    int h_low = (int)floorf(h), w_low = (int)floorf(w);
    int h_high = h_low + 1, w_high = w_low + 1;
    float lh = h - (float)h_low, lw = w - (float)w_low;
    float hh = 1.0f - lh, hw = 1.0f - lw;
    float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    if (h_low >= 0 && w_low >= 0) v1 = plane[h_low * width + w_low];
    if (h_low >= 0 && w_high <= width - 1) v2 = plane[h_low * width + w_high];
    if (h_high <= height - 1 && w_low >= 0) v3 = plane[h_high * width + w_low];
    if (h_high <= height - 1 && w_high <= width - 1) v4 = plane[h_high * width + w_high];
    float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    float val = w1 * v1;
    val = val + w2 * v2;
    val = val + w3 * v3;
    val = val + w4 * v4;
    return val;
}

void orc_deform_im2col(const float *im, const float *offset, const float *mask /* NULL => v1 */,
                       int channels, int height, int width, int kh, int kw, int pad_h, int pad_w,
                       int stride_h, int stride_w, int dil_h, int dil_w, int deformable_group,
                       float *col)
{
    const int height_col = (height + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1; /* :270 */
    const int width_col = (width + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
    const int cpg = channels / deformable_group;
    const size_t plane_col = (size_t)height_col * width_col;
#pragma omp parallel for schedule(static) num_threads(orc_threads)
    for (int c = 0; c < channels; ++c) {
        const int g = c / cpg;
        const float *plane = im + (size_t)c * height * width;
        const float *off_g = offset + (size_t)g * 2 * kh * kw * plane_col;
        const float *mask_g = mask ? mask + (size_t)g * kh * kw * plane_col : NULL;
        for (int h_col = 0; h_col < height_col; ++h_col)
            for (int w_col = 0; w_col < width_col; ++w_col) {
                const int h_in = h_col * stride_h - pad_h, w_in = w_col * stride_w - pad_w;
                for (int i = 0; i < kh; ++i)
                    for (int j = 0; j < kw; ++j) {
                        const size_t pix = (size_t)h_col * width_col + w_col;
                        const float off_h = off_g[(size_t)(2 * (i * kw + j)) * plane_col + pix];
                        const float off_w = off_g[(size_t)(2 * (i * kw + j) + 1) * plane_col + pix];
                        /* :227-228 integer part converted to float before the add */
                        const float h_im = (float)(h_in + i * dil_h) + off_h;
                        const float w_im = (float)(w_in + j * dil_w) + off_w;
                        float val = 0.0f;
                        if (h_im > -1 && w_im > -1 && h_im < (float)height && w_im < (float)width)
                            val = orc_dcn_bilinear(plane, height, width, h_im, w_im);
                        if (mask_g) val = val * mask_g[(size_t)(i * kw + j) * plane_col + pix];
                        col[((size_t)(c * kh * kw + i * kw + j)) * plane_col + pix] = val;
                    }
            }
    }
}

/* ------------------------------------------------------------------------------------------
 * Backward natives. The scatter kernels use atomics in the reference, so their
 * summation order is unspecified there; the restatement accumulates in kernel-index order and the tests
 * compare within 1e-5-level tolerances. The gather kernels (col2im_coord) have a fixed order: bit-exact.
 *   RoIAlignBackwardFeature + bilinear_interpolate_gradient   roi_align_kernel.cu:97-160, 238-349
 *   deformable_col2im / get_gradient_weight                   deform_conv_kernel.cu:120-143, 293-358
 *   deformable_col2im_coord / get_coordinate_weight           deform_conv_kernel.cu:146-184, 391-450
 *   modulated versions                                        mod_deform_conv_kernel.cu:251-381
 * ---------------------------------------------------------------------------------------- */
void orc_roi_align_backward(const float *top_diff, const float *rois, int num_rois, int channels, int height, int width,
                            int pooled_h, int pooled_w, int sampling_ratio, float spatial_scale, float *bottom_diff)
{
    for (int n = 0; n < num_rois; ++n) {
        const float *r = rois + n * 5;
        int b = (int)roundf(r[0]);
        float roi_start_w = r[1] * spatial_scale, roi_start_h = r[2] * spatial_scale;
        float roi_end_w = r[3] * spatial_scale, roi_end_h = r[4] * spatial_scale;
        float roi_width = fmaxf(roi_end_w - roi_start_w, 1.0f), roi_height = fmaxf(roi_end_h - roi_start_h, 1.0f);
        float bin_size_h = roi_height / (float)pooled_h, bin_size_w = roi_width / (float)pooled_w;
        int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)pooled_h);
        int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)pooled_w);
        const float count = (float)(grid_h * grid_w);
        for (int c = 0; c < channels; ++c) {
            float *plane = bottom_diff + ((size_t)b * channels + c) * height * width;
            for (int ph = 0; ph < pooled_h; ++ph)
                for (int pw = 0; pw < pooled_w; ++pw) {
                    const float top = top_diff[(((size_t)n * channels + c) * pooled_h + ph) * pooled_w + pw];
                    for (int iy = 0; iy < grid_h; ++iy) {
                        float y0 = roi_start_h + (float)ph * bin_size_h + ((float)iy + .5f) * bin_size_h / (float)grid_h;
                        for (int ix = 0; ix < grid_w; ++ix) {
                            float x = roi_start_w + (float)pw * bin_size_w + ((float)ix + .5f) * bin_size_w / (float)grid_w;
                            float y = y0;
                            if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) continue;   /* :112-117 */
                            if (y <= 0) y = 0;
                            if (x <= 0) x = 0;
                            int y_low = (int)y, x_low = (int)x, y_high, x_high;
                            if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else y_high = y_low + 1;
                            if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else x_high = x_low + 1;
                            float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.0f - ly, hx = 1.0f - lx;
                            float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                            plane[y_low * width + x_low] += top * w1 / count;     /* :324-327 */
                            plane[y_low * width + x_high] += top * w2 / count;
                            plane[y_high * width + x_low] += top * w3 / count;
                            plane[y_high * width + x_high] += top * w4 / count;
                        }
                    }
                }
        }
    }
}

static float orc_dcn_grad_weight(float ah, float aw, int h, int w, int height, int width)
{
    if (ah <= -1 || ah >= (float)height || aw <= -1 || aw >= (float)width) return 0.f;
    int hl = (int)floorf(ah), wl = (int)floorf(aw), hh = hl + 1, wh = wl + 1;
    float weight = 0.f;
    if (h == hl && w == wl) weight = ((float)(h + 1) - ah) * ((float)(w + 1) - aw);
    if (h == hl && w == wh) weight = ((float)(h + 1) - ah) * (aw + 1.0f - (float)w);
    if (h == hh && w == wl) weight = (ah + 1.0f - (float)h) * ((float)(w + 1) - aw);
    if (h == hh && w == wh) weight = (ah + 1.0f - (float)h) * (aw + 1.0f - (float)w);
    return weight;
}

/* col [C,kh,kw,B,Hc,Wc], offset [B,dg*2*kh*kw,Hc,Wc], mask [B,dg*kh*kw,Hc,Wc] or NULL (v1) -> grad_im [B,C,H,W] (accumulated) */
void orc_deform_col2im(const float *col, const float *offset, const float *mask, int batch, int channels, int height, int width,
                       int height_col, int width_col, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                       int dil_w, int deformable_group, float *grad_im)
{
    const int cpg = channels / deformable_group;
    const size_t plane_col = (size_t)height_col * width_col;
    size_t index = 0;
    for (int c = 0; c < channels; ++c)
        for (int i = 0; i < kh; ++i)
            for (int j = 0; j < kw; ++j)
                for (int b = 0; b < batch; ++b)
                    for (int h_out = 0; h_out < height_col; ++h_out)
                        for (int w_out = 0; w_out < width_col; ++w_out, ++index) {
                            const int g = c / cpg;
                            const size_t pix = (size_t)h_out * width_col + w_out;
                            const float *off = offset + ((size_t)b * deformable_group + g) * 2 * kh * kw * plane_col;
                            const float off_h = off[(size_t)(2 * (i * kw + j)) * plane_col + pix];
                            const float off_w = off[(size_t)(2 * (i * kw + j) + 1) * plane_col + pix];
                            const float ih = (float)(h_out * stride_h - pad_h + i * dil_h) + off_h;
                            const float iw = (float)(w_out * stride_w - pad_w + j * dil_w) + off_w;
                            float top = col[index];
                            if (mask) top = top * mask[(((size_t)b * deformable_group + g) * kh * kw + (i * kw + j)) * plane_col + pix];
                            const int cur_h = (int)ih, cur_w = (int)iw;        /* truncation, :339-340 */
                            for (int dy = -2; dy <= 2; ++dy)
                                for (int dx = -2; dx <= 2; ++dx) {
                                    const int h = cur_h + dy, w = cur_w + dx;
                                    if (h >= 0 && h < height && w >= 0 && w < width && fabsf(ih - (float)h) < 1 && fabsf(iw - (float)w) < 1)
                                        grad_im[(((size_t)b * channels + c) * height + h) * width + w] +=
                                            orc_dcn_grad_weight(ih, iw, h, w, height, width) * top;
                                }
                        }
}

static float orc_dcn_coord_weight(float ah, float aw, int height, int width, const float *im, int bp_dir)
{
    if (ah <= -1 || ah >= (float)height || aw <= -1 || aw >= (float)width) return 0.f;
    int hl = (int)floorf(ah), wl = (int)floorf(aw), hh = hl + 1, wh = wl + 1;
    float weight = 0.f;
    if (bp_dir == 0) {
        if (hl >= 0 && wl >= 0) weight += -1.0f * ((float)(wl + 1) - aw) * im[hl * width + wl];
        if (hl >= 0 && wh <= width - 1) weight += -1.0f * (aw - (float)wl) * im[hl * width + wh];
        if (hh <= height - 1 && wl >= 0) weight += ((float)(wl + 1) - aw) * im[hh * width + wl];
        if (hh <= height - 1 && wh <= width - 1) weight += (aw - (float)wl) * im[hh * width + wh];
    } else {
        if (hl >= 0 && wl >= 0) weight += -1.0f * ((float)(hl + 1) - ah) * im[hl * width + wl];
        if (hl >= 0 && wh <= width - 1) weight += ((float)(hl + 1) - ah) * im[hl * width + wh];
        if (hh <= height - 1 && wl >= 0) weight += -1.0f * (ah - (float)hl) * im[hh * width + wl];
        if (hh <= height - 1 && wh <= width - 1) weight += (ah - (float)hl) * im[hh * width + wh];
    }
    return weight;
}

/* + im [B,C,H,W] -> grad_offset [B,dg*2*kh*kw,Hc,Wc] and (mask != NULL) grad_mask [B,dg*kh*kw,Hc,Wc] */
void orc_deform_col2im_coord(const float *col, const float *im, const float *offset, const float *mask, int batch, int channels,
                             int height, int width, int height_col, int width_col, int kh, int kw, int pad_h, int pad_w,
                             int stride_h, int stride_w, int dil_h, int dil_w, int deformable_group, float *grad_offset,
                             float *grad_mask)
{
    const int taps = kh * kw, cpg = channels / deformable_group;
    const size_t plane_col = (size_t)height_col * width_col;
    for (int b = 0; b < batch; ++b)
        for (int g = 0; g < deformable_group; ++g)
            for (int oc = 0; oc < 2 * taps; ++oc)
                for (int h = 0; h < height_col; ++h)
                    for (int w = 0; w < width_col; ++w) {
                        const int tap = oc / 2, bp_dir = oc % 2, i = tap / kw, j = tap % kw;
                        const size_t pix = (size_t)h * width_col + w;
                        const float *off = offset + ((size_t)b * deformable_group + g) * 2 * taps * plane_col;
                        const float off_h = off[(size_t)(2 * tap) * plane_col + pix], off_w = off[(size_t)(2 * tap + 1) * plane_col + pix];
                        float ih = (float)(h * stride_h - pad_h + i * dil_h) + off_h;
                        float iw = (float)(w * stride_w - pad_w + j * dil_w) + off_w;
                        const int outside = ih <= -1 || iw <= -1 || ih >= (float)height || iw >= (float)width;
                        if (outside) ih = iw = -2.f;
                        const float m = mask ? mask[(((size_t)b * deformable_group + g) * taps + tap) * plane_col + pix] : 1.f;
                        float val = 0.f, mval = 0.f;
                        for (int cc = 0; cc < cpg; ++cc) {
                            const int c_im = g * cpg + cc;
                            const float cv = col[(((size_t)(c_im * taps + tap)) * batch + b) * plane_col + pix];
                            const float *plane = im + ((size_t)b * channels + c_im) * height * width;
                            if (mask && !outside) mval += cv * orc_dcn_bilinear(plane, height, width, ih, iw);
                            const float weight = orc_dcn_coord_weight(ih, iw, height, width, plane, bp_dir);
                            if (mask) val += weight * cv * m;
                            else val += weight * cv;
                        }
                        grad_offset[(((size_t)b * deformable_group + g) * 2 * taps + oc) * plane_col + pix] = val;
                        if (mask && bp_dir == 0) grad_mask[(((size_t)b * deformable_group + g) * taps + tap) * plane_col + pix] = mval;
                    }
}

/* ------------------------------------------------------------------------------------------
 * Hard NMS on score-sorted boxes [N,5]: restates _nms + nms_kernel + devIoU,
 * upsnet/nms/nms_kernel.cu:30-38 (IoU, +1 convention), :73-80 (strict ">"), :130-146 (greedy scan).
 * Returns the number kept; keep_out[] = indices into the SORTED array in visiting order.
 * ---------------------------------------------------------------------------------------- */
static float orc_iou(const float *a, const float *b)
{
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float width = fmaxf(right - left + 1.0f, 0.f), height = fmaxf(bottom - top + 1.0f, 0.f);
    float interS = width * height;
    float Sa = (a[2] - a[0] + 1.0f) * (a[3] - a[1] + 1.0f);
    float Sb = (b[2] - b[0] + 1.0f) * (b[3] - b[1] + 1.0f);
    return interS / (Sa + Sb - interS);
}

int orc_nms_sorted(const float *boxes, int n, int box_dim, float thresh, int *keep_out)
{
    uint8_t *removed = (uint8_t *)calloc(n > 0 ? n : 1, 1);
    int num = 0;
    for (int i = 0; i < n; ++i) {
        if (removed[i]) continue;
        keep_out[num++] = i;
        for (int j = i + 1; j < n; ++j)
            if (!removed[j] && orc_iou(boxes + (size_t)i * box_dim, boxes + (size_t)j * box_dim) > thresh)
                removed[j] = 1;
    }
    free(removed);
    return num;
}

/* cpu_nms (upsnet/nms/cpu_nms.pyx:29-80) on score-sorted boxes: same IoU expression (areas precomputed, :39; inter / (iarea +
 * areas[j] - inter), :76), but suppression at ">=" and against the PYTHON float threshold, i.e. the fp32 overlap is compared
 * in double (:77, `np.float thresh` is a Python object in the compiled module). */
int orc_cpu_nms_sorted(const float *boxes, int n, int box_dim, double thresh, int *keep_out)
{
    uint8_t *removed = (uint8_t *)calloc(n > 0 ? n : 1, 1);
    int num = 0;
    for (int i = 0; i < n; ++i) {
        if (removed[i]) continue;
        keep_out[num++] = i;
        for (int j = i + 1; j < n; ++j)
            if (!removed[j] && (double)orc_iou(boxes + (size_t)i * box_dim, boxes + (size_t)j * box_dim) >= thresh)
                removed[j] = 1;
    }
    free(removed);
    return num;
}

/* ------------------------------------------------------------------------------------------
 * Soft-NMS: upsnet/nms/cpu_nms.pyx:91-196 (cdef float arithmetic; gaussian weight through a
 * double exp, :170).  boxes [N,5] is modified in place, inds[N] must hold 0..N-1 on entry.
 * Returns the new N.
 * ---------------------------------------------------------------------------------------- */
int orc_soft_nms(float *boxes, int64_t *inds, int n, float sigma, float Nt, float threshold,
                 unsigned method)
{
    int N = n;
    for (int i = 0; i < N; ++i) {
        float maxscore = boxes[i * 5 + 4];
        int maxpos = i;
        float tx1 = boxes[i * 5 + 0], ty1 = boxes[i * 5 + 1], tx2 = boxes[i * 5 + 2],
              ty2 = boxes[i * 5 + 3], ts = boxes[i * 5 + 4];
        int64_t ti = inds[i];
        for (int pos = i + 1; pos < N; ++pos)
            if (maxscore < boxes[pos * 5 + 4]) { maxscore = boxes[pos * 5 + 4]; maxpos = pos; }
        for (int k = 0; k < 5; ++k) boxes[i * 5 + k] = boxes[maxpos * 5 + k];
        inds[i] = inds[maxpos];
        boxes[maxpos * 5 + 0] = tx1; boxes[maxpos * 5 + 1] = ty1; boxes[maxpos * 5 + 2] = tx2;
        boxes[maxpos * 5 + 3] = ty2; boxes[maxpos * 5 + 4] = ts;
        inds[maxpos] = ti;
        tx1 = boxes[i * 5 + 0]; ty1 = boxes[i * 5 + 1]; tx2 = boxes[i * 5 + 2]; ty2 = boxes[i * 5 + 3];
        int pos = i + 1;
        while (pos < N) {
            float x1 = boxes[pos * 5 + 0], y1 = boxes[pos * 5 + 1], x2 = boxes[pos * 5 + 2],
                  y2 = boxes[pos * 5 + 3];
            /* Cython coerces the integer literal in `x2 - x1 + 1` (cdef float operands, :155-163) to the C constant 1.0, a
             * DOUBLE: the float difference is widened, the two-factor products and the union are evaluated in double and
             * rounded to float once on assignment. Found by pinning against the compiled reference (tests/test_ref_cpu_nms.py);
             * a pure-fp32 restatement is off by an ulp in `area` / `ua` and hence in the linear / gaussian scores. */
            float area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0));
            float iw = (float)((double)(fminf(tx2, x2) - fmaxf(tx1, x1)) + 1.0);
            if (iw > 0) {
                float ih = (float)((double)(fminf(ty2, y2) - fmaxf(ty1, y1)) + 1.0);
                if (ih > 0) {
                    float ua = (float)(((((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0)) + (double)area) - (double)(iw * ih));
                    float ov = iw * ih / ua;
                    float weight;
                    if (method == 1) weight = ov > Nt ? (float)(1.0 - (double)ov) : 1;
                    else if (method == 2) weight = (float)exp((double)(-(ov * ov) / sigma));
                    else weight = ov > Nt ? 0 : 1;
                    boxes[pos * 5 + 4] = weight * boxes[pos * 5 + 4];
                    if (boxes[pos * 5 + 4] < threshold) {
                        for (int k = 0; k < 5; ++k) boxes[pos * 5 + k] = boxes[(N - 1) * 5 + k];
                        inds[pos] = inds[N - 1];
                        N = N - 1;
                        pos = pos - 1;
                    }
                }
            }
            pos = pos + 1;
        }
    }
    return N;
}

/* ------------------------------------------------------------------------------------------
 * Stand-in for cv2.resize(src 28x28 fp32, (w, h)) INTER_LINEAR (mask_removal.py:68).
 * OpenCV resize.cpp: scale = src/dst (double); fx = (float)((dx+0.5)*scale - 0.5); sx = floor(fx);
 * fx -= sx; sx<0 -> (0, fx=0); sx >= S-1 -> (S-1, fx=0); horizontal pass on both source rows,
 * then vertical pass; fp32 weights (1-f, f).  PARITY UNPINNED (cv2 not installed here).
 * ---------------------------------------------------------------------------------------- */
static void orc_lin_coef(int d, int dsize, int ssize, int *s0, int *s1, float *f)
{
    double scale = (double)ssize / (double)dsize;
    float fx = (float)(((double)d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= ssize - 1) { fx = 0.f; sx = ssize - 1; }
    *s0 = sx;
    *s1 = sx + 1 < ssize ? sx + 1 : ssize - 1;
    *f = fx;
}

float orc_resize_at(const float *src, int ssize, int dw, int dh, int dx, int dy)
{
    int x0, x1, y0, y1;
    float fx, fy;
    orc_lin_coef(dx, dw, ssize, &x0, &x1, &fx);
    orc_lin_coef(dy, dh, ssize, &y0, &y1, &fy);
    float a0 = 1.0f - fx, a1 = fx, b0 = 1.0f - fy, b1 = fy;
    float r0 = src[y0 * ssize + x0] * a0 + src[y0 * ssize + x1] * a1;
    float r1 = src[y1 * ssize + x0] * a0 + src[y1 * ssize + x1] * a1;
    return r0 * b0 + r1 * b1;
}

void orc_resize_bilinear(const float *src, int ssize, int dw, int dh, float *dst)
{
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) dst[(size_t)y * dw + x] = orc_resize_at(src, ssize, dw, dh, x, y);
}

/* ------------------------------------------------------------------------------------------
 * MaskRemoval: upsnet/operators/modules/mask_removal.py:50-93.
 *   rois [m,4] (x1,y1,x2,y2 image coords), mask_logit [m, msize*msize], cls_idx [m] (1-based class),
 *   order [m] = visiting order (the caller applies the argsort()[::-1] tie rule, :50).
 *   occupancy: uint8 [num_cls, H, W] scratch, zeroed here.
 * Outputs keep_inds (original indices, visiting order) and, if mask_energy != NULL, the pasted
 * logits [k, H, W] (zero elsewhere).  Returns k (0 means the reference's "empty" branch).
 * ---------------------------------------------------------------------------------------- */
int orc_mask_removal(const float *rois, const float *mask_logit, const int64_t *cls_idx,
                     const int64_t *order, int m, int msize, int H, int W, int num_cls,
                     double fraction_threshold, uint8_t *occupancy, int64_t *keep_inds,
                     float *mask_energy)
{
    memset(occupancy, 0, (size_t)num_cls * H * W);
    int k = 0;
    for (int ii = 0; ii < m; ++ii) {
        const int64_t i = order[ii];
        const int cls = (int)cls_idx[i] - 1;             /* :54 */
        int bx[4];
        for (int q = 0; q < 4; ++q) bx[q] = (int)rois[i * 4 + q]; /* astype(int32): trunc, :60 */
        int w = bx[2] - bx[0] + 1, h = bx[3] - bx[1] + 1;
        if (w < 1) w = 1;
        if (h < 1) h = 1;
        int x_0 = bx[0] > 0 ? bx[0] : 0, x_1 = bx[2] + 1 < W ? bx[2] + 1 : W;
        int y_0 = bx[1] > 0 ? bx[1] : 0, y_1 = bx[3] + 1 < H ? bx[3] + 1 : H;
        const float *src = mask_logit + (size_t)i * msize * msize;
        uint8_t *occ = occupancy + (size_t)cls * H * W;
        long mask_sum = 0, overlap = 0;
        /* python slicing mask[(y_0-by):(y_1-by), (x_0-bx):(x_1-bx)] clamps the stop to (h, w) */
        for (int y = y_0; y < y_1; ++y) {
            int ly = y - bx[1];
            if (ly < 0 || ly >= h) continue;
            for (int x = x_0; x < x_1; ++x) {
                int lx = x - bx[0];
                if (lx < 0 || lx >= w) continue;
                if (orc_resize_at(src, msize, w, h, lx, ly) > 0) {
                    ++mask_sum;
                    if (occ[(size_t)y * W + x] >= 1) ++overlap;
                }
            }
        }
        if (mask_sum == 0) continue;                                        /* :82 */
        if ((double)overlap / (double)mask_sum > fraction_threshold) continue; /* int/int true division vs python float */
        keep_inds[k] = i;
        float *plane = mask_energy ? mask_energy + (size_t)k * H * W : NULL;
        if (plane) memset(plane, 0, (size_t)H * W * sizeof(float));
        for (int y = y_0; y < y_1; ++y) {
            int ly = y - bx[1];
            if (ly < 0 || ly >= h) continue;
            for (int x = x_0; x < x_1; ++x) {
                int lx = x - bx[0];
                if (lx < 0 || lx >= w) continue;
                float v = orc_resize_at(src, msize, w, h, lx, ly);
                if (v > 0) occ[(size_t)y * W + x] += 1;     /* uint8 += , :85 */
                if (plane) plane[(size_t)y * W + x] = v;   /* :86 */
            }
        }
        ++k;
    }
    return k;
}

/* ------------------------------------------------------------------------------------------
 * SegTerm instance planes: upsnet/operators/modules/unary_logits.py:95-103.
 * boxes [k,4] already multiplied by box_scale (image coords); cls [k]; fcn [S,H,W];
 * class_mapping: cls c -> channel (S - num_inst_classes - 1 ... ) passed in as map[c].
 * ---------------------------------------------------------------------------------------- */
static int orc_py_slice_lo(long v, int n) { if (v < 0) { v += n; if (v < 0) v = 0; } if (v > n) v = n; return (int)v; }

void orc_seg_term(const float *fcn, int S, int H, int W, const float *boxes, const int64_t *cls,
                  const int64_t *class_map, int k, float *seg_inst)
{
    (void)S;
    memset(seg_inst, 0, (size_t)k * H * W * sizeof(float));
    for (int i = 0; i < k; ++i) {
        if (cls[i] == 0) continue;                          /* :97-98 */
        long y0 = (long)boxes[i * 4 + 1];                    /* int() truncates */
        long y1 = (long)(rintf(boxes[i * 4 + 3]) + 1.0f);    /* numpy round = half-even, :100 */
        long x0 = (long)boxes[i * 4 + 0];
        long x1 = (long)(rintf(boxes[i * 4 + 2]) + 1.0f);
        int ys = orc_py_slice_lo(y0, H), ye = orc_py_slice_lo(y1, H);
        int xs = orc_py_slice_lo(x0, W), xe = orc_py_slice_lo(x1, W);
        const float *src = fcn + (size_t)class_map[cls[i]] * H * W;
        float *dst = seg_inst + (size_t)i * H * W;
        for (int y = ys; y < ye; ++y)
            for (int x = xs; x < xe; ++x) dst[(size_t)y * W + x] = src[(size_t)y * W + x];
    }
}

/* ------------------------------------------------------------------------------------------
 * Panoptic fusion: upsnet/models/resnet_upsnet.py:234-243.
 *   fcn [S,H,W]; s_stuff = S - (num_classes-1); seg_inst, mask_energy [k,H,W].
 *   enable_void: logits = [stuff | seg_inst+mask | void], argmax (first max), void -> 255.
 *   else       : argmax(softmax([stuff | seg_inst+mask])), softmax done explicitly.
 * Also emits the semantic argmax (resnet_upsnet.py:213) if sem_out != NULL.
 * ---------------------------------------------------------------------------------------- */
void orc_panoptic_fuse(const float *fcn, int S, int H, int W, int s_stuff, const float *seg_inst,
                       const float *mask_energy, int k, int enable_void, int64_t *pan_out,
                       int64_t *sem_out)
{
    const size_t HW = (size_t)H * W;
    float *logit = (float *)malloc(sizeof(float) * (size_t)(s_stuff + k + 1));
    for (size_t p = 0; p < HW; ++p) {
        int nch = 0;
        for (int c = 0; c < s_stuff; ++c) logit[nch++] = fcn[(size_t)c * HW + p];
        for (int i = 0; i < k; ++i) logit[nch++] = seg_inst[(size_t)i * HW + p] + mask_energy[(size_t)i * HW + p];
        if (enable_void) {
            float mt = fcn[(size_t)s_stuff * HW + p];
            for (int c = s_stuff + 1; c < S; ++c) { float v = fcn[(size_t)c * HW + p]; if (v > mt) mt = v; }
            float mi = seg_inst[p];
            for (int i = 1; i < k; ++i) { float v = seg_inst[(size_t)i * HW + p]; if (v > mi) mi = v; }
            logit[nch++] = mt - mi;
            int best = 0;
            for (int c = 1; c < nch; ++c) if (logit[c] > logit[best]) best = c;
            pan_out[p] = best == nch - 1 ? 255 : best;
        } else {
            float m = logit[0];
            for (int c = 1; c < nch; ++c) if (logit[c] > m) m = logit[c];
            float s = 0.f;
            for (int c = 0; c < nch; ++c) { logit[c] = (float)exp((double)(logit[c] - m)); s += logit[c]; }
            int best = 0;
            float bp = logit[0] / s;
            for (int c = 1; c < nch; ++c) { float pr = logit[c] / s; if (pr > bp) { bp = pr; best = c; } }
            pan_out[p] = best;
        }
        if (sem_out) {
            int best = 0;
            for (int c = 1; c < S; ++c) if (fcn[(size_t)c * HW + p] > fcn[(size_t)best * HW + p]) best = c;
            sem_out[p] = best;
        }
    }
    free(logit);
}

/* ------------------------------------------------------------------------------------------
 * F.interpolate(score, None, scale, mode='bilinear', align_corners=False) (upsnet/models/fcn.py:101), planar
 * [S,Hs,Ws] -> [S,Hs*scale,Ws*scale]. Restates PyTorch's upsample_bilinear2d arithmetic (ATen UpSample.cuh:
 * src = r*(dst+0.5)-0.5 clamped at 0 with r = 1/scale; h1 = (int)src; h1p = h1 < Hs-1; lambda = src - h1;
 * val = h0*(w0*v00 + w1*v01) + h1l*(w0*v10 + w1*v11)) in fp32 without FMA. PyTorch's own kernel is
 * compiled with FMA contraction, so this equals torch to ~1 ulp, not bit-for-bit ("parity unpinned" at the
 * ulp level, like every library op of the reference).
 * ---------------------------------------------------------------------------------------- */
void orc_upsample_bilinear(const float *src, int S, int Hs, int Ws, int scale, float *dst)
{
    const int H = Hs * scale, W = Ws * scale;
    const float r = 1.0f / (float)scale;
    for (int c = 0; c < S; ++c)
        for (int y = 0; y < H; ++y) {
            float h1r = r * ((float)y + 0.5f) - 0.5f;
            if (h1r < 0) h1r = 0;
            const int h1 = (int)h1r, h1p = h1 < Hs - 1 ? 1 : 0;
            const float h1l = h1r - (float)h1, h0l = 1.0f - h1l;
            for (int x = 0; x < W; ++x) {
                float w1r = r * ((float)x + 0.5f) - 0.5f;
                if (w1r < 0) w1r = 0;
                const int w1 = (int)w1r, w1p = w1 < Ws - 1 ? 1 : 0;
                const float w1l = w1r - (float)w1, w0l = 1.0f - w1l;
                const float *p = src + (size_t)c * Hs * Ws;
                const float top = w0l * p[h1 * Ws + w1] + w1l * p[h1 * Ws + w1 + w1p];
                const float bot = w0l * p[(h1 + h1p) * Ws + w1] + w1l * p[(h1 + h1p) * Ws + w1 + w1p];
                dst[((size_t)c * H + y) * W + x] = h0l * top + h1l * bot;
            }
        }
}

/* ------------------------------------------------------------------------------------------
 * Semantic-head tail with the 1x1 score conv commuted below the bilinear upsampling (fcn.py:94-100 restated through
 * linearity): out[y,x,s] = bias[s] + part0[y,x,s] + sum_{l>=1} up_{2^l}(part_l)[y,x,s]; all maps NHWC [H>>l, W>>l, S].
 * Interpolation terms as in orc_upsample_bilinear (PyTorch upsample_bilinear2d, align_corners=False).
 * ---------------------------------------------------------------------------------------- */
void orc_fcn_score_combine(const float *const *part, int nlev, int S, int H, int W, const float *bias, float *out)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int s = 0; s < S; ++s) {
                const size_t idx = ((size_t)y * W + x) * S + s;
                float acc = part[0][idx];
                if (bias) acc = acc + bias[s];
                for (int l = 1; l < nlev; ++l) {
                    const int Hs = H >> l, Ws = W >> l;
                    const float r = 1.0f / (float)(1 << l);
                    float h1r = r * ((float)y + 0.5f) - 0.5f;
                    if (h1r < 0) h1r = 0;
                    const int h1 = (int)h1r, h1p = h1 < Hs - 1 ? 1 : 0;
                    const float h1l = h1r - (float)h1, h0l = 1.0f - h1l;
                    float w1r = r * ((float)x + 0.5f) - 0.5f;
                    if (w1r < 0) w1r = 0;
                    const int w1 = (int)w1r, w1p = w1 < Ws - 1 ? 1 : 0;
                    const float w1l = w1r - (float)w1, w0l = 1.0f - w1l;
                    const float *q = part[l] + s;
                    const float top = w0l * q[((size_t)h1 * Ws + w1) * S] + w1l * q[((size_t)h1 * Ws + w1 + w1p) * S];
                    const float bot = w0l * q[((size_t)(h1 + h1p) * Ws + w1) * S] + w1l * q[((size_t)(h1 + h1p) * Ws + w1 + w1p) * S];
                    acc = acc + (h0l * top + h1l * bot);
                }
                out[idx] = acc;
            }
}

/* ------------------------------------------------------------------------------------------
 * Input blob: BaseDataset.prep_im_for_blob + im_list_to_blob (upsnet/dataset/base_dataset.py:143-173, 898-923).
 *   im uint8 [H,W,3] -> float32; im -= pixel_means (a float64 array: numpy computes float32 - float64 in double and rounds
 *   the result to float32); cv2.resize(im, None, None, fx=s, fy=s, INTER_LINEAR) with dsize = cvRound(size * s) (given by the
 *   caller as Hr, Wr) and coordinate scale 1/s; HWC -> CHW; zero pad to [Hp, Wp]. cv2 is absent: OpenCV's published
 *   INTER_LINEAR formula (see orc_lin_coef) -- PARITY UNPINNED for s != 1; for s == 1 it is an exact copy, which is pinned
 *   against the reference's own Python (tests/golden/make_golden.py).
 * out: planar [3,Hp,Wp].
 * ---------------------------------------------------------------------------------------- */
static void orc_lin_coef_scale(int d, double scale, int ssize, int *s0, int *s1, float *f)
{
    float fx = (float)(((double)d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= ssize - 1) { fx = 0.f; sx = ssize - 1; }
    *s0 = sx;
    *s1 = sx + 1 < ssize ? sx + 1 : ssize - 1;
    *f = fx;
}

void orc_prep_image(const uint8_t *im, int H, int W, const double *means, double im_scale, int Hr, int Wr, int Hp, int Wp, float *out)
{
    const double inv = 1.0 / im_scale;
    memset(out, 0, sizeof(float) * 3 * (size_t)Hp * Wp);
    for (int y = 0; y < Hr; ++y) {
        int y0, y1; float fy;
        orc_lin_coef_scale(y, inv, H, &y0, &y1, &fy);
        for (int x = 0; x < Wr; ++x) {
            int x0, x1; float fx;
            orc_lin_coef_scale(x, inv, W, &x0, &x1, &fx);
            const float a0 = 1.0f - fx, a1 = fx, b0 = 1.0f - fy, b1 = fy;
            for (int c = 0; c < 3; ++c) {
                const float s00 = (float)((double)im[((size_t)y0 * W + x0) * 3 + c] - means[c]);
                const float s01 = (float)((double)im[((size_t)y0 * W + x1) * 3 + c] - means[c]);
                const float s10 = (float)((double)im[((size_t)y1 * W + x0) * 3 + c] - means[c]);
                const float s11 = (float)((double)im[((size_t)y1 * W + x1) * 3 + c] - means[c]);
                const float r0 = s00 * a0 + s01 * a1;
                const float r1 = s10 * a0 + s11 * a1;
                out[((size_t)c * Hp + y) * Wp + x] = r0 * b0 + r1 * b1;
            }
        }
    }
}
