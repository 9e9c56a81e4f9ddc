// Bilinear image resize for uint8 HWC frames, arithmetic-compatible with OpenCV's
// cv::resize(..., fx, fy, INTER_LINEAR) on 8-bit images (the call at the reference's
// code/detection/detection.py:77):
//   * destination size = round-half-even(src size * scale);
//   * source coordinate of destination column dx: fx = (float)((dx + 0.5) / scale - 0.5),
//     sx = floor(fx), weight fx - sx; clamped to the first / last source column;
//   * weights are quantised to 11 bits (x 2048, round half even) and the two passes are
//     integer:  row value = S[sx] * a0 + S[sx+1] * a1,
//               out = (((b0 * (R0 >> 4)) >> 16) + ((b1 * (R1 >> 4)) >> 16) + 2) >> 2;
//   * an exact 2x reduction in both directions averages 2x2 blocks: (a + b + c + d + 2) >> 2.
// HBM-bound: every source byte is read about once and every destination byte written once.
// The coordinate arithmetic must round like the host's, so this file is built with
// floating-point contraction off.
#include "fd_common.h"

#include <cmath>

#pragma clang fp contract(off)

namespace fd {

struct Tap2 {
    int s0, s1;      // the two source indices
    int w0, w1;      // 11-bit weights
};

__device__ __forceinline__ Tap2 linear_tap(int d, double scale, int ssize) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    Tap2 t;
    t.s0 = s;
    t.s1 = min(s + 1, ssize - 1);
    t.w0 = (int)rintf((1.f - f) * 2048.f);
    t.w1 = (int)rintf(f * 2048.f);
    return t;
}

__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, int c,
                                                              double scale_x, double scale_y, uint8_t* __restrict__ dst,
                                                              int dh, int dw, int area2, int reverse) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (dx >= dw || dy >= dh) return;
    const uint8_t* s = src + (size_t)b * sh * sw * c;
    uint8_t* o = dst + (((size_t)b * dh + dy) * dw + dx) * c;
    // reverse: destination channel k takes source channel c-1-k (BGR <-> RGB)
    const int o0 = reverse ? c - 1 : 0, os = reverse ? -1 : 1;
    if (area2) {
        const uint8_t* r0 = s + ((size_t)(2 * dy) * sw + 2 * dx) * c;
        const uint8_t* r1 = r0 + (size_t)sw * c;
        for (int k = 0; k < c; ++k) o[o0 + os * k] = (uint8_t)((r0[k] + r0[c + k] + r1[k] + r1[c + k] + 2) >> 2);
        return;
    }
    const Tap2 tx = linear_tap(dx, scale_x, sw);
    const Tap2 ty = linear_tap(dy, scale_y, sh);
    const uint8_t* r0 = s + (size_t)ty.s0 * sw * c;
    const uint8_t* r1 = s + (size_t)ty.s1 * sw * c;
    for (int k = 0; k < c; ++k) {
        const int h0 = r0[tx.s0 * c + k] * tx.w0 + r0[tx.s1 * c + k] * tx.w1;
        const int h1 = r1[tx.s0 * c + k] * tx.w0 + r1[tx.s1 * c + k] * tx.w1;
        o[o0 + os * k] = (uint8_t)((((ty.w0 * (h0 >> 4)) >> 16) + ((ty.w1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
}

// ---------------------------------------------------------------------------------
// Face crops for the recognition stage: the reference slices a padded box out of the frame
// (code/common/Function.py:49-70) and later letterboxes every crop to a fixed size
// (code/common/Function.py:95-113).  Here both happen in one launch for all boxes of a
// frame: block z handles box z, every thread one destination pixel.
//   rect:  (x1,y1,x2,y2) truncated to int; pad = round-half-even(side * 0.1) per direction;
//          x1 = max(0, x1 - pad), x2 = min(W, x2 + pad), same for y (x2/y2 exclusive).
//   scale: crop h/w > dest h/w ? dest_h / crop_h : dest_w / crop_w  (double);
//          resized size = round-half-even(crop size * scale), centred, zero border.
// ---------------------------------------------------------------------------------
struct CropGeom {
    int x1, y1, x2, y2;     // source rectangle in the frame
    int rh, rw;             // size of the resized crop inside the tile
    int top, left;          // its offset inside the tile
    double inv_scale;       // source step per destination pixel
    int area2;              // exact 2x reduction -> 2x2 average
};

__device__ __forceinline__ CropGeom crop_geom(const double* __restrict__ box, int H, int W, int dh, int dw) {
    CropGeom g;
    int x1 = (int)box[0], y1 = (int)box[1], x2 = (int)box[2], y2 = (int)box[3];
    const int pw = (int)rint((double)(x2 - x1) * 0.1);
    const int ph = (int)rint((double)(y2 - y1) * 0.1);
    g.x1 = max(0, x1 - pw);
    g.y1 = max(0, y1 - ph);
    g.x2 = min(W, x2 + pw);
    g.y2 = min(H, y2 + ph);
    const int ch = g.y2 - g.y1, cw = g.x2 - g.x1;
    g.rh = g.rw = g.top = g.left = 0;
    g.inv_scale = 1.0;
    g.area2 = 0;
    if (ch <= 0 || cw <= 0) return g;
    const double ratio_src = (double)ch / (double)cw, ratio_des = (double)dh / (double)dw;
    const bool tall = ratio_src > ratio_des;
    const double scale = tall ? (double)dh / (double)ch : (double)dw / (double)cw;
    g.rh = (int)rint((double)ch * scale);
    g.rw = (int)rint((double)cw * scale);
    g.inv_scale = 1.0 / scale;
    g.area2 = (g.inv_scale == 2.0 && ch % 2 == 0 && cw % 2 == 0) ? 1 : 0;
    if (tall) g.left = (dw - g.rw) >> 1;      // arithmetic shift == Python's floor division
    else g.top = (dh - g.rh) >> 1;
    return g;
}

__global__ __launch_bounds__(256) void crop_resize_pad_kernel(const uint8_t* __restrict__ frame, int H, int W, int c,
                                                             const double* __restrict__ boxes, int box_stride,
                                                             uint8_t* __restrict__ tiles, int dh, int dw,
                                                             int32_t* __restrict__ rects) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    const CropGeom g = crop_geom(boxes + (size_t)b * box_stride, H, W, dh, dw);
    if (rects && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        rects[b * 4 + 0] = g.x1; rects[b * 4 + 1] = g.y1; rects[b * 4 + 2] = g.x2; rects[b * 4 + 3] = g.y2;
    }
    if (!tiles || ox >= dw || oy >= dh) return;
    uint8_t* o = tiles + (((size_t)b * dh + oy) * dw + ox) * c;
    const int ix = ox - g.left, iy = oy - g.top;
    if (ix < 0 || iy < 0 || ix >= g.rw || iy >= g.rh) {
        for (int k = 0; k < c; ++k) o[k] = 0;
        return;
    }
    const int ch = g.y2 - g.y1, cw = g.x2 - g.x1;
    const uint8_t* s = frame + ((size_t)g.y1 * W + g.x1) * c;          // crop origin; row pitch W * c
    const size_t pitch = (size_t)W * c;
    if (g.area2) {
        const uint8_t* r0 = s + (size_t)(2 * iy) * pitch + (size_t)(2 * ix) * c;
        const uint8_t* r1 = r0 + pitch;
        for (int k = 0; k < c; ++k) o[k] = (uint8_t)((r0[k] + r0[c + k] + r1[k] + r1[c + k] + 2) >> 2);
        return;
    }
    const Tap2 tx = linear_tap(ix, g.inv_scale, cw);
    const Tap2 ty = linear_tap(iy, g.inv_scale, ch);
    const uint8_t* r0 = s + (size_t)ty.s0 * pitch;
    const uint8_t* r1 = s + (size_t)ty.s1 * pitch;
    for (int k = 0; k < c; ++k) {
        const int h0 = r0[tx.s0 * c + k] * tx.w0 + r0[tx.s1 * c + k] * tx.w1;
        const int h1 = r1[tx.s0 * c + k] * tx.w0 + r1[tx.s1 * c + k] * tx.w1;
        o[k] = (uint8_t)((((ty.w0 * (h0 >> 4)) >> 16) + ((ty.w1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
}

}  // namespace fd

extern "C" {

int fd_resize_dims(int src_h, int src_w, double fx, double fy, int* dst_h, int* dst_w) {
    FD_REQUIRE(dst_h && dst_w && src_h > 0 && src_w > 0 && fx > 0 && fy > 0, "fd_resize_dims: bad arguments");
    *dst_w = (int)std::nearbyint((double)src_w * fx);
    *dst_h = (int)std::nearbyint((double)src_h * fy);
    FD_REQUIRE(*dst_w > 0 && *dst_h > 0, "fd_resize_dims: empty destination");
    return FD_OK;
}

int fd_resize_u8(fd_ctx* ctx, const uint8_t* src, int batch, int src_h, int src_w, int channels, double fx, double fy,
                 uint8_t* dst, int dst_h, int dst_w, void* stream) {
    return fd_resize_cvt_u8(ctx, src, batch, src_h, src_w, channels, fx, fy, 0, dst, dst_h, dst_w, stream);
}

int fd_resize_cvt_u8(fd_ctx* ctx, const uint8_t* src, int batch, int src_h, int src_w, int channels, double fx,
                     double fy, int reverse_channels, uint8_t* dst, int dst_h, int dst_w, void* stream) {
    (void)ctx;
    FD_REQUIRE(src && dst && batch > 0 && channels > 0 && channels <= 4, "fd_resize_u8: bad arguments");
    int eh = 0, ew = 0;
    FD_TRY(fd_resize_dims(src_h, src_w, fx, fy, &eh, &ew));
    FD_REQUIRE(eh == dst_h && ew == dst_w, "fd_resize_u8: destination must be %d x %d for this scale (got %d x %d)", eh,
               ew, dst_h, dst_w);
    const double scale_x = 1.0 / fx, scale_y = 1.0 / fy;
    const int area2 = (scale_x == 2.0 && scale_y == 2.0 && src_w % 2 == 0 && src_h % 2 == 0) ? 1 : 0;
    fd::ProfScope scope(fd::as_stream(stream), "resize_linear_u8", 0.0,
                        (double)batch * channels * ((double)src_h * src_w + (double)dst_h * dst_w));
    dim3 grid(fd::ceil_div(dst_w, 64), fd::ceil_div(dst_h, 4), batch);
    hipLaunchKernelGGL(fd::resize_linear_u8_kernel, grid, dim3(256), 0, fd::as_stream(stream), src, src_h, src_w,
                       channels, scale_x, scale_y, dst, dst_h, dst_w, area2, reverse_channels ? 1 : 0);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

int fd_crop_resize_pad(fd_ctx* ctx, const uint8_t* frame, int height, int width, int channels, const double* boxes,
                       int box_stride, int n_boxes, int dst_h, int dst_w, uint8_t* tiles, int32_t* rects,
                       void* stream) {
    (void)ctx;
    FD_REQUIRE(n_boxes >= 0 && box_stride >= 4, "fd_crop_resize_pad: bad box layout");
    if (n_boxes == 0) return FD_OK;
    FD_REQUIRE(boxes && (tiles || rects) && (frame || !tiles) && height > 0 && width > 0 && channels > 0 &&
                   channels <= 4 && dst_h > 0 && dst_w > 0 && n_boxes <= 65535,
               "fd_crop_resize_pad: bad arguments");
    fd::ProfScope scope(fd::as_stream(stream), "crop_resize_pad", 0.0,
                        2.0 * (double)n_boxes * channels * (double)dst_h * dst_w);
    dim3 grid(tiles ? fd::ceil_div(dst_w, 64) : 1, tiles ? fd::ceil_div(dst_h, 4) : 1, n_boxes);
    hipLaunchKernelGGL(fd::crop_resize_pad_kernel, grid, dim3(256), 0, fd::as_stream(stream), frame, height, width,
                       channels, boxes, box_stride, tiles, dst_h, dst_w, rects);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

}  // extern "C"
