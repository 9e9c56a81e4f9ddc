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

#include <algorithm>
#include <cmath>
#include <cstdlib>

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
// The same resize for 3-channel frames with the source staged through LDS.  A workgroup owns a
// tile of RZ_TY x RZ_TX destination pixels.  The source rectangle the tile's taps touch (about
// RZ_TY / fy + 2 rows of RZ_TX / fx + 2 pixels) is read with 16-byte buffer loads -- whole aligned
// chunks of every source row, consecutive lanes on consecutive chunks, so each row segment is one
// coalesced burst and every source byte is fetched once per tile instead of once per tap and channel
// with single-byte loads -- into LDS rows of `pitch` bytes; chunks past the end of the image return
// zero through the buffer's range check.  The column and row taps of the tile are computed once
// (one thread each) into LDS tables.  Then a thread produces 4 neighbouring destination pixels:
// per pixel and source row three aligned LDS dwords and two byte-align operations give the 6 bytes of
// the two tap pixels; the 12 output bytes leave as one 12-byte store (consecutive lanes write
// consecutive addresses).  Arithmetic is that of resize_linear_u8_kernel, bit for bit.
constexpr int RZ_TX = 128, RZ_TY = 8;

template <bool AREA2, bool REVERSE>
__global__ __launch_bounds__(256) void resize3_lds_kernel(const uint8_t* __restrict__ src, int sh, int sw, double scale_x,
                                                         double scale_y, uint8_t* __restrict__ dst, int dh, int dw,
                                                         int pitch, int max_rows) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) uint8_t rz_lds[];
    int2* coltab = reinterpret_cast<int2*>(rz_lds + (size_t)max_rows * pitch);   // (first tap column, w0 | w1 << 16)
    int4* rowtab = reinterpret_cast<int4*>(coltab + RZ_TX);                       // (tap rows, b0 << 16, b1 << 16)
    const int tid = threadIdx.x, b = blockIdx.z;
    const int dy0 = blockIdx.y * RZ_TY, dx0 = blockIdx.x * RZ_TX;
    const int ny = min(RZ_TY, dh - dy0), nx = min(RZ_TX, dw - dx0);
    // one tap per thread: columns of the tile, then its rows
    if (tid < nx) {
        if (AREA2) {
            coltab[tid] = make_int2(2 * (dx0 + tid), 0);
        } else {
            const Tap2 t = linear_tap(dx0 + tid, scale_x, sw);
            coltab[tid] = make_int2(t.s0, t.w0 | (t.w1 << 16));
        }
    } else if (tid >= RZ_TX && tid - RZ_TX < ny) {
        const int r = tid - RZ_TX;
        if (AREA2) {
            rowtab[r] = make_int4(2 * (dy0 + r), 2 * (dy0 + r) + 1, 0, 0);
        } else {
            const Tap2 t = linear_tap(dy0 + r, scale_y, sh);
            rowtab[r] = make_int4(t.s0, t.s1, t.w0 << 16, t.w1 << 16);
        }
    }
    __syncthreads();
    // taps are monotonic: the tile's source rectangle starts at the first column / row tap
    const int c_lo = coltab[0].x, r_lo = rowtab[0].x;
    const int nrows = min(rowtab[ny - 1].y - r_lo + 1, max_rows);
    // the image as a buffer whose base is 16-byte aligned (the bytes before the image inside its first
    // chunk belong to the same allocation)
    const uint8_t* img = src + (size_t)b * sh * sw * 3;
    const unsigned mis = (unsigned)(reinterpret_cast<uintptr_t>(img) & 15u);
    // The range ends at the end of the image's last 16-byte chunk: a 16-byte load is dropped as a whole
    // when it crosses the end of the range, and the bytes up to the chunk boundary lie in the same
    // allocation granule as the image's last byte.
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(img - mis), 0, (unsigned)(((size_t)sh * sw * 3 + mis + 15) & ~(size_t)15), 0x00020000);
    const int cpr = pitch >> 4;                       // 16-byte chunks per staged row
    const unsigned origin = mis + (unsigned)(r_lo * sw + c_lo) * 3u;     // byte offset of (r_lo, c_lo) from the buffer base
    const unsigned row_bytes = (unsigned)sw * 3u;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    for (int i = tid; i < nrows * cpr; i += 256) {
        const int r = i / cpr, ch = i - r * cpr;
        const unsigned start = origin + (unsigned)r * row_bytes;
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (start & ~15u) + (unsigned)ch * 16u, 0, 0);
        *reinterpret_cast<u32x4_t*>(rz_lds + (size_t)r * pitch + ch * 16) = v;
    }
    __syncthreads();
    constexpr int GX = RZ_TX / 4;
    for (int item = tid; item < ny * GX; item += 256) {
        const int ry = item / GX, gx = (item - ry * GX) * 4;
        if (gx >= nx) continue;
        const int4 rt = rowtab[ry];
        const int r0 = rt.x - r_lo, r1 = rt.y - r_lo;
        // byte position of source column c_lo inside each of the two staged rows (their alignment differs)
        const unsigned base0 = (unsigned)r0 * (unsigned)pitch + ((origin + (unsigned)r0 * row_bytes) & 15u);
        const unsigned base1 = (unsigned)r1 * (unsigned)pitch + ((origin + (unsigned)r1 * row_bytes) & 15u);
        unsigned vout[12];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const int2 ct = coltab[min(gx + px, nx - 1)];
            const unsigned coff = (unsigned)(ct.x - c_lo) * 3u;
            unsigned lo[2], hi[2];                     // bytes 0..3 and 4..7 from the first tap pixel on, per row
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned a = (r ? base1 : base0) + coff;
                const unsigned* q = reinterpret_cast<const unsigned*>(rz_lds + (a & ~3u));
                const unsigned d0 = q[0], d1 = q[1], d2 = q[2];
                lo[r] = __builtin_amdgcn_alignbyte(d1, d0, a & 3u);
                hi[r] = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // (byte k) | (byte k + 3) << 16: channel k of the two tap pixels as a pair of 16-bit values
                const unsigned sel = 0x0c000c00u | (unsigned)k | ((unsigned)(k + 3) << 16);
                const unsigned p0 = __builtin_amdgcn_perm(hi[0], lo[0], sel);
                const unsigned p1 = __builtin_amdgcn_perm(hi[1], lo[1], sel);
                unsigned v;
                if (AREA2) {
                    v = ((p0 & 0xffffu) + (p0 >> 16) + (p1 & 0xffffu) + (p1 >> 16) + 2u) >> 2;
                } else {
                    const unsigned w0 = (unsigned)ct.y & 0xffffu, w1 = (unsigned)ct.y >> 16;
                    const unsigned h0 = (p0 & 0xffffu) * w0 + (p0 >> 16) * w1;
                    const unsigned h1 = (p1 & 0xffffu) * w0 + (p1 >> 16) * w1;
                    // (b * (h >> 4)) >> 16 with b < 2^12, h >> 4 < 2^15: the high word of (b << 16) * (h >> 4)
                    v = (__umulhi((unsigned)rt.z, h0 >> 4) + __umulhi((unsigned)rt.w, h1 >> 4) + 2u) >> 2;
                }
                vout[px * 3 + (REVERSE ? 2 - k : k)] = v;
            }
        }
        uint8_t* o = dst + (((size_t)b * dh + dy0 + ry) * dw + dx0 + gx) * 3;
        if (gx + 4 <= nx && (reinterpret_cast<uintptr_t>(o) & 3u) == 0) {
            typedef unsigned u32x3_t __attribute__((ext_vector_type(3)));
            u32x3_t w;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                w[j] = vout[4 * j] | (vout[4 * j + 1] << 8) | (vout[4 * j + 2] << 16) | (vout[4 * j + 3] << 24);
            *reinterpret_cast<u32x3_t*>(o) = w;
        } else {
            const int nb = 3 * min(4, nx - gx);
#pragma unroll
            for (int j = 0; j < 12; ++j)
                if (j < nb) o[j] = (uint8_t)vout[j];
        }
    }
#endif
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
    // 3-channel frames take the LDS-staged kernel whenever a tile's source rectangle fits in LDS (any
    // scale from enlargements down to about 1/7); the per-pixel kernel is the general fallback.
    static const bool allow_lds = !(getenv("FD_RESIZE_LDS") && atoi(getenv("FD_RESIZE_LDS")) == 0);
    if (channels == 3 && allow_lds && (size_t)src_h * src_w * 3 < (1ull << 31) && batch <= 65535) {
        const int max_rows = std::min(src_h, (int)std::ceil(fd::RZ_TY * scale_y) + 3);
        const int max_cols = std::min(src_w, (int)std::ceil(fd::RZ_TX * scale_x) + 3);
        const int pitch = ((max_cols * 3 + 15 + 8) + 15) / 16 * 16;     // worst lead + the taps' 8-byte window, in whole chunks
        const size_t smem = (size_t)max_rows * pitch + fd::RZ_TX * sizeof(int2) + fd::RZ_TY * sizeof(int4);
        if (smem <= 64 * 1024) {
            dim3 grid(fd::ceil_div(dst_w, fd::RZ_TX), fd::ceil_div(dst_h, fd::RZ_TY), batch);
            const hipStream_t st = fd::as_stream(stream);
#define FD_RZ(A2, REV)                                                                                          \
    hipLaunchKernelGGL((fd::resize3_lds_kernel<A2, REV>), grid, dim3(256), smem, st, src, src_h, src_w, scale_x, \
                       scale_y, dst, dst_h, dst_w, pitch, max_rows)
            if (area2) { if (reverse_channels) FD_RZ(true, true); else FD_RZ(true, false); }
            else { if (reverse_channels) FD_RZ(false, true); else FD_RZ(false, false); }
#undef FD_RZ
            FD_LAUNCH_CHECK();
            return FD_OK;
        }
    }
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
