// 3x3 stride-1 convolution by Winograd's minimal filtering F(2x2, 3x3) in float32, for the long-K
// layers of the float32 path (the RPN's 3x3 convolution, reference: code/detection/symbol/
// faster_rcnn.py:85-87).  A 2x2 block of outputs is computed from its 4x4 input patch d and the
// filter g as
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A
// with 16 multiplies per channel pair instead of 36; summed over input channels the element-wise
// products become 16 independent GEMMs  M[xi] = V[xi] U[xi]^T  (rows: patches, K: input channels),
// which run on the fp32 matrix cores through the ordinary 1x1 convolution kernel, launched once
// for all 16.  Three kernels:
//     wino_input_kernel    x [B][H][W][C]            -> V [16][T_pad][C]        (HBM-bound)
//     batched GEMM         V, U [16][cout_pad][C]    -> M [16][T_pad][cout]     (matrix-bound, 4/9 of the direct FLOPs)
//     wino_output_kernel   M                         -> y [B][H][W][cout] (+ bias, ReLU)   (HBM-bound)
// The transform matrices hold only 0, +-1 (and +-1/2 in G, applied to the weights in double on the
// host), so the transforms are additions; measured against float64 the result carries about twice
// the rounding noise of a direct float32 convolution (rms 4.8e-7 against 2.1e-7 of the output rms at
// K = 9216).  F(4x4, 3x3) would need a quarter of the multiplies but ten times the noise: not used.
#include "fd_conv_split.h"
#include "fd_split.h"

#include <cmath>
#include <vector>

namespace fd {

// one thread: one patch, 4 consecutive channels
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int B, int H, int W, int C, int th,
                                                        int tw, long long T_pad, float* __restrict__ V) {
    const int c4n = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long t = idx / c4n;
    const int c = (int)(idx - t * c4n) * 4;
    if (t >= (long long)B * th * tw) return;
    const int tx = (int)(t % tw);
    const int ty = (int)((t / tw) % th);
    const int b = (int)(t / ((long long)tw * th));
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yy = y0 + i, xx = x0 + j;
            d[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                d[i][j] = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + yy) * W + xx) * C + c);
        }
    // B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3); then the same along the columns
    f32x4 r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[0][j] = d[0][j] - d[2][j];
        r[1][j] = d[1][j] + d[2][j];
        r[2][j] = d[2][j] - d[1][j];
        r[3][j] = d[1][j] - d[3][j];
    }
    const size_t plane = (size_t)T_pad * C;
    float* o = V + (size_t)t * C + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4*>(o + (i * 4 + 0) * plane) = r[i][0] - r[i][2];
        *reinterpret_cast<f32x4*>(o + (i * 4 + 1) * plane) = r[i][1] + r[i][2];
        *reinterpret_cast<f32x4*>(o + (i * 4 + 2) * plane) = r[i][2] - r[i][1];
        *reinterpret_cast<f32x4*>(o + (i * 4 + 3) * plane) = r[i][1] - r[i][3];
    }
}

// one thread: one patch, 4 consecutive output channels
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ M, int B, int H, int W, int Cout,
                                                         int th, int tw, long long T_pad, const float* __restrict__ bias,
                                                         int relu, float* __restrict__ y) {
    const int c4n = Cout >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long t = idx / c4n;
    const int c = (int)(idx - t * c4n) * 4;
    if (t >= (long long)B * th * tw) return;
    const int tx = (int)(t % tw);
    const int ty = (int)((t / tw) % th);
    const int b = (int)(t / ((long long)tw * th));
    const size_t plane = (size_t)T_pad * Cout;
    const float* p = M + (size_t)t * Cout + c;
    f32x4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const f32x4*>(p + (i * 4 + j) * plane);
    // A^T m: rows (m0 + m1 + m2, m1 - m2 - m3); then the same along the columns
    f32x4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[0][j] = m[0][j] + m[1][j] + m[2][j];
        s[1][j] = m[1][j] - m[2][j] - m[3][j];
    }
    const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int yy = 2 * ty + i;
        if (yy >= H) continue;
        f32x4 v[2];
        v[0] = s[i][0] + s[i][1] + s[i][2] + bv;
        v[1] = s[i][1] - s[i][2] - s[i][3] + bv;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int xx = 2 * tx + j;
            if (xx >= W) continue;
            f32x4 o = v[j];
            if (relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = fmaxf(o[q], 0.f);
            }
            *reinterpret_cast<f32x4*>(y + (((size_t)b * H + yy) * W + xx) * Cout + c) = o;
        }
    }
}

// U[xi = i * 4 + j][o][c] = (G g G^T)[i][j] of filter (o, c), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],
// computed in double from the OIHW weights (times out_scale[o] when given), rounded once to float32.
int wino_pack_weights(const float* w_oihw, int cout, int cin, const float* out_scale, std::vector<float>* packed,
                      int* cout_pad_out) {
    FD_REQUIRE(w_oihw && packed && cout > 0 && cin > 0, "wino_pack_weights: bad arguments");
    const int cout_pad = (cout + 127) / 128 * 128;
    packed->assign((size_t)16 * cout_pad * cin, 0.f);
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int o = 0; o < cout; ++o) {
        const double sc = out_scale ? (double)out_scale[o] : 1.0;
        for (int c = 0; c < cin; ++c) {
            const float* g = w_oihw + ((size_t)o * cin + c) * 9;
            double gg[4][3];
            for (int i = 0; i < 4; ++i)
                for (int k = 0; k < 3; ++k)
                    gg[i][k] = G[i][0] * g[0 * 3 + k] + G[i][1] * g[1 * 3 + k] + G[i][2] * g[2 * 3 + k];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double u = gg[i][0] * G[j][0] + gg[i][1] * G[j][1] + gg[i][2] * G[j][2];
                    (*packed)[((size_t)(i * 4 + j) * cout_pad + o) * cin + c] = (float)(u * sc);
                }
        }
    }
    *cout_pad_out = cout_pad;
    return FD_OK;
}

long long wino_patches_padded(int batch, int h, int w) {
    const long long T = (long long)batch * ((h + 1) / 2) * ((w + 1) / 2);
    return (T + 127) / 128 * 128;
}

bool wino_fits(const WinoConvParams& p, int batch, int h, int w) {
    const long long T_pad = wino_patches_padded(batch, h, w);
    return p.u && p.cin % 32 == 0 && p.cout % 8 == 0 && 16 * T_pad * p.cin * 4 < (1ll << 31) &&
           16 * T_pad * (long long)p.cout * 4 < (1ll << 33) && 16 * T_pad < (1ll << 29) &&
           16ll * p.cout_pad * p.cin * 4 < (1ll << 31);
}

int wino_conv_forward(const WinoConvParams& p, const float* x, int batch, int h, int w, float* v_ws, float* m_ws,
                      float* y, hipStream_t stream) {
    FD_REQUIRE(x && y && v_ws && m_ws && wino_fits(p, batch, h, w), "winograd convolution: bad arguments or sizes");
    const int th = (h + 1) / 2, tw = (w + 1) / 2;
    const long long T = (long long)batch * th * tw, T_pad = wino_patches_padded(batch, h, w);
    {
        ProfScope scope(stream, "wino_input", 0.0, 4.0 * p.cin * ((double)batch * h * w + 16.0 * T));
        const long long items = T * (p.cin / 4);
        hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)ceil_div(items, 256)), dim3(256), 0, stream, x, batch, h, w,
                           p.cin, th, tw, T_pad, v_ws);
        FD_LAUNCH_CHECK();
    }
    FD_TRY(gemm_f32_batched(v_ws, p.u, 16, (int)T_pad, p.cin, p.cout, p.cout_pad, m_ws, stream));
    {
        ProfScope scope(stream, "wino_output", 0.0, 4.0 * p.cout * (16.0 * T + (double)batch * h * w));
        const long long items = T * (p.cout / 4);
        hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)ceil_div(items, 256)), dim3(256), 0, stream, m_ws, batch, h,
                           w, p.cout, th, tw, T_pad, p.bias, p.relu, y);
        FD_LAUNCH_CHECK();
    }
    return FD_OK;
}

}  // namespace fd

extern "C" {

int fd_conv2d_winograd(fd_ctx* ctx, const float* x_nhwc, int batch, int height, int width, int cin, const float* w_oihw,
                       int cout, const float* bias, int relu, float* y_nhwc, void* stream) {
    FD_REQUIRE(ctx && x_nhwc && w_oihw && y_nhwc, "fd_conv2d_winograd: NULL argument");
    FD_REQUIRE(batch > 0 && height > 0 && width > 0 && cin > 0 && cin % 32 == 0 && cout > 0 && cout % 8 == 0,
               "fd_conv2d_winograd: cin must be a multiple of 32 and cout of 8 (got %d, %d)", cin, cout);
    hipStream_t st = fd::as_stream(stream);
    // weights are transformed on the host (blocking read-back: this entry point is for tests and one-off use;
    // the detector transforms its weights once, at creation)
    const size_t nw = (size_t)cout * cin * 9;
    std::vector<float> host(nw), packed;
    FD_HIP(hipStreamSynchronize(st));
    FD_HIP(hipMemcpy(host.data(), w_oihw, nw * sizeof(float), hipMemcpyDeviceToHost));
    fd::WinoConvParams p;
    p.cin = cin; p.cout = cout; p.bias = bias; p.relu = relu;
    FD_TRY(fd::wino_pack_weights(host.data(), cout, cin, nullptr, &packed, &p.cout_pad));
    void *u = nullptr, *v = nullptr, *m = nullptr;
    FD_TRY(fd::scratch(ctx, fd::SLOT_MISC0, packed.size() * sizeof(float), &u));
    FD_HIP(hipMemcpyAsync(u, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, st));
    FD_HIP(hipStreamSynchronize(st));               // `packed` goes out of scope below
    p.u = static_cast<const float*>(u);
    FD_REQUIRE(fd::wino_fits(p, batch, height, width),
               "fd_conv2d_winograd: problem beyond the 32-bit reach of the kernels (batch %d, %d x %d, %d -> %d)", batch,
               height, width, cin, cout);
    const size_t tp = (size_t)fd::wino_patches_padded(batch, height, width);
    FD_TRY(fd::scratch(ctx, fd::SLOT_MISC1, 16 * tp * cin * sizeof(float), &v));
    FD_TRY(fd::scratch(ctx, fd::SLOT_MISC2, 16 * tp * cout * sizeof(float), &m));
    return fd::wino_conv_forward(p, x_nhwc, batch, height, width, static_cast<float*>(v), static_cast<float*>(m),
                                 y_nhwc, st);
}

}  // extern "C"
