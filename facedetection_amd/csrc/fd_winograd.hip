// 3x3 stride-1 convolution by Winograd's minimal filtering, for the wide 3x3 layers of the network (the RPN's
// 3x3 convolution, reference: code/detection/symbol/faster_rcnn.py:85-87, and the stride-1 3x3 layers of the
// bottleneck units, faster_rcnn.py:41-43).  A tile x tile block of outputs is computed from its
// (tile + 2) x (tile + 2) input patch d and the filter g as
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A
// with n^2 = (tile + 2)^2 multiplies per channel pair instead of 9 tile^2; summed over input channels the
// element-wise products become n^2 independent GEMMs  M[xi] = V[xi] U[xi]^T  (rows: patches, K: input channels),
// which run on the matrix cores through the ordinary 1x1 convolution kernel, launched once for all of them.
// Three kernels:
//     input transform      x [B][H][W][C]              -> V [n^2][T_pad][C]        (HBM-bound)
//     batched GEMM         V, U [n^2][cout_pad][C]     -> M [n^2][T_pad][cout]     (matrix-bound)
//     output transform     M                           -> y [B][H][W][cout] (+ bias, ReLU)   (HBM-bound)
// F(2x2, 3x3) (tile = 2): 16 GEMMs, 4/9 of the direct FLOPs, Winograd-domain tensors 4x the plain ones.  The
// transform matrices hold only 0, +-1 (and +-1/2 in G, applied to the weights in double on the host), so the
// transforms are additions.
// F(4x4, 3x3) (tile = 4, the detector's default): 36 GEMMs over a quarter as many patches, 1/4 of the direct
// FLOPs, Winograd-domain tensors 2.25x the plain ones.  With the textbook interpolation points 0, +-1, +-2 its
// transforms amplify rounding errors about six times more than F(2x2)'s; the points used here, 0, +-5/8, +-3/2
// (all transform entries stay exact binary fractions), bring that to about three times (simulated with float32
// transforms and sequential float32 accumulation over 1024 channels: rms error 2.5e-6 of the output rms against
// 5.7e-6 for the textbook points, 0.9e-6 for F(2x2) and 1.5e-6 for the direct sum over 9216 products; measured on
// the GPU on the RPN layer's shape with non-negative inputs: 2.3e-6, 0.8e-6 for F(2x2), 1.2e-6 direct).
// Float32 form (FD_PRECISION_F32) and split-precision form (FD_PRECISION_F16X3, tile 4 only: activations and
// the Winograd-domain input as fp16 hi+lo pairs, transforms in float32, GEMMs on the f16 matrix cores).
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

// ---------------------------------------------------------------------------------
// F(4x4, 3x3) with the interpolation points 0, 5/8, -5/8, 3/2, -3/2 and infinity (Cook-Toom construction:
// row j of B^T holds the coefficients of the product of (x - p_l) over the finite points l != j, the last row
// that of the product over all of them; A^T is the Vandermonde matrix of the points; G the same scaled by
// 1 / prod_{l != j} (p_j - p_l), computed on the host).
constexpr float W4_BT[6][6] = {
    {0.87890625f, 0.0f, -2.640625f, 0.0f, 1.0f, 0.0f},
    {0.0f, -1.40625f, -2.25f, 0.625f, 1.0f, 0.0f},
    {0.0f, 1.40625f, -2.25f, -0.625f, 1.0f, 0.0f},
    {0.0f, -0.5859375f, -0.390625f, 1.5f, 1.0f, 0.0f},
    {0.0f, 0.5859375f, -0.390625f, -1.5f, 1.0f, 0.0f},
    {0.0f, 0.87890625f, 0.0f, -2.640625f, 0.0f, 1.0f},
};
constexpr float W4_AT[4][6] = {
    {1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 0.0f},
    {0.0f, 0.625f, -0.625f, 1.5f, -1.5f, 0.0f},
    {0.0f, 0.390625f, 0.390625f, 2.25f, 2.25f, 0.0f},
    {0.0f, 0.244140625f, -0.244140625f, 3.375f, -3.375f, 1.0f},
};

// sum_k coef[k] * v[k] over the non-zero coefficients (the table is a compile-time constant: after unrolling
// only the multiplies that exist remain)
template <int N>
__device__ __forceinline__ f32x4 wino_dot(const float (&coef)[N], const f32x4 (&v)[N]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bool first = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (coef[k] == 0.f) continue;
        if (first) acc = coef[k] == 1.f ? v[k] : v[k] * coef[k];
        else acc += v[k] * coef[k];
        first = false;
    }
    return acc;
}

// one thread: one 6x6 patch (4x4 outputs), 4 consecutive channels
__global__ __launch_bounds__(256) void wino4_input_kernel(const float* __restrict__ x, int B, int H, int W, int C, int th,
                                                         int tw, long long T_pad, float* __restrict__ V) {
    const int c4n = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long t = idx / c4n;
    const int c = (int)(idx - t * c4n) * 4;
    if (t >= (long long)B * th * tw) return;
    const int tx = (int)(t % tw);
    const int ty = (int)((t / tw) % th);
    const int b = (int)(t / ((long long)tw * th));
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    f32x4 r[6][6];                                   // B^T d, column by column
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x4 d[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int yy = y0 + i, xx = x0 + j;
            d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                d[i] = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + yy) * W + xx) * C + c);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i][j] = wino_dot(W4_BT[i], d);
    }
    const size_t plane = (size_t)T_pad * C;
    float* o = V + (size_t)t * C + c;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(o + (i * 6 + j) * plane) = wino_dot(W4_BT[j], r[i]);
}

// one thread: one patch (4x4 outputs), 4 consecutive output channels
__global__ __launch_bounds__(256) void wino4_output_kernel(const float* __restrict__ M, int B, int H, int W, int Cout,
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
    f32x4 s[4][6];                                   // A^T m, column by column
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x4 m[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const f32x4*>(p + (i * 6 + j) * plane);
#pragma unroll
        for (int a = 0; a < 4; ++a) s[a][j] = wino_dot(W4_AT[a], m);
    }
    const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = 4 * ty + a;
        if (yy >= H) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xx = 4 * tx + q;
            if (xx >= W) continue;
            f32x4 o = wino_dot(W4_AT[q], s[a]) + bv;
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
            }
            *reinterpret_cast<f32x4*>(y + (((size_t)b * H + yy) * W + xx) * Cout + c) = o;
        }
    }
}

// The same two transforms for the split-precision path (fd_split.h): activations arrive and leave as fp16 hi+lo
// pairs (a value is hi + lo, 22 significant bits), the transforms themselves are float32 arithmetic, and the
// Winograd-domain input V is written as pairs again, for the f16 matrix cores.  |V| can reach 20 times the
// largest activation of the patch (the row sums of |B^T| are up to 4.5); values beyond the fp16 range are
// clamped and counted like everywhere else in that path.
// one thread: one 6x6 patch, 4 consecutive channels (8 bytes of hi halves and 8 of lo halves per pixel)
__global__ __launch_bounds__(256) void wino4_input_split_kernel(const uint8_t* __restrict__ x, int B, int H, int W, int C,
                                                               int th, int tw, long long T_pad, uint8_t* __restrict__ V,
                                                               unsigned* __restrict__ clamped) {
    const int c4n = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long t = idx / c4n;
    const int c = (int)(idx - t * c4n) * 4;
    if (t >= (long long)B * th * tw) return;
    const int tx = (int)(t % tw);
    const int ty = (int)((t / tw) % th);
    const int b = (int)(t / ((long long)tw * th));
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    const size_t pitch = (size_t)C * 4, coff = split_offset(c);
    f32x4 r[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x4 d[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int yy = y0 + i, xx = x0 + j;
            d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                const uint8_t* q = x + (((size_t)b * H + yy) * W + xx) * pitch + coff;
                const f16x4 hi = *reinterpret_cast<const f16x4*>(q), lo = *reinterpret_cast<const f16x4*>(q + 64);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[i][e] = (float)hi[e] + (float)lo[e];
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i][j] = wino_dot(W4_BT[i], d);
    }
    const size_t plane = (size_t)T_pad * pitch;
    uint8_t* o = V + (size_t)t * pitch + coff;
    unsigned bad = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const f32x4 v = wino_dot(W4_BT[j], r[i]);
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 hh, ll;
                split_f32(v[e], hh, ll, bad);
                hi[e] = hh; lo[e] = ll;
            }
            *reinterpret_cast<f16x4*>(o + (i * 6 + j) * plane) = hi;
            *reinterpret_cast<f16x4*>(o + (i * 6 + j) * plane + 64) = lo;
        }
    split_report(clamped, bad);
}

// one thread: one patch (4x4 outputs), 4 consecutive output channels; M is float32, y a split tensor
__global__ __launch_bounds__(256) void wino4_output_split_kernel(const float* __restrict__ M, int B, int H, int W, int Cout,
                                                                int th, int tw, long long T_pad,
                                                                const float* __restrict__ bias, int relu,
                                                                uint8_t* __restrict__ y, unsigned* __restrict__ clamped) {
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
    f32x4 s[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x4 m[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const f32x4*>(p + (i * 6 + j) * plane);
#pragma unroll
        for (int a = 0; a < 4; ++a) s[a][j] = wino_dot(W4_AT[a], m);
    }
    const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    const size_t pitch = (size_t)Cout * 4, coff = split_offset(c);
    unsigned bad = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = 4 * ty + a;
        if (yy >= H) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xx = 4 * tx + q;
            if (xx >= W) continue;
            const f32x4 o = wino_dot(W4_AT[q], s[a]) + bv;
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 hh, ll;
                if (relu) split_f32_relu(o[e], hh, ll, bad);
                else split_f32(o[e], hh, ll, bad);
                hi[e] = hh; lo[e] = ll;
            }
            uint8_t* dst = y + (((size_t)b * H + yy) * W + xx) * pitch + coff;
            *reinterpret_cast<f16x4*>(dst) = hi;
            *reinterpret_cast<f16x4*>(dst + 64) = lo;
        }
    }
    split_report(clamped, bad);
}

// U[xi = i * n + j][o][c] = (G g G^T)[i][j] of filter (o, c), n = tile + 2, computed in double from the OIHW
// weights (times out_scale[o] when given) and rounded once to float32.
// tile 2: G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]];  tile 4: row j = (1, p_j, p_j^2) / prod_{l != j} (p_j - p_l).
int wino_pack_weights(const float* w_oihw, int cout, int cin, const float* out_scale, int tile,
                      std::vector<float>* packed, int* cout_pad_out) {
    FD_REQUIRE(w_oihw && packed && cout > 0 && cin > 0 && (tile == 2 || tile == 4), "wino_pack_weights: bad arguments");
    const int cout_pad = (cout + 127) / 128 * 128;
    const int n = tile + 2;
    packed->assign((size_t)n * n * cout_pad * cin, 0.f);
    double G[6][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}, {0, 0, 0}, {0, 0, 0}};
    if (tile == 4) {
        const double pts[5] = {0.0, 0.625, -0.625, 1.5, -1.5};
        for (int j = 0; j < 5; ++j) {
            double nj = 1.0;
            for (int l = 0; l < 5; ++l)
                if (l != j) nj *= pts[j] - pts[l];
            G[j][0] = 1.0 / nj; G[j][1] = pts[j] / nj; G[j][2] = pts[j] * pts[j] / nj;
        }
        G[5][0] = 0.0; G[5][1] = 0.0; G[5][2] = 1.0;
    }
    for (int o = 0; o < cout; ++o) {
        const double sc = out_scale ? (double)out_scale[o] : 1.0;
        for (int c = 0; c < cin; ++c) {
            const float* g = w_oihw + ((size_t)o * cin + c) * 9;
            double gg[6][3];
            for (int i = 0; i < n; ++i)
                for (int k = 0; k < 3; ++k)
                    gg[i][k] = G[i][0] * g[0 * 3 + k] + G[i][1] * g[1 * 3 + k] + G[i][2] * g[2 * 3 + k];
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    const double u = gg[i][0] * G[j][0] + gg[i][1] * G[j][1] + gg[i][2] * G[j][2];
                    (*packed)[((size_t)(i * n + j) * cout_pad + o) * cin + c] = (float)(u * sc);
                }
        }
    }
    *cout_pad_out = cout_pad;
    return FD_OK;
}

long long wino_patches_padded(int batch, int h, int w, int tile, int align) {
    const long long T = (long long)batch * ((h + tile - 1) / tile) * ((w + tile - 1) / tile);
    return (T + align - 1) / align * align;
}

static ConvShape wino_gemm_shape(int cin, int cout) {
    ConvShape s;
    s.cin = cin; s.cout = cout; s.kh = 1; s.kw = 1; s.kwp = 1; s.stride = 1; s.pad = 0;
    return s;
}

size_t wino_split_weight_floats(int cin, int cout) { return wino_gemm_shape(cin, cout).packed_floats(); }

int wino_pack_weights_split(const float* u_dev, int cin, int cout, int cout_pad, float wscale, void* packed,
                            hipStream_t stream) {
    FD_REQUIRE(u_dev && packed && cin % 32 == 0 && cout > 0 && cout <= cout_pad, "wino_pack_weights_split: bad arguments");
    const ConvShape s = wino_gemm_shape(cin, cout);
    for (int xi = 0; xi < 36; ++xi)
        FD_TRY(pack_conv_weights_split(u_dev + (size_t)xi * cout_pad * cin, cin, s, nullptr, wscale,
                                       static_cast<float*>(packed) + (size_t)xi * s.packed_floats(), stream));
    return FD_OK;
}

bool wino_fits_split(const WinoConvParams& p, int batch, int h, int w) {
    const long long T_pad = wino_patches_padded(batch, h, w, 4, 256);
    return p.u_split && p.tile == 4 && p.cin % 32 == 0 && p.cout % 32 == 0 && 36 * T_pad * p.cin * 4 < (1ll << 31) &&
           36 * T_pad * (long long)p.cout * 4 < (1ll << 33) && 36 * T_pad < (1ll << 29) &&
           36ll * p.u_split_bytes < (1ll << 31);
}

int wino_conv_forward_split(const WinoConvParams& p, const void* x_split, int batch, int h, int w, void* v_ws, float* m_ws,
                            void* y_split, unsigned* clamped, hipStream_t stream) {
    FD_REQUIRE(x_split && y_split && v_ws && m_ws && wino_fits_split(p, batch, h, w),
               "winograd convolution (split precision): bad arguments or sizes");
    const int th = (h + 3) / 4, tw = (w + 3) / 4;
    const long long T = (long long)batch * th * tw, T_pad = wino_patches_padded(batch, h, w, 4, 256);
    {
        ProfScope scope(stream, "wino_input", 0.0, 4.0 * p.cin * ((double)batch * h * w + 36.0 * T));
        const long long items = T * (p.cin / 4);
        hipLaunchKernelGGL(wino4_input_split_kernel, dim3((unsigned)ceil_div(items, 256)), dim3(256), 0, stream,
                           static_cast<const uint8_t*>(x_split), batch, h, w, p.cin, th, tw, T_pad,
                           static_cast<uint8_t*>(v_ws), clamped);
        FD_LAUNCH_CHECK();
    }
    FD_TRY(gemm_split_batched(v_ws, p.u_split, p.u_split_bytes, p.u_inv_scale, 36, (int)T_pad, p.cin, p.cout, m_ws, stream));
    {
        ProfScope scope(stream, "wino_output", 0.0, 4.0 * p.cout * (36.0 * T + (double)batch * h * w));
        const long long items = T * (p.cout / 4);
        hipLaunchKernelGGL(wino4_output_split_kernel, dim3((unsigned)ceil_div(items, 256)), dim3(256), 0, stream, m_ws, batch,
                           h, w, p.cout, th, tw, T_pad, p.bias, p.relu, static_cast<uint8_t*>(y_split), clamped);
        FD_LAUNCH_CHECK();
    }
    return FD_OK;
}

bool wino_fits(const WinoConvParams& p, int batch, int h, int w) {
    const long long T_pad = wino_patches_padded(batch, h, w, p.tile);
    const long long nb = (p.tile + 2) * (p.tile + 2);
    return p.u && (p.tile == 2 || p.tile == 4) && p.cin % 32 == 0 && p.cout % 8 == 0 && nb * T_pad * p.cin * 4 < (1ll << 31) &&
           nb * T_pad * (long long)p.cout * 4 < (1ll << 33) && nb * T_pad < (1ll << 29) &&
           nb * p.cout_pad * p.cin * 4 < (1ll << 31);
}

int wino_conv_forward(const WinoConvParams& p, const float* x, int batch, int h, int w, float* v_ws, float* m_ws,
                      float* y, hipStream_t stream, bool b3, int b3_balance) {
    FD_REQUIRE(x && y && v_ws && m_ws && wino_fits(p, batch, h, w), "winograd convolution: bad arguments or sizes");
    const int tile = p.tile, nb = (tile + 2) * (tile + 2);
    const int th = (h + tile - 1) / tile, tw = (w + tile - 1) / tile;
    const long long T = (long long)batch * th * tw, T_pad = wino_patches_padded(batch, h, w, tile);
    {
        ProfScope scope(stream, "wino_input", 0.0, 4.0 * p.cin * ((double)batch * h * w + (double)nb * T));
        const long long items = T * (p.cin / 4);
        if (tile == 4)
            hipLaunchKernelGGL(wino4_input_kernel, dim3((unsigned)ceil_div(items, 256)), dim3(256), 0, stream, x, batch, h,
                               w, p.cin, th, tw, T_pad, v_ws);
        else
            hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)ceil_div(items, 256)), dim3(256), 0, stream, x, batch, h,
                               w, p.cin, th, tw, T_pad, v_ws);
        FD_LAUNCH_CHECK();
    }
    FD_TRY(gemm_f32_batched(v_ws, p.u, nb, (int)T_pad, p.cin, p.cout, p.cout_pad, m_ws, stream, b3, b3_balance));
    {
        ProfScope scope(stream, "wino_output", 0.0, 4.0 * p.cout * ((double)nb * T + (double)batch * h * w));
        const long long items = T * (p.cout / 4);
        if (tile == 4)
            hipLaunchKernelGGL(wino4_output_kernel, dim3((unsigned)ceil_div(items, 256)), dim3(256), 0, stream, m_ws, batch,
                               h, w, p.cout, th, tw, T_pad, p.bias, p.relu, y);
        else
            hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)ceil_div(items, 256)), dim3(256), 0, stream, m_ws, batch,
                               h, w, p.cout, th, tw, T_pad, p.bias, p.relu, y);
        FD_LAUNCH_CHECK();
    }
    return FD_OK;
}

}  // namespace fd

extern "C" {

int fd_conv2d_winograd(fd_ctx* ctx, const float* x_nhwc, int batch, int height, int width, int cin, const float* w_oihw,
                       int cout, const float* bias, int relu, int tile, float* y_nhwc, void* stream) {
    FD_REQUIRE(ctx && x_nhwc && w_oihw && y_nhwc, "fd_conv2d_winograd: NULL argument");
    FD_REQUIRE(tile == 2 || tile == 4, "fd_conv2d_winograd: tile must be 2 (F(2x2,3x3)) or 4 (F(4x4,3x3)), got %d", tile);
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
    p.cin = cin; p.cout = cout; p.bias = bias; p.relu = relu; p.tile = tile;
    FD_TRY(fd::wino_pack_weights(host.data(), cout, cin, nullptr, tile, &packed, &p.cout_pad));
    void *u = nullptr, *v = nullptr, *m = nullptr;
    FD_TRY(fd::scratch(ctx, fd::SLOT_MISC0, packed.size() * sizeof(float), &u));
    FD_HIP(hipMemcpyAsync(u, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, st));
    FD_HIP(hipStreamSynchronize(st));               // `packed` goes out of scope below
    p.u = static_cast<const float*>(u);
    FD_REQUIRE(fd::wino_fits(p, batch, height, width),
               "fd_conv2d_winograd: problem beyond the 32-bit reach of the kernels (batch %d, %d x %d, %d -> %d)", batch,
               height, width, cin, cout);
    const size_t tp = (size_t)fd::wino_patches_padded(batch, height, width, tile), nb = (size_t)(tile + 2) * (tile + 2);
    FD_TRY(fd::scratch(ctx, fd::SLOT_MISC1, nb * tp * cin * sizeof(float), &v));
    FD_TRY(fd::scratch(ctx, fd::SLOT_MISC2, nb * tp * cout * sizeof(float), &m));
    return fd::wino_conv_forward(p, x_nhwc, batch, height, width, static_cast<float*>(v), static_cast<float*>(m),
                                 y_nhwc, st);
}

}  // extern "C"
