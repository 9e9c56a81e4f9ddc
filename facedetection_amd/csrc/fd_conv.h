// Convolution as an implicit GEMM on the fp32 matrix cores, plus the small layers of the
// network (pooling, input staging, classifier head).  Activations are channels-last
// (NHWC) float32; weights are packed [Cout_pad][KH][KWp][Cin_p] so that the GEMM's K
// dimension (tap-major, channel-minor) is contiguous for both operands.
#pragma once
#include "fd_common.h"

namespace fd {

constexpr int CONV_COUT_ALIGN = 128;   // packed weight rows are padded to a multiple of this

struct ConvShape {
    int cin = 0;      // channels of the stored input tensor (4 for the staged image)
    int cout = 0;
    int kh = 1, kw = 1;
    int kwp = 1;      // kernel width as packed (8 for the 7x7 stem, else kw)
    int stride = 1, pad = 0;
    int cout_pad() const { return (cout + CONV_COUT_ALIGN - 1) / CONV_COUT_ALIGN * CONV_COUT_ALIGN; }
    int k() const { return kh * kwp * cin; }
    size_t packed_floats() const { return (size_t)cout_pad() * k(); }
    int out_h(int h) const { return (h + 2 * pad - kh) / stride + 1; }
    int out_w(int w) const { return (w + 2 * pad - kw) / stride + 1; }
};

struct ConvParams {
    ConvShape s;
    const float* w = nullptr;          // packed weights (device)
    const float* bias = nullptr;       // [cout] added to the output, or NULL
    const float* pre_scale = nullptr;  // [cin]: input is relu(x * pre_scale + pre_shift), or NULL
    const float* pre_shift = nullptr;
    int relu = 0;                      // relu on the output (after bias)
};

// Packs OIHW weights [cout][cin_w][kh][kw] (device) into the GEMM layout, multiplying row o
// by out_scale[o] when given (BatchNorm folding).  cin_w may be smaller than s.cin (stem: 3 -> 4).
int pack_conv_weights(const float* w_oihw, int cin_w, const ConvShape& s, const float* out_scale, float* packed,
                      hipStream_t stream);

// y[B][Ho][Wo][cout] = conv(x[B][H][W][cin]) (+ bias) (relu) (+ residual[B][Ho][Wo][cout]).
int conv_forward(const ConvParams& p, const float* x, int batch, int h, int w, const float* residual, float* y,
                 hipStream_t stream);

// 3x3 / stride 2 / pad 1 max pooling, channels-last (channels % 4 == 0).
int maxpool_3x3s2(const float* x, int batch, int h, int w, int c, float* y, hipStream_t stream);

// uint8 HWC frames [B][H][W][3] -> float32 [B][H][W][4]: (x - mean) * inv_std + beta per channel
// (the bn_data layer, gamma fixed to 1), 4th channel zero.  norm = {scale[3], shift[3]}.
struct InputNorm {
    float scale[3];
    float shift[3];
};
int stage_input(const uint8_t* frames, int batch, int h, int w, const InputNorm& norm, float* y, hipStream_t stream);

// relu(x * scale[c] + shift[c]) averaged over the hw positions: x [N][hw][C] -> y [N][C].
int bn_relu_avgpool(const float* x, int n, int hw, int c, const float* scale, const float* shift, float* y,
                    hipStream_t stream);

// Classifier head: flat [N][C]; w_cls [2][C], w_box [8][C] (FullyConnected layout, device).
// cls_prob [N][2] = softmax(flat w_cls^T + b_cls); bbox [N][8] = flat w_box^T + b_box.
int fc_head(const float* flat, int n, int c, const float* w_cls, const float* b_cls, const float* w_box,
            const float* b_box, float* cls_prob, float* bbox, hipStream_t stream);

}  // namespace fd
