// Split-precision (two-way fp16) convolution path; see fd_split.h for the number format.
#pragma once
#include "fd_conv.h"

#include <vector>

namespace fd {

struct SplitConvParams {
    ConvShape s;
    const void* w = nullptr;       // packed split weights [cout_pad][K/32][32 hi | 32 lo], scaled by 1/w_inv_scale
    float w_inv_scale = 1.f;       // power of two; the accumulator is multiplied by it
    const float* bias = nullptr;   // [cout] or NULL
    int relu = 0;
    bool f32 = false;              // float32 operands: w is the float32 packing of fd_conv.h ([cout_pad][K] floats,
                                   // w_inv_scale 1), the input and the "split" output are plain float32 tensors
};

struct SplitConvOut {
    float* y32 = nullptr;              // float32 output [M][cout] (after bias/relu/residual) or NULL
    void* ysplit = nullptr;            // split output or NULL
    const float* post_scale = nullptr; // split output = split(relu(y * post_scale + post_shift)); NULL: split(y)
    const float* post_shift = nullptr;
    float* splitk_ws = nullptr;        // optional workspace for split-K partial sums (small grids only)
    size_t splitk_ws_floats = 0;
    unsigned* clamped = nullptr;       // device counter of values that left the fp16 range while being split, or NULL
    bool b3 = false;                   // float32 operands only: products on the bf16 matrix cores (three-way bf16 split of
                                       // both operands in registers, six partial products per multiply)
    int b3_balance = 63;               // ... with this percentage of a tile's K-steps accumulated as +(sum), the rest as
                                       // -(sum) (cancels the rounding bias of the bf16 matrix unit; 0: off)
};

int pack_conv_weights_split(const float* w_oihw, int cin_w, const ConvShape& s, const float* out_scale, float wscale,
                            void* packed, hipStream_t stream);

// x_split: split activations [B][H][W][cin] (or the split staged image for the 7x7 stem).
int split_conv_forward(const SplitConvParams& p, const void* x_split, int batch, int h, int w, const float* residual,
                       const SplitConvOut& out, hipStream_t stream);

// A bottleneck unit's last 1x1 convolution and its projection shortcut as ONE float32 GEMM (direct-to-LDS kernel with
// two-source rows): y[m] = [x1[m] | x2[pixel (ho * stride2, wo * stride2)]] . w_cat[o], w_cat = [cout][conv3.cin + cin2]
// floats (conv3's row, then the shortcut's).  The shortcut tensor is neither written nor read back as a residual.
// *taken = false (and nothing launched) when the layer is outside what the kernel addresses.
int split_conv_forward_two(const SplitConvParams& conv3, const float* w_cat, const float* x1, int batch, int ho, int wo,
                           const float* x2, int h2, int w2, int cin2, int stride2, const SplitConvOut& out, hipStream_t stream,
                           bool* taken);

// Two 1x1 / stride-1 float32 convolutions of the same input in one launch (a projection unit's shortcut and its first
// convolution): w_pair = pa's packed weight rows (pa.s.cout of them, a multiple of 64) followed by 64 rows of pb's
// (pb.s.cout <= 64 real rows, the rest zero).  ya [M][pa.s.cout] gets pa's result (pa.bias / pa.relu), yb [M][pb.s.cout]
// pb's (pb.bias / pb.relu).  Same accumulation order and epilogue as two split_conv_forward calls: the same bits.
// *taken is false (and nothing is launched) when the shapes are not the ones this form is for.
int split_conv_forward_pair(const SplitConvParams& pa, const SplitConvParams& pb, const float* w_pair, size_t w_pair_floats,
                            const float* x, int batch, int h, int w, float* ya, float* yb, hipStream_t stream, bool* taken);

// The same convolution on a float32 input [B][H][W][cin] that enters as
// split(relu(x * pre_scale[c] + pre_shift[c])), converted while the tile is staged.
int split_conv_forward_f32(const SplitConvParams& p, const float* x32, const float* pre_scale, const float* pre_shift,
                           int batch, int h, int w, const float* residual, const SplitConvOut& out,
                           hipStream_t stream);

// conv2 (3x3/1, 64 -> 64, ReLU) and conv3 (1x1, 64 -> cout) of a residual unit in one kernel; the
// tensor between them stays in LDS.  *taken is false (and nothing is launched) when the shapes are
// not the ones this kernel is written for; the caller then runs the two convolutions separately.
int conv_tail_fused_forward(const SplitConvParams& conv2, const SplitConvParams& conv3, const void* x_split, int batch,
                            int h, int w, const float* residual, const SplitConvOut& out, hipStream_t stream,
                            bool* taken);

// The same tail with the 3x3 in Winograd F(2x2, 3x3) form, float32 only (fd_conv_tailw.hip): u_wino = the transformed
// weights written by tail_wino_pack (host: OIHW 64 x 64 x 3 x 3 weights times out_scale[o] -> 16 x 2 x 64 x 32 floats in the kernel's reading order),
// bias2 = the 3x3's bias; *taken as above.  With sc_act and w3_sc (and residual == NULL) the projection shortcut of
// the unit runs inside the 1x1: sc_act = the unit's activated input [B][H][W][64], w3_sc = [cout][128] floats, row o =
// conv3's 64 weights then the shortcut's 64 -- the shortcut tensor is neither written nor read.  With conv1 as well
// (the unit's 64 -> 64 1x1 with ReLU, float32 form) the kernel computes t1 itself on the input patch and t1 may be NULL.
int tail_wino_pack(const float* w_oihw, const float* out_scale, std::vector<float>* packed);
int conv_tail_wino_forward(const float* u_wino, const float* bias2, const SplitConvParams& conv3, const float* t1, int batch,
                           int h, int w, const float* residual, const SplitConvOut& out, hipStream_t stream, bool* taken,
                           const float* sc_act = nullptr, const float* w3_sc = nullptr, const SplitConvParams* conv1 = nullptr);

// A whole identity-shortcut unit of the 64-channel stage (conv1 on the float32 input through
// BatchNorm+ReLU, conv2, conv3 + the input as residual) in one kernel; *taken as above.
int unit_fused_forward(const SplitConvParams& conv1, const SplitConvParams& conv2, const SplitConvParams& conv3,
                       const float* x32, const float* pre_scale, const float* pre_shift, int batch, int h, int w,
                       const SplitConvOut& out, hipStream_t stream, bool* taken);

// The projection-shortcut unit of the 64-channel stage (conv1, conv2, conv3 and the shortcut
// convolution, all on the split input act) in one kernel writing the float32 sum; *taken as above.
int unit_proj_fused_forward(const SplitConvParams& conv1, const SplitConvParams& conv2, const SplitConvParams& conv3,
                            const SplitConvParams& shortcut, const void* act_split, int batch, int h, int w, float* y32,
                            unsigned* clamped, hipStream_t stream, bool* taken);

// float32 [rows][channels] -> split, optionally through relu(x * scale[c] + shift[c]).
// (`clamped` in this and the following functions: the device counter described at SplitConvOut, may be NULL)
// (f32 = true in this and the following functions: the output is the plain float32 tensor of the activated
// values -- the operand format of the float32 convolutions -- instead of the split form)
int to_split(const float* x, long long rows, int channels, const float* scale, const float* shift, void* y,
             unsigned* clamped, hipStream_t stream, bool f32 = false);

int stage_input_split(const uint8_t* frames, int batch, int h, int w, const InputNorm& norm, void* y,
                      unsigned* clamped, hipStream_t stream);

// The 7x7/2 stem (bn0+ReLU folded) and the 3x3/2 max-pool in one kernel: staged split image ->
// pooled split activations with relu(x * post_scale + post_shift) applied (and float32 pooled
// values when y32 is given).
// (input: the planar staged image written by stage_input_planar)
int staged_planar_width(int w);
int stage_input_planar(const uint8_t* frames, int batch, int h, int w, const InputNorm& norm, void* y,
                       unsigned* clamped, hipStream_t stream);
// float32 operands (p.f32): the staged image is the single float32 plane written by stage_input_padded_f32
// (same zero columns, 16 bytes per column) and `ysplit` receives the plain float32 activated tensor.
int stage_input_padded_f32(const uint8_t* frames, int batch, int h, int w, const InputNorm& norm, void* y,
                           hipStream_t stream);
int stem_pool_forward(const SplitConvParams& p, const void* staged_planar, int batch, int h, int w,
                      const float* post_scale, const float* post_shift, float* y32, void* ysplit,
                      unsigned* clamped, hipStream_t stream);

// 3x3 stride-1 pad-1 convolution by Winograd F(2x2, 3x3) or F(4x4, 3x3) in float32 (fd_winograd.hip).
struct WinoConvParams {
    int cin = 0, cout = 0, cout_pad = 0;
    int tile = 2;                  // outputs per patch and direction: 2 (4x4 patches, 16 GEMMs) or 4 (6x6 patches, 36 GEMMs)
    const float* u = nullptr;      // transformed weights [(tile + 2)^2][cout_pad][cin] (device)
    const float* bias = nullptr;   // [cout] or NULL
    int relu = 0;
    // split-precision form of the same transformed weights (fp16 hi+lo pairs, packed per GEMM like 1x1
    // convolution weights, times a power of two 1 / u_inv_scale); NULL: float32 only
    const void* u_split = nullptr;
    float u_inv_scale = 1.f;
    unsigned u_split_bytes = 0;    // bytes per GEMM's weight set
};
// host: OIHW weights -> the transformed layout above (float32 values computed in double)
int wino_pack_weights(const float* w_oihw, int cout, int cin, const float* out_scale, int tile, std::vector<float>* packed,
                      int* cout_pad);
// number of tile x tile output patches of a batch, rounded up to whole 128-row GEMM tiles
long long wino_patches_padded(int batch, int h, int w, int tile, int align = 128);
// whether the 32-bit offsets of the kernels reach every tensor of this problem
bool wino_fits(const WinoConvParams& p, int batch, int h, int w);
// x [B][H][W][cin] float32 -> y [B][H][W][cout] float32 (+ bias, ReLU).  Workspaces: v_ws holds
// (tile + 2)^2 * wino_patches_padded * cin floats, m_ws (tile + 2)^2 * wino_patches_padded * cout floats.
// b3: the (tile + 2)^2 GEMMs with bf16x3 products (the transforms stay float32).
int wino_conv_forward(const WinoConvParams& p, const float* x, int batch, int h, int w, float* v_ws, float* m_ws,
                      float* y, hipStream_t stream, bool b3 = false, int b3_balance = 63);
// Split-precision form (tile 4 only): x and y are split tensors (fd_split.h), the Winograd-domain input V is a
// split tensor too (same bytes as float32), the GEMMs run on the f16 matrix cores (three MFMAs per product) and
// leave M in float32.  Patches are padded to whole 256-row GEMM tiles: wino_patches_padded(..., 256).
// device: packs the float32 transformed weights `u` (device, [36][cout_pad][cin]) into `packed` (36 sets of
// split_weight_floats(cin, cout) floats) scaled by `wscale`
size_t wino_split_weight_floats(int cin, int cout);
int wino_pack_weights_split(const float* u_dev, int cin, int cout, int cout_pad, float wscale, void* packed, hipStream_t stream);
bool wino_fits_split(const WinoConvParams& p, int batch, int h, int w);
int wino_conv_forward_split(const WinoConvParams& p, const void* x_split, int batch, int h, int w, void* v_ws, float* m_ws,
                            void* y_split, unsigned* clamped, hipStream_t stream);
// nb independent split-precision GEMMs in one launch (256 x 256 tiles): y[b] [rows_pad][cout] float32 =
// a[b] [rows_pad][k] (split) * w[b]^T * w_inv_scale; rows_pad % 256 == 0, k % 32 == 0
int gemm_split_batched(const void* a_split, const void* w_split, unsigned w_set_bytes, float w_inv_scale, int nb, int rows_pad,
                       int k, int cout, float* y, hipStream_t stream);
// nb independent float32 GEMMs in one launch on the 128 x 128 direct-to-LDS kernel:
// y[b] [rows_pad][cout] = a[b] [rows_pad][k] * w[b]^T, w[b] = [cout_pad][k]; rows_pad % 128 == 0, k % 32 == 0.
// b3: products on the bf16 matrix cores (three-way bf16 split of both operands), b3_balance as in SplitConvOut.
int gemm_f32_batched(const float* a, const float* w, int nb, int rows_pad, int k, int cout, int cout_pad, float* y,
                     hipStream_t stream, bool b3 = false, int b3_balance = 63);

int maxpool_3x3s2_split(const float* x, int batch, int h, int w, int c, const float* scale, const float* shift,
                        float* y32, void* ysplit, unsigned* clamped, hipStream_t stream, bool f32 = false);

}  // namespace fd
