// Device-side pieces shared by the split-precision convolution kernels (fd_conv_split.hip: the
// general kernels; fd_conv_fused.hip: the fused stem, bottleneck tail and whole-unit kernels):
// the kernel argument block, tile geometry, the MFMA step of a wave, the epilogues and the
// synchronisation helpers.
#pragma once
#include "fd_conv_split.h"
#include "fd_split.h"

namespace fd {

struct SplitKArgs {
    const uint8_t* x;        // split activations (or the split staged image for the stem)
    const uint8_t* w;        // split packed weights
    const float* bias;
    const float* res;        // float32 residual [M][Cout] or NULL
    float* y32;              // float32 output or NULL
    uint8_t* ysplit;         // split output or NULL
    const float* post_scale; // BatchNorm+ReLU applied to the split output only (NULL: raw)
    const float* post_shift;
    const float* pre_scale;  // PRE kernels: x is float32 and enters as split(relu(x * pre_scale[c] + pre_shift[c]))
    const float* pre_shift;
    float w_inv_scale;
    int H, W, Cin, Ho, Wo, Cout, KW, stride, pad, K, M, relu, tiles_n;
    // split-K (small grids): workgroup ks of a tile sums K-steps [ks * kchunk, (ks + 1) * kchunk) and
    // writes its scaled partial sums to partial[ks][M][Cout]; split_k_finish_kernel adds them up
    int ksplit, kchunk;
    float* partial;
    unsigned* clamped;       // device counter of values outside the fp16 range met while splitting (may be NULL)
    // batched GEMM (direct-to-LDS kernel only): row tiles [b * wbatch_tiles, (b + 1) * wbatch_tiles) multiply
    // the weight set at byte offset b * wbatch_bytes (0 tiles: one weight set)
    int wbatch_tiles = 0;
    unsigned wbatch_bytes = 0;
    int b3_balance = 63;     // bf16x3: percentage of a tile's K-steps accumulated as +(sum); the rest as -(sum), which
                             // cancels the matrix unit's rounding bias (0: everything as +(sum))
    // two convolutions of the same input in one launch (direct-to-LDS kernel only): the weight rows from alt_from on
    // (a multiple of the tile width) belong to a second layer with its own output tensor [M][alt_cout] (written
    // like `ysplit`: float32 in float32 mode), bias and ReLU flag; columns below alt_from form the first layer, whose
    // output row length is alt_from
    int alt_from = 0;
    uint8_t* alt_y = nullptr;
    int alt_cout = 0;
    const float* alt_bias = nullptr;
    int alt_relu = 0;
    int stream_res = 0;      // residual values of large tensors are read with streaming loads
    // two-source 1x1 (direct-to-LDS kernel, float32): the K = Cin + Cin2 channels of an output row are row m of x
    // (Cin channels) followed by the Cin2 channels of pixel (ho * stride2, wo * stride2) of the [.][H2][W2][Cin2] tensor x2
    const uint8_t* x2 = nullptr;
    int Cin2 = 0, H2 = 0, W2 = 0, stride2 = 1;
    unsigned x2_bytes = 0;
    int f32;                 // float32 operands (1; 2: with bf16x3 products): x, w and the "split" output are plain float32 tensors (same
                             // 4 bytes per element, so the same addressing; 128-byte rows of 32 floats)
};

// Row-panel kernel for the short-K 1x1 layers of the float32 pass (fd_conv_panel.hip): whether it takes the layer
// described by `a` (tiles_n / ksplit already set), and its launch.
bool conv_panel_applies(const SplitKArgs& a);
int conv_panel_forward(const SplitKArgs& a, size_t x_bytes, size_t w_bytes, hipStream_t stream);

constexpr int SBK = 32;      // k per K-step
constexpr int SLDK = 36;     // staged row: 128 B of data + 16 B pad, in dwords

template <int BM, int BN>
struct SplitTile {
    static constexpr int STAGE = (BM + BN) * 32;      // dwords per pipeline stage
    static constexpr int WN = BN / 2;
    static constexpr int LDC = WN + 4;
    static constexpr int EPI = 4 * 32 * LDC;          // epilogue staging, dwords
    static constexpr int SMEM = (2 * STAGE > EPI) ? 2 * STAGE : EPI;
    static constexpr int AM = BM / 32, AN = BN / 32;  // 16x16 blocks per wave
};

// Geometry of the epilogue's row passes: after the transpose through LDS a lane owns 8
// consecutive channels of one row per pass.
template <int BM, int BN>
struct SplitEpi {
    static constexpr int WN = BN / 2;
    static constexpr int L8 = WN / 8;           // lanes per row
    static constexpr int RPP = 64 / L8;         // rows per pass
    static constexpr int PASSES = 32 / RPP;     // passes per 32-row group
    static constexpr int GROUPS = BM / 64;      // 32-row groups per wave
};

// all MFMAs of one K-step (32 k) of a wave's tile from the stage at `stage`
template <int BM, int BN>
__device__ __forceinline__ void split_mma_step(const char* stage, const int (&a_off)[2], const int (&b_off)[2],
                                               f32x4 (&acc)[BM / 32][BN / 32]) {
    constexpr int AM = BM / 32, AN = BN / 32;
    constexpr int NG = AN > 4 ? 4 : AN;          // column blocks held in registers at a time
    f16x8 ah[AM], al[AM];
#pragma unroll
    for (int mi = 0; mi < AM; ++mi) {
        ah[mi] = *reinterpret_cast<const f16x8*>(stage + a_off[0] + mi * 2048);
        al[mi] = *reinterpret_cast<const f16x8*>(stage + a_off[1] + mi * 2048);
    }
#pragma unroll
    for (int n0 = 0; n0 < AN; n0 += NG) {
        f16x8 bh[NG], bl[NG];
#pragma unroll
        for (int ni = 0; ni < NG; ++ni) {
            bh[ni] = *reinterpret_cast<const f16x8*>(stage + b_off[0] + (n0 + ni) * 2048);
            bl[ni] = *reinterpret_cast<const f16x8*>(stage + b_off[1] + (n0 + ni) * 2048);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < AM; ++mi)
#pragma unroll
            for (int ni = 0; ni < NG; ++ni) {
                acc[mi][n0 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mi], bh[ni], acc[mi][n0 + ni], 0, 0, 0);
                acc[mi][n0 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bl[ni], acc[mi][n0 + ni], 0, 0, 0);
                acc[mi][n0 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bh[ni], acc[mi][n0 + ni], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
    }
}

// The same K-step with float32 operands (v_mfma_f32_16x16x4_f32): a staged row is 32 floats, chunk
// part * 4 + (lane >> 4) of a row gives a lane k = 4 * chunk .. + 3, and MFMA t of a part multiplies
// element t of both operands, i.e. the k set {4 * (part * 4 + g) + t : g = 0..3} -- the same
// permutation of k for both operands.  Blocks are visited innermost so that consecutive MFMAs
// write different accumulators.
template <int BM, int BN>
__device__ __forceinline__ void f32_mma_step(const char* stage, const int (&a_off)[2], const int (&b_off)[2],
                                             f32x4 (&acc)[BM / 32][BN / 32]) {
    constexpr int AM = BM / 32, AN = BN / 32;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        f32x4 av[AM], bv[AN];
#pragma unroll
        for (int mi = 0; mi < AM; ++mi) av[mi] = *reinterpret_cast<const f32x4*>(stage + a_off[part] + mi * 2048);
#pragma unroll
        for (int ni = 0; ni < AN; ++ni) bv[ni] = *reinterpret_cast<const f32x4*>(stage + b_off[part] + ni * 2048);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mi = 0; mi < AM; ++mi)
#pragma unroll
                for (int ni = 0; ni < AN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi][t], bv[ni][t], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
}

// The float32 K-step on a raw input: BatchNorm + ReLU (v -> max(v * scale[k] + shift[k], 0)) on the activation
// fragments between the LDS reads and the MFMAs.  `pre` points at this lane's four scale values of the step's
// part 0 (channels 4 * (lane >> 4) .. + 3 of the 32); part 1 is 16 channels on, the shifts `shift_bytes` on.
template <int BM, int BN>
__device__ __forceinline__ void f32_mma_step_pre(const char* stage, const int (&a_off)[2], const int (&b_off)[2],
                                                 f32x4 (&acc)[BM / 32][BN / 32], const char* pre, int shift_bytes) {
    constexpr int AM = BM / 32, AN = BN / 32;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        f32x4 av[AM], bv[AN];
        const f32x4 sc = *reinterpret_cast<const f32x4*>(pre + part * 64);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(pre + shift_bytes + part * 64);
#pragma unroll
        for (int mi = 0; mi < AM; ++mi) av[mi] = *reinterpret_cast<const f32x4*>(stage + a_off[part] + mi * 2048);
#pragma unroll
        for (int ni = 0; ni < AN; ++ni) bv[ni] = *reinterpret_cast<const f32x4*>(stage + b_off[part] + ni * 2048);
#pragma unroll
        for (int mi = 0; mi < AM; ++mi)
#pragma unroll
            for (int t = 0; t < 4; ++t) av[mi][t] = fmaxf(av[mi][t] * sc[t] + sh[t], 0.f);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mi = 0; mi < AM; ++mi)
#pragma unroll
                for (int ni = 0; ni < AN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi][t], bv[ni][t], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
}

// The same K-step on the bf16 matrix cores with float32-grade products ("bf16x3"): the staged operands are
// the same float32 rows; each value v is cut in registers into three bf16 numbers
//     hi = bf16(v),  mid = bf16(v - hi),  lo = bf16(v - hi - mid)        (round-to-nearest-even)
// whose sum reproduces v to 2^-27 |v| (both subtractions are exact in float32), with the float32 exponent
// range (nothing is clamped).  A product a * b is accumulated as the six partial products down to 2^-18 of
// its size,  lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi,  each exact in the MFMA's float32 accumulator;
// the three dropped ones (mid*lo, lo*mid, lo*lo) are below 2^-26 |a b|, less than the rounding of a float32
// multiply.  v_mfma_f32_16x16x32_bf16 takes 8 consecutive k per lane: the lane's two 16-byte chunks of the
// row (part 0 and 1), i.e. the same permutation of k for both operands as in f32_mma_step.
// NEG: the activation operand enters negated (its sign bit flipped before the cut), for callers that
// accumulate the second half of K as -(sum) to cancel the matrix unit's rounding direction (see the kernel).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Bf16x3 {
    u32x4 hi, mid, lo;      // 8 bf16 each
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));     // v_cvt_pk_bf16_f32
}

__device__ __forceinline__ void split_bf16x3(const f32x4& p0, const f32x4& p1, Bf16x3& o, unsigned flip = 0u) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = j < 2 ? p0[2 * j] : p1[2 * j - 4], b = j < 2 ? p0[2 * j + 1] : p1[2 * j - 3];
        a = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) ^ flip);
        b = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) ^ flip);
        const unsigned h = pack_bf16(a, b);
        const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
        const unsigned m = pack_bf16(ra, rb);
        const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
        o.hi[j] = h;
        o.mid[j] = m;
        o.lo[j] = pack_bf16(sa, sb);
    }
}

template <int BM, int BN>
__device__ __forceinline__ void bf16x3_mma_step(const char* stage, const int (&a_off)[2], const int (&b_off)[2],
                                                f32x4 (&acc)[BM / 32][BN / 32], unsigned flip = 0u) {
    constexpr int AM = BM / 32, AN = BN / 32;
    Bf16x3 a[AM];
#pragma unroll
    for (int mi = 0; mi < AM; ++mi)
        split_bf16x3(*reinterpret_cast<const f32x4*>(stage + a_off[0] + mi * 2048),
                     *reinterpret_cast<const f32x4*>(stage + a_off[1] + mi * 2048), a[mi], flip);
#pragma unroll
    for (int ni = 0; ni < AN; ++ni) {
        Bf16x3 b;
        split_bf16x3(*reinterpret_cast<const f32x4*>(stage + b_off[0] + ni * 2048),
                     *reinterpret_cast<const f32x4*>(stage + b_off[1] + ni * 2048), b);
        const bf16x8 bh = __builtin_bit_cast(bf16x8, b.hi), bm = __builtin_bit_cast(bf16x8, b.mid),
                     bl = __builtin_bit_cast(bf16x8, b.lo);
#pragma unroll
        for (int mi = 0; mi < AM; ++mi) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, a[mi].hi), am = __builtin_bit_cast(bf16x8, a[mi].mid),
                         al = __builtin_bit_cast(bf16x8, a[mi].lo);
            f32x4 c = acc[mi][ni];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
            acc[mi][ni] = c;
        }
    }
}

// Which row of the [M][Cout] output a tile-local row is (-1: none).  Linear tiles are BM
// consecutive rows; pixel tiles are 8 x 16 patches of one image.
struct LinearRows {
    int base, M;
    __device__ __forceinline__ int operator()(int local) const {
        const int r = base + local;
        return r < M ? r : -1;
    }
};
struct PatchRows {
    int pix0;          // output row of the tile's top-left pixel
    int W;             // image width in pixels
    int ymax, xmax;    // valid tile rows / columns (image edge)
    __device__ __forceinline__ int operator()(int local) const {
        const int ty = local >> 4, tx = local & 15;
        return (ty < ymax && tx < xmax) ? pix0 + ty * W + tx : -1;
    }
};

// The residual values a lane adds in the epilogue, fetched ahead of time by kernels whose
// per-workgroup chain of memory round trips would otherwise be exposed.
template <int BM, int BN, class RM>
__device__ __forceinline__ void split_residual_prefetch(const SplitKArgs& a, const RM& rm, int tile_n, int wave, int lane,
                                                        f32x4 (&res)[SplitEpi<BM, BN>::GROUPS][SplitEpi<BM, BN>::PASSES][2]) {
    using E = SplitEpi<BM, BN>;
    const int wm = wave >> 1, wn = wave & 1;
    const int erow = lane / E::L8, ecol = (lane % E::L8) * 8;
    const int gcol = tile_n * BN + wn * E::WN + ecol;
    const bool col_ok = gcol < a.Cout;
    // Every lane issues every load (rows outside the output and columns beyond Cout read a clamped
    // address and are zeroed afterwards): callers count these loads in their s_waitcnt vmcnt
    // values, so their number must not depend on the tile.
    const int safe_col = col_ok ? gcol : 0;
#pragma unroll
    for (int mi = 0; mi < E::GROUPS; ++mi)
#pragma unroll
        for (int ps = 0; ps < E::PASSES; ++ps) {
            const int row = rm(wm * (BM / 2) + mi * 32 + ps * E::RPP + erow);
            const bool ok = row >= 0 && col_ok;
            const float* rp = a.res + (size_t)(row >= 0 ? row : 0) * a.Cout + safe_col;
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
            res[mi][ps][0] = ok ? r0 : f32x4{0.f, 0.f, 0.f, 0.f};
            res[mi][ps][1] = ok ? r1 : f32x4{0.f, 0.f, 0.f, 0.f};
        }
}

// Epilogue of a 16x16-block tile: transpose each wave's rows through its slice of LDS (32 rows
// at a time); every lane then owns 8 consecutive channels of a row.
template <int BM, int BN, bool PREFETCHED, class RM>
__device__ __forceinline__ void split_epilogue16_rows(const SplitKArgs& a, float* smem, const RM& rm, int tile_n, int wave,
                                                      int lane, f32x4 (&acc)[BM / 32][BN / 32],
                                                      f32x4 (*pres)[SplitEpi<BM, BN>::PASSES][2] = nullptr) {
    using T = SplitTile<BM, BN>;
    constexpr int WN = T::WN, LDC = T::LDC, AN = T::AN;
    constexpr int L8 = WN / 8;
    constexpr int RPP = 64 / L8;
    constexpr int PASSES = 32 / RPP;
    const int wm = wave >> 1, wn = wave & 1;
    float* Cs = smem + wave * (32 * LDC);
    const int erow = lane / L8, ecol = (lane % L8) * 8;
    const int gcol = tile_n * BN + wn * WN + ecol;
    const bool col_ok = gcol < a.Cout;
    float bias_v[AN];
#pragma unroll
    for (int ni = 0; ni < AN; ++ni) {
        const int col = tile_n * BN + wn * WN + ni * 16 + (lane & 15);
        bias_v[ni] = (a.bias && col < a.Cout) ? a.bias[col] : 0.f;
    }
    f32x4 ps0 = {1.f, 1.f, 1.f, 1.f}, ps1 = ps0, pt0 = {0.f, 0.f, 0.f, 0.f}, pt1 = pt0;
    // float32 outputs larger than the memory-side cache are written with streaming stores
    const bool stream_out = (long long)a.M * a.Cout > (80ll << 20);
    const bool stream_res = stream_out && a.stream_res;
    const bool post = a.ysplit && a.post_scale && col_ok;
    if (post) {
        ps0 = *reinterpret_cast<const f32x4*>(a.post_scale + gcol);
        ps1 = *reinterpret_cast<const f32x4*>(a.post_scale + gcol + 4);
        pt0 = *reinterpret_cast<const f32x4*>(a.post_shift + gcol);
        pt1 = *reinterpret_cast<const f32x4*>(a.post_shift + gcol + 4);
    }
#pragma unroll
    for (int mi = 0; mi < BM / 64; ++mi) {
        const int lrow0 = wm * (BM / 2) + mi * 32;
        unsigned bad = 0;      // per row group: keeps the count out of the registers live across groups
        f32x4 res0[PASSES], res1[PASSES];
        if (PREFETCHED) {
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) { res0[ps] = pres[mi][ps][0]; res1[ps] = pres[mi][ps][1]; }
        } else if (a.res) {
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int row = rm(lrow0 + ps * RPP + erow);
                res0[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
                res1[ps] = res0[ps];
                if (row >= 0 && col_ok) {
                    const float* rp = a.res + (size_t)row * a.Cout + gcol;
                    if (stream_res) {
                        res0[ps] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp));
                        res1[ps] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp + 4));
                    } else {
                        res0[ps] = *reinterpret_cast<const f32x4*>(rp);
                        res1[ps] = *reinterpret_cast<const f32x4*>(rp + 4);
                    }
                }
            }
        }
        // The staging slice is private to the wave (LDS operations of a wave execute in order), so
        // only the first use needs a workgroup barrier: the slice overlays the pipeline stages other
        // waves may still be reading.
        if (mi == 0) __syncthreads();
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int ni = 0; ni < AN; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[mi * 2 + mb][ni][r] * a.w_inv_scale + bias_v[ni];
                    if (a.relu) v = fmaxf(v, 0.f);
                    Cs[(mb * 16 + (lane >> 4) * 4 + r) * LDC + ni * 16 + (lane & 15)] = v;
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int lrow2 = ps * RPP + erow;
            const int row = rm(lrow0 + lrow2);
            f32x4 v0 = *reinterpret_cast<const f32x4*>(&Cs[lrow2 * LDC + ecol]);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(&Cs[lrow2 * LDC + ecol + 4]);
            if (a.res) { v0 += res0[ps]; v1 += res1[ps]; }
            if (row >= 0 && col_ok) {
                if (a.y32) {
                    float* yp = a.y32 + (size_t)row * a.Cout + gcol;
                    // streaming stores: the float32 tensor is far larger than the caches and is next
                    // read several kernels later (+0.5 % frames/s against plain stores, same box)
                    if (stream_out) {
                        __builtin_nontemporal_store(v0, reinterpret_cast<f32x4*>(yp));
                        __builtin_nontemporal_store(v1, reinterpret_cast<f32x4*>(yp + 4));
                    } else {
                        *reinterpret_cast<f32x4*>(yp) = v0;
                        *reinterpret_cast<f32x4*>(yp + 4) = v1;
                    }
                }
                if (a.ysplit) {
                    if (post) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v0[j] = fmaxf(v0[j] * ps0[j] + pt0[j], 0.f);
                            v1[j] = fmaxf(v1[j] * ps1[j] + pt1[j], 0.f);
                        }
                    }
                    if (a.f32) {
                        // float32 operands: the activated tensor is a plain float32 tensor
                        float* ap = reinterpret_cast<float*>(a.ysplit) + (size_t)row * a.Cout + gcol;
                        *reinterpret_cast<f32x4*>(ap) = v0;
                        *reinterpret_cast<f32x4*>(ap + 4) = v1;
                    } else {
                        f16x8 hi, lo;
                        split8(v0, v1, hi, lo, bad);
                        uint8_t* sp = a.ysplit + (size_t)row * a.Cout * 4 + split_offset(gcol);
                        *reinterpret_cast<f16x8*>(sp) = hi;
                        *reinterpret_cast<f16x8*>(sp + 64) = lo;
                    }
                }
            }
        }
        if (a.ysplit) split_report(a.clamped, bad);
    }
}

// the same for a tile of BM consecutive output rows
template <int BM, int BN, bool PREFETCHED = false>
__device__ __forceinline__ void split_epilogue16(const SplitKArgs& a, float* smem, int tile_m, int tile_n, int wave,
                                                 int lane, f32x4 (&acc)[BM / 32][BN / 32],
                                                 f32x4 (*pres)[SplitEpi<BM, BN>::PASSES][2] = nullptr) {
    const LinearRows rm{tile_m * BM, a.M};
    split_epilogue16_rows<BM, BN, PREFETCHED>(a, smem, rm, tile_n, wave, lane, acc, pres);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Workgroup barrier for the multi-stage LDS-DMA loops.  __syncthreads() carries fences that make
// the compiler drain every vector-memory operation (s_waitcnt vmcnt(0)) while LDS-DMA is in
// flight, which turns a three- or four-stage pipeline into a two-stage one; here only this wave's
// LDS operations are drained, and LDS-DMA completion is the caller's business (wait_vmcnt before
// the barrier).
__device__ __forceinline__ void pipeline_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

}  // namespace fd
