// Split-precision convolution: the implicit GEMM of fd_conv.hip with both operands carried as
// two-way fp16 splits (fd_split.h) and the products formed on the f16 matrix cores (three MFMAs
// per block: hi*hi + hi*lo + lo*hi, fp32 accumulate).
//
// Kernels in this file, all on the same GEMM view (M = output pixels, N = output channels,
// K = kh * kw * cin in steps of 32, one 128-byte row [32 hi | 32 lo] per pixel and K-step):
//   conv_split_glds_kernel  - input already split: both operand tiles go global -> LDS by LDS-DMA;
//                             v_mfma_f32_16x16x32_f16 blocks; 2..4 LDS stages.  The workhorse.
//   conv_split_reg_kernel   - activations through registers into the same LDS image: float32
//                             input with BatchNorm+ReLU+split on the way (REG_PRE), or the staged
//                             4-channel image of the 7x7 stem (REG_STEM); weights by LDS-DMA.
//   conv_split_kernel       - pointer-addressed, register-staged form with padded LDS rows and
//                             32x32x16 blocks; taken only when a tensor exceeds the 2^31-byte
//                             reach of the buffer-addressed kernels above.
// Epilogues rescale by the weights' power-of-two factor, add bias / residual, and write float32
// and/or the split form for the next layer (with that layer's BatchNorm+ReLU applied).  The fused
// stem, bottleneck-tail and whole-unit kernels live in fd_conv_fused.hip; the device code both
// files share is in fd_conv_split_dev.h.
#include "fd_conv_split_dev.h"

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <string>
#include <vector>

namespace fd {


// PRE: the input tensor is float32 (same 4 bytes per element, so the same addressing) and the
// consumer's BatchNorm+ReLU and the fp16 split happen between the global load and the LDS
// store.  This lets the producer write the float32 tensor only instead of both forms.
template <int BM, int BN, bool STEM, bool PRE = false>
__global__ __launch_bounds__(256, 2) void conv_split_kernel(SplitKArgs a) {
    constexpr int RA = BM / 32, RB = BN / 32;
    constexpr int MT = BM / 64, NT = BN / 64;
    __shared__ __attribute__((aligned(16))) float As[2][BM * SLDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * SLDK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int wgid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_n = wgid % a.tiles_n, tile_m = wgid / a.tiles_n;

    const int c4 = tid & 7, r0 = tid >> 3;
    // all offsets in 4-byte units: a split element occupies 4 bytes like a float
    int hi0[RA], wi0[RA], rowoff[RA];
    const f32x4* xrow[RA];
    bool mval[RA];
    const float* xf = reinterpret_cast<const float*>(a.x);
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = tile_m * BM + r0 + 32 * i;
        mval[i] = m < a.M;
        const int mm = mval[i] ? m : 0;
        const int hw = a.Ho * a.Wo;
        const int img = mm / hw;
        const int rem = mm - img * hw;
        const int ho = rem / a.Wo;
        const int wo = rem - ho * a.Wo;
        hi0[i] = ho * a.stride - a.pad;
        wi0[i] = wo * a.stride - a.pad;
        rowoff[i] = (hi0[i] * a.W + wi0[i]) * a.Cin + c4 * 4;
        xrow[i] = reinterpret_cast<const f32x4*>(xf + (size_t)img * a.H * a.W * a.Cin);
    }
    const float* wf = reinterpret_cast<const float*>(a.w);
    const float* wrow[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) wrow[i] = wf + (size_t)(tile_n * BN + r0 + 32 * i) * a.K + c4 * 4;

    f32x4 ra[RA], rb[RB];
    f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
    unsigned okbits = 0, bad = 0;
    int kh = 0, kw = 0, coff = 0;
    const int nk = a.K / SBK;

#define FD_GLOAD(KT)                                                                              \
    do {                                                                                          \
        if (PRE) {                                                                                \
            psc = *reinterpret_cast<const f32x4*>(a.pre_scale + coff + c4 * 4);                   \
            psh = *reinterpret_cast<const f32x4*>(a.pre_shift + coff + c4 * 4);                   \
        }                                                                                         \
        const int th_ = STEM ? (KT) : kh;                                                         \
        const int tw_ = STEM ? c4 : kw;                                                           \
        const int toff_ = STEM ? (KT) * a.W * 4 : (kh * a.W + kw) * a.Cin + coff;                 \
        okbits = 0;                                                                               \
        _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                          \
            const int hi_ = hi0[i] + th_, wi_ = wi0[i] + tw_;                                     \
            const bool ok_ = mval[i] && (unsigned)hi_ < (unsigned)a.H && (unsigned)wi_ < (unsigned)a.W; \
            okbits |= ok_ ? (1u << i) : 0u;                                                       \
            const float* p_ = ok_ ? reinterpret_cast<const float*>(xrow[i]) + (rowoff[i] + toff_) : xf; \
            ra[i] = *reinterpret_cast<const f32x4*>(p_);                                          \
        }                                                                                         \
        _Pragma("unroll") for (int i = 0; i < RB; ++i)                                            \
            rb[i] = *reinterpret_cast<const f32x4*>(wrow[i] + (size_t)(KT) * SBK);                \
        if (!STEM) {                                                                              \
            coff += SBK;                                                                          \
            if (coff >= a.Cin) {                                                                  \
                coff = 0;                                                                         \
                if (++kw == a.KW) { kw = 0; ++kh; }                                               \
            }                                                                                     \
        }                                                                                         \
    } while (0)

    // Stem: a thread's 16 bytes are one pixel [4 hi | 4 lo]; its halves go to the hi and lo
    // sections of the staged row (k = 4 kw + c).  Otherwise the 16-byte chunk is copied as is.
#define FD_LSTORE(BUF)                                                                            \
    do {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                          \
            f32x4 v_ = ra[i];                                                                     \
            float* row_ = &As[BUF][(r0 + 32 * i) * SLDK];                                         \
            if (PRE) {                                                                            \
                f16x4 h_, l_;                                                                     \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                   \
                    float t_ = fmaxf(v_[j] * psc[j] + psh[j], 0.f);                               \
                    if (!((okbits >> i) & 1u)) t_ = 0.f;                                          \
                    _Float16 hh_, ll_;                                                            \
                    split_f32(t_, hh_, ll_, bad);                                                 \
                    h_[j] = hh_; l_[j] = ll_;                                                     \
                }                                                                                 \
                *reinterpret_cast<f16x4*>(row_ + c4 * 2) = h_;                                    \
                *reinterpret_cast<f16x4*>(row_ + 16 + c4 * 2) = l_;                               \
                continue;                                                                         \
            }                                                                                     \
            if (!((okbits >> i) & 1u)) v_ = f32x4{0.f, 0.f, 0.f, 0.f};                            \
            if (STEM) {                                                                           \
                *reinterpret_cast<float2*>(row_ + c4 * 2) = make_float2(v_[0], v_[1]);            \
                *reinterpret_cast<float2*>(row_ + 16 + c4 * 2) = make_float2(v_[2], v_[3]);       \
            } else {                                                                              \
                *reinterpret_cast<f32x4*>(row_ + c4 * 4) = v_;                                    \
            }                                                                                     \
        }                                                                                         \
        _Pragma("unroll") for (int i = 0; i < RB; ++i)                                            \
            *reinterpret_cast<f32x4*>(&Bs[BUF][(r0 + 32 * i) * SLDK + c4 * 4]) = rb[i];           \
    } while (0)

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // fragment of k-group kg (16 k): lane (row = lane & 31, half = lane >> 5) reads halves
    // 16 kg + 8 half .. + 7 of the hi section (dwords 0..15) and of the lo section (dwords 16..31)
    const int frow = lane & 31, fcol = (lane >> 5) * 4;
    const int a_frag = (wm * (BM / 2) + frow) * SLDK + fcol;
    const int b_frag = (wn * (BN / 2) + frow) * SLDK + fcol;

    FD_GLOAD(0);
    FD_LSTORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const int ktn = (kt + 1 < nk) ? kt + 1 : nk - 1;
        FD_GLOAD(ktn);
        __builtin_amdgcn_sched_barrier(0);
        const float* Ab = &As[cur][a_frag];
        const float* Bb = &Bs[cur][b_frag];
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            f16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                ah[mi] = *reinterpret_cast<const f16x8*>(Ab + mi * 32 * SLDK + kg * 8);
                al[mi] = *reinterpret_cast<const f16x8*>(Ab + mi * 32 * SLDK + 16 + kg * 8);
            }
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                bh[ni] = *reinterpret_cast<const f16x8*>(Bb + ni * 32 * SLDK + kg * 8);
                bl[ni] = *reinterpret_cast<const f16x8*>(Bb + ni * 32 * SLDK + 16 + kg * 8);
            }
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        FD_LSTORE(cur ^ 1);
        __syncthreads();
    }
#undef FD_GLOAD
#undef FD_LSTORE

    // Epilogue: transpose each wave's rows through its slice of the A staging buffer (32 rows at
    // a time); every lane then owns 8 consecutive channels of a row: two 16-byte float32 stores
    // and/or one 16-byte hi store plus one 16-byte lo store of the split form.
    constexpr int WN = BN / 2;
    constexpr int LDC = WN + 4;
    constexpr int L8 = WN / 8;                 // lanes per row
    constexpr int RPP = 64 / L8;               // rows per pass
    constexpr int PASSES = 32 / RPP;
    static_assert(2 * BM * SLDK >= 4 * 32 * LDC, "A staging buffer too small for the epilogue");
    float* Cs = &As[0][0] + wave * (32 * LDC);
    const int erow = lane / L8, ecol = (lane % L8) * 8;
    const int gcol = tile_n * BN + wn * WN + ecol;
    const bool col_ok = gcol < a.Cout;         // Cout is a multiple of 8
    float bias_v[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int col = tile_n * BN + wn * WN + ni * 32 + (lane & 31);
        bias_v[ni] = (a.bias && col < a.Cout) ? a.bias[col] : 0.f;
    }
    f32x4 ps0 = {1.f, 1.f, 1.f, 1.f}, ps1 = ps0, pt0 = {0.f, 0.f, 0.f, 0.f}, pt1 = pt0;
    const bool post = a.ysplit && a.post_scale && col_ok;
    if (post) {
        ps0 = *reinterpret_cast<const f32x4*>(a.post_scale + gcol);
        ps1 = *reinterpret_cast<const f32x4*>(a.post_scale + gcol + 4);
        pt0 = *reinterpret_cast<const f32x4*>(a.post_shift + gcol);
        pt1 = *reinterpret_cast<const f32x4*>(a.post_shift + gcol + 4);
    }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int grow0 = tile_m * BM + wm * (BM / 2) + mi * 32;
        f32x4 res0[PASSES], res1[PASSES];
        if (a.res) {
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int row = grow0 + ps * RPP + erow;
                res0[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
                res1[ps] = res0[ps];
                if (row < a.M && col_ok) {
                    const float* rp = a.res + (size_t)row * a.Cout + gcol;
                    res0[ps] = *reinterpret_cast<const f32x4*>(rp);
                    res1[ps] = *reinterpret_cast<const f32x4*>(rp + 4);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mi][ni][r] * a.w_inv_scale + bias_v[ni];
                if (a.relu) v = fmaxf(v, 0.f);
                Cs[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDC + ni * 32 + (lane & 31)] = v;
            }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int lrow = ps * RPP + erow;
            const int row = grow0 + lrow;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(&Cs[lrow * LDC + ecol]);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(&Cs[lrow * LDC + ecol + 4]);
            if (a.res) { v0 += res0[ps]; v1 += res1[ps]; }
            if (row < a.M && col_ok) {
                if (a.y32) {
                    float* yp = a.y32 + (size_t)row * a.Cout + gcol;
                    *reinterpret_cast<f32x4*>(yp) = v0;
                    *reinterpret_cast<f32x4*>(yp + 4) = v1;
                }
                if (a.ysplit) {
                    if (post) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v0[j] = fmaxf(v0[j] * ps0[j] + pt0[j], 0.f);
                            v1[j] = fmaxf(v1[j] * ps1[j] + pt1[j], 0.f);
                        }
                    }
                    f16x8 hi, lo;
                    split8(v0, v1, hi, lo, bad);
                    uint8_t* sp = a.ysplit + (size_t)row * a.Cout * 4 + split_offset(gcol);
                    *reinterpret_cast<f16x8*>(sp) = hi;
                    *reinterpret_cast<f16x8*>(sp + 64) = lo;
                }
            }
        }
    }
    split_report(a.clamped, bad);
}

// ---------------------------------------------------------------------------------
// Direct-to-LDS kernel (every layer whose input is already split).  Both operand tiles are
// fetched with `buffer_load_dwordx4 ... lds`: no staging registers, no ds_write, and padding
// taps / rows beyond M are zero-filled by the buffer's range check (their offset is pushed out
// of range).  One wave instruction fills 8 rows x 128 B of the LDS image contiguously, so rows
// cannot be padded; instead the 16-byte chunk c of row r is stored at chunk position
// c ^ ((r >> 1) & 7), which makes the fragment reads conflict-free.  The permutation is applied
// on the global source address of each lane and undone when reading.
// NS pipeline stages: a K-step's loads are issued NS-1 steps before their use.  Two stages for
// the 128-row tiles (two workgroups per CU share 160 KB of LDS; three stages measured slower on
// the 128x64 tile), four for the 64x64 tile, whose small grids leave one wave per SIMD.
// WAVES = 4: a 2 x 2 grid of waves, two workgroups per CU.  WAVES = 8: a 4 x 2 grid of waves over
// a tile twice as tall, one workgroup per CU; per multiply the larger tile moves 25 % fewer bytes
// from L2 into LDS, which is what bounds the long-K layers (the 128 x 128 loop of the RPN 3x3
// takes 976 us with the MFMAs removed and 1141 us with the LDS-DMA removed, 1476 us as it is).
// F32: the operands are float32 (a staged row = 32 floats) and the products run on the fp32 matrix
// cores; loads, LDS image, pipeline and epilogue are the same code.
// B3 (with F32): the same float32 operand images, products on the bf16 matrix cores (bf16x3_mma_step).  The bf16
// matrix unit rounds its sums downwards (measured: the mean error of a K = 4608 layer is -2e-7 of the output rms,
// against 3e-10 for the float32 unit, at equal rms error), a bias that survives averaging layers.  The second half of
// a tile's K-steps is therefore accumulated as -(sum): the accumulators change sign at the midpoint and at the end and
// the activation operand enters negated in between, so the two halves' biases cancel.
// PREK > 0 (float32, 1x1 layers of at most PREK input channels): the input is the unit's raw float32 tensor and
// BatchNorm + ReLU are applied to the activation fragments in registers, between their LDS reads and the MFMAs
// (scale and shift of all K channels sit in LDS behind the pipeline stages); the loads stay direct-to-LDS with
// counted waits, where the register-staged kernel below drains its loads at every K-step.  The activated values and
// the order of the products are those of the register-staged kernel, so both give the same bits.
// TWO (float32, 1x1): the K channels of a row come from two tensors, x (row m, Cin channels) and x2 (a strided pixel,
// Cin2 channels) -- a bottleneck unit's last convolution and its projection shortcut as one GEMM over [t2 | act] with the
// weight rows [W3 | Wsc], so the shortcut tensor is neither written nor read back as a residual.
template <int BM, int BN, int NS, int WAVES, bool F32, int OCC, bool B3, int PREK, bool TWO = false>
__device__ __forceinline__ void conv_split_glds_body(const SplitKArgs& a, unsigned x_bytes, unsigned w_bytes) {
    static_assert(!B3 || F32, "bf16x3 products are formed from float32 operand images");
    static_assert(!TWO || (F32 && !B3 && PREK == 0), "two-source rows: float32 products only");
    static_assert(PREK == 0 || (F32 && !B3 && PREK % 32 == 0), "BatchNorm on the fragments: float32 products only");
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins exist in device code only; the host pass needs just the stub
    constexpr int WROWS = BM / (WAVES / 2);     // rows of the tile per wave row
    using T = SplitTile<2 * WROWS, BN>;         // per-wave geometry: that of a 2 x 2 grid over 2 WROWS rows
    constexpr int GA = BM / (8 * WAVES), GB = BN / (8 * WAVES);   // 8-row groups per wave per K-step
    constexpr int PIECES = GA + GB;             // LDS-DMA instructions per wave and stage
    constexpr int STAGE = (BM + BN) * 32;       // dwords per pipeline stage
    constexpr int EPI = WAVES * 32 * T::LDC;    // epilogue staging: one 32-row slice per wave
    constexpr int SMEM = (NS * STAGE > EPI) ? NS * STAGE : EPI;
    static_assert(BM % (8 * WAVES) == 0 && BN % (8 * WAVES) == 0 && WROWS % 32 == 0 && SMEM * 4 <= 160 * 1024, "tile shape");
    static_assert(NS >= 2 && NS <= 4 && (NS - 2) * PIECES < 64, "pipeline depth");
    static_assert((SMEM + 2 * PREK) * 4 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) float smem[SMEM + 2 * PREK];

    // the wave index is uniform: keeping it in a scalar register makes every LDS destination
    // address of the loads below scalar arithmetic
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int wgid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int ks = wgid % a.ksplit;
    const int tile_n = (wgid / a.ksplit) % a.tiles_n, tile_m = wgid / (a.ksplit * a.tiles_n);

    if constexpr (PREK > 0) {
        // scale | shift of the input channels this workgroup multiplies (all of them unless the launch splits K;
        // before any LDS-DMA is issued: these are ordinary loads, and the first barrier of the pipeline below
        // publishes them)
        const int c0 = ks * a.kchunk * SBK, c1 = min(a.K, c0 + a.kchunk * SBK);
        for (int i = c0 + tid * 4; i < c1; i += WAVES * 256) {
            *reinterpret_cast<f32x4*>(smem + SMEM + i) = *reinterpret_cast<const f32x4*>(a.pre_scale + i);
            *reinterpret_cast<f32x4*>(smem + SMEM + PREK + i) = *reinterpret_cast<const f32x4*>(a.pre_shift + i);
        }
    }

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;        // beyond any tensor this kernel is launched for

    // this lane's row inside an 8-row group and the chunk position it fills
    const int lrow = lane >> 3, lpos = lane & 7;
    int hi0[GA], wi0[GA];
    unsigned abase[GA];                          // byte offset of the row's window origin (+ its chunk)
    bool mval[GA];
    unsigned abase2[TWO ? GA : 1];               // TWO: the row's pixel in the second tensor (+ its chunk)
    const bool linear_rows = TWO || (a.K == a.Cin && a.stride == 1 && a.pad == 0 && a.Ho == a.H && a.Wo == a.W);
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int row = (wave + WAVES * i) * 8 + lrow;              // row inside the tile
        const int chunk = lpos ^ ((row >> 1) & 7);
        const int m = tile_m * BM + row;
        mval[i] = m < a.M;
        const int mm = mval[i] ? m : 0;
        if constexpr (TWO) {
            const int hw = a.Ho * a.Wo;
            const int img = mm / hw;
            const int rem = mm - img * hw;
            const int ho = rem / a.Wo;
            const int wo = rem - ho * a.Wo;
            abase2[i] = (unsigned)(((img * a.H2 + ho * a.stride2) * a.W2 + wo * a.stride2) * a.Cin2 * 4 + chunk * 16);
        }
        if (linear_rows) {
            // 1x1, stride 1: output row m reads input row m, no pixel coordinates (and no divisions) needed
            hi0[i] = 0;
            wi0[i] = 0;
            abase[i] = (unsigned)(mm * a.Cin * 4 + chunk * 16);
            continue;
        }
        const int hw = a.Ho * a.Wo;
        const int img = mm / hw;
        const int rem = mm - img * hw;
        const int ho = rem / a.Wo;
        const int wo = rem - ho * a.Wo;
        hi0[i] = ho * a.stride - a.pad;
        wi0[i] = wo * a.stride - a.pad;
        abase[i] = (unsigned)(((img * a.H + hi0[i]) * a.W + wi0[i]) * a.Cin * 4 + chunk * 16);
    }
    unsigned bvo[GB];                            // weights: per-lane offset, constant over the K loop
#pragma unroll
    for (int i = 0; i < GB; ++i) {
        const int row = (wave + WAVES * i) * 8 + lrow;
        const int chunk = lpos ^ ((row >> 1) & 7);
        bvo[i] = (unsigned)((tile_n * BN + row) * a.K * 4 + chunk * 16);
        if (a.wbatch_tiles > 0) bvo[i] += (unsigned)(tile_m / a.wbatch_tiles) * a.wbatch_bytes;
    }
    // this workgroup's K-steps: kb .. kb + nk - 1 (everything unless the launch splits K)
    const int kb = ks * a.kchunk;
    const int nk = min(a.K / SBK, kb + a.kchunk) - kb;
    // Order of the K-steps: 32-channel chunk outermost, filter taps inside.  All taps of a chunk
    // read nearly the same activation rows (shifted by one pixel or one image row), 128 bytes
    // each, so a workgroup's working set between re-uses is a few dozen KB and stays in L2; with
    // the taps outermost every tap streamed the rows' full channel depth and a 3x3 layer fetched
    // its input nine times from memory (PMC: 612 MB for a 67 MB input).
    const int ntaps = TWO ? 1 : a.K / a.Cin, cchunks = (TWO ? a.K : a.Cin) / SBK;
    const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc((void*)(TWO ? a.x2 : a.x), 0, TWO ? a.x2_bytes : x_bytes, 0x00020000);
    int kh, kw, coff, tapi;
    {
        const int chunk = kb / ntaps;
        tapi = kb - chunk * ntaps;
        coff = chunk * SBK;
        kh = tapi / a.KW;
        kw = tapi - kh * a.KW;
    }

    // Per-lane offsets of the activation rows change only with the filter tap (validity of the
    // tap for the row's window); the channel position inside the tap and the K position in the
    // weights are uniform and travel in the instruction's scalar offset.
    unsigned avo[GA];
#define FD_TAP()                                                                                  \
    do {                                                                                          \
        const unsigned toff_ = (unsigned)((kh * a.W + kw) * a.Cin * 4);                           \
        _Pragma("unroll") for (int i = 0; i < GA; ++i) {                                          \
            const int hi_ = hi0[i] + kh, wi_ = wi0[i] + kw;                                       \
            const bool ok_ = mval[i] && (unsigned)hi_ < (unsigned)a.H && (unsigned)wi_ < (unsigned)a.W; \
            avo[i] = ok_ ? abase[i] + toff_ : OOB;                                                \
        }                                                                                         \
    } while (0)

#define FD_ISSUE(KT, STG)                                                                         \
    do {                                                                                          \
        float* sa_ = smem + (STG) * STAGE;                                                        \
        if (TWO && coff >= a.Cin) {                                                               \
            _Pragma("unroll") for (int i = 0; i < GA; ++i)                                        \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                         \
                    xr2, (__attribute__((address_space(3))) void*)(sa_ + (wave + WAVES * i) * 256), 16, \
                    mval[i] ? abase2[TWO ? i : 0] : OOB, (coff - a.Cin) * 4, 0, 0);               \
        } else {                                                                                  \
            _Pragma("unroll") for (int i = 0; i < GA; ++i)                                        \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                         \
                    xr, (__attribute__((address_space(3))) void*)(sa_ + (wave + WAVES * i) * 256), 16, avo[i], coff * 4, 0, 0); \
        }                                                                                         \
        _Pragma("unroll") for (int i = 0; i < GB; ++i)                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                             \
                wr, (__attribute__((address_space(3))) void*)(sa_ + BM * 32 + (wave + WAVES * i) * 256), 16, \
                bvo[i], (tapi * cchunks + (coff >> 5)) * 128, 0, 0);                              \
        if (ntaps > 1) {                                                                          \
            if (++tapi == ntaps) { tapi = 0; kh = 0; kw = 0; coff += SBK; }                       \
            else if (++kw == a.KW) { kw = 0; ++kh; }                                              \
            FD_TAP();                                                                             \
        } else {                                                                                  \
            coff += SBK;                                                                          \
        }                                                                                         \
    } while (0)

    // wait until at most AHEAD later stages of this wave's loads are still in flight
#define FD_WAIT(AHEAD)                                                                            \
    do {                                                                                          \
        const int ahead_ = (AHEAD);                                                               \
        if (NS >= 4 && ahead_ >= 2) wait_vmcnt<(NS >= 4 ? 2 * PIECES : 0)>();                     \
        else if (NS >= 3 && ahead_ >= 1) wait_vmcnt<(NS >= 3 ? PIECES : 0)>();                    \
        else wait_vmcnt<0>();                                                                     \
    } while (0)

    f32x4 acc[T::AM][T::AN];
#pragma unroll
    for (int mi = 0; mi < T::AM; ++mi)
#pragma unroll
        for (int ni = 0; ni < T::AN; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addressing (16x16x32 blocks): row R with R & 15 == lane & 15, chunk
    // part * 4 + (lane >> 4) (hi chunks 0..3, lo chunks 4..7) stored at position
    // chunk ^ ((R >> 1) & 7); stage and block index are immediates
    const int frow = lane & 15, fh = lane >> 4;
    const int sw = (frow >> 1) & 7;
    const char* sbase = reinterpret_cast<const char*>(smem);
    int a_off[2], b_off[2];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int pos = ((part * 4 + fh) ^ sw) * 16;
        a_off[part] = (wm * WROWS + frow) * 128 + pos;
        b_off[part] = (BM + wn * (BN / 2) + frow) * 128 + pos;
    }

    // BatchNorm on the fragments: this lane's scale values of K-step 0, part 0 (the shifts sit PREK floats on)
    const char* pre_base = sbase + (SMEM + fh * 4) * 4;

    // bf16x3: K-steps from `half` on are accumulated negated
    const int half = (B3 && a.b3_balance > 0 && nk > 1) ? max(1, min(nk - 1, (nk * a.b3_balance + 50) / 100)) : nk + 1;
    unsigned flip = 0u;
#define FD_NEGATE()                                                                               \
    do {                                                                                          \
        _Pragma("unroll") for (int mi = 0; mi < T::AM; ++mi)                                      \
            _Pragma("unroll") for (int ni = 0; ni < T::AN; ++ni) acc[mi][ni] = -acc[mi][ni];      \
    } while (0)
#define FD_STEP(CUR, KT)                                                                          \
    do {                                                                                          \
        if ((KT) + NS - 1 < nk) FD_ISSUE((KT) + NS - 1, ((CUR) + NS - 1) % NS);                   \
        if (B3) {                                                                                 \
            if ((KT) == half) { FD_NEGATE(); flip = 0x80000000u; }                                \
            bf16x3_mma_step<2 * WROWS, BN>(sbase + (CUR) * STAGE * 4, a_off, b_off, acc, flip);   \
        } else if (PREK > 0) {                                                                    \
            f32_mma_step_pre<2 * WROWS, BN>(sbase + (CUR) * STAGE * 4, a_off, b_off, acc,         \
                                            pre_base + (kb + (KT)) * 128, PREK * 4);              \
        } else if (F32) f32_mma_step<2 * WROWS, BN>(sbase + (CUR) * STAGE * 4, a_off, b_off, acc); \
        else split_mma_step<2 * WROWS, BN>(sbase + (CUR) * STAGE * 4, a_off, b_off, acc);         \
        FD_WAIT(min(NS - 2, nk - 2 - (KT)));                                                      \
        pipeline_barrier();                                                                       \
    } while (0)

    FD_TAP();
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < nk) FD_ISSUE(st, st);
    FD_WAIT(min(NS - 2, nk - 1));
    pipeline_barrier();
    for (int kt = 0; kt < nk; kt += NS) {
        FD_STEP(0, kt);
        if (kt + 1 < nk) FD_STEP(1, kt + 1);
        if constexpr (NS > 2) {
            if (kt + 2 < nk) FD_STEP(2, kt + 2);
        }
        if constexpr (NS > 3) {
            if (kt + 3 < nk) FD_STEP(3, kt + 3);
        }
    }
    if (B3 && flip) FD_NEGATE();
#undef FD_STEP
#undef FD_NEGATE
#undef FD_WAIT
#undef FD_ISSUE
#undef FD_TAP
    if (a.ksplit > 1) {
        SplitKArgs e = a;            // raw scaled partial sums; bias, ReLU, residual and split follow in the finish kernel
        e.y32 = a.partial + (size_t)ks * a.M * a.Cout;
        e.ysplit = nullptr; e.bias = nullptr; e.res = nullptr; e.relu = 0;
        split_epilogue16_rows<2 * WROWS, BN, false>(e, smem, LinearRows{tile_m * BM, a.M}, tile_n, wave, lane, acc);
    } else if (a.alt_y) {
        // two layers in one launch: this tile's columns belong to the first (row length alt_from) or the second
        SplitKArgs e = a;
        int tn = tile_n;
        if (tile_n * BN >= a.alt_from) {
            tn = tile_n - a.alt_from / BN;
            e.Cout = a.alt_cout; e.y32 = nullptr; e.ysplit = a.alt_y; e.bias = a.alt_bias; e.relu = a.alt_relu;
            e.res = nullptr; e.post_scale = nullptr; e.post_shift = nullptr;
        } else {
            e.Cout = a.alt_from;
        }
        split_epilogue16_rows<2 * WROWS, BN, false>(e, smem, LinearRows{tile_m * BM, a.M}, tn, wave, lane, acc);
    } else {
        split_epilogue16_rows<2 * WROWS, BN, false>(a, smem, LinearRows{tile_m * BM, a.M}, tile_n, wave, lane, acc);
    }
#endif
}

template <int BM, int BN, int NS, int WAVES = 4, bool F32 = false, int OCC = 0, bool B3 = false>
__global__ __launch_bounds__(WAVES * 64, OCC ? OCC : (WAVES == 4 ? 2 : 1)) void conv_split_glds_kernel(SplitKArgs a, unsigned x_bytes,
                                                                                      unsigned w_bytes) {
    conv_split_glds_body<BM, BN, NS, WAVES, F32, OCC, B3, 0>(a, x_bytes, w_bytes);
}

// float32, two-source rows (a unit's last convolution and its projection shortcut as one GEMM)
template <int BM, int BN, int NS>
__global__ __launch_bounds__(256, 2) void conv_split_glds_two_kernel(SplitKArgs a, unsigned x_bytes, unsigned w_bytes) {
    conv_split_glds_body<BM, BN, NS, 4, true, 0, false, 0, true>(a, x_bytes, w_bytes);
}

template <int BM, int BN, int NS>
static int launch_split_glds_two(const SplitKArgs& a, size_t x_bytes, size_t w_bytes, hipStream_t stream) {
    static const std::string name = std::string("conv_f32_") + std::to_string(BM) + "x" + std::to_string(BN) + "_two";
    ProfScope scope(stream, name.c_str(), 2.0 * a.M * a.Cout * (double)a.K,
                    4.0 * a.M * (double)a.K + 4.0 * a.Cout * (double)a.K +
                        4.0 * a.M * a.Cout * ((a.res ? 1.0 : 0.0) + (a.y32 ? 1.0 : 0.0) + (a.ysplit ? 1.0 : 0.0)));
    SplitKArgs k = a;
    k.tiles_n = ceil_div(a.Cout, BN);
    hipLaunchKernelGGL((conv_split_glds_two_kernel<BM, BN, NS>), dim3(ceil_div(a.M, BM) * k.tiles_n), dim3(256), 0, stream, k,
                       (unsigned)x_bytes, (unsigned)w_bytes);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

// float32, BatchNorm + ReLU on the activation fragments (input channels <= PREK)
template <int BM, int BN, int NS, int OCC, int PREK>
__global__ __launch_bounds__(256, OCC ? OCC : 2) void conv_split_glds_pre_kernel(SplitKArgs a, unsigned x_bytes, unsigned w_bytes) {
    conv_split_glds_body<BM, BN, NS, 4, true, OCC, false, PREK>(a, x_bytes, w_bytes);
}

template <int BM, int BN, int NS, int OCC, int PREK>
static int launch_split_glds_pre(const SplitKArgs& a, size_t x_bytes, size_t w_bytes, hipStream_t stream) {
    static const std::string name = std::string("conv_f32_") + std::to_string(BM) + "x" + std::to_string(BN) +
                                    (OCC ? "_o" + std::to_string(OCC) : std::string()) + "_bn";
    FD_REQUIRE(a.f32 == 1 && a.pre_scale && a.pre_shift && a.K == a.Cin && a.pad == 0 && a.stride == 1 && a.K <= PREK && !a.alt_y,
               "conv_split_glds_pre: not a 1x1 float32 layer of at most %d channels", PREK);
    ProfScope scope(stream, name.c_str(), 2.0 * a.M * a.Cout * (double)a.K,
                    4.0 * a.M * a.Cin + 4.0 * a.Cout * (double)a.K +
                        4.0 * a.M * a.Cout * ((a.res ? 1.0 : 0.0) + (a.y32 ? 1.0 : 0.0) + (a.ysplit ? 1.0 : 0.0)));
    SplitKArgs k = a;
    const int tiles_m = ceil_div(a.M, BM);
    k.tiles_n = ceil_div(a.Cout, BN);
    hipLaunchKernelGGL((conv_split_glds_pre_kernel<BM, BN, NS, OCC, PREK>), dim3(tiles_m * k.tiles_n * k.ksplit), dim3(256), 0,
                       stream, k, (unsigned)x_bytes, (unsigned)w_bytes);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

template <int BM, int BN, int NS, int WAVES = 4, bool F32 = false, int OCC = 0, bool B3 = false>
static int launch_split_glds(const SplitKArgs& a, size_t x_bytes, size_t w_bytes, hipStream_t stream) {
    if constexpr (F32 && !B3) {
        // bf16x3: same tensors, tiles and pipeline, products on the bf16 matrix cores
        if (a.f32 == 2) return launch_split_glds<BM, BN, NS, WAVES, true, OCC, true>(a, x_bytes, w_bytes, stream);
    }
    static const std::string name = std::string(B3 ? "conv_b3_" : F32 ? "conv_f32_" : "conv_split_") + std::to_string(BM) + "x" + std::to_string(BN) +
                                    (OCC ? "_o" + std::to_string(OCC) : std::string());
    ProfScope scope(stream, name.c_str(), 2.0 * a.M * a.Cout * (double)a.K,
                    (double)a.H * a.W * (a.M / ((double)a.Ho * a.Wo)) * 4.0 * a.Cin + 4.0 * a.Cout * (double)a.K +
                        4.0 * a.M * a.Cout * ((a.res ? 1.0 : 0.0) + (a.y32 ? 1.0 : 0.0) + (a.ysplit ? 1.0 : 0.0)));
    SplitKArgs k = a;
    const int tiles_m = ceil_div(a.M, BM);
    k.tiles_n = ceil_div(a.Cout, BN);
    hipLaunchKernelGGL((conv_split_glds_kernel<BM, BN, NS, WAVES, F32, OCC, B3>), dim3(tiles_m * k.tiles_n * k.ksplit),
                       dim3(WAVES * 64), 0, stream, k, (unsigned)x_bytes, (unsigned)w_bytes);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

// ---------------------------------------------------------------------------------
// Register-staged activations on the same LDS image as the direct-to-LDS kernel (swizzled
// 128-byte rows, 16x16x32 blocks, weights by LDS-DMA).  Two uses:
//   REG_PRE  - the input is float32; BatchNorm+ReLU and the fp16 split happen between the
//              global load and the LDS store (first convolution of identity-shortcut units);
//   REG_STEM - the 7x7/2 stem on the staged 4-channel image: a K-step is one kernel row,
//              k = 4 * kw + c, a thread's 16 bytes are one pixel [4 hi | 4 lo].
// In both a thread's 16 global bytes become 8 bytes of the hi half of a row (chunk c4 >> 1,
// byte (c4 & 1) * 8) and 8 bytes of the lo half (chunk 4 + (c4 >> 1)).  Loads are buffer
// loads: per-lane offsets change with the filter tap only, the channel offset is scalar, and
// padding taps read zeros through the range check.
enum { REG_PRE = 0, REG_STEM = 1 };

// F32 (REG_PRE only): float32 operands -- the activated values go to LDS as floats (one 16-byte chunk
// per thread) and the products run on the fp32 matrix cores.
template <int BM, int BN, int MODE, bool F32 = false, bool B3 = false>
__global__ __launch_bounds__(256, 2) void conv_split_reg_kernel(SplitKArgs a, unsigned x_bytes, unsigned w_bytes) {
    static_assert(!B3 || F32, "bf16x3 products are formed from float32 operand images");
#if defined(__HIP_DEVICE_COMPILE__)
    using T = SplitTile<BM, BN>;
    constexpr int RA = BM / 32;                  // activation rows per thread and K-step
    constexpr int GB = BN / 32;                  // weight 8-row groups per wave and K-step
    constexpr int STAGE = T::STAGE;
    __shared__ __attribute__((aligned(16))) float smem[T::SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int wgid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int ks = wgid % a.ksplit;
    const int tile_n = (wgid / a.ksplit) % a.tiles_n, tile_m = wgid / (a.ksplit * a.tiles_n);

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // activations: thread -> rows r0 + 32 i, global 16-byte chunk c4 of the K-step
    const int c4 = tid & 7, r0 = tid >> 3;
    int hi0[RA], wi0[RA];
    unsigned abase[RA], a_lds[RA];
    bool mval[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int row = r0 + 32 * i;
        const int m = tile_m * BM + row;
        mval[i] = m < a.M;
        const int mm = mval[i] ? m : 0;
        const int hw = a.Ho * a.Wo;
        const int img = mm / hw;
        const int rem = mm - img * hw;
        const int ho = rem / a.Wo;
        const int wo = rem - ho * a.Wo;
        hi0[i] = ho * a.stride - a.pad;
        wi0[i] = wo * a.stride - a.pad;
        abase[i] = (unsigned)(((img * a.H + hi0[i]) * a.W + wi0[i]) * a.Cin * 4 + c4 * 16);
        a_lds[i] = F32 ? (unsigned)(row * 128 + ((c4 ^ ((row >> 1) & 7)) * 16))
                       : (unsigned)(row * 128 + (((c4 >> 1) ^ ((row >> 1) & 7)) * 16) + (c4 & 1) * 8);
    }
    // weights: as in the direct-to-LDS kernel
    const int lrow = lane >> 3, lpos = lane & 7;
    unsigned bvo[GB];
#pragma unroll
    for (int i = 0; i < GB; ++i) {
        const int row = (wave + 4 * i) * 8 + lrow;
        const int chunk = lpos ^ ((row >> 1) & 7);
        bvo[i] = (unsigned)((tile_n * BN + row) * a.K * 4 + chunk * 16);
    }
    const int kb = ks * a.kchunk;                // first K-step of this workgroup (0 unless K is split)
    const int nk = min(a.K / SBK, kb + a.kchunk) - kb;
    int kh, kw, coff;
    if (MODE == REG_STEM) {
        kh = kb; kw = 0; coff = 0;
    } else {
        const int tap = (kb * SBK) / a.Cin;
        coff = kb * SBK - tap * a.Cin;
        kh = tap / a.KW;
        kw = tap - kh * a.KW;
    }
    unsigned avo[RA], okbits = 0, bad = 0;
    f32x4 ra[RA];
    f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};

#define FD_TAP()                                                                                  \
    do {                                                                                          \
        const unsigned toff_ = (unsigned)((kh * a.W + kw) * a.Cin * 4);                           \
        okbits = 0;                                                                               \
        _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                          \
            const int hi_ = hi0[i] + kh, wi_ = wi0[i] + (MODE == REG_STEM ? c4 : kw);             \
            const bool ok_ = mval[i] && (unsigned)hi_ < (unsigned)a.H && (unsigned)wi_ < (unsigned)a.W; \
            avo[i] = ok_ ? abase[i] + toff_ : OOB;                                                \
            okbits |= ok_ ? (1u << i) : 0u;                                                       \
        }                                                                                         \
    } while (0)

    // global loads of K-step KT: activations into registers, weights straight into stage STG
#define FD_GLOAD(KT, STG)                                                                         \
    do {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < RA; ++i)                                            \
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, avo[i], coff * 4, 0)); \
        if (MODE == REG_PRE) {                                                                    \
            psc = *reinterpret_cast<const f32x4*>(a.pre_scale + coff + c4 * 4);                   \
            psh = *reinterpret_cast<const f32x4*>(a.pre_shift + coff + c4 * 4);                   \
        }                                                                                         \
        lok = okbits;                                                                             \
        float* sa_ = smem + (STG) * STAGE;                                                        \
        _Pragma("unroll") for (int i = 0; i < GB; ++i)                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                             \
                wr, (__attribute__((address_space(3))) void*)(sa_ + BM * 32 + (wave + 4 * i) * 256), 16, \
                bvo[i], (kb + (KT)) * 128, 0, 0);                                                        \
        if (MODE == REG_STEM) {                                                                   \
            ++kh;                                                                                 \
            FD_TAP();                                                                             \
        } else {                                                                                  \
            coff += SBK;                                                                          \
            if (coff >= a.Cin) {                                                                  \
                coff = 0;                                                                         \
                if (++kw == a.KW) { kw = 0; ++kh; }                                               \
                FD_TAP();                                                                         \
            }                                                                                     \
        }                                                                                         \
    } while (0)

    // registers -> LDS stage STG: BatchNorm+ReLU+split (PRE) or the staged halves as they are (STEM)
#define FD_LSTORE(STG)                                                                            \
    do {                                                                                          \
        char* sb_ = reinterpret_cast<char*>(smem) + (STG) * STAGE * 4;                            \
        _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                          \
            f16x4 h_, l_;                                                                         \
            if (F32) {                                                                            \
                f32x4 v_;                                                                         \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                   \
                    float t_ = ra[i][j] * psc[j] + psh[j];                                        \
                    if (!((lok >> i) & 1u)) t_ = 0.f;                                             \
                    v_[j] = fmaxf(t_, 0.f);                                                       \
                }                                                                                 \
                *reinterpret_cast<f32x4*>(sb_ + a_lds[i]) = v_;                                   \
                continue;                                                                         \
            }                                                                                     \
            if (MODE == REG_PRE) {                                                                \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                   \
                    float t_ = ra[i][j] * psc[j] + psh[j];                                        \
                    if (!((lok >> i) & 1u)) t_ = 0.f;                                             \
                    _Float16 hh_, ll_;                                                            \
                    split_f32_relu(t_, hh_, ll_, bad);                                            \
                    h_[j] = hh_; l_[j] = ll_;                                                     \
                }                                                                                 \
            } else {                                                                              \
                const float2 p0_ = make_float2(ra[i][0], ra[i][1]), p1_ = make_float2(ra[i][2], ra[i][3]); \
                h_ = __builtin_bit_cast(f16x4, p0_);                                              \
                l_ = __builtin_bit_cast(f16x4, p1_);                                              \
            }                                                                                     \
            *reinterpret_cast<f16x4*>(sb_ + a_lds[i]) = h_;                                       \
            *reinterpret_cast<f16x4*>(sb_ + (a_lds[i] ^ 64u)) = l_;                               \
        }                                                                                         \
    } while (0)

    f32x4 acc[T::AM][T::AN];
#pragma unroll
    for (int mi = 0; mi < T::AM; ++mi)
#pragma unroll
        for (int ni = 0; ni < T::AN; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fh = lane >> 4;
    const int sw = (frow >> 1) & 7;
    const char* sbase = reinterpret_cast<const char*>(smem);
    int a_off[2], b_off[2];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int pos = ((part * 4 + fh) ^ sw) * 16;
        a_off[part] = (wm * (BM / 2) + frow) * 128 + pos;
        b_off[part] = (BM + wn * (BN / 2) + frow) * 128 + pos;
    }
    unsigned lok = 0;     // validity bits that belong to the registers in flight

    // bf16x3: K-steps from `half` on are accumulated negated (see conv_split_glds_kernel)
    const int half = (B3 && a.b3_balance > 0 && nk > 1) ? max(1, min(nk - 1, (nk * a.b3_balance + 50) / 100)) : nk + 1;
    unsigned flip = 0u;
#define FD_NEGATE()                                                                               \
    do {                                                                                          \
        _Pragma("unroll") for (int mi = 0; mi < T::AM; ++mi)                                      \
            _Pragma("unroll") for (int ni = 0; ni < T::AN; ++ni) acc[mi][ni] = -acc[mi][ni];      \
    } while (0)
#define FD_STEP(CUR, KT)                                                                          \
    do {                                                                                          \
        const bool more_ = (KT) + 1 < nk;                                                         \
        if (more_) FD_GLOAD((KT) + 1, (CUR) ^ 1);                                                 \
        if (B3) {                                                                                 \
            if ((KT) == half) { FD_NEGATE(); flip = 0x80000000u; }                                \
            bf16x3_mma_step<BM, BN>(sbase + (CUR) * STAGE * 4, a_off, b_off, acc, flip);          \
        } else if (F32) f32_mma_step<BM, BN>(sbase + (CUR) * STAGE * 4, a_off, b_off, acc);       \
        else split_mma_step<BM, BN>(sbase + (CUR) * STAGE * 4, a_off, b_off, acc);                \
        if (more_) FD_LSTORE((CUR) ^ 1);                                                          \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
        __syncthreads();                                                                          \
    } while (0)

    FD_TAP();
    FD_GLOAD(0, 0);
    FD_LSTORE(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        FD_STEP(0, kt);
        if (kt + 1 < nk) FD_STEP(1, kt + 1);
    }
    if (B3 && flip) FD_NEGATE();
#undef FD_STEP
#undef FD_NEGATE
#undef FD_LSTORE
#undef FD_GLOAD
#undef FD_TAP
    if (MODE == REG_PRE && !F32) split_report(a.clamped, bad);
    if (a.ksplit > 1) {
        SplitKArgs e = a;
        e.y32 = a.partial + (size_t)ks * a.M * a.Cout;
        e.ysplit = nullptr; e.bias = nullptr; e.res = nullptr; e.relu = 0;
        split_epilogue16<BM, BN>(e, smem, tile_m, tile_n, wave, lane, acc);
    } else {
        split_epilogue16<BM, BN>(a, smem, tile_m, tile_n, wave, lane, acc);
    }
#endif
}

template <int BM, int BN, int MODE, bool F32 = false, bool B3 = false>
static int launch_split_reg(const SplitKArgs& a, size_t x_bytes, size_t w_bytes, hipStream_t stream) {
    static_assert(!F32 || MODE == REG_PRE, "the float32 form exists for the BatchNorm-on-load kernel only");
    if constexpr (F32 && !B3) {
        if (a.f32 == 2) return launch_split_reg<BM, BN, MODE, true, true>(a, x_bytes, w_bytes, stream);
    }
    static const std::string name = std::string(B3 ? "conv_b3_" : F32 ? "conv_f32_" : "conv_split_") + std::to_string(BM) + "x" +
                                    std::to_string(BN) + (MODE == REG_STEM ? "_stem" : "_pre");
    const double taps = MODE == REG_STEM ? 147.0 : (double)a.K;
    const double cin_bytes = MODE == REG_STEM ? 16.0 : 4.0 * a.Cin;
    ProfScope scope(stream, name.c_str(), 2.0 * a.M * a.Cout * taps,
                    (double)a.H * a.W * (a.M / ((double)a.Ho * a.Wo)) * cin_bytes + 4.0 * a.Cout * taps +
                        4.0 * a.M * a.Cout * ((a.res ? 1.0 : 0.0) + (a.y32 ? 1.0 : 0.0) + (a.ysplit ? 1.0 : 0.0)));
    SplitKArgs k = a;
    const int tiles_m = ceil_div(a.M, BM);
    k.tiles_n = ceil_div(a.Cout, BN);
    hipLaunchKernelGGL((conv_split_reg_kernel<BM, BN, MODE, F32, B3>), dim3(tiles_m * k.tiles_n * k.ksplit), dim3(256), 0,
                       stream, k, (unsigned)x_bytes, (unsigned)w_bytes);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

template <int BM, int BN, bool STEM, bool PRE = false>
static int launch_split(const SplitKArgs& a, hipStream_t stream) {
    static const std::string name = std::string("conv_split_") + std::to_string(BM) + "x" + std::to_string(BN) +
                                    (STEM ? "_stem" : "") + (PRE ? "_pre" : "");
    const double taps = STEM ? 147.0 : (double)a.K;
    const double cin_bytes = STEM ? 16.0 : 4.0 * a.Cin;
    ProfScope scope(stream, name.c_str(), 2.0 * a.M * a.Cout * taps,
                    (double)a.H * a.W * (a.M / ((double)a.Ho * a.Wo)) * cin_bytes + 4.0 * a.Cout * taps +
                        4.0 * a.M * a.Cout * ((a.res ? 1.0 : 0.0) + (a.y32 ? 1.0 : 0.0) + (a.ysplit ? 1.0 : 0.0)));
    SplitKArgs k = a;
    const int tiles_m = ceil_div(a.M, BM);
    k.tiles_n = ceil_div(a.Cout, BN);
    hipLaunchKernelGGL((conv_split_kernel<BM, BN, STEM, PRE>), dim3(tiles_m * k.tiles_n), dim3(256), 0, stream, k);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

// Sum of the split-K partial tensors, then the part of the epilogue the partial launches left out:
// bias, ReLU, residual, float32 and/or split output.  One thread per 8 channels of a row.
__global__ __launch_bounds__(256) void split_k_finish_kernel(SplitKArgs a) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c8n = a.Cout / 8;
    if (t >= (long long)a.M * c8n) return;
    const int col = (int)(t % c8n) * 8;
    const long long row = t / c8n;
    const size_t off = (size_t)row * a.Cout + col, plane = (size_t)a.M * a.Cout;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    for (int s = 0; s < a.ksplit; ++s) {
        v0 += *reinterpret_cast<const f32x4*>(a.partial + s * plane + off);
        v1 += *reinterpret_cast<const f32x4*>(a.partial + s * plane + off + 4);
    }
    if (a.bias) {
        v0 += *reinterpret_cast<const f32x4*>(a.bias + col);
        v1 += *reinterpret_cast<const f32x4*>(a.bias + col + 4);
    }
    if (a.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v0[j] = fmaxf(v0[j], 0.f); v1[j] = fmaxf(v1[j], 0.f); }
    }
    if (a.res) {
        v0 += *reinterpret_cast<const f32x4*>(a.res + off);
        v1 += *reinterpret_cast<const f32x4*>(a.res + off + 4);
    }
    if (a.y32) {
        *reinterpret_cast<f32x4*>(a.y32 + off) = v0;
        *reinterpret_cast<f32x4*>(a.y32 + off + 4) = v1;
    }
    if (a.ysplit) {
        if (a.post_scale) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(a.post_scale + col);
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(a.post_scale + col + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(a.post_shift + col);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(a.post_shift + col + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v0[j] = fmaxf(v0[j] * s0[j] + t0[j], 0.f);
                v1[j] = fmaxf(v1[j] * s1[j] + t1[j], 0.f);
            }
        }
        if (a.f32) {
            float* ap = reinterpret_cast<float*>(a.ysplit) + off;
            *reinterpret_cast<f32x4*>(ap) = v0;
            *reinterpret_cast<f32x4*>(ap + 4) = v1;
            return;
        }
        f16x8 hi, lo;
        unsigned bad = 0;
        split8(v0, v1, hi, lo, bad);
        split_report(a.clamped, bad);
        uint8_t* sp = a.ysplit + (size_t)row * a.Cout * 4 + split_offset(col);
        *reinterpret_cast<f16x8*>(sp) = hi;
        *reinterpret_cast<f16x8*>(sp + 64) = lo;
    }
}

// The convolutions of the ROI head see 7x7 or 4x4 pixels per roi: a few hundred GEMM rows, so a
// 64x64-tile launch leaves most CUs idle while each workgroup runs one long serial K loop.  There
// K is cut into ranges summed by a second kernel.  The number of ranges is a function of K alone
// (as long as the partial sums fit the workspace), never of the batch size, so a frame's result
// does not depend on how many frames share the launch.
static int choose_split_k(int pixels_per_image, int nk, long long out_floats, size_t ws_floats, int* kchunk) {
    *kchunk = nk;
    if (pixels_per_image > 49 || nk < 16 || ws_floats == 0) return 1;
    int want = nk / 8;
    if (want > 8) want = 8;
    if ((size_t)want * (size_t)out_floats > ws_floats) return 1;
    *kchunk = ceil_div(nk, want);
    return ceil_div(nk, *kchunk);
}

static int split_k_finish(const SplitKArgs& a, hipStream_t stream) {
    const long long items = (long long)a.M * (a.Cout / 8);
    ProfScope scope(stream, "split_k_finish", 0.0, 4.0 * a.M * a.Cout * (a.ksplit + 2.0));
    hipLaunchKernelGGL(split_k_finish_kernel, dim3((unsigned)ceil_div(items, 256)), dim3(256), 0, stream, a);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

static int split_conv_dispatch(const SplitConvParams& p, const void* x_split, const float* pre_scale,
                               const float* pre_shift, int batch, int h, int w, const float* residual,
                               const SplitConvOut& out, hipStream_t stream) {
    const ConvShape& s = p.s;
    FD_REQUIRE(x_split && p.w && (out.y32 || out.ysplit), "split conv: NULL tensor");
    FD_REQUIRE(batch > 0 && h > 0 && w > 0, "split conv: bad input shape %d x %d x %d", batch, h, w);
    const bool stem = (s.cin == 4);
    FD_REQUIRE(stem ? (s.kwp == 8) : (s.cin % 32 == 0 && s.kwp == s.kw),
               "split conv: input channels must be a multiple of 32 (or the staged image), got %d", s.cin);
    FD_REQUIRE(s.cout % 8 == 0, "split conv: output channels must be a multiple of 8, got %d", s.cout);
    FD_REQUIRE(!out.ysplit || s.cout % 32 == 0, "split conv: a split output needs a multiple of 32 channels");
    const int ho = s.out_h(h), wo = s.out_w(w);
    FD_REQUIRE(ho > 0 && wo > 0, "split conv: empty output");
    const long long m_ll = (long long)batch * ho * wo;
    FD_REQUIRE(m_ll < (1ll << 31) / 4, "split conv: too many output pixels (%lld)", m_ll);
    SplitKArgs a;
    a.x = static_cast<const uint8_t*>(x_split);
    a.w = static_cast<const uint8_t*>(p.w);
    a.bias = p.bias; a.res = residual; a.y32 = out.y32; a.ysplit = static_cast<uint8_t*>(out.ysplit);
    a.post_scale = out.post_scale; a.post_shift = out.post_shift; a.w_inv_scale = p.w_inv_scale;
    a.pre_scale = pre_scale; a.pre_shift = pre_shift;
    a.H = h; a.W = w; a.Cin = s.cin; a.Ho = ho; a.Wo = wo; a.Cout = s.cout; a.KW = s.kwp;
    a.stride = s.stride; a.pad = s.pad; a.K = s.k(); a.M = (int)m_ll; a.relu = p.relu; a.tiles_n = 0;
    a.ksplit = 1; a.kchunk = a.K / SBK; a.partial = nullptr; a.clamped = out.clamped; a.f32 = p.f32 ? 1 : 0;
    // FD_BF16X3=1 / FD_B3_BALANCE=<percent>: overrides for probes of the operator entry points
    static const int b3_env = getenv("FD_BF16X3") ? atoi(getenv("FD_BF16X3")) : 0;
    static const int b3_bal = getenv("FD_B3_BALANCE") ? atoi(getenv("FD_B3_BALANCE")) : -1;
    if (p.f32 && (out.b3 || b3_env)) a.f32 = 2;
    a.b3_balance = b3_bal >= 0 ? b3_bal : out.b3_balance;
    static const int stream_res_env = getenv("FD_STREAM_RES") ? atoi(getenv("FD_STREAM_RES")) : 0;
    a.stream_res = stream_res_env;
    FD_REQUIRE(!(stem && pre_scale), "split conv: the stem takes the staged image only");
    // the buffer-addressed kernels use 32-bit byte offsets below 2^31 into both tensors
    const size_t x_bytes = (size_t)batch * h * w * s.cin * 4, w_bytes = s.packed_floats() * 4;
    const bool small = x_bytes < (1ull << 31) && w_bytes < (1ull << 31);
    if (stem) {
        if (small && s.cout <= 64) return launch_split_reg<128, 64, REG_STEM>(a, x_bytes, w_bytes, stream);
        return launch_split<128, 64, true>(a, stream);
    }
    const long long t128 = (long long)ceil_div(a.M, 128) * ceil_div(s.cout, 128);
    const long long t12864 = (long long)ceil_div(a.M, 128) * ceil_div(s.cout, 64);
    if (p.f32) {
        // float32 operands on the fp32 matrix cores: the same two kernel families (direct-to-LDS for
        // activated inputs, BatchNorm-on-load through registers).  Measured on the layers of the 1080p
        // network, 128 x 128 tiles with two workgroups per CU are as fast as the eight-wave 256-row tiles
        // (the fp32 matrix pipe, not the path into LDS, is the limit), so only the four-wave shapes are used.
        FD_REQUIRE(small && !stem, "float32 convolution: tensor beyond the 2^31-byte reach of the buffer-addressed kernels");
        a.ksplit = choose_split_k(ho * wo, a.K / SBK, (long long)a.M * s.cout, out.splitk_ws_floats, &a.kchunk);
        if (a.ksplit > 1) {
            a.partial = out.splitk_ws;
            // (BatchNorm on load with K split: the register-staged kernel is the faster one, 58 against 63 us on the
            // ROI head's K = 2048 layers at 160 rois, the same at 10)
            if (pre_scale) FD_TRY((launch_split_reg<64, 64, REG_PRE, true>(a, x_bytes, w_bytes, stream)));
            else if (a.M >= 1024 && s.cout >= 512) FD_TRY((launch_split_glds<128, 128, 2, 4, true>(a, x_bytes, w_bytes, stream)));
            else FD_TRY((launch_split_glds<64, 64, 4, 4, true>(a, x_bytes, w_bytes, stream)));
            return split_k_finish(a, stream);
        }
        // FD_F32_TILE / FD_F32_TILE_PRE (probes): one tile shape for every layer of the kernel family that can take it
        static const int tile_env = getenv("FD_F32_TILE") ? atoi(getenv("FD_F32_TILE")) : 0;
        static const int tile_pre_env = getenv("FD_F32_TILE_PRE") ? atoi(getenv("FD_F32_TILE_PRE")) : 0;
        if (pre_scale) {
            if (tile_pre_env == 1 && s.cout > 64) return launch_split_reg<128, 128, REG_PRE, true>(a, x_bytes, w_bytes, stream);
            if (tile_pre_env == 2) return launch_split_reg<128, 64, REG_PRE, true>(a, x_bytes, w_bytes, stream);
            if (tile_pre_env == 3) return launch_split_reg<64, 64, REG_PRE, true>(a, x_bytes, w_bytes, stream);
            // FD_F32_BN_FRAG=0 (read at every call, so that a test can compare the two forms in one process): the
            // register-staged kernel for every such layer
            const char* frag_env = getenv("FD_F32_BN_FRAG");
            const bool frag = !(frag_env && atoi(frag_env) == 0) && a.f32 == 1 && tile_pre_env == 0 && a.stride == 1 && a.pad == 0 &&
                              a.K == a.Cin;
            if (frag && s.cout > 64 && t128 >= 384 && a.K <= 1024)
                return launch_split_glds_pre<128, 128, 2, 0, 1024>(a, x_bytes, w_bytes, stream);
            // (K = 2048: the ROI head's layers with thousands of rois; 80 KB of LDS, still two workgroups per CU)
            if (frag && s.cout > 64 && t128 >= 384 && a.K <= 2048)
                return launch_split_glds_pre<128, 128, 2, 0, 2048>(a, x_bytes, w_bytes, stream);
            if (frag && t12864 >= 1536 && a.K <= 256) return launch_split_glds_pre<128, 64, 2, 3, 256>(a, x_bytes, w_bytes, stream);
            if (s.cout > 64 && t128 >= 384) return launch_split_reg<128, 128, REG_PRE, true>(a, x_bytes, w_bytes, stream);
            // a single frame's 64-channel layers (1 013 tiles of 128 x 64): 59 -> 52 us on 64 x 64 tiles; from two
            // frames on the taller tile is as fast or faster
            if (t12864 >= (s.cout <= 64 ? 1536 : 384)) {
                if (frag && a.K <= 1024) return launch_split_glds_pre<128, 64, 3, 0, 1024>(a, x_bytes, w_bytes, stream);
                return launch_split_reg<128, 64, REG_PRE, true>(a, x_bytes, w_bytes, stream);
            }
            // (64 output channels on 64 x 64 tiles, a single frame's layers of the 64-channel stage: 52-53 us on the
            // register-staged kernel, 56-58 us on this one; with 256 channels 41 against 48.5 us)
            if (frag && s.cout > 64 && a.K <= 2048) return launch_split_glds_pre<64, 64, 4, 0, 2048>(a, x_bytes, w_bytes, stream);
            return launch_split_reg<64, 64, REG_PRE, true>(a, x_bytes, w_bytes, stream);
        }
        if (a.f32 != 2 && tile_env) {
            if (tile_env == 1 && s.cout > 64) return launch_split_glds<128, 128, 2, 4, true>(a, x_bytes, w_bytes, stream);
            if (tile_env == 2 && s.cout > 64) return launch_split_glds<128, 64, 2, 4, true, 3>(a, x_bytes, w_bytes, stream);
            if (tile_env == 3) return launch_split_glds<128, 64, 3, 4, true>(a, x_bytes, w_bytes, stream);
            if (tile_env == 4) return launch_split_glds<64, 64, 4, 4, true>(a, x_bytes, w_bytes, stream);
        }
        if (a.f32 == 2) {
            // bf16x3 products: a K-step lasts 3/8 as long as with float32 products, so the two-stage four-wave tiles
            // no longer keep a step's loads a memory round trip ahead; the eight-wave 256-row tiles (one workgroup per
            // CU, as for f16x3) run longer steps and convert each operand value for twice as many multiplies
            static const int b3_big = getenv("FD_B3_BIG") ? atoi(getenv("FD_B3_BIG")) : 1;
            const long long t256 = (long long)ceil_div(a.M, 256) * ceil_div(s.cout, 128);
            if (b3_big == 1 && s.cout >= 256 && a.K >= 128 && t256 >= 448)
                return launch_split_glds<256, 256, 2, 8, true, 0, true>(a, x_bytes, w_bytes, stream);
            if (b3_big && s.cout > 64 && a.K >= 128 && t256 >= 448)
                return launch_split_glds<256, 128, 3, 8, true, 0, true>(a, x_bytes, w_bytes, stream);
        }
        // K = 256 layers (8 K-steps per tile): the per-tile prologue and epilogue weigh as much as the loop, and
        // three workgroups per CU on 128 x 64 tiles (two pipeline stages, 48 KB of LDS, 105 registers) cover
        // them better than two on 128 x 128: 107-117 -> 125 TF on those layers, +1.6 % frames/s.  Shorter K
        // (the layers bound by their own output traffic; also tried: 64 x 64 tiles at five workgroups per CU)
        // and longer K measured no gain.
        if (a.f32 == 1 && conv_panel_applies(a)) return conv_panel_forward(a, x_bytes, w_bytes, stream);
        // Measured again layer by layer at micro-batch 16 with one tile shape forced per run (FD_F32_TILE): the
        // three-workgroup tile is also 3-5 % faster on the K = 128 layers of the 128-channel stage (789 -> 745 us) and
        // on the K = 512 projection layers (1 073 -> 1 032 us), which is +0.2 % frames/s over the whole pass -- inside
        // the box-to-box spread, so those layers stay on 128 x 128 (FD_F32_SHORTK=512 moves them).
        // Small grids (single frames): a launch whose last round of tiles is mostly empty takes the shape that fills
        // its rounds better -- 1 024 tiles of 128 x 64 are 1.33 rounds of 768 resident workgroups, the same layer as
        // 512 tiles of 128 x 128 is exactly one round of 512 (50 -> 43.5 us).  The tile shape does not change the
        // order in which an output element accumulates over K, so results do not depend on the choice.
        static const int shortk = getenv("FD_F32_SHORTK") ? atoi(getenv("FD_F32_SHORTK")) : 256;
        if ((a.K > 128 || shortk > 256) && a.K <= shortk && s.cout > 64 && t12864 >= 768) {
            const double fill_o3 = (double)t12864 / ((double)ceil_div(t12864, 768) * 768.0);
            const double fill_128 = (double)t128 / ((double)ceil_div(t128, 512) * 512.0);
            if (!(fill_o3 < 0.8 && fill_128 >= 0.95 && t128 >= 384))
                return launch_split_glds<128, 64, 2, 4, true, 3>(a, x_bytes, w_bytes, stream);
        }
        if (s.cout > 64 && t128 >= 384) return launch_split_glds<128, 128, 2, 4, true>(a, x_bytes, w_bytes, stream);
        if (t12864 >= 384) return launch_split_glds<128, 64, 3, 4, true>(a, x_bytes, w_bytes, stream);
        return launch_split_glds<64, 64, 4, 4, true>(a, x_bytes, w_bytes, stream);
    }
    if (small) {
        // ROI-head layers: split K whenever the workspace allows, whatever the batch size
        a.ksplit = choose_split_k(ho * wo, a.K / SBK, (long long)a.M * s.cout, out.splitk_ws_floats, &a.kchunk);
        if (a.ksplit > 1) {
            a.partial = out.splitk_ws;
            if (pre_scale) FD_TRY((launch_split_reg<64, 64, REG_PRE>(a, x_bytes, w_bytes, stream)));
            // larger tiles re-read the weights (the dominant traffic here) half as often; the tile shape
            // does not change the order in which an output element accumulates its K range, so a
            // frame's result still does not depend on how many frames share the launch
            else if (a.M >= 1024 && s.cout >= 512) FD_TRY((launch_split_glds<128, 128, 2>(a, x_bytes, w_bytes, stream)));
            else FD_TRY((launch_split_glds<64, 64, 4>(a, x_bytes, w_bytes, stream)));
            return split_k_finish(a, stream);
        }
    }
    if (pre_scale) {
        if (small) {
            if (s.cout > 64 && t128 >= 384) return launch_split_reg<128, 128, REG_PRE>(a, x_bytes, w_bytes, stream);
            if (t12864 >= 384) return launch_split_reg<128, 64, REG_PRE>(a, x_bytes, w_bytes, stream);
            return launch_split_reg<64, 64, REG_PRE>(a, x_bytes, w_bytes, stream);
        }
        if (s.cout > 64 && t128 >= 384) return launch_split<128, 128, false, true>(a, stream);
        if (t12864 >= 384) return launch_split<128, 64, false, true>(a, stream);
        return launch_split<64, 64, false, true>(a, stream);
    }
    if (small) {
        // Large grids run eight-wave workgroups on 256-row tiles (one per CU): per multiply they move
        // half (256 x 256) or three quarters (256 x 128) of the bytes from L2 into LDS and out of
        // LDS into registers that the 128 x 128 tile does (RPN 3x3: 1476 -> 1319 us; 752 -> 805
        // frames/s over all layers, same box, two streams).
        const long long t256 = (long long)ceil_div(a.M, 256) * ceil_div(s.cout, 128);
        if (s.cout >= 256 && a.K >= 128 && t256 >= 448) return launch_split_glds<256, 256, 2, 8>(a, x_bytes, w_bytes, stream);
        if (s.cout > 64 && a.K >= 128 && t256 >= 448) return launch_split_glds<256, 128, 3, 8>(a, x_bytes, w_bytes, stream);
        if (s.cout > 64 && t128 >= 384) return launch_split_glds<128, 128, 2>(a, x_bytes, w_bytes, stream);
        if (t12864 >= 384) return launch_split_glds<128, 64, 3>(a, x_bytes, w_bytes, stream);
        return launch_split_glds<64, 64, 4>(a, x_bytes, w_bytes, stream);
    }
    if (s.cout > 64 && t128 >= 384) return launch_split<128, 128, false>(a, stream);
    if (t12864 >= 384) return launch_split<128, 64, false>(a, stream);
    return launch_split<64, 64, false>(a, stream);
}

int gemm_f32_batched(const float* a_mat, const float* w, int nb, int rows_pad, int k, int cout, int cout_pad, float* y,
                     hipStream_t stream, bool b3, int b3_balance) {
    FD_REQUIRE(a_mat && w && y && nb > 0 && rows_pad > 0 && rows_pad % 128 == 0 && k % 32 == 0 && cout % 8 == 0 &&
                   cout_pad % 128 == 0 && cout <= cout_pad, "gemm_f32_batched: bad arguments");
    const size_t x_bytes = (size_t)nb * rows_pad * k * 4, w_bytes = (size_t)nb * cout_pad * k * 4;
    FD_REQUIRE(x_bytes < (1ull << 31) && w_bytes < (1ull << 31) && (long long)nb * rows_pad < (1ll << 29),
               "gemm_f32_batched: operands beyond the 2^31-byte reach of the buffer-addressed kernel");
    SplitKArgs a;
    a.x = reinterpret_cast<const uint8_t*>(a_mat);
    a.w = reinterpret_cast<const uint8_t*>(w);
    a.bias = nullptr; a.res = nullptr; a.y32 = nullptr; a.ysplit = reinterpret_cast<uint8_t*>(y);
    a.post_scale = a.post_shift = nullptr; a.pre_scale = a.pre_shift = nullptr; a.w_inv_scale = 1.f;
    a.H = 1; a.W = 1; a.Cin = k; a.Ho = 1; a.Wo = 1; a.Cout = cout; a.KW = 1; a.stride = 1; a.pad = 0; a.K = k;
    a.M = nb * rows_pad; a.relu = 0; a.tiles_n = 0; a.ksplit = 1; a.kchunk = k / SBK; a.partial = nullptr;
    a.clamped = nullptr; a.f32 = 1;
    static const int b3_env = getenv("FD_BF16X3") ? atoi(getenv("FD_BF16X3")) : 0;
    static const int b3_bal = getenv("FD_B3_BALANCE") ? atoi(getenv("FD_B3_BALANCE")) : -1;
    if (b3 || b3_env) a.f32 = 2;
    a.b3_balance = b3_bal >= 0 ? b3_bal : b3_balance;
    a.wbatch_tiles = rows_pad / 128;
    a.wbatch_bytes = (unsigned)((size_t)cout_pad * k * 4);
    // K <= 256 (the 256-channel stage): three workgroups per CU on 128 x 64 tiles, as for the K = 256 layers
    // of split_conv_dispatch (+0.5 % frames/s)
    static const int gemm_tile = getenv("FD_F32_GEMM_TILE") ? atoi(getenv("FD_F32_GEMM_TILE")) : 0;     // probes
    if (gemm_tile == 1) return launch_split_glds<128, 128, 2, 4, true>(a, x_bytes, w_bytes, stream);
    if (gemm_tile == 2) return launch_split_glds<128, 64, 2, 4, true, 3>(a, x_bytes, w_bytes, stream);
    if (gemm_tile == 3) return launch_split_glds<128, 64, 3, 4, true>(a, x_bytes, w_bytes, stream);
    if (gemm_tile == 4) {
        a.wbatch_tiles = rows_pad / 64;
        return launch_split_glds<64, 64, 4, 4, true>(a, x_bytes, w_bytes, stream);
    }
    static const int gemm_o3 = getenv("FD_F32_GEMM_O3") ? atoi(getenv("FD_F32_GEMM_O3")) : 1;
    if (gemm_o3 && k <= 256) return launch_split_glds<128, 64, 2, 4, true, 3>(a, x_bytes, w_bytes, stream);
    // long K on a small grid (the RPN's layer for a single frame: 576 tiles of 128 x 128 are 1.13 rounds of 512
    // resident workgroups): the three-workgroup tile fills its rounds better (197 -> 164 us)
    const long long t128 = (long long)(a.M / 128) * ceil_div(cout, 128), t12864 = (long long)(a.M / 128) * ceil_div(cout, 64);
    const double fill_128 = (double)t128 / ((double)ceil_div(t128, 512) * 512.0);
    const double fill_o3 = (double)t12864 / ((double)ceil_div(t12864, 768) * 768.0);
    if (gemm_o3 && fill_128 < 0.7 && fill_o3 > fill_128) return launch_split_glds<128, 64, 2, 4, true, 3>(a, x_bytes, w_bytes, stream);
    return launch_split_glds<128, 128, 2, 4, true>(a, x_bytes, w_bytes, stream);
}

int gemm_split_batched(const void* a_split, const void* w_split, unsigned w_set_bytes, float w_inv_scale, int nb, int rows_pad,
                       int k, int cout, float* y, hipStream_t stream) {
    FD_REQUIRE(a_split && w_split && y && nb > 0 && rows_pad > 0 && rows_pad % 256 == 0 && k % 32 == 0 && cout % 8 == 0,
               "gemm_split_batched: bad arguments");
    const size_t x_bytes = (size_t)nb * rows_pad * k * 4, w_bytes = (size_t)nb * w_set_bytes;
    FD_REQUIRE(x_bytes < (1ull << 31) && w_bytes < (1ull << 31) && (long long)nb * rows_pad < (1ll << 29),
               "gemm_split_batched: operands beyond the 2^31-byte reach of the buffer-addressed kernel");
    SplitKArgs a;
    a.x = static_cast<const uint8_t*>(a_split);
    a.w = static_cast<const uint8_t*>(w_split);
    a.bias = nullptr; a.res = nullptr; a.y32 = y; a.ysplit = nullptr;
    a.post_scale = a.post_shift = nullptr; a.pre_scale = a.pre_shift = nullptr; a.w_inv_scale = w_inv_scale;
    a.H = 1; a.W = 1; a.Cin = k; a.Ho = 1; a.Wo = 1; a.Cout = cout; a.KW = 1; a.stride = 1; a.pad = 0; a.K = k;
    a.M = nb * rows_pad; a.relu = 0; a.tiles_n = 0; a.ksplit = 1; a.kchunk = k / SBK; a.partial = nullptr;
    a.clamped = nullptr; a.f32 = 0;
    a.wbatch_tiles = rows_pad / 256;
    a.wbatch_bytes = w_set_bytes;
    // (256 x 128 and 128 x 128 tiles measured the same on the K = 256 GEMMs of the 256-channel stage)
    if (cout >= 256) return launch_split_glds<256, 256, 2, 8>(a, x_bytes, w_bytes, stream);
    return launch_split_glds<256, 128, 3, 8>(a, x_bytes, w_bytes, stream);
}

int split_conv_forward_pair(const SplitConvParams& pa, const SplitConvParams& pb, const float* w_pair, size_t w_pair_floats,
                            const float* x, int batch, int h, int w, float* ya, float* yb, hipStream_t stream, bool* taken) {
    *taken = false;
    const ConvShape &sa = pa.s, &sb = pb.s;
    // Opt-in (FD_F32_PAIR=1, read at every call so that a test can compare the two forms in one process): the pair
    // is bound by its 2.6 GB of output, saves 0.1 ms per 16 frames (+0.3-0.5 % frames/s) and one launch, and runs
    // on the three-workgroup tile -- the pass's dominant kernel, whose per-launch average then mixes a memory-bound
    // launch (76 TF) with its matrix-bound ones; kept off so that the kernel's roofline record stays what its
    // matrix-bound layers reach.
    const char* env = getenv("FD_F32_PAIR");
    if (!(env && atoi(env) != 0) || !pa.f32 || !pb.f32 || !w_pair || sa.cin != sb.cin || sa.cin % 32 != 0 || sa.kh != 1 || sa.kw != 1 ||
        sb.kh != 1 || sb.kw != 1 || sa.stride != 1 || sb.stride != 1 || sa.pad != 0 || sb.pad != 0 || sa.cout % 64 != 0 ||
        sb.cout % 32 != 0 || sb.cout > 64)
        return FD_OK;
    const long long m_ll = (long long)batch * h * w;
    const size_t x_bytes = (size_t)m_ll * sa.cin * 4, w_bytes = w_pair_floats * 4;
    if (x_bytes >= (1ull << 31) || w_bytes >= (1ull << 31) || m_ll >= (1ll << 29) || ceil_div(m_ll, 128) < 768) return FD_OK;
    FD_REQUIRE(x && ya && yb, "paired convolution: NULL tensor");
    SplitKArgs a;
    a.x = reinterpret_cast<const uint8_t*>(x);
    a.w = reinterpret_cast<const uint8_t*>(w_pair);
    a.bias = pa.bias; a.res = nullptr; a.y32 = ya; a.ysplit = nullptr; a.post_scale = a.post_shift = nullptr;
    a.pre_scale = a.pre_shift = nullptr; a.w_inv_scale = 1.f;
    a.H = h; a.W = w; a.Cin = sa.cin; a.Ho = h; a.Wo = w; a.Cout = sa.cout + 64; a.KW = 1; a.stride = 1; a.pad = 0;
    a.K = sa.cin; a.M = (int)m_ll; a.relu = pa.relu; a.tiles_n = 0; a.ksplit = 1; a.kchunk = a.K / SBK; a.partial = nullptr;
    a.clamped = nullptr; a.f32 = 1;
    a.alt_from = sa.cout; a.alt_y = reinterpret_cast<uint8_t*>(yb); a.alt_cout = sb.cout; a.alt_bias = pb.bias;
    a.alt_relu = pb.relu;
    *taken = true;
    return launch_split_glds<128, 64, 2, 4, true, 3>(a, x_bytes, w_bytes, stream);
}

int split_conv_forward_two(const SplitConvParams& p3, const float* w_cat, const float* x1, int batch, int ho, int wo,
                           const float* x2, int h2, int w2, int cin2, int stride2, const SplitConvOut& out, hipStream_t stream,
                           bool* taken) {
    *taken = false;
    const ConvShape& s = p3.s;
    const long long m_ll = (long long)batch * ho * wo;
    const int k = s.cin + cin2;
    const size_t x_bytes = (size_t)m_ll * s.cin * 4, x2_bytes = (size_t)batch * h2 * w2 * cin2 * 4, w_bytes = (size_t)s.cout * k * 4;
    if (!p3.f32 || out.b3 || !w_cat || !x2 || s.kh != 1 || s.kw != 1 || s.stride != 1 || s.pad != 0 || s.cin % 32 || cin2 % 32 ||
        s.cout % 8 || x_bytes >= (1ull << 31) || x2_bytes >= (1ull << 31) || w_bytes >= (1ull << 31) || m_ll >= (1ll << 29) ||
        (ho - 1) * stride2 >= h2 || (wo - 1) * stride2 >= w2)
        return FD_OK;
    FD_REQUIRE(x1 && (out.y32 || out.ysplit) && batch > 0, "two-source convolution: NULL tensor");
    SplitKArgs a;
    a.x = reinterpret_cast<const uint8_t*>(x1);
    a.w = reinterpret_cast<const uint8_t*>(w_cat);
    a.bias = p3.bias; a.res = nullptr; a.y32 = out.y32; a.ysplit = static_cast<uint8_t*>(out.ysplit);
    a.post_scale = out.post_scale; a.post_shift = out.post_shift; a.w_inv_scale = p3.w_inv_scale;
    a.pre_scale = a.pre_shift = nullptr;
    a.H = ho; a.W = wo; a.Cin = s.cin; a.Ho = ho; a.Wo = wo; a.Cout = s.cout; a.KW = 1; a.stride = 1; a.pad = 0; a.K = k;
    a.M = (int)m_ll; a.relu = p3.relu; a.tiles_n = 0; a.ksplit = 1; a.kchunk = k / SBK; a.partial = nullptr; a.clamped = out.clamped;
    a.f32 = 1;
    a.x2 = reinterpret_cast<const uint8_t*>(x2); a.Cin2 = cin2; a.H2 = h2; a.W2 = w2; a.stride2 = stride2; a.x2_bytes = (unsigned)x2_bytes;
    // the tile shape does not change the order in which an output element accumulates over K
    const long long t128 = (long long)ceil_div(a.M, 128) * ceil_div(s.cout, 128);
    const long long t12864 = (long long)ceil_div(a.M, 128) * ceil_div(s.cout, 64);
    *taken = true;
    if (s.cout > 64 && t128 >= 384) return launch_split_glds_two<128, 128, 2>(a, x_bytes, w_bytes, stream);
    if (t12864 >= 384) return launch_split_glds_two<128, 64, 3>(a, x_bytes, w_bytes, stream);
    return launch_split_glds_two<64, 64, 4>(a, x_bytes, w_bytes, stream);
}

int split_conv_forward(const SplitConvParams& p, const void* x_split, int batch, int h, int w, const float* residual,
                       const SplitConvOut& out, hipStream_t stream) {
    return split_conv_dispatch(p, x_split, nullptr, nullptr, batch, h, w, residual, out, stream);
}

int split_conv_forward_f32(const SplitConvParams& p, const float* x32, const float* pre_scale, const float* pre_shift,
                           int batch, int h, int w, const float* residual, const SplitConvOut& out,
                           hipStream_t stream) {
    FD_REQUIRE(pre_scale && pre_shift, "split conv: the float32-input form needs the BatchNorm scale and shift");
    return split_conv_dispatch(p, x32, pre_scale, pre_shift, batch, h, w, residual, out, stream);
}

// ---------------------------------------------------------------------------------
// Weight packing: OIHW float32 -> [cout_pad][K/32][32 hi | 32 lo] of w * out_scale[o] * wscale
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ w, int cout, int cin_w, int kh, int kw,
                                                        int kwp, int cin_p, int cout_pad,
                                                        const float* __restrict__ scale, float wscale,
                                                        uint8_t* __restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long K = (long long)kh * kwp * cin_p;
    if (t >= (long long)cout_pad * K) return;
    const int k = (int)(t % K);
    const int o = (int)(t / K);
    const int c = k % cin_p;
    const int x = (k / cin_p) % kwp;
    const int y = k / (cin_p * kwp);
    float v = 0.f;
    if (o < cout && x < kw && c < cin_w) {
        v = w[(((size_t)o * cin_w + c) * kh + y) * kw + x];
        if (scale) v *= scale[o];
        v *= wscale;
    }
    _Float16 hi, lo;
    unsigned bad = 0;                  // the host chose wscale so that every weight is in range
    split_f32(v, hi, lo, bad);
    uint8_t* row = out + (size_t)o * K * 4 + (size_t)(k >> 5) * 128 + (size_t)(k & 31) * 2;
    *reinterpret_cast<_Float16*>(row) = hi;
    *reinterpret_cast<_Float16*>(row + 64) = lo;
}

int pack_conv_weights_split(const float* w_oihw, int cin_w, const ConvShape& s, const float* out_scale, float wscale,
                            void* packed, hipStream_t stream) {
    FD_REQUIRE(w_oihw && packed && cin_w > 0 && cin_w <= s.cin && s.k() % 32 == 0, "pack_conv_weights_split: bad arguments");
    const long long total = (long long)s.cout_pad() * s.k();
    hipLaunchKernelGGL(pack_split_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, w_oihw, s.cout, cin_w, s.kh,
                       s.kw, s.kwp, s.cin, s.cout_pad(), out_scale, wscale, static_cast<uint8_t*>(packed));
    FD_LAUNCH_CHECK();
    return FD_OK;
}

// ---------------------------------------------------------------------------------
// float32 [rows][C] -> split [rows][C], optionally through relu(x * scale[c] + shift[c])
__global__ __launch_bounds__(256) void to_split_kernel(const float* __restrict__ x, long long rows, int C,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      uint8_t* __restrict__ y, unsigned* __restrict__ clamped, int f32) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;     // one thread per 8 channels
    const int c8n = C / 8;
    if (t >= rows * c8n) return;
    const int c = (int)(t % c8n) * 8;
    const long long r = t / c8n;
    f32x4 v0 = *reinterpret_cast<const f32x4*>(x + r * C + c);
    f32x4 v1 = *reinterpret_cast<const f32x4*>(x + r * C + c + 4);
    if (scale) {
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + c), s1 = *reinterpret_cast<const f32x4*>(scale + c + 4);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(shift + c), t1 = *reinterpret_cast<const f32x4*>(shift + c + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v0[j] = fmaxf(v0[j] * s0[j] + t0[j], 0.f);
            v1[j] = fmaxf(v1[j] * s1[j] + t1[j], 0.f);
        }
    }
    if (f32) {
        float* ap = reinterpret_cast<float*>(y) + r * C + c;
        *reinterpret_cast<f32x4*>(ap) = v0;
        *reinterpret_cast<f32x4*>(ap + 4) = v1;
        return;
    }
    f16x8 hi, lo;
    unsigned bad = 0;
    split8(v0, v1, hi, lo, bad);
    split_report(clamped, bad);
    uint8_t* sp = y + (size_t)r * C * 4 + split_offset(c);
    *reinterpret_cast<f16x8*>(sp) = hi;
    *reinterpret_cast<f16x8*>(sp + 64) = lo;
}

int to_split(const float* x, long long rows, int channels, const float* scale, const float* shift, void* y,
             unsigned* clamped, hipStream_t stream, bool f32) {
    FD_REQUIRE(x && y && channels % 32 == 0, "to_split: channels must be a multiple of 32");
    if (rows <= 0) return FD_OK;
    ProfScope scope(stream, "to_split", 0.0, 8.0 * (double)rows * channels);
    hipLaunchKernelGGL(to_split_kernel, dim3(ceil_div(rows * (channels / 8), 256)), dim3(256), 0, stream, x, rows,
                       channels, scale, shift, static_cast<uint8_t*>(y), clamped, f32 ? 1 : 0);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

// ---------------------------------------------------------------------------------
// uint8 frames -> split staged image [B][H][W][4 hi | 4 lo] with bn_data applied
__global__ __launch_bounds__(256) void stage_input_split_kernel(const uint8_t* __restrict__ f, long long pixels,
                                                               InputNorm n, f16x8* __restrict__ y,
                                                               unsigned* __restrict__ clamped) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= pixels) return;
    const uint8_t* p = f + t * 3;
    f16x8 o;
    unsigned bad = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        _Float16 hi, lo;
        split_f32((float)p[c] * n.scale[c] + n.shift[c], hi, lo, bad);
        o[c] = hi;
        o[4 + c] = lo;
    }
    o[3] = (_Float16)0.f;
    o[7] = (_Float16)0.f;
    y[t] = o;
    split_report(clamped, bad);
}

int stage_input_split(const uint8_t* frames, int batch, int h, int w, const InputNorm& norm, void* y,
                      unsigned* clamped, hipStream_t stream) {
    FD_REQUIRE(frames && y, "stage_input_split: NULL tensor");
    const long long pixels = (long long)batch * h * w;
    ProfScope scope(stream, "stage_input", 0.0, 19.0 * (double)pixels);
    hipLaunchKernelGGL(stage_input_split_kernel, dim3(ceil_div(pixels, 256)), dim3(256), 0, stream, frames, pixels, norm,
                       static_cast<f16x8*>(y), clamped);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

// ---------------------------------------------------------------------------------
// uint8 frames -> planar staged image for the fused stem: a plane of hi halves and a plane of
// lo halves, each [B][H][Wpad][4] fp16 with the pixel (y, x) in column x + 3 and zeros in the
// other columns (the convolution's horizontal padding, stored), bn_data applied.
int staged_planar_width(int w) { return (w + 8 + 1) & ~1; }

__global__ __launch_bounds__(256) void stage_input_planar_kernel(const uint8_t* __restrict__ f, int W, int Wpad,
                                                                long long slots, InputNorm n, f16x4* __restrict__ hi,
                                                                f16x4* __restrict__ lo, unsigned* __restrict__ clamped) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;      // one slot (row, column) of a plane
    if (t >= slots) return;
    const int col = (int)(t % Wpad);
    const long long row = t / Wpad;                                      // image row over the whole batch
    f16x4 h = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f}, l = h;
    const int x = col - 3;
    unsigned bad = 0;
    if ((unsigned)x < (unsigned)W) {
        const uint8_t* p = f + (row * W + x) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            _Float16 a, b;
            split_f32((float)p[c] * n.scale[c] + n.shift[c], a, b, bad);
            h[c] = a;
            l[c] = b;
        }
    }
    hi[t] = h;
    lo[t] = l;
    split_report(clamped, bad);
}

// The float32 form of that image: one plane [B][H][Wpad][4] floats (16 bytes per column), same zero columns.
__global__ __launch_bounds__(256) void stage_input_padded_f32_kernel(const uint8_t* __restrict__ f, int W, int Wpad,
                                                                    long long slots, InputNorm n, float4* __restrict__ y) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= slots) return;
    const int col = (int)(t % Wpad);
    const long long row = t / Wpad;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int x = col - 3;
    if ((unsigned)x < (unsigned)W) {
        const uint8_t* p = f + (row * W + x) * 3;
        v = make_float4((float)p[0] * n.scale[0] + n.shift[0], (float)p[1] * n.scale[1] + n.shift[1],
                        (float)p[2] * n.scale[2] + n.shift[2], 0.f);
    }
    y[t] = v;
}

int stage_input_padded_f32(const uint8_t* frames, int batch, int h, int w, const InputNorm& norm, void* y,
                           hipStream_t stream) {
    FD_REQUIRE(frames && y, "stage_input_padded_f32: NULL tensor");
    const int wpad = staged_planar_width(w);
    const long long slots = (long long)batch * h * wpad;
    ProfScope scope(stream, "stage_input", 0.0, 3.0 * batch * h * (double)w + 16.0 * (double)slots);
    hipLaunchKernelGGL(stage_input_padded_f32_kernel, dim3((unsigned)ceil_div(slots, 256)), dim3(256), 0, stream, frames,
                       w, wpad, slots, norm, static_cast<float4*>(y));
    FD_LAUNCH_CHECK();
    return FD_OK;
}

int stage_input_planar(const uint8_t* frames, int batch, int h, int w, const InputNorm& norm, void* y,
                       unsigned* clamped, hipStream_t stream) {
    FD_REQUIRE(frames && y, "stage_input_planar: NULL tensor");
    const int wpad = staged_planar_width(w);
    const long long slots = (long long)batch * h * wpad;
    ProfScope scope(stream, "stage_input", 0.0, 3.0 * batch * h * (double)w + 16.0 * (double)slots);
    f16x4* hi = static_cast<f16x4*>(y);
    hipLaunchKernelGGL(stage_input_planar_kernel, dim3((unsigned)ceil_div(slots, 256)), dim3(256), 0, stream, frames, w,
                       wpad, slots, norm, hi, hi + slots, clamped);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

// ---------------------------------------------------------------------------------
// 3x3/2 max pooling of a float32 map, written as float32 (optional) and as the split of
// relu(max * scale[c] + shift[c]) for the first residual unit.
__global__ __launch_bounds__(256) void maxpool_split_kernel(const float* __restrict__ x, int B, int H, int W, int C, int Ho,
                                                           int Wo, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float* __restrict__ y32,
                                                           uint8_t* __restrict__ ysplit, unsigned* __restrict__ clamped,
                                                           int f32) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c8n = C / 8;
    if (t >= (long long)B * Ho * Wo * c8n) return;
    const int c = (int)(t % c8n) * 8;
    const long long pix = t / c8n;
    const int wo = (int)(pix % Wo);
    const int ho = (int)((pix / Wo) % Ho);
    const int b = (int)(pix / ((long long)Wo * Ho));
    f32x4 m0 = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX}, m1 = m0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int hi = ho * 2 - 1 + dy;
        if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int wi = wo * 2 - 1 + dx;
            if ((unsigned)wi >= (unsigned)W) continue;
            const float* p = x + (((size_t)b * H + hi) * W + wi) * C + c;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { m0[j] = fmaxf(m0[j], v0[j]); m1[j] = fmaxf(m1[j], v1[j]); }
        }
    }
    if (y32) {
        *reinterpret_cast<f32x4*>(y32 + pix * C + c) = m0;
        *reinterpret_cast<f32x4*>(y32 + pix * C + c + 4) = m1;
    }
    if (ysplit) {
        if (scale) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + c), s1 = *reinterpret_cast<const f32x4*>(scale + c + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(shift + c), t1 = *reinterpret_cast<const f32x4*>(shift + c + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m0[j] = fmaxf(m0[j] * s0[j] + t0[j], 0.f);
                m1[j] = fmaxf(m1[j] * s1[j] + t1[j], 0.f);
            }
        }
        if (f32) {
            float* ap = reinterpret_cast<float*>(ysplit) + pix * C + c;
            *reinterpret_cast<f32x4*>(ap) = m0;
            *reinterpret_cast<f32x4*>(ap + 4) = m1;
            return;
        }
        f16x8 hi, lo;
        unsigned bad = 0;
        split8(m0, m1, hi, lo, bad);
        split_report(clamped, bad);
        uint8_t* sp = ysplit + (size_t)pix * C * 4 + split_offset(c);
        *reinterpret_cast<f16x8*>(sp) = hi;
        *reinterpret_cast<f16x8*>(sp + 64) = lo;
    }
}

int maxpool_3x3s2_split(const float* x, int batch, int h, int w, int c, const float* scale, const float* shift,
                        float* y32, void* ysplit, unsigned* clamped, hipStream_t stream, bool f32) {
    FD_REQUIRE(x && (y32 || ysplit) && c % 32 == 0, "maxpool (split): bad arguments");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)batch * ho * wo * (c / 8);
    ProfScope scope(stream, "maxpool_3x3s2", 0.0,
                    4.0 * c * ((double)batch * h * w + (double)batch * ho * wo * ((y32 ? 1 : 0) + (ysplit ? 1 : 0))));
    hipLaunchKernelGGL(maxpool_split_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, x, batch, h, w, c, ho, wo,
                       scale, shift, y32, static_cast<uint8_t*>(ysplit), clamped, f32 ? 1 : 0);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

}  // namespace fd

// =================================================================================
// C ABI: the convolution operator of fd_conv2d computed with split-precision operands
// =================================================================================
extern "C" {

int fd_conv2d_f16x3(fd_ctx* ctx, const float* x_nhwc, int batch, int height, int width, int cin, const float* w_oihw,
                    int cout, int kh, int kw, int stride, int pad, const float* bias, const float* pre_scale,
                    const float* pre_shift, int relu, const float* residual, float* y_nhwc, void* stream) {
    FD_REQUIRE(ctx && x_nhwc && w_oihw && y_nhwc, "fd_conv2d_f16x3: NULL argument");
    FD_REQUIRE(cin % 32 == 0 && cin > 0 && cout > 0 && cout % 8 == 0,
               "fd_conv2d_f16x3: cin must be a multiple of 32 and cout of 8 (got %d, %d)", cin, cout);
    hipStream_t st = fd::as_stream(stream);
    fd::SplitConvParams p;
    p.s.cin = cin; p.s.cout = cout; p.s.kh = kh; p.s.kw = kw; p.s.kwp = kw; p.s.stride = stride; p.s.pad = pad;
    // weight scale from the largest magnitude (blocking read-back: this entry point is for tests)
    const size_t nw = (size_t)cout * cin * kh * kw;
    std::vector<float> host(nw);
    FD_HIP(hipStreamSynchronize(st));
    FD_HIP(hipMemcpy(host.data(), w_oihw, nw * sizeof(float), hipMemcpyDeviceToHost));
    double wmax = 0.0;
    for (float v : host) wmax = std::fmax(wmax, std::fabs((double)v));
    float wscale = 1.f;
    if (wmax > 0.0 && std::isfinite(wmax)) wscale = (float)std::ldexp(1.0, 13 - (int)std::floor(std::log2(wmax)));
    void* packed = nullptr;
    void* xs = nullptr;
    FD_TRY(fd::scratch(ctx, fd::SLOT_MISC0, p.s.packed_floats() * sizeof(float), &packed));
    FD_TRY(fd::scratch(ctx, fd::SLOT_MISC1, (size_t)batch * height * width * cin * sizeof(float), &xs));
    FD_TRY(fd::pack_conv_weights_split(w_oihw, cin, p.s, nullptr, wscale, packed, st));
    p.w = packed; p.w_inv_scale = 1.0f / wscale; p.bias = bias; p.relu = relu;
    fd::SplitConvOut o;
    o.y32 = y_nhwc;
    // with a BatchNorm+ReLU prologue the kernel converts float32 input while staging it
    if (pre_scale && pre_shift)
        return fd::split_conv_forward_f32(p, x_nhwc, pre_scale, pre_shift, batch, height, width, residual, o, st);
    FD_TRY(fd::to_split(x_nhwc, (long long)batch * height * width, cin, nullptr, nullptr, xs, nullptr, st));
    return fd::split_conv_forward(p, xs, batch, height, width, residual, o, st);
}

}  // extern "C"
