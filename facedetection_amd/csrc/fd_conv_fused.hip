// Fused kernels of the split-precision path (operand format: fd_split.h, shared device code:
// fd_conv_split_dev.h):
//   conv_tail_fused_kernel  - 3x3 + following 1x1 of a 64-channel bottleneck (tensor between them in LDS).
//   unit_fused_kernel       - a whole identity-shortcut unit of the 64-channel stage (8 x 16 pixel
//                             tiles; both intermediate tensors stay in LDS).
//   unit_proj_fused_kernel  - the projection-shortcut unit of that stage, shortcut convolution
//                             accumulated with the main branch.
//   stem_pool_kernel        - the 7x7/2 stem fused with the 3x3/2 max-pool (2-D tiles, operands read
//                             straight from the input image in LDS).
#include "fd_conv_split_dev.h"

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <string>

namespace fd {

// ---------------------------------------------------------------------------------
// Bottleneck tail in one kernel: the 3x3 convolution of a residual unit and the 1x1 convolution
// that follows it, for units whose middle width is 64 channels (stage 1).  A workgroup computes a
// 128 x 64 tile of the 3x3 output -- all 64 channels of 128 pixels, i.e. the complete K of the
// 1x1 -- keeps it in LDS as the split A operand (bias + ReLU applied), and runs the 1x1 against
// it in four chunks of 64 output channels with the usual epilogue (residual add, float32 and/or
// split output).  The intermediate tensor never goes to HBM (530 MB per 8 frames at 1080p and
// unit), and the compute-heavy and the traffic-heavy phase of co-resident workgroups overlap.
// LDS: [epilogue staging 18 KB | A operand 32 KB | weight chunk 16 KB]; the K loop of the 3x3
// uses the first 48 KB as its two pipeline stages before that.
// F32: float32 operands on the fp32 matrix cores (same LDS image: a row of a K-step is 32 floats).  That
// variant runs THREE workgroups per CU: its weight chunk shares the LDS of the epilogue staging (52 KB per
// workgroup), so a chunk's weights are fetched after the previous chunk's epilogue instead of during it, and
// only one set of residual registers is live (<= 168 registers); what a workgroup no longer overlaps inside
// itself, the third workgroup covers.
// B3 (with F32): products on the bf16 matrix cores (bf16x3_mma_step), both loops sign-balanced as in the
// direct-to-LDS kernel.
template <int BM, int MID, bool F32, bool B3 = false>
__global__ __launch_bounds__(256, F32 && !B3 ? 3 : 2) void conv_tail_fused_kernel(SplitKArgs a, SplitKArgs c, unsigned x_bytes,
                                                                unsigned w_bytes, unsigned w3_bytes) {
    static_assert(!B3 || F32, "bf16x3 products are formed from float32 operand images");
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(BM == 128 && MID == 64, "written for 128-pixel tiles of a 64-channel middle layer");
    constexpr int BN = MID;
    using T = SplitTile<BM, BN>;
    using E = SplitEpi<BM, BN>;
    constexpr int GA = BM / 32, GB = BN / 32;
    constexpr int STAGE = T::STAGE;                       // dwords
    constexpr int CS_BYTES = 20 * 1024;                   // epilogue staging (18 KB used)
    constexpr int AIMG_BYTES = (MID / 32) * BM * 128;     // the 1x1's A operand: MID/32 K-steps of BM rows
    constexpr int WCH_BYTES = (MID / 32) * 64 * 128;      // one chunk of 64 output channels of the 1x1's weights
    constexpr bool SHARED_WCH = F32;                      // the weight chunk overlays the epilogue staging
    constexpr int WCH_OFF = SHARED_WCH ? 0 : CS_BYTES + AIMG_BYTES;
    constexpr int SMEM_BYTES = CS_BYTES + AIMG_BYTES + (SHARED_WCH ? 0 : WCH_BYTES);
    static_assert(2 * STAGE * 4 <= SMEM_BYTES && T::EPI * 4 <= CS_BYTES && WCH_BYTES <= (SHARED_WCH ? CS_BYTES : WCH_BYTES) &&
                      SMEM_BYTES <= (SHARED_WCH ? 53 : 80) * 1024,
                  "LDS plan");
    __shared__ __attribute__((aligned(16))) float smem[SMEM_BYTES / 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile_m = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3r = __builtin_amdgcn_make_buffer_rsrc((void*)c.w, 0, w3_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- phase 1: the 3x3 convolution, as in conv_split_glds_kernel<128, 64, 2> ----
    const int lrow = lane >> 3, lpos = lane & 7;
    int hi0[GA], wi0[GA];
    unsigned abase[GA];
    bool mval[GA];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int row = (wave + 4 * i) * 8 + lrow;
        const int chunk = lpos ^ ((row >> 1) & 7);
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
        abase[i] = (unsigned)(((img * a.H + hi0[i]) * a.W + wi0[i]) * a.Cin * 4 + chunk * 16);
    }
    unsigned bvo[GB];
#pragma unroll
    for (int i = 0; i < GB; ++i) {
        const int row = (wave + 4 * i) * 8 + lrow;
        bvo[i] = (unsigned)(row * a.K * 4 + (lpos ^ ((row >> 1) & 7)) * 16);
    }
    int kh = 0, kw = 0, coff = 0;
    const int nk = a.K / SBK;
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
        _Pragma("unroll") for (int i = 0; i < GA; ++i)                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                             \
                xr, (__attribute__((address_space(3))) void*)(sa_ + (wave + 4 * i) * 256), 16, avo[i], coff * 4, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < GB; ++i)                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                             \
                wr, (__attribute__((address_space(3))) void*)(sa_ + BM * 32 + (wave + 4 * i) * 256), 16, \
                bvo[i], (KT) * 128, 0, 0);                                                        \
        coff += SBK;                                                                              \
        if (coff >= a.Cin) {                                                                      \
            coff = 0;                                                                             \
            if (++kw == a.KW) { kw = 0; ++kh; }                                                   \
            FD_TAP();                                                                             \
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
    const int half = (B3 && a.b3_balance > 0) ? max(1, min(nk - 1, (nk * a.b3_balance + 50) / 100)) : nk + 1;
    unsigned flip = 0u;
#define FD_STEP(CUR, KT)                                                                          \
    do {                                                                                          \
        if ((KT) + 1 < nk) FD_ISSUE((KT) + 1, (CUR) ^ 1);                                         \
        if (B3) {                                                                                 \
            if ((KT) == half) {                                                                   \
                _Pragma("unroll") for (int mi = 0; mi < T::AM; ++mi)                              \
                    _Pragma("unroll") for (int ni = 0; ni < T::AN; ++ni) acc[mi][ni] = -acc[mi][ni]; \
                flip = 0x80000000u;                                                               \
            }                                                                                     \
            bf16x3_mma_step<BM, BN>(sbase + (CUR) * STAGE * 4, a_off, b_off, acc, flip);          \
        } else if (F32) f32_mma_step<BM, BN>(sbase + (CUR) * STAGE * 4, a_off, b_off, acc);       \
        else split_mma_step<BM, BN>(sbase + (CUR) * STAGE * 4, a_off, b_off, acc);                \
        wait_vmcnt<0>();                                                                          \
        __syncthreads();                                                                          \
    } while (0)
    FD_TAP();
    FD_ISSUE(0, 0);
    wait_vmcnt<0>();
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        FD_STEP(0, kt);
        if (kt + 1 < nk) FD_STEP(1, kt + 1);
    }
    if (B3 && flip) {
#pragma unroll
        for (int mi = 0; mi < T::AM; ++mi)
#pragma unroll
            for (int ni = 0; ni < T::AN; ++ni) acc[mi][ni] = -acc[mi][ni];
    }
#undef FD_STEP
#undef FD_ISSUE
#undef FD_TAP

    // ---- phase 2: bias + ReLU + split of the 128 x 64 tile into the 1x1's A operand in LDS ----
    // (transposed through the staging area like the epilogue, 32 rows at a time; a wave's 32
    // columns are exactly one K-step of the 1x1: K-step index = wn)
    char* aimg = reinterpret_cast<char*>(smem) + CS_BYTES;
    {
        unsigned bad2 = 0;
        constexpr int LDC = T::LDC;
        float* Cs = smem + wave * (32 * LDC);
        const int erow = lane / E::L8, ecol = (lane % E::L8) * 8;
        float bias_v[T::AN];
#pragma unroll
        for (int ni = 0; ni < T::AN; ++ni) bias_v[ni] = a.bias ? a.bias[wn * 32 + ni * 16 + (lane & 15)] : 0.f;
#pragma unroll
        for (int mi = 0; mi < E::GROUPS; ++mi) {
            if (mi > 0) __syncthreads();
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int ni = 0; ni < T::AN; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Cs[(mb * 16 + (lane >> 4) * 4 + r) * LDC + ni * 16 + (lane & 15)] =
                            fmaxf(acc[mi * 2 + mb][ni][r] * a.w_inv_scale + bias_v[ni], 0.f);
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < E::PASSES; ++ps) {
                const int lrow2 = ps * E::RPP + erow;
                const int row = wm * (BM / 2) + mi * 32 + lrow2;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(&Cs[lrow2 * LDC + ecol]);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(&Cs[lrow2 * LDC + ecol + 4]);
                char* kstep = aimg + wn * (BM * 128);
                if (F32) {
                    // 8 channels = two 16-byte chunks of the row's 32 floats
                    const int c4 = ecol >> 2, swz = (row >> 1) & 7;
                    *reinterpret_cast<f32x4*>(kstep + row * 128 + ((c4 ^ swz) * 16)) = v0;
                    *reinterpret_cast<f32x4*>(kstep + row * 128 + (((c4 + 1) ^ swz) * 16)) = v1;
                } else {
                    f16x8 hi, lo;
                    split8(v0, v1, hi, lo, bad2);
                    const int off = row * 128 + (((ecol >> 3) ^ ((row >> 1) & 7)) * 16);
                    *reinterpret_cast<f16x8*>(kstep + off) = hi;
                    *reinterpret_cast<f16x8*>(kstep + (off ^ 64)) = lo;      // lo chunk = hi chunk + 4
                }
            }
        }
        if (!F32) split_report(c.clamped, bad2);
    }

    // ---- phase 3: the 1x1 convolution, 64 output channels at a time ----
    float* wch = smem + WCH_OFF / 4;
    int a3_off[MID / 32][2], b3_off[MID / 32][2];
#pragma unroll
    for (int kk = 0; kk < MID / 32; ++kk)
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const int pos = ((part * 4 + fh) ^ sw) * 16;
            a3_off[kk][part] = CS_BYTES + kk * (BM * 128) + (wm * (BM / 2) + frow) * 128 + pos;
            b3_off[kk][part] = WCH_OFF + kk * (64 * 128) + (wn * 32 + frow) * 128 + pos;
        }
    // Software pipeline over the chunks: the weights and the residual values of chunk ch + 1 are
    // requested before the epilogue of chunk ch runs, so neither round trip is exposed.  vmcnt
    // counts in order: the weight loads of a chunk are always older than its residual loads.
    constexpr int RL = E::GROUPS * E::PASSES * 2;        // residual loads per chunk and lane
#define FD_WEIGHTS(CH)                                                                            \
    do {                                                                                          \
        _Pragma("unroll") for (int kk = 0; kk < MID / 32; ++kk)                                   \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                       \
                const int row = (wave + 4 * i) * 8 + lrow;                                        \
                const unsigned vo = (unsigned)((((CH) * 64 + row) * MID + kk * 32) * 4 + (lpos ^ ((row >> 1) & 7)) * 16); \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                         \
                    w3r, (__attribute__((address_space(3))) void*)(wch + kk * (64 * 32) + (wave + 4 * i) * 256), 16, vo, \
                    0, 0, 0);                                                                     \
            }                                                                                     \
    } while (0)
#define FD_CHUNK(CH, RCUR, RNEXT)                                                                 \
    do {                                                                                          \
        wait_vmcnt<RL>();              /* weights of this chunk landed; its residual may fly */   \
        __syncthreads();               /* ... for every wave; first chunk: A operand complete */  \
        f32x4 acc3[T::AM][T::AN];                                                                 \
        _Pragma("unroll") for (int mi = 0; mi < T::AM; ++mi)                                      \
            _Pragma("unroll") for (int ni = 0; ni < T::AN; ++ni) acc3[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f}; \
        _Pragma("unroll") for (int kk = 0; kk < MID / 32; ++kk) {                                 \
            if (F32) FD_MMA3(kk);                                                                 \
            else split_mma_step<BM, BN>(sbase, a3_off[kk], b3_off[kk], acc3);                     \
        }                                                                                         \
        __syncthreads();               /* every wave is done reading the weight chunk */          \
        if ((CH) + 1 < nch) {                                                                     \
            FD_WEIGHTS((CH) + 1);                                                                 \
            split_residual_prefetch<BM, BN>(c, rows, (CH) + 1, wave, lane, RNEXT);              \
            __builtin_amdgcn_sched_barrier(0);                                                    \
        }                                                                                         \
        split_epilogue16<BM, BN, true>(c, smem, tile_m, (CH), wave, lane, acc3, RCUR);            \
    } while (0)

    // K-step kk of the 1x1 (two in all); bf16x3: the second one accumulated negated
#define FD_MMA3(KK)                                                                               \
    do {                                                                                          \
        if (B3) {                                                                                 \
            const bool neg_ = c.b3_balance > 0 && (KK) == MID / 32 - 1 && MID / 32 > 1;           \
            if (neg_) {                                                                           \
                _Pragma("unroll") for (int mi = 0; mi < T::AM; ++mi)                              \
                    _Pragma("unroll") for (int ni = 0; ni < T::AN; ++ni) acc3[mi][ni] = -acc3[mi][ni]; \
            }                                                                                     \
            bf16x3_mma_step<BM, BN>(sbase, a3_off[KK], b3_off[KK], acc3, neg_ ? 0x80000000u : 0u); \
            if (neg_) {                                                                           \
                _Pragma("unroll") for (int mi = 0; mi < T::AM; ++mi)                              \
                    _Pragma("unroll") for (int ni = 0; ni < T::AN; ++ni) acc3[mi][ni] = -acc3[mi][ni]; \
            }                                                                                     \
        } else {                                                                                  \
            f32_mma_step<BM, BN>(sbase, a3_off[KK], b3_off[KK], acc3);                            \
        }                                                                                         \
    } while (0)

    const int nch = c.Cout / 64;
    const LinearRows rows{tile_m * BM, c.M};
    if constexpr (SHARED_WCH) {
        f32x4 res[E::GROUPS][E::PASSES][2];
        for (int ch = 0; ch < nch; ++ch) {
            __syncthreads();               // the shared region is free: phase 2 / the previous epilogue are done with it
            FD_WEIGHTS(ch);
            split_residual_prefetch<BM, BN>(c, rows, ch, wave, lane, res);
            __builtin_amdgcn_sched_barrier(0);
            wait_vmcnt<RL>();              // weights of this chunk landed; its residual may fly
            __syncthreads();
            f32x4 acc3[T::AM][T::AN];
#pragma unroll
            for (int mi = 0; mi < T::AM; ++mi)
#pragma unroll
                for (int ni = 0; ni < T::AN; ++ni) acc3[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < MID / 32; ++kk) FD_MMA3(kk);
            // (the epilogue's own first barrier separates the last weight read from the staging writes)
            split_epilogue16<BM, BN, true>(c, smem, tile_m, ch, wave, lane, acc3, res);
        }
    } else {
        f32x4 resA[E::GROUPS][E::PASSES][2], resB[E::GROUPS][E::PASSES][2];
        FD_WEIGHTS(0);
        split_residual_prefetch<BM, BN>(c, rows, 0, wave, lane, resA);
        __builtin_amdgcn_sched_barrier(0);
        for (int ch = 0; ch < nch; ch += 2) {
            FD_CHUNK(ch, resA, resB);
            if (ch + 1 < nch) FD_CHUNK(ch + 1, resB, resA);
        }
    }
#undef FD_CHUNK
#undef FD_MMA3
#undef FD_WEIGHTS
#endif
}

int conv_tail_fused_forward(const SplitConvParams& p2, const SplitConvParams& p3, const void* x_split, int batch, int h,
                            int w, const float* residual, const SplitConvOut& out, hipStream_t stream, bool* taken) {
    *taken = false;
    if (p2.f32 != p3.f32) return FD_OK;
    const bool f32 = p2.f32;
    const ConvShape &s2 = p2.s, &s3 = p3.s;
    if (!(s2.cin == 64 && s2.cout == 64 && s2.kh == 3 && s2.kw == 3 && s2.kwp == 3 && s2.stride == 1 && s2.pad == 1 &&
          p2.relu && s3.cin == 64 && s3.kh == 1 && s3.kw == 1 && s3.stride == 1 && s3.pad == 0 && s3.cout % 64 == 0))
        return FD_OK;
    const long long m_ll = (long long)batch * h * w;
    const size_t x_bytes = (size_t)m_ll * 64 * 4, w_bytes = s2.packed_floats() * 4, w3_bytes = s3.packed_floats() * 4;
    if (x_bytes >= (1ull << 31) || (size_t)m_ll * s3.cout * 4 >= (1ull << 33) || m_ll >= (1ll << 29)) return FD_OK;
    FD_REQUIRE(x_split && p2.w && p3.w && (out.y32 || out.ysplit), "fused tail: NULL tensor");
    if (!residual) return FD_OK;       // the kernel's load counting assumes a residual tensor
    SplitKArgs a, c;
    a.x = static_cast<const uint8_t*>(x_split); a.w = static_cast<const uint8_t*>(p2.w);
    a.bias = p2.bias; a.res = nullptr; a.y32 = nullptr; a.ysplit = nullptr; a.post_scale = a.post_shift = nullptr;
    a.pre_scale = a.pre_shift = nullptr; a.w_inv_scale = p2.w_inv_scale;
    a.H = h; a.W = w; a.Cin = 64; a.Ho = h; a.Wo = w; a.Cout = 64; a.KW = 3; a.stride = 1; a.pad = 1; a.K = s2.k();
    a.M = (int)m_ll; a.relu = 1; a.tiles_n = 1;
    a.ksplit = 1; a.kchunk = a.K / SBK; a.partial = nullptr; a.clamped = out.clamped; a.f32 = f32 ? 1 : 0;
    c = a;
    c.x = nullptr; c.w = static_cast<const uint8_t*>(p3.w); c.bias = p3.bias; c.res = residual; c.y32 = out.y32;
    c.ysplit = static_cast<uint8_t*>(out.ysplit); c.post_scale = out.post_scale; c.post_shift = out.post_shift;
    c.w_inv_scale = p3.w_inv_scale; c.Cin = 64; c.Cout = s3.cout; c.KW = 1; c.pad = 0; c.K = 64; c.relu = p3.relu;
    c.tiles_n = s3.cout / 64;
    const int tiles_m = ceil_div(a.M, 128);
    a.b3_balance = c.b3_balance = out.b3_balance;
    ProfScope scope(stream, f32 ? (out.b3 ? "conv_tail_fused_b3" : "conv_tail_fused_f32") : "conv_tail_fused", 2.0 * a.M * 64.0 * (double)a.K + 2.0 * a.M * (double)s3.cout * 64.0,
                    4.0 * 64.0 * a.M + 4.0 * 64.0 * (a.K + s3.cout) +
                        4.0 * a.M * s3.cout * ((c.res ? 1.0 : 0.0) + (c.y32 ? 1.0 : 0.0) + (c.ysplit ? 1.0 : 0.0)));
    if (f32 && out.b3)
        hipLaunchKernelGGL((conv_tail_fused_kernel<128, 64, true, true>), dim3(tiles_m), dim3(256), 0, stream, a, c,
                           (unsigned)x_bytes, (unsigned)w_bytes, (unsigned)w3_bytes);
    else if (f32)
        hipLaunchKernelGGL((conv_tail_fused_kernel<128, 64, true>), dim3(tiles_m), dim3(256), 0, stream, a, c,
                           (unsigned)x_bytes, (unsigned)w_bytes, (unsigned)w3_bytes);
    else
        hipLaunchKernelGGL((conv_tail_fused_kernel<128, 64, false>), dim3(tiles_m), dim3(256), 0, stream, a, c,
                           (unsigned)x_bytes, (unsigned)w_bytes, (unsigned)w3_bytes);
    FD_LAUNCH_CHECK();
    *taken = true;
    return FD_OK;
}

// ---------------------------------------------------------------------------------
// A whole identity-shortcut bottleneck unit of the 64-channel stage in one kernel:
//   t1 = relu(conv1x1(relu(bn1(x))) + b1)   (256 -> 64, BatchNorm bn2 folded into the weights)
//   t2 = relu(conv3x3(t1) + b2)             (64 -> 64, bn3 folded)
//   y  = conv1x1(t2) + x                    (64 -> 256)
// A workgroup owns an 8 x 16 pixel tile of one image.  It computes t1 on the tile's 10 x 18 halo
// patch straight from the float32 input (BatchNorm+ReLU+split between the global load and the MFMA
// operand registers), keeps t1 in LDS as a split image, runs the 3x3 out of that image, keeps t2 in
// LDS as the last convolution's operand and finishes like conv_tail_fused_kernel.  Neither t1 nor
// t2 goes to HBM, and x is read from HBM once (the residual read at the end hits the caches).
//
// In the first two phases the weights are the MFMA's row operand and the pixels its column
// operand, so a lane ends up with 4 consecutive channels of one pixel: 8 bytes of hi and 8 bytes
// of lo halves that go straight into the LDS image without a transpose.
//
// LDS, phases 0-1: [t1 patch: 2 K-step planes x 192 rows x 128 B | 2 weight stages x 8 KB |
//                   bn1 scale and shift 2 KB]
//      phases 2-3: [epilogue staging 20 KB | t2 operand 32 KB | weight chunk 16 KB]
struct UnitFusedArgs {
    SplitKArgs c1, c2, c3;      // c1.x: float32 input; c3.res: the same tensor; c3.y32 / c3.ysplit: outputs
    unsigned x_bytes, w1_bytes, w2_bytes, w3_bytes;
    int tiles_x, tiles_y;
};

__global__ __launch_bounds__(256, 2) void unit_fused_kernel(UnitFusedArgs u) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128, MID = 64, BN = 64, CIN = 256;
    using T = SplitTile<BM, BN>;
    using E = SplitEpi<BM, BN>;
    constexpr int PW = 18, PPIX = 180, PROWS = 192;        // patch width, pixels, rows incl. padding
    constexpr int PLANE = PROWS * 128;                     // bytes per K-step plane of the patch
    constexpr int WST_OFF = 2 * PLANE, WSTAGE = 64 * 128;  // weight stages of phases 0 and 1
    constexpr int PS_OFF = WST_OFF + 2 * WSTAGE;           // bn1 scale [256], shift [256]
    constexpr int CS_BYTES = 20 * 1024;
    constexpr int AIMG_BYTES = (MID / 32) * BM * 128;
    constexpr int WCH_BYTES = (MID / 32) * 64 * 128;
    constexpr int WBLK = 2 * WSTAGE;                       // phase 1: both K-steps of a filter tap
    constexpr int WA = WST_OFF, WB = WST_OFF + WBLK;
    constexpr int SMEM_BYTES = WB + WBLK;
    static_assert(PS_OFF + 2 * CIN * 4 <= SMEM_BYTES && T::EPI * 4 <= CS_BYTES &&
                      CS_BYTES + AIMG_BYTES + WCH_BYTES <= SMEM_BYTES && SMEM_BYTES <= 80 * 1024,
                  "LDS plan");
    __shared__ __attribute__((aligned(16))) float smem[SMEM_BYTES / 4];
    char* const sbytes = reinterpret_cast<char*>(smem);

    const SplitKArgs &a1 = u.c1, &a2 = u.c2, &c = u.c3;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int wgid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int H = a1.H, W = a1.W;
    const int per_img = u.tiles_x * u.tiles_y;
    const int img = wgid / per_img;
    const int trem = wgid - img * per_img;
    const int tyi = trem / u.tiles_x, txi = trem - tyi * u.tiles_x;
    const int y0 = tyi * 8, x0 = txi * 16;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a1.x, 0, u.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc((void*)a1.w, 0, u.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc((void*)a2.w, 0, u.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3r = __builtin_amdgcn_make_buffer_rsrc((void*)c.w, 0, u.w3_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    const int frow = lane & 15, fh = lane >> 4;
    const int sw = (frow >> 1) & 7;
    const int lrow = lane >> 3, lpos = lane & 7;

    // bn1 scale and shift into LDS
    if (tid < 64) *reinterpret_cast<f32x4*>(sbytes + PS_OFF + tid * 16) = *reinterpret_cast<const f32x4*>(a1.pre_scale + tid * 4);
    else if (tid < 128)
        *reinterpret_cast<f32x4*>(sbytes + PS_OFF + CIN * 4 + (tid - 64) * 16) =
            *reinterpret_cast<const f32x4*>(a1.pre_shift + (tid - 64) * 4);

    // weight rows this lane moves by LDS-DMA in phases 0 and 1 (64 rows per K-step: 2 pieces per wave)
    unsigned wvo1[2], wvo2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave + 4 * i) * 8 + lrow;
        const unsigned chunk = (unsigned)((lpos ^ ((row >> 1) & 7)) * 16);
        wvo1[i] = (unsigned)(row * CIN * 4) + chunk;
        wvo2[i] = (unsigned)(row * (9 * MID) * 4) + chunk;
    }
#define FD_WDMA(RSRC, VO, KT, STG)                                                                \
    do {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                             \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                             \
                RSRC, (__attribute__((address_space(3))) void*)(sbytes + WST_OFF + (STG) * WSTAGE + (wave + 4 * i) * 1024), \
                16, VO[i], (KT) * 128, 0, 0);                                                     \
    } while (0)
    // weight fragment of 16 rows NF*16.. of stage STG, part 0: hi, part 1: lo
    int w_off[2];
#pragma unroll
    for (int part = 0; part < 2; ++part) w_off[part] = WST_OFF + frow * 128 + (((part * 4 + fh) ^ sw) * 16);

    // ---- phase 0: t1 on the halo patch; wave w owns patch rows 48 w .. 48 w + 47 ----
    unsigned xvo[3];
    bool pval[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const int pp = (wave * 3 + f) * 16 + frow;
        const int py = pp / PW, px = pp - py * PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        pval[f] = pp < PPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        xvo[f] = pval[f] ? (unsigned)(((img * H + gy) * W + gx) * (CIN * 4) + fh * 32) : OOB;
    }
    f32x4 ra[3][2];
#define FD_XLOAD(KT)                                                                              \
    do {                                                                                          \
        _Pragma("unroll") for (int f = 0; f < 3; ++f) {                                           \
            ra[f][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, xvo[f], (KT) * 128, 0)); \
            ra[f][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, xvo[f] + 16u, (KT) * 128, 0)); \
        }                                                                                         \
    } while (0)
    f32x4 acc0[4][3];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int f = 0; f < 3; ++f) acc0[nf][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned bad = 0;          // values outside the fp16 range met while splitting (input, t1, t2)

    FD_WDMA(w1r, wvo1, 0, 0);
    FD_XLOAD(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < CIN / 32; ++kt) {
        const int cur = kt & 1;
        // operand registers of this K-step from the raw values: relu(x * scale + shift), split
        f16x8 bh[3], bl[3];
        {
            const char* ps = sbytes + PS_OFF + (kt * 32 + fh * 8) * 4;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(ps), s1 = *reinterpret_cast<const f32x4*>(ps + 16);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(ps + CIN * 4), t1 = *reinterpret_cast<const f32x4*>(ps + CIN * 4 + 16);
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    _Float16 h_, l_;
                    split_f32_relu(ra[f][0][j] * s0[j] + t0[j], h_, l_, bad);
                    bh[f][j] = h_; bl[f][j] = l_;
                    split_f32_relu(ra[f][1][j] * s1[j] + t1[j], h_, l_, bad);
                    bh[f][4 + j] = h_; bl[f][4 + j] = l_;
                }
        }
        const bool more = kt + 1 < CIN / 32;
        if (more) {
            FD_WDMA(w1r, wvo1, kt + 1, cur ^ 1);
            asm volatile("" ::: "memory");
            FD_XLOAD(kt + 1);
            asm volatile("" ::: "memory");
        }
        f16x8 ah[4], al[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            ah[nf] = *reinterpret_cast<const f16x8*>(sbytes + w_off[0] + cur * WSTAGE + nf * 2048);
            al[nf] = *reinterpret_cast<const f16x8*>(sbytes + w_off[1] + cur * WSTAGE + nf * 2048);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                acc0[nf][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[nf], bh[f], acc0[nf][f], 0, 0, 0);
                acc0[nf][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[nf], bl[f], acc0[nf][f], 0, 0, 0);
                acc0[nf][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[nf], bh[f], acc0[nf][f], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        // the weights of the next K-step have landed once only the 6 younger loads are in flight
        if (more) wait_vmcnt<6>();
        pipeline_barrier();
    }
#undef FD_XLOAD
    // Phase 1 moves its weights in 16 KB blocks (both K-steps of a filter tap); tap t reads block
    // (t + 1) & 1.  Tap 0 goes out while t1 is written (the scale/shift table and the conv1 stages
    // are dead: every wave has passed the last barrier of the loop).
#define FD_WBLOCK(KT0, DST)                                                                       \
    do {                                                                                          \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_)                                       \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                         \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                         \
                    w2r, (__attribute__((address_space(3))) void*)(sbytes + (DST) + kk_ * WSTAGE + (wave + 4 * i) * 1024), \
                    16, wvo2[i], ((KT0) + kk_) * 128, 0, 0);                                      \
    } while (0)
    FD_WBLOCK(0, WB);
    {
        // a lane holds channels nf * 16 + fh * 4 + j of patch row (3 wave + f) * 16 + frow
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const f32x4 b = a1.bias ? *reinterpret_cast<const f32x4*>(a1.bias + nf * 16 + fh * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const int pp = (wave * 3 + f) * 16 + frow;
                f16x4 h4, l4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc0[nf][f][j] * a1.w_inv_scale + b[j];
                    if (!pval[f]) v = 0.f;                  // zero padding of the 3x3 applies to t1
                    _Float16 h_, l_;
                    split_f32_relu(v, h_, l_, bad);
                    h4[j] = h_; l4[j] = l_;
                }
                const int chunk = (nf & 1) * 2 + (fh >> 1);
                const int off = (nf >> 1) * PLANE + pp * 128 + ((chunk ^ ((pp >> 1) & 7)) * 16) + (fh & 1) * 8;
                *reinterpret_cast<f16x4*>(sbytes + off) = h4;
                *reinterpret_cast<f16x4*>(sbytes + (off ^ 64)) = l4;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- phase 1: the 3x3 out of the patch; wave w owns tile rows 2 w and 2 w + 1 ----
    f32x4 acc1[4][2];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) acc1[nf][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - dy * 3;
        const int rd = ((tap + 1) & 1) ? WB : WA, wr_ = ((tap + 1) & 1) ? WA : WB;
        if (tap + 1 < 9) FD_WBLOCK(2 * (tap + 1), wr_);
        int p_off[2][2];
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
            const int pp = (2 * wave + pf + dy) * PW + frow + dx;
            const int sp = (pp >> 1) & 7;
#pragma unroll
            for (int part = 0; part < 2; ++part) p_off[pf][part] = pp * 128 + (((part * 4 + fh) ^ sp) * 16);
        }
        f16x8 ah[2][4], al[2][4], bh[2][2], bl[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                ah[kk][nf] = *reinterpret_cast<const f16x8*>(sbytes + rd - WST_OFF + kk * WSTAGE + w_off[0] + nf * 2048);
                al[kk][nf] = *reinterpret_cast<const f16x8*>(sbytes + rd - WST_OFF + kk * WSTAGE + w_off[1] + nf * 2048);
            }
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                bh[kk][pf] = *reinterpret_cast<const f16x8*>(sbytes + kk * PLANE + p_off[pf][0]);
                bl[kk][pf] = *reinterpret_cast<const f16x8*>(sbytes + kk * PLANE + p_off[pf][1]);
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int pf = 0; pf < 2; ++pf) {
                    acc1[nf][pf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kk][nf], bh[kk][pf], acc1[nf][pf], 0, 0, 0);
                    acc1[nf][pf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kk][nf], bl[kk][pf], acc1[nf][pf], 0, 0, 0);
                    acc1[nf][pf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kk][nf], bh[kk][pf], acc1[nf][pf], 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
        wait_vmcnt<0>();
        pipeline_barrier();
    }
#undef FD_WBLOCK
#undef FD_WDMA

    // ---- phase 2: t2 = relu(. + b2), split, into the last convolution's operand image ----
    // (the patch and the weight stages are dead: every wave has passed the last barrier)
    float* wch = smem + (CS_BYTES + AIMG_BYTES) / 4;
#define FD_WEIGHTS(CH)                                                                            \
    do {                                                                                          \
        _Pragma("unroll") for (int kk = 0; kk < MID / 32; ++kk)                                   \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                       \
                const int row = (wave + 4 * i) * 8 + lrow;                                        \
                const unsigned vo = (unsigned)((((CH) * 64 + row) * MID + kk * 32) * 4 + (lpos ^ ((row >> 1) & 7)) * 16); \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                         \
                    w3r, (__attribute__((address_space(3))) void*)(wch + kk * (64 * 32) + (wave + 4 * i) * 256), 16, vo, \
                    0, 0, 0);                                                                     \
            }                                                                                     \
    } while (0)
    const PatchRows rows{(img * H + y0) * W + x0, W, H - y0, W - x0};
    constexpr int RL = E::GROUPS * E::PASSES * 2;        // residual loads per chunk and lane
    f32x4 resA[E::GROUPS][E::PASSES][2], resB[E::GROUPS][E::PASSES][2];
    FD_WEIGHTS(0);
    split_residual_prefetch<BM, BN>(c, rows, 0, wave, lane, resA);
    __builtin_amdgcn_sched_barrier(0);
    {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const f32x4 b = a2.bias ? *reinterpret_cast<const f32x4*>(a2.bias + nf * 16 + fh * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                const int row = (2 * wave + pf) * 16 + frow;
                f16x4 h4, l4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    _Float16 h_, l_;
                    split_f32_relu(acc1[nf][pf][j] * a2.w_inv_scale + b[j], h_, l_, bad);
                    h4[j] = h_; l4[j] = l_;
                }
                const int chunk = (nf & 1) * 2 + (fh >> 1);
                const int off = CS_BYTES + (nf >> 1) * (BM * 128) + row * 128 + ((chunk ^ ((row >> 1) & 7)) * 16) + (fh & 1) * 8;
                *reinterpret_cast<f16x4*>(sbytes + off) = h4;
                *reinterpret_cast<f16x4*>(sbytes + (off ^ 64)) = l4;
            }
        }
    }

    split_report(c.clamped, bad);

    // ---- phase 3: the last 1x1 convolution, 64 output channels at a time (as conv_tail_fused_kernel) ----
    const char* sbase = sbytes;
    int a3_off[MID / 32][2], b3_off[MID / 32][2];
#pragma unroll
    for (int kk = 0; kk < MID / 32; ++kk)
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const int pos = ((part * 4 + fh) ^ sw) * 16;
            a3_off[kk][part] = CS_BYTES + kk * (BM * 128) + (wm * (BM / 2) + frow) * 128 + pos;
            b3_off[kk][part] = CS_BYTES + AIMG_BYTES + kk * (64 * 128) + (wn * 32 + frow) * 128 + pos;
        }
#define FD_CHUNK(CH, RCUR, RNEXT)                                                                 \
    do {                                                                                          \
        wait_vmcnt<RL>();              /* weights of this chunk landed; its residual may fly */   \
        __syncthreads();               /* ... for every wave; first chunk: operand image complete */ \
        f32x4 acc3[T::AM][T::AN];                                                                 \
        _Pragma("unroll") for (int mi = 0; mi < T::AM; ++mi)                                      \
            _Pragma("unroll") for (int ni = 0; ni < T::AN; ++ni) acc3[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f}; \
        _Pragma("unroll") for (int kk = 0; kk < MID / 32; ++kk)                                   \
            split_mma_step<BM, BN>(sbase, a3_off[kk], b3_off[kk], acc3);                          \
        __syncthreads();               /* every wave is done reading the weight chunk */          \
        if ((CH) + 1 < nch) {                                                                     \
            FD_WEIGHTS((CH) + 1);                                                                 \
            split_residual_prefetch<BM, BN>(c, rows, (CH) + 1, wave, lane, RNEXT);                \
            __builtin_amdgcn_sched_barrier(0);                                                    \
        }                                                                                         \
        split_epilogue16_rows<BM, BN, true>(c, smem, rows, (CH), wave, lane, acc3, RCUR);         \
    } while (0)
    const int nch = c.Cout / 64;
    for (int ch = 0; ch < nch; ch += 2) {
        FD_CHUNK(ch, resA, resB);
        if (ch + 1 < nch) FD_CHUNK(ch + 1, resB, resA);
    }
#undef FD_CHUNK
#undef FD_WEIGHTS
#endif
}

int unit_fused_forward(const SplitConvParams& p1, const SplitConvParams& p2, const SplitConvParams& p3, const float* x32,
                       const float* pre_scale, const float* pre_shift, int batch, int h, int w, const SplitConvOut& out,
                       hipStream_t stream, bool* taken) {
    *taken = false;
    if (p1.f32 || p2.f32 || p3.f32) return FD_OK;   // split operands only
    const ConvShape &s1 = p1.s, &s2 = p2.s, &s3 = p3.s;
    if (!(s1.cin == 256 && s1.cout == 64 && s1.kh == 1 && s1.kw == 1 && s1.stride == 1 && s1.pad == 0 && p1.relu &&
          s2.cin == 64 && s2.cout == 64 && s2.kh == 3 && s2.kw == 3 && s2.kwp == 3 && s2.stride == 1 && s2.pad == 1 &&
          p2.relu && s3.cin == 64 && s3.cout == 256 && s3.kh == 1 && s3.kw == 1 && s3.stride == 1 && s3.pad == 0))
        return FD_OK;
    const long long m_ll = (long long)batch * h * w;
    const size_t x_bytes = (size_t)m_ll * 256 * 4;
    if (x_bytes >= (1ull << 31)) return FD_OK;
    FD_REQUIRE(x32 && pre_scale && pre_shift && p1.w && p2.w && p3.w && (out.y32 || out.ysplit), "fused unit: NULL tensor");
    UnitFusedArgs u;
    SplitKArgs a;
    a.x = reinterpret_cast<const uint8_t*>(x32); a.w = static_cast<const uint8_t*>(p1.w);
    a.bias = p1.bias; a.res = nullptr; a.y32 = nullptr; a.ysplit = nullptr; a.post_scale = a.post_shift = nullptr;
    a.pre_scale = pre_scale; a.pre_shift = pre_shift; a.w_inv_scale = p1.w_inv_scale;
    a.H = h; a.W = w; a.Cin = 256; a.Ho = h; a.Wo = w; a.Cout = 64; a.KW = 1; a.stride = 1; a.pad = 0; a.K = 256;
    a.M = (int)m_ll; a.relu = 1; a.tiles_n = 1; a.ksplit = 1; a.kchunk = a.K / SBK; a.partial = nullptr;
    a.clamped = out.clamped; a.f32 = 0;
    u.c1 = a;
    u.c2 = a;
    u.c2.x = nullptr; u.c2.w = static_cast<const uint8_t*>(p2.w); u.c2.bias = p2.bias; u.c2.pre_scale = u.c2.pre_shift = nullptr;
    u.c2.w_inv_scale = p2.w_inv_scale; u.c2.Cin = 64; u.c2.KW = 3; u.c2.pad = 1; u.c2.K = 576; u.c2.kchunk = 18;
    u.c3 = a;
    u.c3.x = nullptr; u.c3.w = static_cast<const uint8_t*>(p3.w); u.c3.bias = p3.bias; u.c3.res = x32; u.c3.y32 = out.y32;
    u.c3.ysplit = static_cast<uint8_t*>(out.ysplit); u.c3.post_scale = out.post_scale; u.c3.post_shift = out.post_shift;
    u.c3.pre_scale = u.c3.pre_shift = nullptr; u.c3.w_inv_scale = p3.w_inv_scale; u.c3.Cin = 64; u.c3.Cout = 256; u.c3.K = 64;
    u.c3.kchunk = 2; u.c3.relu = p3.relu; u.c3.tiles_n = 4;
    u.x_bytes = (unsigned)x_bytes;
    u.w1_bytes = (unsigned)(s1.packed_floats() * 4); u.w2_bytes = (unsigned)(s2.packed_floats() * 4);
    u.w3_bytes = (unsigned)(s3.packed_floats() * 4);
    u.tiles_x = ceil_div(w, 16); u.tiles_y = ceil_div(h, 8);
    const long long wgs = (long long)batch * u.tiles_x * u.tiles_y;
    if (wgs >= (1ll << 31)) return FD_OK;
    const double M = (double)m_ll;
    ProfScope scope(stream, "unit_fused", 2.0 * M * 64.0 * (256.0 + 576.0 + 256.0),
                    4.0 * M * 256.0 * ((out.y32 ? 2.0 : 1.0) + (out.ysplit ? 1.0 : 0.0)) + 4.0 * 64.0 * (256.0 + 576.0 + 256.0));
    hipLaunchKernelGGL(unit_fused_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, u);
    FD_LAUNCH_CHECK();
    *taken = true;
    return FD_OK;
}

// ---------------------------------------------------------------------------------
// The projection-shortcut unit of the 64-channel stage (first unit after the max-pool) in one
// kernel.  Input: act = split(relu(bn1(x))) with 64 channels.
//   t1 = relu(conv1x1(act) + b1),  t2 = relu(conv3x3(t1) + b2),  y = conv1x1(t2) + conv1x1_sc(act)
// Same 8 x 16 pixel tiles and 10 x 18 halo patch as unit_fused_kernel.  The patch of act arrives by
// LDS-DMA; each wave turns its own 48 patch rows into t1 in place (the first convolution is
// pointwise), the 3x3 runs out of that image, and the last phase accumulates the main branch and
// the shortcut convolution for 64 output channels at a time -- the shortcut tensor is never
// written, so the unit reads act and writes y and nothing else.  The tile's own act pixels (the
// patch was overwritten by t1) are fetched a second time, straight into the operand registers of
// the shortcut convolution.  The output leaves the accumulators directly: with the weights as
// the MFMA's row operand a lane owns 4 consecutive channels of a pixel, one 16-byte store.
//
// Weights move in 16 KB blocks (64 output channels x 64 K: both K-steps of a filter tap of the
// 3x3, or of a 64-channel chunk of a 1x1) through LDS-DMA, one block ahead in the 3x3 and two
// ahead in the last phase.
// LDS, phases 0-1: [patch: 2 K-step planes x 192 rows x 128 B | 2 weight blocks]
//      phase 3:    [t2 operand 32 KB | 3 weight blocks]
struct UnitProjArgs {
    SplitKArgs c1, c2, c3, sc;  // c1.x: act (split); c3.y32: output
    unsigned x_bytes, y_bytes, w1_bytes, w2_bytes, w3_bytes, wsc_bytes;
    int tiles_x, tiles_y;
};

__global__ __launch_bounds__(256, 2) void unit_proj_fused_kernel(UnitProjArgs u) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128, MID = 64;
    constexpr int PW = 18, PPIX = 180, PROWS = 192;
    constexpr int PLANE = PROWS * 128;
    constexpr int WBLK = 2 * 64 * 128;                      // one weight block: [K-step][64 rows][128 B]
    constexpr int WA = 2 * PLANE, WB = WA + WBLK;           // the two blocks of phases 0-1
    constexpr int T2_OFF = 0, WC = 2 * BM * 128;            // phase 3: third block where the patch tail was
    constexpr int SMEM_BYTES = WB + WBLK;
    static_assert(WC + WBLK <= WA && SMEM_BYTES <= 80 * 1024, "LDS plan");
    __shared__ __attribute__((aligned(16))) float smem[SMEM_BYTES / 4];
    char* const sbytes = reinterpret_cast<char*>(smem);

    const SplitKArgs &a1 = u.c1, &a2 = u.c2, &c = u.c3;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int wgid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int H = a1.H, W = a1.W;
    const int per_img = u.tiles_x * u.tiles_y;
    const int img = wgid / per_img;
    const int trem = wgid - img * per_img;
    const int tyi = trem / u.tiles_x, txi = trem - tyi * u.tiles_x;
    const int y0 = tyi * 8, x0 = txi * 16;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a1.x, 0, u.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)c.y32, 0, u.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc((void*)a1.w, 0, u.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc((void*)a2.w, 0, u.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3r = __builtin_amdgcn_make_buffer_rsrc((void*)c.w, 0, u.w3_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc((void*)u.sc.w, 0, u.wsc_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    const int frow = lane & 15, fh = lane >> 4;
    const int sw = (frow >> 1) & 7;
    const int lrow = lane >> 3, lpos = lane & 7;

    // ---- phase 0: the act patch by LDS-DMA (each wave its own 48 rows), then t1 in place ----
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int pp = wave * 48 + i * 8 + lrow;
        const int py = pp / PW, px = pp - py * PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool ok = pp < PPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const unsigned vo = ok ? (unsigned)(((img * H + gy) * W + gx) * (MID * 4) + (lpos ^ ((pp >> 1) & 7)) * 16) : OOB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                xr, (__attribute__((address_space(3))) void*)(sbytes + kk * PLANE + (wave * 48 + i * 8) * 128), 16, vo,
                kk * 128, 0, 0);
    }
    // a weight block = 64 rows x 2 K-steps by LDS-DMA: 4 instructions per wave, always
    unsigned wvo1[2], wvo2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave + 4 * i) * 8 + lrow;
        const unsigned chunk = (unsigned)((lpos ^ ((row >> 1) & 7)) * 16);
        wvo1[i] = (unsigned)(row * MID * 4) + chunk;          // conv1, conv3 and the shortcut: K = 64
        wvo2[i] = (unsigned)(row * (9 * MID) * 4) + chunk;    // conv2: K = 576
    }
#define FD_WBLOCK(RSRC, VO, ROWOFF, KT0, DST)                                                     \
    do {                                                                                          \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_)                                       \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                         \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                         \
                    RSRC, (__attribute__((address_space(3))) void*)(sbytes + (DST) + kk_ * (64 * 128) + (wave + 4 * i) * 1024), \
                    16, VO[i] + (unsigned)(ROWOFF), ((KT0) + kk_) * 128, 0, 0);                   \
    } while (0)
    FD_WBLOCK(w1r, wvo1, 0, 0, WA);
    FD_WBLOCK(w2r, wvo2, 0, 0, WB);                           // filter tap 0 of the 3x3
    int w_off[2];
#pragma unroll
    for (int part = 0; part < 2; ++part) w_off[part] = frow * 128 + (((part * 4 + fh) ^ sw) * 16);
    bool pval[3];
    int prow_off[3][2];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const int pp = (wave * 3 + f) * 16 + frow;
        const int py = pp / PW, px = pp - py * PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        pval[f] = pp < PPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
#pragma unroll
        for (int part = 0; part < 2; ++part) prow_off[f][part] = pp * 128 + (((part * 4 + fh) ^ ((pp >> 1) & 7)) * 16);
    }
    wait_vmcnt<0>();
    __syncthreads();
    unsigned bad = 0;           // values outside the fp16 range met while splitting t1 and t2
    f32x4 acc0[4][3];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int f = 0; f < 3; ++f) acc0[nf][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        f16x8 ah[4], al[4], bh[3], bl[3];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            ah[nf] = *reinterpret_cast<const f16x8*>(sbytes + WA + kk * (64 * 128) + w_off[0] + nf * 2048);
            al[nf] = *reinterpret_cast<const f16x8*>(sbytes + WA + kk * (64 * 128) + w_off[1] + nf * 2048);
        }
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            bh[f] = *reinterpret_cast<const f16x8*>(sbytes + kk * PLANE + prow_off[f][0]);
            bl[f] = *reinterpret_cast<const f16x8*>(sbytes + kk * PLANE + prow_off[f][1]);
        }
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                acc0[nf][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[nf], bh[f], acc0[nf][f], 0, 0, 0);
                acc0[nf][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[nf], bl[f], acc0[nf][f], 0, 0, 0);
                acc0[nf][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[nf], bh[f], acc0[nf][f], 0, 0, 0);
            }
    }
    // the tile's own act pixels as the shortcut's column operand: rows (2 wave + pf) * 16 + frow,
    // this lane's 8 channels of each K-step (hi at fh * 16, lo 64 bytes further)
    f16x8 sact[2][2][2];
#pragma unroll
    for (int pf = 0; pf < 2; ++pf) {
        const int gy = y0 + 2 * wave + pf, gx = x0 + frow;
        const unsigned vo = (gy < H && gx < W) ? (unsigned)(((img * H + gy) * W + gx) * (MID * 4) + fh * 16) : OOB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            sact[pf][kk][0] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(xr, vo, kk * 128, 0));
            sact[pf][kk][1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(xr, vo + 64u, kk * 128, 0));
        }
    }
    {
        // t1 over act, in place: a lane holds channels nf * 16 + fh * 4 + j of patch row (3 wave + f) * 16 + frow
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const f32x4 b = a1.bias ? *reinterpret_cast<const f32x4*>(a1.bias + nf * 16 + fh * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const int pp = (wave * 3 + f) * 16 + frow;
                f16x4 h4, l4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc0[nf][f][j] * a1.w_inv_scale + b[j];
                    if (!pval[f]) v = 0.f;                  // zero padding of the 3x3 applies to t1
                    _Float16 h_, l_;
                    split_f32_relu(v, h_, l_, bad);
                    h4[j] = h_; l4[j] = l_;
                }
                const int chunk = (nf & 1) * 2 + (fh >> 1);
                const int off = (nf >> 1) * PLANE + pp * 128 + ((chunk ^ ((pp >> 1) & 7)) * 16) + (fh & 1) * 8;
                *reinterpret_cast<f16x4*>(sbytes + off) = h4;
                *reinterpret_cast<f16x4*>(sbytes + (off ^ 64)) = l4;
            }
        }
    }
    __syncthreads();        // t1 complete; every wave has read the conv1 weights

    // ---- phase 1: the 3x3 out of the patch; wave w owns tile rows 2 w and 2 w + 1.  Tap t reads
    //      block (t + 1) & 1; the next tap's block (after the last tap: the first block of
    //      phase 3) is requested at the top of the step ----
    f32x4 acc1[4][2];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) acc1[nf][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - dy * 3;
        const int rd = ((tap + 1) & 1) ? WB : WA, wr_ = ((tap + 1) & 1) ? WA : WB;
        if (tap + 1 < 9) FD_WBLOCK(w2r, wvo2, 0, 2 * (tap + 1), wr_);
        else FD_WBLOCK(w3r, wvo1, 0, 0, wr_);
        int p_off[2][2];
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
            const int pp = (2 * wave + pf + dy) * PW + frow + dx;
            const int sp = (pp >> 1) & 7;
#pragma unroll
            for (int part = 0; part < 2; ++part) p_off[pf][part] = pp * 128 + (((part * 4 + fh) ^ sp) * 16);
        }
        f16x8 ah[2][4], al[2][4], bh[2][2], bl[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                ah[kk][nf] = *reinterpret_cast<const f16x8*>(sbytes + rd + kk * (64 * 128) + w_off[0] + nf * 2048);
                al[kk][nf] = *reinterpret_cast<const f16x8*>(sbytes + rd + kk * (64 * 128) + w_off[1] + nf * 2048);
            }
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                bh[kk][pf] = *reinterpret_cast<const f16x8*>(sbytes + kk * PLANE + p_off[pf][0]);
                bl[kk][pf] = *reinterpret_cast<const f16x8*>(sbytes + kk * PLANE + p_off[pf][1]);
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int pf = 0; pf < 2; ++pf) {
                    acc1[nf][pf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kk][nf], bh[kk][pf], acc1[nf][pf], 0, 0, 0);
                    acc1[nf][pf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kk][nf], bl[kk][pf], acc1[nf][pf], 0, 0, 0);
                    acc1[nf][pf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kk][nf], bh[kk][pf], acc1[nf][pf], 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
        wait_vmcnt<0>();
        pipeline_barrier();
    }

    // ---- phase 2: t2 = relu(. + b2), split, into the operand image of the last phase (the patch is
    //      dead: every wave has passed the last barrier) ----
    // Phase 3 runs 2 steps per 64-channel chunk (main branch, shortcut), step s reading block
    // ring[s % 3] = {WA, WC, WB}; step 0's block arrived during the last tap.
    FD_WBLOCK(wsr, wvo1, 0, 0, WC);                           // step 1
    {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const f32x4 b = a2.bias ? *reinterpret_cast<const f32x4*>(a2.bias + nf * 16 + fh * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                const int row = (2 * wave + pf) * 16 + frow;
                f16x4 h4, l4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    _Float16 h_, l_;
                    split_f32_relu(acc1[nf][pf][j] * a2.w_inv_scale + b[j], h_, l_, bad);
                    h4[j] = h_; l4[j] = l_;
                }
                const int chunk = (nf & 1) * 2 + (fh >> 1);
                const int off = T2_OFF + (nf >> 1) * (BM * 128) + row * 128 + ((chunk ^ ((row >> 1) & 7)) * 16) + (fh & 1) * 8;
                *reinterpret_cast<f16x4*>(sbytes + off) = h4;
                *reinterpret_cast<f16x4*>(sbytes + (off ^ 64)) = l4;
            }
        }
    }
    split_report(c.clamped, bad);
    pipeline_barrier();

    // ---- phase 3: y = W3 t2 + Wsc act ----
    int t_off[2][2];            // t2 operand of tile rows (2 wave + pf) * 16 + frow
    unsigned out_vo[2];         // byte offset of those rows in y (out of range: row outside the image)
#pragma unroll
    for (int pf = 0; pf < 2; ++pf) {
        const int row = (2 * wave + pf) * 16 + frow;
#pragma unroll
        for (int part = 0; part < 2; ++part) t_off[pf][part] = row * 128 + (((part * 4 + fh) ^ sw) * 16);
        const int gy = y0 + 2 * wave + pf, gx = x0 + frow;
        out_vo[pf] = (gy < H && gx < W) ? (unsigned)(((img * H + gy) * W + gx) * (c.Cout * 4) + fh * 16) : OOB;
    }
    constexpr int NSTEP = 8;    // 4 chunks x (main, shortcut)
    f32x4 accm[4][2], accs[4][2];
#pragma unroll
    for (int s3 = 0; s3 < NSTEP; ++s3) {
        const int ch = s3 >> 1, st = s3 & 1;
        const int ring[3] = {WA, WC, WB};
        const int rd = ring[s3 % 3];
        if (s3 + 2 < NSTEP) {
            const int wr_ = ring[(s3 + 2) % 3];
            if (st == 0) FD_WBLOCK(w3r, wvo1, (ch + 1) * 64 * MID * 4, 0, wr_);
            else FD_WBLOCK(wsr, wvo1, (ch + 1) * 64 * MID * 4, 0, wr_);
        }
        if (st == 0) {
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int pf = 0; pf < 2; ++pf) { accm[nf][pf] = f32x4{0.f, 0.f, 0.f, 0.f}; accs[nf][pf] = accm[nf][pf]; }
        }
        f16x8 ah[2][4], al[2][4], bh[2][2], bl[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                ah[kk][nf] = *reinterpret_cast<const f16x8*>(sbytes + rd + kk * (64 * 128) + w_off[0] + nf * 2048);
                al[kk][nf] = *reinterpret_cast<const f16x8*>(sbytes + rd + kk * (64 * 128) + w_off[1] + nf * 2048);
            }
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                if (st == 0) {
                    bh[kk][pf] = *reinterpret_cast<const f16x8*>(sbytes + T2_OFF + kk * (BM * 128) + t_off[pf][0]);
                    bl[kk][pf] = *reinterpret_cast<const f16x8*>(sbytes + T2_OFF + kk * (BM * 128) + t_off[pf][1]);
                } else {
                    bh[kk][pf] = sact[pf][kk][0];
                    bl[kk][pf] = sact[pf][kk][1];
                }
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int pf = 0; pf < 2; ++pf) {
                    f32x4& d = st == 0 ? accm[nf][pf] : accs[nf][pf];
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kk][nf], bh[kk][pf], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kk][nf], bl[kk][pf], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kk][nf], bh[kk][pf], d, 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
        // The next step's block must have landed.  Younger than it, and allowed to stay in flight:
        // the block requested at the top of this step (4 instructions) and, in the step after an
        // epilogue, that epilogue's 8 stores (vector-memory operations retire in issue order).
        if (s3 + 1 < NSTEP) {
            constexpr int DMA = 4, STORES = 8;
            if (s3 + 2 < NSTEP) {
                if (st == 0 && ch > 0) wait_vmcnt<DMA + STORES>();
                else wait_vmcnt<DMA>();
            } else {
                wait_vmcnt<0>();
            }
            pipeline_barrier();
        }
        if (st == 1) {
            // a lane holds channels ch * 64 + nf * 16 + fh * 4 + j of tile rows (2 wave + pf) * 16 + frow;
            // always 8 store instructions (rows outside the image are dropped by the range check)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const int col = ch * 64 + nf * 16 + fh * 4;
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (c.bias) b = *reinterpret_cast<const f32x4*>(c.bias + col);
                if (u.sc.bias) b += *reinterpret_cast<const f32x4*>(u.sc.bias + col);
#pragma unroll
                for (int pf = 0; pf < 2; ++pf) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[j] = accm[nf][pf][j] * c.w_inv_scale + (accs[nf][pf][j] * u.sc.w_inv_scale + b[j]);
                    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                    // The channel offset goes into the instruction's immediate offset, not into an SGPR:
                    // with a register soffset the compiler inserts no wait state between a 16-byte
                    // store and the next write of its data registers, and on this chip the store
                    // then picks up part of the next value (seen as wrong values in lanes 12-15 of
                    // every 16).
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), yr,
                                                           out_vo[pf] + (unsigned)((ch * 64 + nf * 16) * 4), 0,
                                                           2 /* streaming store */);
                }
            }
        }
    }
#undef FD_WBLOCK
#endif
}

int unit_proj_fused_forward(const SplitConvParams& p1, const SplitConvParams& p2, const SplitConvParams& p3,
                            const SplitConvParams& psc, const void* act, int batch, int h, int w, float* y32,
                            unsigned* clamped, hipStream_t stream, bool* taken) {
    *taken = false;
    if (p1.f32 || p2.f32 || p3.f32 || psc.f32) return FD_OK;   // split operands only
    const ConvShape &s1 = p1.s, &s2 = p2.s, &s3 = p3.s, &ss = psc.s;
    if (!(s1.cin == 64 && s1.cout == 64 && s1.kh == 1 && s1.kw == 1 && s1.stride == 1 && s1.pad == 0 && p1.relu &&
          s2.cin == 64 && s2.cout == 64 && s2.kh == 3 && s2.kw == 3 && s2.kwp == 3 && s2.stride == 1 && s2.pad == 1 &&
          p2.relu && s3.cin == 64 && s3.cout == 256 && s3.kh == 1 && s3.kw == 1 && s3.stride == 1 && s3.pad == 0 &&
          !p3.relu && ss.cin == 64 && ss.cout == 256 && ss.kh == 1 && ss.kw == 1 && ss.stride == 1 && ss.pad == 0 &&
          !psc.relu))
        return FD_OK;
    const long long m_ll = (long long)batch * h * w;
    const size_t x_bytes = (size_t)m_ll * 64 * 4, y_bytes = (size_t)m_ll * 256 * 4;
    if (y_bytes >= (1ull << 31) || !y32) return FD_OK;
    FD_REQUIRE(act && p1.w && p2.w && p3.w && psc.w, "fused projection unit: NULL tensor");
    UnitProjArgs u;
    SplitKArgs a;
    a.x = static_cast<const uint8_t*>(act); a.w = static_cast<const uint8_t*>(p1.w);
    a.bias = p1.bias; a.res = nullptr; a.y32 = nullptr; a.ysplit = nullptr; a.post_scale = a.post_shift = nullptr;
    a.pre_scale = a.pre_shift = nullptr; a.w_inv_scale = p1.w_inv_scale;
    a.H = h; a.W = w; a.Cin = 64; a.Ho = h; a.Wo = w; a.Cout = 64; a.KW = 1; a.stride = 1; a.pad = 0; a.K = 64;
    a.M = (int)m_ll; a.relu = 1; a.tiles_n = 1; a.ksplit = 1; a.kchunk = 2; a.partial = nullptr; a.clamped = clamped; a.f32 = 0;
    u.c1 = a;
    u.c2 = a;
    u.c2.x = nullptr; u.c2.w = static_cast<const uint8_t*>(p2.w); u.c2.bias = p2.bias; u.c2.w_inv_scale = p2.w_inv_scale;
    u.c2.KW = 3; u.c2.pad = 1; u.c2.K = 576; u.c2.kchunk = 18;
    u.c3 = a;
    u.c3.x = nullptr; u.c3.w = static_cast<const uint8_t*>(p3.w); u.c3.bias = p3.bias; u.c3.y32 = y32;
    u.c3.w_inv_scale = p3.w_inv_scale; u.c3.Cout = 256; u.c3.relu = 0; u.c3.tiles_n = 4;
    u.sc = u.c3;
    u.sc.w = static_cast<const uint8_t*>(psc.w); u.sc.bias = psc.bias; u.sc.w_inv_scale = psc.w_inv_scale; u.sc.y32 = nullptr;
    u.x_bytes = (unsigned)x_bytes; u.y_bytes = (unsigned)y_bytes;
    u.w1_bytes = (unsigned)(s1.packed_floats() * 4); u.w2_bytes = (unsigned)(s2.packed_floats() * 4);
    u.w3_bytes = (unsigned)(s3.packed_floats() * 4); u.wsc_bytes = (unsigned)(ss.packed_floats() * 4);
    u.tiles_x = ceil_div(w, 16); u.tiles_y = ceil_div(h, 8);
    const long long wgs = (long long)batch * u.tiles_x * u.tiles_y;
    if (wgs >= (1ll << 31)) return FD_OK;
    const double M = (double)m_ll;
    ProfScope scope(stream, "unit_proj_fused", 2.0 * M * 64.0 * (64.0 + 576.0 + 256.0 + 256.0),
                    4.0 * M * (64.0 + 256.0) + 4.0 * 64.0 * (64.0 + 576.0 + 256.0 + 256.0));
    hipLaunchKernelGGL(unit_proj_fused_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, u);
    FD_LAUNCH_CHECK();
    *taken = true;
    return FD_OK;
}

// ---------------------------------------------------------------------------------
// Stem convolution (7x7/2, bn0+ReLU folded) fused with the 3x3/2 max-pool that follows it.
// A workgroup owns PH x PW pooled pixels of one image.  Their windows cover a patch of
// (2 PH + 1) x (2 PW + 1) convolution outputs (one-pixel halo, recomputed by the neighbouring
// workgroups: 15 % extra MFMA work), which is the M tile of the GEMM: row m of the tile is
// patch pixel (m / CW, m % CW).  The patch stays in LDS as float32; the pooled values leave as
// the split activation of the first residual unit (its bn1+ReLU applied) and, on request, as
// float32.  Patch positions outside the convolution's output are 0, which cannot win a maximum
// of post-ReLU values.  This removes the write and the re-read of the 64-channel full-resolution
// tensor (2.1 GB per 8 frames at 1080p).
struct StemPoolArgs {
    const uint8_t* x;        // planar staged image: hi plane [B][H][Wpad][4 halves], then the lo plane
    const uint8_t* w;        // split packed stem weights [64][7][8][4]
    const float* bias;       // bn0 shift
    const float* post_scale; // bn1 of the first unit, applied to the split output
    const float* post_shift;
    float* y32;              // pooled float32 [B][Hp][Wp][64] or NULL
    uint8_t* ysplit;         // pooled split output
    unsigned* clamped;       // device counter of out-of-range values met while splitting, or NULL
    float w_inv_scale;
    int H, W, Ho, Wo, Hp, Wp, tiles_y, tiles_x;
    int Wpad;                // columns per staged row: 3 zero columns, W pixels, zero columns up to Wpad
    unsigned plane_bytes;    // size of one plane
};

// F32: float32 operands.  The staged image is one plane of 16-byte columns (4 floats per pixel), so the 8
// window pixels of a kernel row are the 128 contiguous bytes of a K-step row of floats: chunk j is window
// pixel j, and MFMA t of a part (v_mfma_f32_16x16x4_f32) multiplies channel t of pixels part * 4 .. + 3 with the
// matching weights.  The image takes the same LDS bytes as the two half-precision planes.
template <int PH, int PW, int BM, int OCC, bool F32>
__global__ __launch_bounds__(256, OCC) void stem_pool_kernel(StemPoolArgs a, unsigned x_bytes, unsigned w_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BN = 64, COUT = 64;
    constexpr int CH = 2 * PH + 1, CW = 2 * PW + 1;   // convolution patch
    static_assert(CH * CW <= BM, "patch must fit the M tile");
    using T = SplitTile<BM, BN>;
    constexpr int STAGE = T::STAGE;
    constexpr int LDP = COUT + 4;                     // patch row pitch in floats
    constexpr int SMEM = (2 * STAGE > BM * LDP) ? 2 * STAGE : BM * LDP;       // dwords; also holds image + weights
    __shared__ __attribute__((aligned(16))) float smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tx = blockIdx.x % a.tiles_x, ty = (blockIdx.x / a.tiles_x) % a.tiles_y;
    const int img = blockIdx.x / (a.tiles_x * a.tiles_y);
    const int oy0 = 2 * PH * ty - 1, ox0 = 2 * PW * tx - 1;     // patch origin in convolution-output coordinates

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // The patch's receptive field -- IH x IWC columns of the planar staged image, both planes --
    // and all 7 K-steps of the weights are brought into LDS once (LDS-DMA).  The MFMA operands
    // are read straight out of that image: for patch pixel (pr, pc) and kernel row kh the 8 window
    // pixels are 64 contiguous bytes of image row 2 pr + kh starting at column 2 pc (pixel x lives
    // in column x + 3, so the window of output ox starts at column 2 ox: 16-byte aligned), i.e.
    // exactly the four 16-byte hi chunks of a K-step row, and the same bytes of the lo plane are
    // its lo chunks.  Nothing is expanded, so L2 -> LDS traffic is the image once (22 KB per tile)
    // instead of one 128-byte row per patch pixel and kernel row (226 KB).  The zero columns of
    // the image stand in for the horizontal padding; rows above / below the image go out of range
    // and read zeros.
    constexpr int IH = 2 * (CH - 1) + 7, IWC = 2 * (CW - 1) + 8;   // image rows, columns (8 bytes per plane, or 16)
    constexpr int COLB = F32 ? 16 : 8;                             // bytes per column of a plane
    constexpr int NPL = F32 ? 1 : 2;                               // planes
    constexpr int IROW = IWC * COLB;                               // bytes per image row in LDS
    static_assert(IROW % 16 == 0, "image rows are whole 16-byte chunks");
    constexpr int ICH = IROW / 16;                                 // chunks per row
    constexpr int IPLANE = IH * IROW;
    constexpr int IMG_BYTES = ((NPL * IPLANE + 1023) / 1024) * 1024;
    constexpr int WGT_BYTES = 7 * 64 * 128;
    static_assert(IMG_BYTES + WGT_BYTES <= SMEM * 4, "LDS plan");
    char* const sbytes = reinterpret_cast<char*>(smem);
    {
        const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0;                // first image row, first column
        constexpr int NCHUNK = NPL * IH * ICH;
        for (int base = wave * 64; base < NCHUNK; base += 256) {
            const int idx = base + lane;
            const int pl = idx / (IH * ICH), rem = idx - pl * (IH * ICH);
            const int r = rem / ICH, c = rem - r * ICH;
            const int iy = iy0 + r;
            const bool ok = idx < NCHUNK && (unsigned)iy < (unsigned)a.H;
            const unsigned vo = ok ? (unsigned)(pl * a.plane_bytes) + (unsigned)(((img * a.H + iy) * a.Wpad + ix0) * COLB + c * 16) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(sbytes + base * 16), 16, vo, 0,
                                                     0, 0);
        }
        const int lrow = lane >> 3, lpos = lane & 7;
#pragma unroll
        for (int kt = 0; kt < 7; ++kt)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wave + 4 * i) * 8 + lrow;
                const unsigned vo = (unsigned)(row * 224 * 4 + (lpos ^ ((row >> 1) & 7)) * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    wr, (__attribute__((address_space(3))) void*)(sbytes + IMG_BYTES + kt * (64 * 128) + (wave + 4 * i) * 1024), 16,
                    vo, kt * 128, 0, 0);
            }
    }

    f32x4 acc[T::AM][T::AN];
#pragma unroll
    for (int mi = 0; mi < T::AM; ++mi)
#pragma unroll
        for (int ni = 0; ni < T::AN; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fh = lane >> 4;
    const int sw = (frow >> 1) & 7;
    int x_off[T::AM];            // hi chunk of kernel row 0 for patch row wm * (BM / 2) + mi * 16 + frow
#pragma unroll
    for (int mi = 0; mi < T::AM; ++mi) {
        int row = wm * (BM / 2) + mi * 16 + frow;
        if (row >= CH * CW) row = CH * CW - 1;           // rows beyond the patch: any valid address (masked later)
        const int pr = row / CW, pc = row - pr * CW;
        x_off[mi] = F32 ? (2 * pr) * IROW + (2 * pc + fh) * 16 : (2 * pr) * IROW + (2 * pc + 2 * fh) * 8;
    }
    int w_off[2];
#pragma unroll
    for (int part = 0; part < 2; ++part)
        w_off[part] = IMG_BYTES + (wn * (BN / 2) + frow) * 128 + (((part * 4 + fh) ^ sw) * 16);
    wait_vmcnt<0>();
    __syncthreads();
    if constexpr (F32) {
#pragma unroll
        for (int kt = 0; kt < 7; ++kt)
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                f32x4 wv[T::AN];
#pragma unroll
                for (int ni = 0; ni < T::AN; ++ni)
                    wv[ni] = *reinterpret_cast<const f32x4*>(sbytes + w_off[part] + kt * (64 * 128) + ni * 2048);
#pragma unroll
                for (int mh = 0; mh < T::AM; mh += 4) {
                    f32x4 xv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        xv[q] = *reinterpret_cast<const f32x4*>(sbytes + x_off[mh + q] + part * 64 + kt * IROW);
                    // element t of a chunk is channel t of a pixel; the staged image has three real channels and
                    // a zero fourth one, whose MFMA would add nothing: it is not issued (6 instead of 8 per kernel row)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int ni = 0; ni < T::AN; ++ni)
                                acc[mh + q][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ni][t], xv[q][t], acc[mh + q][ni], 0, 0, 0);
                }
            }
    } else {
#pragma unroll
    for (int kt = 0; kt < 7; ++kt) {
        // weights as the MFMA's row operand: a lane ends up with 4 consecutive channels of a
        // patch pixel, which go into the patch as one 16-byte LDS write
        f16x8 wh[T::AN], wl[T::AN];
#pragma unroll
        for (int ni = 0; ni < T::AN; ++ni) {
            wh[ni] = *reinterpret_cast<const f16x8*>(sbytes + w_off[0] + kt * (64 * 128) + ni * 2048);
            wl[ni] = *reinterpret_cast<const f16x8*>(sbytes + w_off[1] + kt * (64 * 128) + ni * 2048);
        }
#pragma unroll
        for (int mi = 0; mi < T::AM; ++mi) {
            const f16x8 xh = *reinterpret_cast<const f16x8*>(sbytes + x_off[mi] + kt * IROW);
            const f16x8 xl = *reinterpret_cast<const f16x8*>(sbytes + IPLANE + x_off[mi] + kt * IROW);
#pragma unroll
            for (int ni = 0; ni < T::AN; ++ni) {
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ni], xh, acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ni], xl, acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ni], xh, acc[mi][ni], 0, 0, 0);
            }
        }
    }
    }
    __syncthreads();             // every wave is done with the image and the weights: the patch overlays them

    // the convolution patch, bias + ReLU applied, as float32 rows of LDP floats: acc[mi][ni] holds
    // channels wn * 32 + ni * 16 + fh * 4 .. + 3 of patch row wm * (BM / 2) + mi * 16 + frow
    float* patch = smem;
    f32x4 bias_v[T::AN];
#pragma unroll
    for (int ni = 0; ni < T::AN; ++ni)
        bias_v[ni] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + wn * 32 + ni * 16 + fh * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < T::AM; ++mi) {
        const int row = wm * (BM / 2) + mi * 16 + frow;
        const int pr = row / CW, pc = row - pr * CW;
        const bool ok = row < CH * CW && (unsigned)(oy0 + pr) < (unsigned)a.Ho && (unsigned)(ox0 + pc) < (unsigned)a.Wo;
#pragma unroll
        for (int ni = 0; ni < T::AN; ++ni) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ok ? fmaxf(acc[mi][ni][j] * a.w_inv_scale + bias_v[ni][j], 0.f) : 0.f;
            *reinterpret_cast<f32x4*>(&patch[row * LDP + wn * 32 + ni * 16 + fh * 4]) = v;
        }
    }
    __syncthreads();

    // pooling: one item = 8 channels of one pooled pixel
    unsigned bad = 0;
    for (int item = tid; item < PH * PW * (COUT / 8); item += 256) {
        const int cg = item & 7, pp = item >> 3;
        const int ppy = pp / PW, ppx = pp - ppy * PW;
        const int py = PH * ty + ppy, px = PW * tx + ppx;
        if (py >= a.Hp || px >= a.Wp) continue;
        f32x4 m0 = {0.f, 0.f, 0.f, 0.f}, m1 = m0;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float* q = patch + ((2 * ppy + dy) * CW + 2 * ppx + dx) * LDP + cg * 8;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(q), v1 = *reinterpret_cast<const f32x4*>(q + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    m0[j] = fmaxf(m0[j], v0[j]);
                    m1[j] = fmaxf(m1[j], v1[j]);
                }
            }
        const size_t orow = ((size_t)img * a.Hp + py) * a.Wp + px;
        const int c = cg * 8;
        if (a.y32) {
            float* yp = a.y32 + orow * COUT + c;
            *reinterpret_cast<f32x4*>(yp) = m0;
            *reinterpret_cast<f32x4*>(yp + 4) = m1;
        }
        if (a.post_scale) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(a.post_scale + c);
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(a.post_scale + c + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(a.post_shift + c);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(a.post_shift + c + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m0[j] = fmaxf(m0[j] * s0[j] + t0[j], 0.f);
                m1[j] = fmaxf(m1[j] * s1[j] + t1[j], 0.f);
            }
        }
        if constexpr (F32) {
            float* ap = reinterpret_cast<float*>(a.ysplit) + orow * COUT + c;
            *reinterpret_cast<f32x4*>(ap) = m0;
            *reinterpret_cast<f32x4*>(ap + 4) = m1;
        } else {
            f16x8 hi, lo;
            split8(m0, m1, hi, lo, bad);
            uint8_t* sp = a.ysplit + orow * COUT * 4 + split_offset(c);
            *reinterpret_cast<f16x8*>(sp) = hi;
            *reinterpret_cast<f16x8*>(sp + 64) = lo;
        }
    }
    if constexpr (!F32) split_report(a.clamped, bad);
#endif
}

// Tile of the fused stem: 5 x 11 pooled pixels = an 11 x 23 patch in a 256-row M tile, two
// workgroups per CU.  128-row tiles (2 x 12 or 3 x 8 pooled pixels, three workgroups per CU)
// measured the same throughput with 13 % more halo work.
constexpr int STEM_PH = 5, STEM_PW = 11, STEM_BM = 256, STEM_OCC = 2;
int stem_pool_forward(const SplitConvParams& p, const void* staged_planar, int batch, int h, int w,
                      const float* post_scale, const float* post_shift, float* y32, void* ysplit,
                      unsigned* clamped, hipStream_t stream) {
    const ConvShape& s = p.s;
    FD_REQUIRE(staged_planar && p.w && ysplit, "stem+pool: NULL tensor");
    FD_REQUIRE(s.cin == 4 && s.kwp == 8 && s.kh == 7 && s.kw == 7 && s.stride == 2 && s.pad == 3 && s.cout == 64 &&
                   p.relu,
               "stem+pool: this kernel is the 7x7/2 stem with 64 output channels and ReLU");
    constexpr int PH = STEM_PH, PW = STEM_PW;
    StemPoolArgs a;
    a.x = static_cast<const uint8_t*>(staged_planar);
    a.w = static_cast<const uint8_t*>(p.w);
    a.bias = p.bias; a.post_scale = post_scale; a.post_shift = post_shift;
    a.Wpad = staged_planar_width(w);
    a.y32 = y32; a.ysplit = static_cast<uint8_t*>(ysplit); a.clamped = clamped; a.w_inv_scale = p.w_inv_scale;
    a.H = h; a.W = w; a.Ho = s.out_h(h); a.Wo = s.out_w(w);
    a.Hp = (a.Ho + 2 - 3) / 2 + 1; a.Wp = (a.Wo + 2 - 3) / 2 + 1;
    a.tiles_y = ceil_div(a.Hp, PH); a.tiles_x = ceil_div(a.Wp, PW);
    const size_t x_bytes = (size_t)batch * h * a.Wpad * 16, w_bytes = s.packed_floats() * 4;
    FD_REQUIRE(x_bytes < (1ull << 31), "stem+pool: staged batch too large for 32-bit offsets");
    a.plane_bytes = p.f32 ? (unsigned)x_bytes : (unsigned)(x_bytes / 2);
    const long long grid = (long long)batch * a.tiles_y * a.tiles_x;
    FD_REQUIRE(grid < (1ll << 31), "stem+pool: too many tiles");
    ProfScope scope(stream, p.f32 ? "stem_pool_f32" : "stem_pool", 2.0 * batch * a.Ho * a.Wo * 64.0 * 147.0,
                    16.0 * batch * h * w + 4.0 * 64.0 * 224.0 +
                        256.0 * batch * a.Hp * a.Wp * ((y32 ? 1.0 : 0.0) + 1.0));
    if (p.f32)
        hipLaunchKernelGGL((stem_pool_kernel<PH, PW, STEM_BM, STEM_OCC, true>), dim3((unsigned)grid), dim3(256), 0, stream,
                           a, (unsigned)x_bytes, (unsigned)w_bytes);
    else
        hipLaunchKernelGGL((stem_pool_kernel<PH, PW, STEM_BM, STEM_OCC, false>), dim3((unsigned)grid), dim3(256), 0, stream,
                           a, (unsigned)x_bytes, (unsigned)w_bytes);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

}  // namespace fd
