// Bottleneck tail of the 64-channel stage with the 3x3 convolution in Winograd F(2x2, 3x3) form, float32:
//   t2 = relu(conv3x3(t1) + b2)      (64 -> 64, stride 1, pad 1; BatchNorm folded into the weights)
//   y  = conv1x1(t2) + residual      (64 -> cout)
// in one kernel, like conv_tail_fused_kernel, but with 4/9 of the 3x3's multiplies.  Unlike the stand-alone
// Winograd layers (fd_winograd.hip) nothing of the Winograd domain ever exists in memory:
//   * a workgroup owns an 8 x 16 pixel tile = 4 x 8 output patches of 2 x 2; its 10 x 18 pixel input patch goes
//     to LDS once by LDS-DMA (two K-step planes of 192 rows x 128 B, the usual swizzled row image, zeros outside
//     the picture);
//   * the input transform V = B^T d B happens in registers, on the MFMA's activation fragments: a lane reads the
//     two input rows a transform row i combines (8 float4 per 16-channel part), forms their combination and from it
//     the fragments of the four transform columns j -- two LDS reads and four additions per fragment;
//   * the 16 GEMMs M[i][j] = V[i][j] U[i][j]^T (32 patches x 64 channels x K = 64 each): wave (wm, wn) owns 16
//     patches x 32 output channels of ALL of them, transform row i outermost, so only the four GEMMs of one row are
//     live as accumulators (32 registers).  The transformed weights never touch LDS: a lane's fragments of a
//     sub-step (GEMM, K half, 16-channel part: 8 MFMAs) are 2 x 16 bytes that it reads straight from L2 into registers
//     five sub-steps ahead, from a copy packed in reading order (every load instruction fetches 1 KB of consecutive
//     bytes) -- phase 1 has no barrier; the activation fragments are formed one 16-channel part at a time (16 registers);
//   * the output transform Y = A^T M A is linear in M: after each transform row the wave folds that row's column
//     combination into its four output pixels per patch (32 registers), entirely in registers; bias + ReLU, and t2
//     lands in LDS as the A operand of the 1x1, which then runs exactly as in conv_tail_fused_kernel (64 output
//     channels at a time, residual values requested ahead, transposed epilogue).
// Where a workgroup's time goes (wave 0's clock at the phase boundaries, averaged over 255 workgroups of a 16-frame
// launch; alone on its CU / sharing it with a second workgroup as in production): input patch 2.0 / 5.1 us, the 16
// GEMMs 9.9 / 14.0 us (6.8 us of matrix time), bias + ReLU + t2 0.65 / 1.5 us, first chunk of the 1x1 4.3 / 6.9 us,
// chunks 2-4 11.4 / 18 us (1.7 us of matrix time each).  The form before this one -- the waves of a patch block
// split the transform rows, weights through a two-stage LDS ring with a barrier per step, halves of the output
// transform exchanged through LDS -- spent 10.6 / 13.3 us in the GEMMs and 2.8 / 7.4 us in the output transform.
// SC (the stride-1 projection unit, the first of the stage): y = conv1x1(t2) + conv1x1_sc(act) is ONE 1x1 over the 128
// channels [t2 | act] with the weight rows [W3 | Wsc]; the shortcut tensor (2.1 GB per 16 frames at 1080p) is neither
// written by a launch of its own nor read back as a residual.  933 + 1 500 us -> 2 003 us for the unit, +1.8 % frames/s.
// MODE 2 (the default for that unit): conv1 too -- the input patch in LDS is the unit's activated input, and each wave
// turns three 16-row blocks of it into t1 in place (192 MFMAs per wave, conv1's weights as register fragments) before
// the GEMM phase; the halo rows are computed by every tile that needs them (1.5 x conv1's FLOPs, 2 % of the unit's).
// The whole unit is one launch and t1 is not in memory either: 263 + 2 003 -> 2 185 us.
// F(2x2, 3x3)'s transforms are additions and halvings only: against float64 its error is the direct form's
// (fd_winograd.hip).  Results differ from conv_tail_fused_kernel's in the last bits.
#include "fd_conv_split_dev.h"

#include <cstdlib>
#include <vector>

namespace fd {

struct TailWinoArgs {
    const float* x;        // t1 [B][H][W][64], activated
    const float* u;        // transformed weights in the lanes' reading order (tail_wino_pack)
    const float* bias2;    // [64]
    SplitKArgs c;          // the 1x1: w, bias, res, y32, ysplit, post_scale / post_shift, Cout, M, w_inv_scale
    int H, W, tiles_x, tiles_y;
    unsigned x_bytes, u_bytes, w3_bytes;
    const float* act;      // MODE >= 1: the unit's activated input [B][H][W][64], the projection shortcut's operand
    const float* w1;       // MODE 2: conv1's weights [64][64] (BatchNorm folded), bias [64] and output scale
    const float* bias1;
    float w1_inv_scale;
};

// MODE 0: the tail; 1: with the projection shortcut inside the 1x1; 2: the whole projection unit -- conv1 as well,
// computed on the input patch in LDS, so t1 never exists in memory either.
template <int MODE>
__global__ __launch_bounds__(256, 2) void conv_tail_wino_kernel(TailWinoArgs k) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool SC = MODE >= 1, C1 = MODE == 2;
    constexpr int BM = 128, BN = 64, MID = 64;
    using T = SplitTile<BM, BN>;
    using E = SplitEpi<BM, BN>;
    constexpr int PW = 18, PPIX = 180, PROWS = 192;
    constexpr int PLANE = PROWS * 128;                  // bytes per K-half plane of the input patch
    constexpr int D_BYTES = 2 * PLANE;                  // 48 KB
    // t2 behind the patch: 80 KB, two workgroups per CU.  O3 (MODE 0 only; with __launch_bounds__(256, 3)): t2 laid
    // over the input patch once every wave is done with it, the 1x1's staging / weight chunk behind it -- 52 KB, three
    // workgroups per CU at 168 registers (13 float4 values of the output transform spilled across the GEMM phase).
    // Measured: 1 621 / 1 761 us per unit against 1 484 / 1 612 us with two workgroups -- the third workgroup costs
    // more in contention than it covers -- so it stays off.
    constexpr bool O3 = false;
    constexpr int KK3 = SC ? 4 : 2;                     // K-steps of the 1x1: t2's 64 channels (+ the shortcut operand's 64)
    constexpr int CS_BYTES = SC ? 32 * 1024 : 20 * 1024; // epilogue staging / weight chunk of the 1x1
    constexpr int A_BYTES = (MID / 32) * BM * 128;      // t2 as the 1x1's A operand: 2 K-steps x 128 rows x 128 B
    constexpr int A_OFF = O3 ? 0 : D_BYTES;
    constexpr int CS_OFF = O3 ? A_BYTES : 0;
    constexpr int SMEM_BYTES = O3 ? A_BYTES + CS_BYTES : D_BYTES + A_BYTES;
    static_assert(SMEM_BYTES >= D_BYTES && T::EPI * 4 <= CS_BYTES && KK3 * 64 * 128 <= CS_BYTES &&
                      (O3 ? 3 * SMEM_BYTES <= 160 * 1024 : (SMEM_BYTES == 80 * 1024 && CS_BYTES <= A_OFF)), "LDS plan");
    __shared__ __attribute__((aligned(16))) float smem[SMEM_BYTES / 4];
    char* const sbytes = reinterpret_cast<char*>(smem);

    const SplitKArgs& c = k.c;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int wgid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int H = k.H, W = k.W;
    const int per_img = k.tiles_x * k.tiles_y;
    const int img = wgid / per_img;
    const int trem = wgid - img * per_img;
    const int tyi = trem / k.tiles_x, txi = trem - tyi * k.tiles_x;
    const int y0 = tyi * 8, x0 = txi * 16;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(C1 ? k.act : k.x), 0, k.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)k.u, 0, k.u_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3r = __builtin_amdgcn_make_buffer_rsrc((void*)c.w, 0, k.w3_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    const int frow = lane & 15, fh = lane >> 4;
    const int sw = (frow >> 1) & 7;
    const int lrow = lane >> 3, lpos = lane & 7;

    // ---- phase 0: the input patch, 24 pieces of 8 rows per plane (rows 180..191 and pixels outside the picture: zeros)
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int q = wave + 4 * i;                  // 0..47, uniform
        const int h = q / 24, grp = q - h * 24;
        const int r = grp * 8 + lrow;
        const int py = r / PW, px = r - py * PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool ok = r < PPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const unsigned vo = ok ? (unsigned)(((img * H + gy) * W + gx) * (MID * 4) + h * 128 + ((lpos ^ ((r >> 1) & 7)) * 16)) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(sbytes + h * PLANE + grp * 1024),
                                                 16, vo, 0, 0, 0);
    }

    // MODE 2: conv1's weights as this lane's B fragments, [K-step][part][16-channel block], straight from memory
    f32x4 w1f[C1 ? 2 : 1][2][C1 ? 4 : 1];
    if constexpr (C1) {
        const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc((void*)k.w1, 0, 64 * 64 * 4, 0x00020000);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    w1f[kk][part][ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        w1r, (unsigned)(((ni * 16 + frow) * MID + kk * 32 + (part * 4 + fh) * 4) * 4), 0, 0));
    }

    // ---- transformed weights: step s = (i, h, j) multiplies by GEMM xi = 4 i + j, K half h.  A lane's four fragments of
    // a step (part pl, 16-channel block ni: f = 2 pl + ni) come straight from memory (L2) into registers, 16 bytes each,
    // in the packed order [s][wn][f][lane] -- every load instruction reads 1 KB of consecutive bytes -- and are requested
    // PD - 1 steps ahead, so phase 1 has no LDS traffic for the weights and no barrier.
    // Sub-step u = (i, h, pl, j): 8 MFMAs on the two fragments f = 2 pl + ni of step s = (i, h, j), requested PD - 1
    // sub-steps ahead.
    constexpr int PD = 6;
    f32x4 bq[PD][2];
    const unsigned blane = (unsigned)(lane * 16);
#define FD_ULOAD(U)                                                                               \
    do {                                                                                          \
        const int s_ = ((U) >> 3) * 4 + ((U) & 3), pl_ = ((U) >> 2) & 1;                          \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                          \
            bq[(U) % PD][ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(   \
                ur, blane + (pl_ * 2 + ni) * 1024, (s_ * 2 + wn) * 4096, 0));                     \
    } while (0)
#pragma unroll
    for (int st = 0; st < PD - 1; ++st) FD_ULOAD(st);
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt<(PD - 1) * 2>();        // the input patch has landed (the weights requested after it may fly)
    __syncthreads();

    if constexpr (C1) {
        // ---- conv1 in place: the patch holds the unit's activated input; every wave turns three 16-row blocks of it
        // into t1 = relu(act W1^T + b1) -- the rows of a block are read (into fragments) and written by that wave
        // alone, and a wave's LDS operations execute in order.  Pixels outside the picture are the 3x3's zero padding:
        // they are written as zeros, not as relu(b1).
        float b1[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) b1[ni] = k.bias1 ? k.bias1[ni * 16 + frow] : 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int rb = wave * 3 + j;
            const int R = rb * 16 + frow;
            f32x4 a1[2][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int part = 0; part < 2; ++part)
                    a1[kk][part] = *reinterpret_cast<const f32x4*>(sbytes + kk * PLANE + R * 128 + (((part * 4 + fh) ^ ((R >> 1) & 7)) << 4));
            f32x4 acc1[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc1[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int part = 0; part < 2; ++part)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            acc1[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kk][part][t], w1f[kk][part][ni][t], acc1[ni], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rb * 16 + fh * 4 + r;
                const int py = row / PW, px = row - py * PW;
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                const bool ok = row < PPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const float v = ok ? fmaxf(acc1[ni][r] * k.w1_inv_scale + b1[ni], 0.f) : 0.f;
                    const int c32 = (ni & 1) * 16 + frow;
                    *reinterpret_cast<float*>(sbytes + (ni >> 1) * PLANE + row * 128 + (((c32 >> 2) ^ ((row >> 1) & 7)) << 4) +
                                              (c32 & 3) * 4) = v;
                }
            }
        }
        __syncthreads();
    }

    // ---- phase 1: the 16 GEMMs, transform row i outermost; wave (wm, wn) owns 16 patches x 32 output channels
    // this lane's patch in the MFMA's row operand: block wm holds patch rows 2 wm, 2 wm + 1 (8 patches each)
    const int pyy = wm * 2 + (frow >> 3), pxx = frow & 7;
    const int r00 = (2 * pyy) * PW + 2 * pxx;       // patch row of the 4 x 4 input patch's top-left pixel (even)
    f32x4 y[4][2];          // [p * 2 + q][ni]: output pixel (p, q) of the patch, Y = A^T M A with A^T = (1, 1, 1, 0; 0, 1, -1, -1)
#pragma unroll
    for (int i_ = 0; i_ < 4; ++i_) {
        // transform row i combines input rows a1 and a2: (0, -2), (1, +2), (2, -1), (1, -3)
        const int a1 = (i_ == 0) ? 0 : (i_ == 2 ? 2 : 1);
        const int a2 = (i_ == 0 || i_ == 1) ? 2 : (i_ == 2 ? 1 : 3);
        const float s2 = (i_ == 1) ? 1.f : -1.f;
        f32x4 acc[4][2];     // [j][ni]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[j][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // one 16-channel part at a time: its activation fragments [j], then the four column GEMMs' MFMAs
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const int cq = pl * 4 + fh;
                f32x4 rr[4], av[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int ra = r00 + a1 * PW + b, rb = r00 + a2 * PW + b;
                    const f32x4 d1 = *reinterpret_cast<const f32x4*>(sbytes + h * PLANE + ra * 128 + ((cq ^ ((ra >> 1) & 7)) << 4));
                    const f32x4 d2 = *reinterpret_cast<const f32x4*>(sbytes + h * PLANE + rb * 128 + ((cq ^ ((rb >> 1) & 7)) << 4));
                    rr[b] = d1 + s2 * d2;
                }
                av[0] = rr[0] - rr[2];
                av[1] = rr[1] + rr[2];
                av[2] = rr[2] - rr[1];
                av[3] = rr[1] - rr[3];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int u = ((i_ * 2 + h) * 2 + pl) * 4 + j;
                    if (u + PD - 1 < 64) FD_ULOAD(u + PD - 1);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            acc[j][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][t], bq[u % PD][ni][t], acc[j][ni], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                }
            }
        }
        // this row's share of the output transform: T[i][q] over the columns j, then Y[p][q] += A^T[p][i] T[i][q]
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const f32x4 t0 = acc[0][ni] + acc[1][ni] + acc[2][ni];
            const f32x4 t1 = acc[1][ni] - acc[2][ni] - acc[3][ni];
            if (i_ == 0) {
                y[0][ni] = t0; y[1][ni] = t1;
            } else if (i_ == 1) {
                y[0][ni] += t0; y[1][ni] += t1; y[2][ni] = t0; y[3][ni] = t1;
            } else if (i_ == 2) {
                y[0][ni] += t0; y[1][ni] += t1; y[2][ni] -= t0; y[3][ni] -= t1;
            } else {
                y[2][ni] -= t0; y[3][ni] -= t1;
            }
        }
    }
#undef FD_ULOAD

    // SC (the stride-1 projection unit): the shortcut convolution of the unit's activated input joins the 1x1 as 64
    // more channels of K -- y = [t2 | act] [W3 | Wsc]^T -- so neither the shortcut tensor nor a residual read exists.
    // The operand's fragments do not depend on the output chunk: a lane fetches its 16 pieces of 16 bytes (4 row
    // blocks x 2 K-steps x 2 parts; pixels outside the picture read zeros) once, here, into the registers phase 1
    // has released, and uses them for all chunks.
    f32x4 xa[SC ? 4 : 1][2][2];
    if constexpr (SC) {
        const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)k.act, 0, k.x_bytes, 0x00020000);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int row = wm * 64 + mi * 16 + frow, ty = row >> 4, tx = row & 15;
            const bool ok = y0 + ty < H && x0 + tx < W;
            const unsigned vo = ok ? (unsigned)((((img * H + y0 + ty) * W + x0 + tx) * MID + fh * 4) * 4) : OOB;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int part = 0; part < 2; ++part)
                    xa[mi][kk][part] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ar, vo + kk * 128 + part * 64, 0, 0));
        }
    }

    if constexpr (O3) __syncthreads();      // every wave is done with the input patch, which t2 overlays
    // ---- phase 2: bias + ReLU; t2 lands in LDS as the A operand of the 1x1 (behind the input patch, which other waves
    // may still be reading): channel half wn, swizzled 128-byte rows like every operand image
    {
        float b2[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b2[ni] = k.bias2 ? k.bias2[wn * 32 + ni * 16 + frow] : 0.f;
#pragma unroll
        for (int pq = 0; pq < 4; ++pq)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = fmaxf(y[pq][ni][r] + b2[ni], 0.f);
                    // accumulator row (lane >> 4) * 4 + r = patch of this block; column = output channel
                    const int pm = fh * 4 + r;
                    const int prow = (2 * (wm * 2 + (pm >> 3)) + (pq >> 1)) * 16 + 2 * (pm & 7) + (pq & 1);
                    const int c32 = ni * 16 + frow;
                    *reinterpret_cast<float*>(sbytes + A_OFF + wn * (BM * 128) + prow * 128 +
                                              (((c32 >> 2) ^ ((prow >> 1) & 7)) << 4) + (c32 & 3) * 4) = v;
                }
    }

    // ---- phase 3: the 1x1 convolution, 64 output channels at a time, as conv_tail_fused_kernel's float32 form: the
    // weight chunk shares the LDS of the epilogue staging and is fetched after the previous chunk's epilogue.
    // Measured and not kept (1.47-1.60 ms per unit as it is): the chunk in LDS of its own with chunk ch + 1 and its
    // residual values requested before the epilogue of chunk ch, barriers and epilogue that do not drain the
    // vector-memory counter; the residual values of all four chunks requested at once into the registers the
    // accumulators leave; the 1x1's weights as register fragments a chunk ahead, like the 3x3's, with the staging
    // slices laid over the dead input patch so that one barrier serves all four chunks (1.57-1.67 ms: a chunk costs
    // 3.5 us alone on a CU with or without its weight wait and barriers -- 1.7 us of matrix time and the round trip
    // of its residual values).  The kernel is within 1.5x of both of its floors -- 0.89 ms of matrix time, 1.04 ms
    // of HBM time for 4.9 GB -- and what separates it from them is that a CU's two workgroups overlap their matrix
    // and memory phases only by chance; staggering their start changed nothing, and without the raised priority of
    // the MFMA blocks the GEMM phase of one workgroup starves behind the other's epilogue (14 -> 24 us).
    float* wch = smem + CS_OFF / 4;
    int a3_off[MID / 32][2], b3_off[KK3][2];
#pragma unroll
    for (int kk = 0; kk < KK3; ++kk)
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const int pos = ((part * 4 + fh) ^ sw) * 16;
            if (kk < MID / 32) a3_off[kk][part] = A_OFF + kk * (BM * 128) + (wm * (BM / 2) + frow) * 128 + pos;
            b3_off[kk][part] = CS_OFF + kk * (64 * 128) + (wn * 32 + frow) * 128 + pos;
        }
    constexpr int RL = SC ? 0 : E::GROUPS * E::PASSES * 2;        // residual loads per chunk and lane
    const int nch = c.Cout / 64;
    const PatchRows rows{(img * H + y0) * W + x0, W, min(8, H - y0), min(16, W - x0)};
    f32x4 res[E::GROUPS][E::PASSES][2];
    for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();               // the shared region is free: phase 2 / the previous epilogue are done with it
#pragma unroll
        for (int kk = 0; kk < KK3; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wave + 4 * i) * 8 + lrow;
                const unsigned vo = (unsigned)(((ch * 64 + row) * (KK3 * 32) + kk * 32) * 4 + (lpos ^ ((row >> 1) & 7)) * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    w3r, (__attribute__((address_space(3))) void*)(wch + kk * (64 * 32) + (wave + 4 * i) * 256), 16, vo, 0, 0, 0);
            }
        if constexpr (!SC) split_residual_prefetch<BM, BN>(c, rows, ch, wave, lane, res);
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<RL>();              // weights of this chunk landed; its residual may fly
        __syncthreads();
        f32x4 acc3[T::AM][T::AN];
#pragma unroll
        for (int mi = 0; mi < T::AM; ++mi)
#pragma unroll
            for (int ni = 0; ni < T::AN; ++ni) acc3[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < MID / 32; ++kk) f32_mma_step<BM, BN>(sbytes, a3_off[kk], b3_off[kk], acc3);
        if constexpr (SC) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    f32x4 bv[T::AN];
#pragma unroll
                    for (int ni = 0; ni < T::AN; ++ni)
                        bv[ni] = *reinterpret_cast<const f32x4*>(sbytes + b3_off[MID / 32 + kk][part] + ni * 2048);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int mi = 0; mi < T::AM; ++mi)
#pragma unroll
                            for (int ni = 0; ni < T::AN; ++ni)
                                acc3[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[mi][kk][part][t], bv[ni][t], acc3[mi][ni], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                }
        }
        // (the epilogue's own first barrier separates the last weight read from the staging writes)
        if constexpr (SC) split_epilogue16_rows<BM, BN, false>(c, wch, rows, ch, wave, lane, acc3);
        else split_epilogue16_rows<BM, BN, true>(c, wch, rows, ch, wave, lane, acc3, res);
    }
#endif
}

// Transformed weights of the 3x3 in the order the kernel's lanes read them: [step s = (i, h, j)][wn][f = 2 pl + ni][lane][t]
// holds (G g G^T)[i][j] of filter (o, c) times out_scale[o] with o = 32 wn + 16 ni + (lane & 15) and
// c = 32 h + 16 pl + 4 (lane >> 4) + t; built from wino_pack_weights' [16][cout_pad][64].
int tail_wino_pack(const float* w_oihw, const float* out_scale, std::vector<float>* packed) {
    std::vector<float> u;
    int cout_pad = 0;
    FD_TRY(wino_pack_weights(w_oihw, 64, 64, out_scale, 2, &u, &cout_pad));
    packed->assign((size_t)32 * 2 * 4 * 64 * 4, 0.f);
    for (int s = 0; s < 32; ++s) {
        const int i = s >> 3, h = (s >> 2) & 1, j = s & 3, xi = i * 4 + j;
        for (int wn = 0; wn < 2; ++wn)
            for (int f = 0; f < 4; ++f)
                for (int lane = 0; lane < 64; ++lane)
                    for (int t = 0; t < 4; ++t) {
                        const int o = wn * 32 + (f & 1) * 16 + (lane & 15);
                        const int c = h * 32 + (f >> 1) * 16 + (lane >> 4) * 4 + t;
                        (*packed)[((((size_t)s * 2 + wn) * 4 + f) * 64 + lane) * 4 + t] = u[((size_t)xi * cout_pad + o) * 64 + c];
                    }
    }
    return FD_OK;
}

int conv_tail_wino_forward(const float* u_wino, const float* bias2, const SplitConvParams& p3, const float* t1, int batch, int h,
                           int w, const float* residual, const SplitConvOut& out, hipStream_t stream, bool* taken,
                           const float* sc_act, const float* w3_sc, const SplitConvParams* conv1) {
    *taken = false;
    const ConvShape& s3 = p3.s;
    const bool sc = sc_act && w3_sc;      // the projection shortcut inside the 1x1 (no residual tensor then)
    if (!u_wino || !p3.f32 || out.b3 || (sc ? residual != nullptr : !residual) ||
        !(s3.cin == 64 && s3.kh == 1 && s3.kw == 1 && s3.stride == 1 && s3.pad == 0 && s3.cout % 64 == 0))
        return FD_OK;
    const long long m_ll = (long long)batch * h * w;
    const size_t x_bytes = (size_t)m_ll * 64 * 4, w3_bytes = s3.packed_floats() * 4 * (sc ? 2 : 1);
    if (x_bytes >= (1ull << 31) || m_ll >= (1ll << 29)) return FD_OK;
    FD_REQUIRE((t1 || (sc && conv1)) && p3.w && (out.y32 || out.ysplit), "Winograd tail: NULL tensor");
    TailWinoArgs k;
    k.x = t1; k.u = u_wino; k.bias2 = bias2;
    SplitKArgs& c = k.c;
    c.x = nullptr; c.w = static_cast<const uint8_t*>(sc ? (const void*)w3_sc : p3.w); c.bias = p3.bias; c.res = residual; c.y32 = out.y32;
    k.act = sc ? sc_act : nullptr;
    // the whole projection unit: conv1 (64 -> 64, 1x1, ReLU) on the input patch inside the kernel; t1 is not read then
    const bool c1 = sc && conv1 && conv1->f32 && conv1->w && conv1->relu && conv1->s.cin == 64 && conv1->s.cout == 64 &&
                    conv1->s.kh == 1 && conv1->s.kw == 1 && conv1->s.stride == 1 && conv1->s.pad == 0;
    k.w1 = c1 ? static_cast<const float*>(conv1->w) : nullptr;
    k.bias1 = c1 ? conv1->bias : nullptr;
    k.w1_inv_scale = c1 ? conv1->w_inv_scale : 1.f;
    c.ysplit = static_cast<uint8_t*>(out.ysplit); c.post_scale = out.post_scale; c.post_shift = out.post_shift;
    c.pre_scale = c.pre_shift = nullptr; c.w_inv_scale = p3.w_inv_scale;
    c.H = h; c.W = w; c.Cin = 64; c.Ho = h; c.Wo = w; c.Cout = s3.cout; c.KW = 1; c.stride = 1; c.pad = 0; c.K = 64;
    c.M = (int)m_ll; c.relu = p3.relu; c.tiles_n = s3.cout / 64; c.ksplit = 1; c.kchunk = 2; c.partial = nullptr;
    c.clamped = nullptr; c.f32 = 1;
    k.H = h; k.W = w; k.tiles_x = ceil_div(w, 16); k.tiles_y = ceil_div(h, 8);
    k.x_bytes = (unsigned)x_bytes; k.u_bytes = 16u * 2u * 64u * 32u * 4u; k.w3_bytes = (unsigned)w3_bytes;
    // FLOPs executed: 16 GEMMs of (patches x 64 x 64) + the 1x1
    const double patches = (double)batch * k.tiles_x * k.tiles_y * 32.0;
    ProfScope scope(stream, c1 ? "unit_proj_wino_f32" : sc ? "conv_tail_wino_sc_f32" : "conv_tail_wino_f32",
                    2.0 * 16.0 * patches * 64.0 * 64.0 + 2.0 * m_ll * (double)s3.cout * (sc ? 128.0 : 64.0) +
                        (c1 ? 2.0 * (patches / 32.0) * 192.0 * 64.0 * 64.0 : 0.0),
                    4.0 * 64.0 * m_ll * (sc ? 2.0 : 1.0) + 4.0 * 64.0 * (16.0 * 64.0 + s3.cout * (sc ? 2.0 : 1.0)) +
                        4.0 * m_ll * s3.cout * ((sc ? 0.0 : 1.0) + (c.y32 ? 1.0 : 0.0) + (c.ysplit ? 1.0 : 0.0)));
    if (c1) hipLaunchKernelGGL(conv_tail_wino_kernel<2>, dim3(batch * k.tiles_x * k.tiles_y), dim3(256), 0, stream, k);
    else if (sc) hipLaunchKernelGGL(conv_tail_wino_kernel<1>, dim3(batch * k.tiles_x * k.tiles_y), dim3(256), 0, stream, k);
    else hipLaunchKernelGGL(conv_tail_wino_kernel<0>, dim3(batch * k.tiles_x * k.tiles_y), dim3(256), 0, stream, k);
    FD_LAUNCH_CHECK();
    *taken = true;
    return FD_OK;
}

}  // namespace fd
