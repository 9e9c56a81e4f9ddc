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
// band of RZ_TY destination rows and walks it in tiles of RZ_TX pixels (128 x 8 or 256 x 4, chosen by the
// host code).  For every tile the two tap rows of each destination row (2 RZ_TY staged rows; rows between
// the taps of a strong reduction are never touched), cut to the columns the tile's taps span (about RZ_TX / fx + 2 pixels), are fetched by LDS-DMA
// (`buffer_load_dwordx4 ... lds`): whole aligned 16-byte chunks of every source row, consecutive
// lanes on consecutive chunks, so each row segment is one coalesced burst and every source byte is
// fetched once per tile instead of once per tap and channel with single-byte loads; chunks past the
// end of the image return zero through the buffer's range check.  A workgroup keeps one
// staging buffer by default (eight workgroups, 32 waves, per CU cover each other's load latency);
// rings of two or three buffers, where the next tiles' rectangles are in flight while the current
// tile is computed, are selectable but measured slower: the kernel is bound by instruction issue
// as much as by memory.  Collecting a band's destination rows in LDS and writing them as whole
// 16-byte chunks was slower as well (fewer workgroups per CU).  The band's row taps and
// each tile's column taps are computed once (one thread each) into LDS tables.  A thread produces
// 4 neighbouring destination pixels: per pixel and source row three aligned LDS dwords and two
// byte-align operations give the 6 bytes of the two tap pixels, a byte permute pairs up each
// channel, and the row value is one 16-bit dot product; the 12 output bytes leave as one 12-byte
// store (consecutive lanes write consecutive addresses).  (One unaligned 8-byte LDS read per window
// instead of three aligned dwords gives the same bytes but runs 1.5x slower.)
// The band is walked along x by one workgroup, or by 2, 4 or 8 that are neighbours in the launch order and
// therefore run in the same XCD at about the same time (small launches: see the host code), on purpose: the
// RZ_TY destination rows of a band are one contiguous piece of memory, so all partial cache lines between
// neighbouring tiles are completed in the same L2, a few microseconds apart.  With neighbouring
// tiles on different workgroups (round-robin over the XCDs, i.e. different L2s) the stores ran at
// a third of the rate (ablation: 110 us for loads + stores against 73 us for the loads alone).
// Arithmetic is that of resize_linear_u8_kernel, bit for bit.
constexpr int RZ_ROUNDS = 8;          // DMA rounds of 256 chunks per tile: up to 32 KB of staged rows

struct Resize3Args {
    const uint8_t* src;
    uint8_t* dst;
    double scale_x, scale_y;
    int sh, sw, dh, dw;
    int pitch;            // bytes per staged row (multiple of 16)
    int max_rows;         // staged rows per tile
    int buf_bytes;        // bytes per staging buffer (whole 4 KB pieces: a DMA round of 256 lanes)
    unsigned inv_cpr;     // ceil(2^20 / chunks per row): chunk index -> row by multiplication
    int dst_aligned;      // every destination row starts on a 4-byte boundary (12-byte stores allowed)
    int stages;           // staging buffers in the ring (1 to 3): stages - 1 tiles are in flight while one is computed
    int xcd_runs;         // every XCD takes a contiguous run of (frame, band) pairs
    int batch;
    int xparts;           // a band's walk along x is cut into this many pieces, one workgroup each
};

// s_waitcnt vmcnt(n) for a wave-uniform n known only at run time (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt_upto(int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    }
#endif
}

// high 32 bits of the 48-bit product of two 24-bit factors (full-rate instruction)
__device__ __forceinline__ unsigned mulhi_u24(unsigned x, unsigned y) {
    unsigned r = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
#endif
    return r;
}

template <bool AREA2, bool REVERSE, int RZ_TX, int RZ_TY>
__global__ __launch_bounds__(256) void resize3_lds_kernel(Resize3Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) uint8_t rz_lds[];
    const int S = a.stages;
    const int tmask = S == 1 ? 1 : 3;                 // column-tap tables - 1
    int4* rowtab = reinterpret_cast<int4*>(rz_lds + S * (size_t)a.buf_bytes);   // [RZ_TY]: (tap rows, b0, b1)
    int2* coltab = reinterpret_cast<int2*>(rowtab + RZ_TY);                      // [4][RZ_TX]: (first tap column, w0 | w1 << 16)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sw = a.sw, sh = a.sh, dh = a.dh, dw = a.dw, pitch = a.pitch;
    // Workgroups are dealt round-robin to the 8 XCDs, each with an L2 of its own.  Neighbouring bands share
    // two or three source rows, so every XCD takes a contiguous run of (frame, band) pairs: the shared rows
    // are then fetched from memory once instead of once per L2.
    const unsigned nbands = (unsigned)((dh + RZ_TY - 1) / RZ_TY);
    const unsigned xparts = (unsigned)a.xparts;
    const unsigned nwork = nbands * (unsigned)a.batch * xparts, per_xcd = (nwork + 7u) >> 3;
    const unsigned unit = a.xcd_runs ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    if (unit >= nwork || (a.xcd_runs && (blockIdx.x >> 3) >= per_xcd)) return;
    const unsigned work = unit / xparts, part = unit - work * xparts;
    const int b = (int)(work / nbands);
    const int dy0 = (int)(work - (unsigned)b * nbands) * RZ_TY;
    const int ny = min(RZ_TY, dh - dy0);
    // this workgroup's tiles: j0 .. ntiles - 1 (the pieces of a band are neighbours in the launch order, so
    // they run in the same XCD at about the same time)
    const int tiles_all = (dw + RZ_TX - 1) / RZ_TX, per_part = (tiles_all + (int)xparts - 1) / (int)xparts;
    const int j0 = (int)part * per_part;
    const int ntiles = min(tiles_all, j0 + per_part);
    if (j0 >= ntiles) return;

    // the image as a buffer whose base is 16-byte aligned (the bytes before the image inside its first
    // chunk belong to the same allocation)
    const uint8_t* img = a.src + (size_t)b * sh * sw * 3;
    uint8_t* dimg = a.dst + (size_t)b * dh * dw * 3;
    const unsigned mis = (unsigned)(reinterpret_cast<uintptr_t>(img) & 15u);
    // The range ends at the end of the image's last 16-byte chunk: a 16-byte load is dropped as a whole
    // when it crosses the end of the range, and the bytes up to the chunk boundary lie in the same
    // allocation granule as the image's last byte.
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(img - mis), 0, (unsigned)(((size_t)sh * sw * 3 + mis + 15) & ~(size_t)15), 0x00020000);
    const int cpr = pitch >> 4;                       // 16-byte chunks per staged row
    const unsigned row_bytes = (unsigned)sw * 3u;

    // column taps of tile j (one thread per destination column) into table j % 4 (j % 2 with a single
    // staging buffer), because tile j + S - 1's is written while slower waves may still be reading tile
    // j - 1's; columns past the image repeat its last one, so a table is always full
    auto col_taps = [&](int j) {
#pragma unroll
        for (int c = tid; c < RZ_TX; c += 256) {
            const int dx = min(j * RZ_TX + c, dw - 1);
            int2 e;
            if (AREA2) {
                e = make_int2(2 * dx, 0);
            } else {
                const Tap2 t = linear_tap(dx, a.scale_x, sw);
                e = make_int2(t.s0, t.w0 | (t.w1 << 16));
            }
            // entry of column dx at (dx % 4) * (RZ_TX / 4) + dx / 4: the lanes of a wave, which own consecutive
            // 4-pixel groups, then read consecutive 8-byte entries for each of their 4 pixels (stored in column
            // order the reads of a wave would be 32 bytes apart and hit 4 of the 32 LDS banks)
            coltab[(j & tmask) * RZ_TX + (c & 3) * (RZ_TX / 4) + (c >> 2)] = e;
        }
    };
    // row taps of the band, by the threads the column taps leave idle (if any)
    constexpr int ROW_T0 = RZ_TX + RZ_TY <= 256 ? RZ_TX : 0;      // first of the RZ_TY threads that compute them
    if (tid >= ROW_T0 && tid - ROW_T0 < RZ_TY) {
        const int dy = min(dy0 + tid - ROW_T0, dh - 1);
        int4 e;
        if (AREA2) {
            e = make_int4(2 * dy, 2 * dy + 1, 0, 0);
        } else {
            const Tap2 t = linear_tap(dy, a.scale_y, sh);
            e = make_int4(t.s0, t.s1, t.w0, t.w1);
        }
        rowtab[tid - ROW_T0] = e;
    }
    col_taps(j0);
    if (S > 2 && ntiles > j0 + 1) col_taps(j0 + 1);
    __syncthreads();
    // Staged rows: 2 i and 2 i + 1 hold the two tap rows of destination row i -- only the source rows the
    // band uses are fetched (a reduction by 3.84 touches 2 rows in 3.84; a contiguous rectangle would also
    // carry the rows in between).  Which staged row and chunk lane `tid` moves in DMA round k does not
    // depend on the tile: byte offset of the source row in the image, and of the chunk inside the row.
    unsigned dma_row[RZ_ROUNDS], dma_chunk[RZ_ROUNDS];
#pragma unroll
    for (int k = 0; k < RZ_ROUNDS; ++k) {
        const unsigned i = (unsigned)(k * 256 + tid);
        const unsigned r = min(__umul24(i, a.inv_cpr) >> 20, (unsigned)(2 * RZ_TY - 1));
        const int4 e = rowtab[r >> 1];
        dma_row[k] = (unsigned)((r & 1u) ? e.y : e.x) * row_bytes;
        dma_chunk[k] = (i - __umul24(r, (unsigned)cpr)) * 16u;
    }
    const int nrows = 2 * ny;
    const int total = nrows * cpr;
    const int rounds = (total + 255) >> 8;            // DMA operations per wave and tile
    // LDS-DMA of tile j's source rectangle into buffer `slot` = j % S (its column table must be visible)
    auto issue = [&](int j, int slot) {
        const int c_lo = coltab[(j & tmask) * RZ_TX].x;
        const unsigned origin = mis + (unsigned)c_lo * 3u;
        uint8_t* buf = rz_lds + (size_t)slot * a.buf_bytes;
#pragma unroll
        for (int k = 0; k < RZ_ROUNDS; ++k) {
            if (k * 256 >= total) break;
            const unsigned vo = k * 256 + tid < total ? ((origin + dma_row[k]) & ~15u) + dma_chunk[k] : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rs, (__attribute__((address_space(3))) void*)(buf + (size_t)(k * 256 + wave * 64) * 16), 16, vo, 0, 0, 0);
        }
    };
    issue(j0, 0);
    if (S > 2 && ntiles > j0 + 1) issue(j0 + 1, 1);
    if (S == 1 && ntiles > j0 + 1) col_taps(j0 + 1);            // into the other table, while tile j0 is in flight

    // this thread's destination row and 4-pixel group inside every tile, and the staged rows of its taps
    constexpr int GX = RZ_TX / 4;
    const int ry = tid / GX, gx = (tid - ry * GX) * 4;
    const int4 rt = rowtab[ry];
    const unsigned rowpos0 = __umul24((unsigned)(2 * ry), (unsigned)pitch), rowpos1 = rowpos0 + (unsigned)pitch;
    const unsigned rowoff0 = (unsigned)rt.x * row_bytes, rowoff1 = (unsigned)rt.y * row_bytes;
    const unsigned orow = __umul24((unsigned)(dy0 + ry), (unsigned)dw);
    const unsigned bz = (unsigned)rt.z << 12, bw = (unsigned)rt.w << 12;     // row weights (<= 2^11) as 24-bit factors

    // a wave whose rows lie inside the band issues one store per tile when rows are 4-byte aligned
    const int st = ((wave * 64) / GX < ny && a.dst_aligned) ? 1 : 0;       // the wave's first row
    int slot = 0;                                               // j % S
    for (int j = j0; j < ntiles; ++j) {
        if (S == 1) {
            // single buffer: latency is covered by the other workgroups of the CU (seven fit) instead
            if (j > j0) {
                __syncthreads();                                // everyone has finished reading tile j - 1
                issue(j, 0);
                if (j + 1 < ntiles) col_taps(j + 1);            // its table was last read for tile j - 1
            }
        } else if (j + S - 1 < ntiles) col_taps(j + S - 1);
        // This wave's part of tile j has landed.  Vector-memory operations retire in issue order; after the
        // loads of tile j this wave issued (S = 3) the store of tile j - 2, the loads of tile j + 1 and the
        // store of tile j - 1, or (S = 2) the store of tile j - 1 only.  The loads of the tile ahead and the
        // youngest store may stay in flight, so no tile waits out a write round trip or the prefetch.
        // (A wave that stores byte by byte counts none of its stores: waiting longer is always safe.)
        // With a single buffer the order is the other way round -- the store of tile j - 1 was issued BEFORE
        // the loads of tile j -- so the youngest operations are the loads themselves and nothing may stay
        // in flight.
        wait_vmcnt_upto(S == 1 ? 0 : (S > 2 && j + 1 < ntiles ? rounds : 0) + (j > j0 ? st : 0));
        __syncthreads();                                        // ... everyone's; the taps written above are visible
        if (S > 1 && j + S - 1 < ntiles) {                      // that buffer was last read while computing tile j - 1
            int ns = slot + S - 1;
            if (ns >= S) ns -= S;
            issue(j + S - 1, ns);
        }
        const int cur = slot;
        if (++slot == S) slot = 0;
        const int dx0 = j * RZ_TX;
        const int nx = min(RZ_TX, dw - dx0);
        if (ry >= ny || gx >= nx) continue;
        const int2* ctab = coltab + (j & tmask) * RZ_TX;
        const uint8_t* buf = rz_lds + (size_t)cur * a.buf_bytes;
        const int c_lo = ctab[0].x;
        const unsigned origin = mis + (unsigned)c_lo * 3u;
        // byte position of source column c_lo inside each of the two staged rows (their alignment differs)
        const unsigned base0 = rowpos0 + ((origin + rowoff0) & 15u);
        const unsigned base1 = rowpos1 + ((origin + rowoff1) & 15u);
        unsigned vout[12];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const int2 ct = ctab[px * GX + (gx >> 2)];
            const unsigned coff = (unsigned)(ct.x - c_lo) * 3u;
            unsigned lo[2], hi[2];                     // bytes 0..3 and 4..7 from the first tap pixel on, per row
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned ad = (r ? base1 : base0) + coff;
                const unsigned* q = reinterpret_cast<const unsigned*>(buf + (ad & ~3u));
                const unsigned d0 = q[0], d1 = q[1], d2 = q[2];
                lo[r] = __builtin_amdgcn_alignbyte(d1, d0, ad & 3u);
                hi[r] = __builtin_amdgcn_alignbyte(d2, d1, ad & 3u);
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
                    // row value S[s0] * w0 + S[s1] * w1: one dot product of 16-bit pairs
                    typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
                    const u16x2_t wp = __builtin_bit_cast(u16x2_t, (unsigned)ct.y);
                    const unsigned h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, p0), wp, 0u, false);
                    const unsigned h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, p1), wp, 0u, false);
                    // b <= 2^11, h >> 4 < 2^15: 24-bit multiplies (32-bit ones run at a quarter of the rate)
                    // (b * (h >> 4)) >> 16 as the high half of a 24-bit product: ((b << 12) * (h & ~15)) >> 32
                    v = (mulhi_u24(bz, h0 & ~15u) + mulhi_u24(bw, h1 & ~15u) + 2u) >> 2;
                }
                vout[px * 3 + (REVERSE ? 2 - k : k)] = v;
            }
        }
        uint8_t* o = dimg + (orow + (unsigned)(dx0 + gx)) * 3u;
        if (gx + 4 <= nx && a.dst_aligned) {
            typedef unsigned u32x3_t __attribute__((ext_vector_type(3)));
            u32x3_t w;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                w[q] = vout[4 * q] | (vout[4 * q + 1] << 8) | (vout[4 * q + 2] << 16) | (vout[4 * q + 3] << 24);
            *reinterpret_cast<u32x3_t*>(o) = w;
        } else {
            const int nb = 3 * min(4, nx - gx);
#pragma unroll
            for (int q = 0; q < 12; ++q)
                if (q < nb) o[q] = (uint8_t)vout[q];
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
    // 3-channel frames take the LDS-staged kernel whenever two tiles' source rectangles fit in LDS (any
    // scale from enlargements down to about 1/6); the per-pixel kernel is the general fallback.
    // Tile of a workgroup: tx x ty destination pixels, one 4-pixel group per thread (tx / 4 * ty = 256).  256 x 4 for
    // moderate reductions of wide frames: row segments twice as long per burst (3 DMA rounds per tile instead of 4 at
    // 1080p -> 562x1000, half as many partial cache lines between tiles): 106-113 us against 111-114 us on 64 frames,
    // the same on 128.  128 x 8 otherwise: a strong reduction (4K -> 562x1000) stages the same number of chunks
    // either way and then the column taps, computed once per tile and shared by its rows, cost twice as much per
    // pixel on the flatter tile (40.5 against 38 us on 16 frames; 512 x 2 was slower everywhere).
    // FD_RESIZE_TILE=128|256 overrides (measurements).
    static const int env_tile = getenv("FD_RESIZE_TILE") ? atoi(getenv("FD_RESIZE_TILE")) : 0;
    const int tx = env_tile == 256 || env_tile == 128 ? env_tile : ((scale_x < 3.0 && dst_w >= 512) ? 256 : 128);
    const int ty = 1024 / tx;
    static const bool allow_lds = !(getenv("FD_RESIZE_LDS") && atoi(getenv("FD_RESIZE_LDS")) == 0);
    if (channels == 3 && allow_lds && (size_t)src_h * src_w * 3 < (1ull << 31) && batch <= 65535) {
        fd::Resize3Args a;
        a.src = src; a.dst = dst; a.scale_x = scale_x; a.scale_y = scale_y;
        a.sh = src_h; a.sw = src_w; a.dh = dst_h; a.dw = dst_w;
        a.max_rows = 2 * ty;                              // the two tap rows of every destination row
        const int max_cols = std::min(src_w, (int)std::ceil(tx * scale_x) + 3);
        a.pitch = ((max_cols * 3 + 15 + 8) + 15) / 16 * 16;     // worst lead + the taps' 8-byte window, in whole chunks
        const int cpr = a.pitch / 16;
        const int chunks = a.max_rows * cpr;
        a.buf_bytes = (chunks + 255) / 256 * 4096;
        a.inv_cpr = (unsigned)(((1u << 20) + cpr - 1) / cpr);
        a.dst_aligned = ((reinterpret_cast<uintptr_t>(dst) & 3u) == 0 && ((size_t)dst_w * 3) % 4 == 0 &&
                         ((size_t)dst_h * dst_w * 3) % 4 == 0) ? 1 : 0;
        // One staging buffer per workgroup by default: the kernel is bound by instruction issue as much as by
        // memory, and eight single-buffer workgroups per CU (32 waves, each other's latency cover) measured
        // 10 % faster on 1080p frames than four double-buffered or three triple-buffered ones.
        // FD_RESIZE_STAGES (1-3) / FD_RESIZE_XCD override (measurements).
        static const int env_stages = getenv("FD_RESIZE_STAGES") ? atoi(getenv("FD_RESIZE_STAGES")) : 0;
        static const int env_xcd = getenv("FD_RESIZE_XCD") ? atoi(getenv("FD_RESIZE_XCD")) : 1;
        a.stages = env_stages >= 1 && env_stages <= 3 ? env_stages : 1;
        const size_t tables = (a.stages == 1 ? 2 : 4) * tx * sizeof(int2) + ty * sizeof(int4);
        static const int env_xparts = getenv("FD_RESIZE_XPARTS") ? atoi(getenv("FD_RESIZE_XPARTS")) : 0;
        a.xcd_runs = env_xcd ? 1 : 0;
        a.batch = batch;
        // A band's walk along x is cut into 1, 2, 4 or 8 pieces so that the launch has at least ~8192 workgroups
        // (four rounds of the 2048 that are resident at a time): 64 1080p frames take 2 pieces (-2 %: a shorter
        // tail; 4 pieces cost more in per-workgroup set-up than they gain), 16 4K frames 8 (-9 %).
        a.xparts = 1;
        while (a.xparts < 8 && (long long)fd::ceil_div(dst_h, ty) * batch * a.xparts < 8192) a.xparts *= 2;
        if (env_xparts >= 1 && env_xparts <= 8) a.xparts = env_xparts;
        const size_t smem = a.stages * (size_t)a.buf_bytes + tables;
        // chunk -> row by multiplication is exact for chunk indices below 2^20 / cpr and needs cpr >= 2
        if (smem <= 64 * 1024 && cpr >= 2 && chunks <= fd::RZ_ROUNDS * 256 && chunks + 256 <= (1 << 20) / cpr &&
            (size_t)dst_h * dst_w * 3 < (1ull << 31) && dst_h < (1 << 23) && dst_w < (1 << 23) &&
            (long long)fd::ceil_div(dst_h, ty) * batch * a.xparts < (1ll << 31) - 8) {
            const long long nwork = (long long)fd::ceil_div(dst_h, ty) * batch * a.xparts;
            dim3 grid((unsigned)((nwork + 7) / 8 * 8));
            const hipStream_t st = fd::as_stream(stream);
#define FD_RZ(A2, REV)                                                                                       \
    do {                                                                                                     \
        if (tx == 256) hipLaunchKernelGGL((fd::resize3_lds_kernel<A2, REV, 256, 4>), grid, dim3(256), smem, st, a); \
        else hipLaunchKernelGGL((fd::resize3_lds_kernel<A2, REV, 128, 8>), grid, dim3(256), smem, st, a);   \
    } while (0)
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
