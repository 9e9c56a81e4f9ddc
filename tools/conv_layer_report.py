#!/usr/bin/env python3
"""Per-layer convolution timing from a rocprofv3 kernel trace of bench.py.

    python tools/conv_layer_report.py <kernel_trace.csv> [micro_batch] [f16x3 | f32 | f32c1 | f32sep | f32pair | layers] [winograd tile: 4 | 2]
       (f32c1: a float32 trace taken with FD_TAIL_C1=0 FD_SC_IN_C3=0; f32sep: with FD_TAIL_SC=0 FD_SC_IN_C3=0; f32pair: ... and FD_F32_PAIR=1)

Takes the convolution launches of the last complete forward pass in the trace, pairs them with
the launch list of faster_rcnn50 at 1080p (same order as the library issues them, see layer_list)
and prints duration and achieved TFLOP/s (FLOPs executed) per launch.  Use a single-stream trace of
one precision (bench.py --streams 1 --no-split-record [--precision f16x3])."""
import csv
import sys


def layer_list(B, H=1080, W=1920, R=10, mode="f16x3", tile=4):
    """Convolution launches of one forward pass in issue order: (name, M, N, K, FLOPs executed).
    mode "f16x3": the stem runs fused with the max-pool, each unit of the 64-channel stage is one kernel, and (tile 4)
    the RPN's 3x3 layer and the stride-1 3x3 layers of the 256-channel stage are one batched Winograd GEMM launch
    each, rows padded to whole 256-row tiles;
    mode "f32": fused stem, the 3x3 + 1x1 tail of each 64-channel unit is one kernel, and the stride-1 3x3
    layers from 128 channels on (256 with tile 2) and the RPN's are one batched Winograd GEMM launch each
    ((tile + 2)^2 GEMMs over the tile x tile patches: 1/4 of the direct FLOPs with tile 4, 4/9 with tile 2,
    rows padded to whole GEMM tiles), and the stride-1 projection unit is one launch (its shortcut inside the tail's
    1x1, its conv1 computed on the tail's input patch: 192 rows per 8 x 16 tile); mode "f32c1": "f32" with that conv1 as
    a launch of its own (FD_TAIL_C1=0); mode "f32sep": ... and the shortcut too (FD_TAIL_SC=0); mode "f32pair":
    "f32sep" with the shortcut and conv1 as one launch (FD_F32_PAIR=1); mode "layers": every layer its own launch."""
    pair = mode == "f32pair"
    sc_in_tail = mode in ("f32", "f32c1")
    c1_in_tail = mode == "f32"
    sc_in_c3 = mode == "f32"          # projection units of the 128- and 256-channel stages: shortcut inside conv3's GEMM
    if mode in ("f32pair", "f32sep", "f32c1"):
        mode = "f32"
    layers = []

    def conv(name, n, h, w, cin, cout, k, s, p, wino=False):
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        if wino:
            t = n * ((h + tile - 1) // tile) * ((w + tile - 1) // tile)
            align = 256 if mode == "f16x3" else 128
            tp = (t + align - 1) // align * align
            nb = (tile + 2) ** 2
            layers.append((name + "w", nb * tp, cout, cin, 2.0 * nb * tp * cout * cin))
        else:
            layers.append((name, n * ho * wo, cout, cin * k * k, 2.0 * n * ho * wo * cout * cin * k * k))
        return ho, wo

    h, w = conv("conv0", B, H, W, 3, 64, 7, 2, 3)
    h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    F, U = [64, 256, 512, 1024, 2048], [3, 4, 6, 3]
    cin = 64
    for s in range(4):
        if s == 3:
            conv("rpn3x3", B, h, w, 1024, 512, 3, 1, 1, wino=(mode == "f32" or (mode == "f16x3" and tile == 4)))
            conv("rpnout", B, h, w, 512, 72, 1, 1, 0)
            n, h, w = B * R, 7, 7
        else:
            n = B
        cout, mid = F[s + 1], F[s + 1] // 4
        for u in range(U[s]):
            st = 2 if (u == 0 and s > 0) else 1
            nm = "s%du%d" % (s + 1, u + 1)
            if u == 0:
                conv(nm + "_sc", n, h, w, cin, cout, 1, st, 0)
            conv(nm + "_c1", n, h, w, cin, mid, 1, 1, 0)
            wino = mode == "f32" and st == 1 and mid >= (128 if tile == 4 else 256) and (h * w > 49 or n * h * w >= 16384)
            wino = wino or (mode == "f16x3" and tile == 4 and st == 1 and mid >= 256 and h * w > 49)
            h2, w2 = conv(nm + "_c2", n, h, w, mid, mid, 3, st, 1, wino=wino)
            conv(nm + "_c3", n, h2, w2, mid, cout, 1, 1, 0)
            if sc_in_c3 and u == 0 and s in (1, 2):
                c3_row, c2_row, c1_row, sc_row = layers.pop(), layers.pop(), layers.pop(), layers.pop()
                layers.extend([c1_row, c2_row, (c3_row[0], c3_row[1], c3_row[2], c3_row[3] + sc_row[3], c3_row[4] + sc_row[4])])
            if mid == 64 and st == 1 and mode != "layers":
                # f16x3: whole unit in one kernel (4 layers for the projection unit); f32: conv2 + conv3
                k = 2 if mode == "f32" else (4 if u == 0 else 3)
                parts = [layers.pop() for _ in range(k)]
                flops = sum(p[4] for p in parts)
                if mode == "f32":
                    # the tail's 3x3 runs as Winograd F(2x2,3x3) inside the kernel: 16 GEMMs over the 2 x 2 patches of
                    # whole 8 x 16 pixel tiles, plus the 1x1 (FLOPs executed, like the other rows)
                    patches = n * ((h + 7) // 8) * ((w + 15) // 16) * 32
                    flops = 2.0 * 16 * patches * mid * mid + 2.0 * n * h2 * w2 * mid * cout
                layers.append((nm + ("_tail" if mode == "f32" else "_unit"), parts[0][1], parts[0][2],
                               sum(p[3] for p in parts), flops))
                if sc_in_tail and u == 0:
                    # the shortcut's 64 channels of K inside the tail's 1x1; no shortcut launch
                    tail_row = layers.pop()
                    c1_row, sc_row = layers.pop(), layers.pop()
                    if c1_in_tail:
                        tiles = n * ((h + 7) // 8) * ((w + 15) // 16)
                        layers.append((nm + "_unit", tail_row[1], tail_row[2], tail_row[3] + sc_row[3] + c1_row[3],
                                       tail_row[4] + sc_row[4] + 2.0 * tiles * 192 * 64 * 64))
                    else:
                        layers.append(c1_row)
                        layers.append((tail_row[0], tail_row[1], tail_row[2], tail_row[3] + sc_row[3], tail_row[4] + sc_row[4]))
                if pair and u == 0:
                    # the stride-1 projection unit's shortcut and conv1 (same input) as one launch
                    tail_row = layers.pop()
                    c1_row, sc_row = layers.pop(), layers.pop()
                    layers.append((nm + "_sc+c1", sc_row[1], sc_row[2] + c1_row[2], sc_row[3], sc_row[4] + c1_row[4]))
                    layers.append(tail_row)
            h, w, cin = h2, w2, cout
    return layers


def main():
    path = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rows = [r for r in csv.DictReader(open(path))
            if any(t in r["Kernel_Name"] for t in ("conv_igemm", "conv_split_", "stem_pool", "conv_tail_fused", "unit_fused", "unit_proj_fused", "conv_f32_panel", "conv_tail_wino"))]
    layers = layer_list(B, mode=sys.argv[3] if len(sys.argv) > 3 else "f16x3", tile=int(sys.argv[4]) if len(sys.argv) > 4 else 4)
    last = rows[-len(layers):]
    tot = totf = 0.0
    groups = {}
    for L, r in zip(layers, last):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += us
        totf += L[4]
        kind = L[0].split("_")[-1].split("+")[0] if "_" in L[0] else L[0]
        g = groups.setdefault(kind, [0.0, 0.0])
        g[0] += us
        g[1] += L[4]
        kn = r["Kernel_Name"]
        tile = (kn.split("<")[1].split(">")[0] if "<" in kn else kn.split("::")[-1].split("(")[0]) + (" glds" if "glds" in kn else "")
        print("%-9s M=%8d N=%5d K=%5d %9.1f us %7.1f TF  tile %s" % (L[0], L[1], L[2], L[3], us, L[4] / us / 1e6, tile))
    print("total %.1f us for %d frames = %.2f ms/frame, %.1f TF" % (tot, B, tot / B / 1e3, totf / tot / 1e6))
    for k, (us, f) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
        print("  %-8s %9.1f us  %5.1f%%  %6.1f TF" % (k, us, 100 * us / tot, f / us / 1e6))


if __name__ == "__main__":
    main()
