#!/usr/bin/env python3
"""Every convolution launch of the float32 pass against BOTH of its floors.

    python tools/ridge_table.py <layers_f32.txt written by tools/conv_layer_report.py> [micro_batch=16]

For each launch of the per-layer table: FLOPs executed, the bytes its tensors hold (input rows it reads, weights,
residual, output -- each once), FLOP per byte, the time its FLOPs need at the fp32 matrix peak (157.3 TFLOP/s), the
time its bytes need at the rate a device-to-device copy reaches on this part (4.5 TB/s; the 8 TB/s data-sheet figure
in brackets), the measured duration, and measured / max(floors).  A launch whose two floors are close sits at the
ridge: it cannot reach either roof, because both pipes would have to be busy all the time."""
import re
import sys

PEAK_TF = 157.3
COPY_TBS = 4.5
SPEC_TBS = 8.0


def layer_bytes(name, M, N, K, B):
    """Bytes of the tensors a launch touches, each once (float32)."""
    w = 4.0 * N * K
    if name == "conv0":
        # staged image: 4 floats per pixel, 1080 x 1920; output: the pooled, activated 64-channel tensor
        return 16.0 * B * 1080 * 1920 + w + 4.0 * (M / 4) * 64
    if name.endswith("_unit") and K > 640:
        # the whole projection unit: the activated input (64 channels; the patch and the shortcut operand), output
        return 4.0 * M * 64 + 4.0 * M * N + 4.0 * 64 * (64 + 576 + 2 * N)
    if name.endswith("_tail"):
        if K > 640:
            # the projection unit: t1 and the activated input (64 channels each), output; no residual tensor
            return 4.0 * M * 128 + 4.0 * M * N + 4.0 * 64 * (576 + 2 * N)
        # t1 (64 channels), residual and output (256 channels); weights of both convolutions
        return 4.0 * M * 64 + 2 * 4.0 * M * N + 4.0 * 64 * (576 + N)
    if name.endswith("_sc+c1"):
        return 4.0 * M * 64 + 4.0 * M * N + w
    kind = name.split("_")[-1] if "_" in name else name
    if kind in ("c2w", "rpn3x3w"):
        return 4.0 * M * K + 4.0 * M * N + 36 * w          # 36 GEMMs: transformed input, transformed output
    if kind == "c2":
        # direct 3x3 with stride 2: every input pixel is read (4 M pixels of K / 9 channels)
        return 4.0 * (4 * M) * (K / 9) + w + 4.0 * M * N
    if kind == "c3":
        if name in ("s2u1_c3", "s3u1_c3") and K > N // 4:
            return 4.0 * M * K + w + 4.0 * M * N            # the projection shortcut inside: two-source rows, no residual
        return 4.0 * M * K + w + 2 * 4.0 * M * N           # + residual
    return 4.0 * M * K + w + 4.0 * M * N                    # sc (its strided rows only), c1, rpnout


def main():
    path = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    pat = re.compile(r"^(\S+)\s+M=\s*(\d+)\s+N=\s*(\d+)\s+K=\s*(\d+)\s+([\d.]+) us\s+([\d.]+) TF")
    print("%-10s %9s %8s %7s %9s %15s %9s %9s" % ("launch", "GFLOP", "MB", "FLOP/B", "matrix us", "memory us (8TB/s)", "measured", "of floor"))
    tot = [0.0, 0.0, 0.0, 0.0]
    for line in open(path):
        m = pat.match(line)
        if not m:
            continue
        name, M, N, K, us, tf = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5)), float(m.group(6))
        flops = tf * us * 1e6
        by = layer_bytes(name, M, N, K, B)
        t_m = flops / (PEAK_TF * 1e6)
        t_h = by / (COPY_TBS * 1e6)
        t_s = by / (SPEC_TBS * 1e6)
        floor = max(t_m, t_h)
        print("%-10s %9.2f %8.1f %7.1f %9.1f %8.1f (%5.1f) %9.1f %9.2f" % (name, flops / 1e9, by / 1e6, flops / by, t_m, t_h, t_s, us, floor / us))
        tot[0] += t_m; tot[1] += t_h; tot[2] += us; tot[3] += floor
    print("sum of matrix floors %.0f us, of memory floors %.0f us, of the larger floor per launch %.0f us, measured %.0f us: %.2f of the floors"
          % (tot[0], tot[1], tot[3], tot[2], tot[3] / tot[2]))


if __name__ == "__main__":
    main()
