"""Seeded synthetic inputs: a random-weight checkpoint with the parameter names of the
reference's ``mxnet-face-fr50`` model and synthetic frames.

The reference ships no weights (``mx.model.load_checkpoint`` at detection.py:28 reads files
that are not in its tree), so tests and benchmarks run on random weights of the same
architecture.  Plain random weights would make activations drift by orders of magnitude
over 50 layers and saturate every score; to keep the numbers in a realistic range each
BatchNorm's running statistics are centred on the scalar mean / variance its input shows on
a calibration frame (table ``CALIB`` below, produced once by tools/calibrate_synthetic.py),
with seeded per-channel jitter on top.  Everything is a pure function of ``seed``.
"""
import zlib

import numpy as np

UNITS = (3, 4, 6, 3)
FILTERS = (64, 256, 512, 1024, 2048)
NUM_ANCHORS = 12

# name -> (mean, variance) of the tensor entering that BatchNorm; for the four output layers
# the standard deviation (and, for the two classifiers, the mean) of their raw outputs --
# foreground minus background logit for the classifiers -- under unit-gain He weights.
CALIB = {
    'bn_data': (119.688, 1178.03),
    'bn0': (0.00762361, 2.25081),
    'stage1_unit1_bn1': (0.652821, 0.55882),
    'stage1_unit1_bn2': (-0.300662, 1.76198),
    'stage1_unit1_bn3': (0.0754063, 1.34167),
    'stage1_unit2_bn1': (0.00791882, 2.75068),
    'stage1_unit2_bn2': (0.068814, 1.34444),
    'stage1_unit2_bn3': (-0.0134308, 1.12103),
    'stage1_unit3_bn1': (-0.0386111, 3.71478),
    'stage1_unit3_bn2': (0.0539666, 1.00908),
    'stage1_unit3_bn3': (-0.146606, 2.03291),
    'stage2_unit1_bn1': (-0.00928955, 4.90233),
    'stage2_unit1_bn2': (0.0160403, 1.21618),
    'stage2_unit1_bn3': (-0.0626986, 1.34859),
    'stage2_unit2_bn1': (0.0184729, 2.55302),
    'stage2_unit2_bn2': (0.0199104, 1.01631),
    'stage2_unit2_bn3': (0.060661, 1.22869),
    'stage2_unit3_bn1': (0.0298522, 3.75697),
    'stage2_unit3_bn2': (-0.0987688, 1.28595),
    'stage2_unit3_bn3': (-0.0196249, 1.25377),
    'stage2_unit4_bn1': (0.105038, 4.80766),
    'stage2_unit4_bn2': (0.0073964, 1.01422),
    'stage2_unit4_bn3': (0.0114107, 1.29514),
    'stage3_unit1_bn1': (0.123822, 6.22848),
    'stage3_unit1_bn2': (-0.0575413, 1.12543),
    'stage3_unit1_bn3': (0.0716235, 1.14704),
    'stage3_unit2_bn1': (0.0555038, 2.28967),
    'stage3_unit2_bn2': (0.0880657, 1.05705),
    'stage3_unit2_bn3': (0.094961, 1.12298),
    'stage3_unit3_bn1': (0.0779717, 3.26961),
    'stage3_unit3_bn2': (0.062095, 1.19447),
    'stage3_unit3_bn3': (-0.070378, 1.41293),
    'stage3_unit4_bn1': (0.119071, 4.19074),
    'stage3_unit4_bn2': (-0.0346638, 1.19446),
    'stage3_unit4_bn3': (0.0673733, 1.00921),
    'stage3_unit5_bn1': (0.119627, 5.31704),
    'stage3_unit5_bn2': (0.00254195, 1.12218),
    'stage3_unit5_bn3': (-0.0204866, 1.12933),
    'stage3_unit6_bn1': (0.126713, 6.34677),
    'stage3_unit6_bn2': (0.0136965, 1.34332),
    'stage3_unit6_bn3': (-0.0196404, 1.15205),
    'rpn_cls_score': (5.26746, 1.50267),
    'rpn_bbox_pred': (3.77877,),
    'stage4_unit1_bn1': (1.2707, 8.04899),
    'stage4_unit1_bn2': (-0.022863, 1.16298),
    'stage4_unit1_bn3': (0.0300611, 0.699524),
    'stage4_unit2_bn1': (0.0632341, 2.28059),
    'stage4_unit2_bn2': (-0.0533246, 1.15755),
    'stage4_unit2_bn3': (-0.0308727, 0.827568),
    'stage4_unit3_bn1': (0.0528561, 3.38589),
    'stage4_unit3_bn2': (-0.0142088, 1.14148),
    'stage4_unit3_bn3': (0.0357061, 0.895522),
    'bn1': (0.0530772, 4.40728),
    'cls_score': (0.231979, 1.37297),
    'bbox_pred': (0.88171,),
}

# what the output layers should produce
TARGETS = {
    "rpn_cls_score": 2.5,    # std of the fg-bg logit difference
    "rpn_bbox_pred": 0.25,   # std of the RPN deltas
    "cls_score": 2.0,        # std of the head's logit difference
    "bbox_pred": 0.15,       # std of the head's deltas
}
RPN_FG_BIAS = -3.0           # pushes most anchors towards background
HEAD_FG_BIAS = 1.0


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode("ascii"))])


def conv_shapes():
    """(name, (Cout, Cin, kh, kw)) of every convolution / FC weight, in graph order."""
    out = [("conv0", (FILTERS[0], 3, 7, 7))]
    cin = FILTERS[0]
    for s in range(4):
        cout = FILTERS[s + 1]
        mid = cout // 4
        for u in range(UNITS[s]):
            name = "stage%d_unit%d" % (s + 1, u + 1)
            out.append((name + "_conv1", (mid, cin, 1, 1)))
            out.append((name + "_conv2", (mid, mid, 3, 3)))
            out.append((name + "_conv3", (cout, mid, 1, 1)))
            if u == 0:
                out.append((name + "_sc", (cout, cin, 1, 1)))
            cin = cout
        if s == 2:
            out.append(("rpn_conv_3x3", (512, cout, 3, 3)))
            out.append(("rpn_cls_score", (2 * NUM_ANCHORS, 512, 1, 1)))
            out.append(("rpn_bbox_pred", (4 * NUM_ANCHORS, 512, 1, 1)))
    out.append(("cls_score", (2, FILTERS[4])))
    out.append(("bbox_pred", (8, FILTERS[4])))
    return out


def bn_shapes():
    """(name, channels) of every BatchNorm, in graph order."""
    out = [("bn_data", 3), ("bn0", FILTERS[0])]
    cin = FILTERS[0]
    for s in range(4):
        cout = FILTERS[s + 1]
        mid = cout // 4
        for u in range(UNITS[s]):
            name = "stage%d_unit%d" % (s + 1, u + 1)
            out += [(name + "_bn1", cin), (name + "_bn2", mid), (name + "_bn3", mid)]
            cin = cout
    out.append(("bn1", FILTERS[4]))
    return out


def bn_params(seed, name, channels, mean, var):
    """gamma / beta / running statistics of one BatchNorm around the scalar (mean, var)."""
    g = _rng(seed, name)
    std = float(np.sqrt(var))
    return {
        name + "_gamma": g.uniform(0.5, 1.5, channels).astype(np.float32),
        name + "_beta": g.normal(0.0, 0.1, channels).astype(np.float32),
        name + "_moving_mean": (mean + 0.1 * std * g.normal(0.0, 1.0, channels)).astype(np.float32),
        name + "_moving_var": (var * g.uniform(0.6, 1.5, channels)).astype(np.float32),
    }


def synthetic_params(seed=0, calib=None):
    """dict name -> float32 array with every parameter / auxiliary state of faster_rcnn50(num_class=2)."""
    calib = CALIB if calib is None else calib
    p = {}
    for name, shape in conv_shapes():
        g = _rng(seed, name + "_weight")
        fan_in = int(np.prod(shape[1:]))
        w = g.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in))
        if name in TARGETS:
            w *= np.float32(TARGETS[name] / calib.get(name, (1.0,))[0])
        p[name + "_weight"] = w
    for name in ("rpn_conv_3x3", "rpn_cls_score", "rpn_bbox_pred", "cls_score", "bbox_pred"):
        g = _rng(seed, name + "_bias")
        n = p[name + "_weight"].shape[0]
        b = g.normal(0.0, 0.05, n).astype(np.float32)
        # classifiers: cancel the common offset random weights leave on the logit difference,
        # then shift it to the wanted operating point
        if name in ("rpn_cls_score", "cls_score"):
            c = calib.get(name, (1.0, 0.0))
            offset = (c[1] if len(c) > 1 else 0.0) * TARGETS[name] / c[0]
            shift = RPN_FG_BIAS if name == "rpn_cls_score" else HEAD_FG_BIAS
            b[n // 2:] += np.float32(shift - offset)
        p[name + "_bias"] = b
    for name, ch in bn_shapes():
        mean, var = calib.get(name, (0.0, 1.0))
        p.update(bn_params(seed, name, ch, mean, var))
    return p


def scaled_params(params, s, eps=2e-5):
    """The same network with every intermediate activation multiplied by ``s`` (a power of two):
    in exact arithmetic the graph outputs do not change, so a float64 evaluation of the returned
    parameters is a yardstick for how an arithmetic copes with large (s >> 1) or tiny (s << 1)
    activations.  bn0 and the three BatchNorms of every unit get gamma, beta and moving_mean
    multiplied by s and their variance replaced by (var + eps) s^2 - eps (the normalised value is
    unchanged, the output is s times larger); convolutions are linear, the residual stream follows.
    The two consumers that must see unscaled values undo it: the RPN's 3x3 weights are divided by
    s, and the final bn1 keeps its gamma and beta."""
    s = float(s)
    q = {k: np.array(v, copy=True) for k, v in params.items()}
    q["bn0_gamma"] = (q["bn0_gamma"] * s).astype(np.float32)
    q["bn0_beta"] = (q["bn0_beta"] * s).astype(np.float32)
    for name, _ in bn_shapes():
        if name in ("bn_data", "bn0"):
            continue
        q[name + "_moving_mean"] = (q[name + "_moving_mean"] * s).astype(np.float32)
        q[name + "_moving_var"] = ((q[name + "_moving_var"].astype(np.float64) + eps) * s * s - eps).astype(np.float32)
        if name != "bn1":
            q[name + "_gamma"] = (q[name + "_gamma"] * s).astype(np.float32)
            q[name + "_beta"] = (q[name + "_beta"] * s).astype(np.float32)
    q["rpn_conv_3x3_weight"] = (q["rpn_conv_3x3_weight"] / s).astype(np.float32)
    return q


def synthetic_image(height, width, seed=0, faces=3):
    """uint8 [H,W,3] frame: smooth low-frequency noise plus a few bright face-like blobs."""
    g = np.random.default_rng([int(seed), 0xFACE])
    gh, gw = height // 32 + 2, width // 32 + 2
    coarse = g.normal(0.0, 1.0, (gh, gw, 3))
    ys = np.linspace(0, gh - 1.001, height)
    xs = np.linspace(0, gw - 1.001, width)
    y0, x0 = ys.astype(np.int64), xs.astype(np.int64)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    c = coarse
    img = ((1 - fy) * (1 - fx) * c[y0][:, x0] + (1 - fy) * fx * c[y0][:, x0 + 1] +
           fy * (1 - fx) * c[y0 + 1][:, x0] + fy * fx * c[y0 + 1][:, x0 + 1])
    img = 118.0 + 45.0 * img + g.normal(0.0, 12.0, (height, width, 3))
    yy, xx = np.mgrid[0:height, 0:width]
    for _ in range(faces):
        cy, cx = g.uniform(0.15, 0.85) * height, g.uniform(0.15, 0.85) * width
        r = g.uniform(0.04, 0.12) * min(height, width)
        blob = np.exp(-(((yy - cy) / r) ** 2 + ((xx - cx) / (0.8 * r)) ** 2))
        img += 70.0 * blob[:, :, None] * np.array([1.0, 0.85, 0.75])
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def crowd_image(height, width, seed=0, faces=5000, size_lo=12, size_hi=40):
    """uint8 [H,W,3] frame densely covered with small bright blobs (dense-crowd stress input)."""
    g = np.random.default_rng([int(seed), 0xC0DE])
    img = 90.0 + g.normal(0.0, 10.0, (height, width, 3))
    for _ in range(faces):
        s = int(g.integers(size_lo, size_hi + 1))
        y = int(g.integers(0, max(1, height - s)))
        x = int(g.integers(0, max(1, width - s)))
        yy, xx = np.mgrid[0:s, 0:s]
        blob = np.exp(-(((yy - s / 2) / (0.35 * s)) ** 2 + ((xx - s / 2) / (0.3 * s)) ** 2))
        img[y:y + s, x:x + s] += 110.0 * blob[:, :, None] * g.uniform(0.7, 1.0, 3)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)
