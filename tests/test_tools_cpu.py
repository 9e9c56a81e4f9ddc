"""The analysis tools behind profiles/: the launch list the per-layer report pairs with a kernel trace
(reference graph: code/detection/symbol/faster_rcnn.py:6-53, 73-135 -- stem, 3 + 4 + 6 bottleneck units, RPN head,
3 ROI-head units) and the two-floor table computed from a committed per-layer table.  No GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_launch_list_of_the_float32_pass():
    import conv_layer_report as rep
    every = rep.layer_list(16, mode="layers")
    # 1 stem + 16 units x 3 convolutions + 4 projection shortcuts + 2 RPN layers
    assert len(every) == 1 + 16 * 3 + 4 + 2
    direct = sum(l[4] for l in every)
    # 354.5 GFLOP per 1080p frame when every layer is a direct convolution (10 rois into the head)
    assert direct / 16 / 1e9 == pytest.approx(354.5, rel=2e-3)
    sep, c1, unit = (rep.layer_list(16, mode=m) for m in ("f32sep", "f32c1", "f32"))
    # shortcut inside the tail: one launch fewer; conv1 as well, and the shortcuts of the two later projection units
    # inside their conv3: five fewer
    assert len(sep) == len(c1) + 1 == len(unit) + 4
    names = [l[0] for l in unit]
    assert "s1u1_unit" in names and "s1u1_sc" not in names and "s1u1_c1" not in names
    assert "s2u1_sc" not in names and "s3u1_sc" not in names and "s4u1_sc" in names
    # the Winograd forms execute fewer FLOPs than the direct ones; the unit kernel a few more than its three layers
    assert sum(l[4] for l in sep) < 0.7 * direct
    assert sum(l[4] for l in unit) > sum(l[4] for l in sep)
    assert sum(l[4] for l in unit) < 1.01 * sum(l[4] for l in sep)


def test_two_floor_table_runs_on_a_committed_layer_table():
    tables = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_layers_f32.txt") and f.startswith("r03_"))
    assert tables
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ridge_table.py"), os.path.join(ROOT, "profiles", tables[-1]), "16"],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0].split()[0] == "launch" and lines[-1].startswith("sum of matrix floors")
    # no launch beats its floors
    for line in lines[1:-1]:
        assert 0.0 < float(line.split()[-1]) <= 1.0, line
