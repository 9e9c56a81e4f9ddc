import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from facedetection_amd import profiling
from facedetection_amd.detection import Detector
from facedetection_amd.synthetic import synthetic_image
det = Detector("synthetic:0", 0, precision="f16x3")
f = det._frames(synthetic_image(1080, 1920, seed=1))
for _ in range(3): det.detect_batch_device(f)
torch.cuda.synchronize()
profiling.begin()
for _ in range(20): det.detect_batch_device(f)
torch.cuda.synchronize()
rows = profiling.end()
tot = 0
for r in sorted(rows, key=lambda r: -r["ms"]):
    print("%-26s %4d launches/frame %8.1f us/frame" % (r["kernel"], r["launches"] // 20, r["ms"] / 20 * 1e3)); tot += r["ms"] / 20 * 1e3
print("sum", tot)
