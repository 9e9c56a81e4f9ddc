"""MI355X-native face-detection hot path (Faster R-CNN, ResNet-50) behind the
call signature of tanmlh/FaceDetection's ``code/detection`` package.

The arithmetic lives in hand-written HIP kernels (``csrc/``) behind a C ABI
(``include/facedet.h``); this package is the Python host side: it mirrors the
reference's ``Detector`` / ``processing`` / ``proposal`` interfaces and uses
PyTorch-ROCm tensors purely as device buffers.  There is no CPU fallback: if
``libfacedet.so`` is missing, importing a compute entry point raises.
"""
__version__ = "0.1.0"
