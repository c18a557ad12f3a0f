"""MI355X-native VLM inference path behind the Vision-Inspection-System agent API.

Scope: the Inspector/Auditor image -> defect-report step.
Host code is Python (the reference's language) on PyTorch-ROCm tensors; all
arithmetic runs in hand-written gfx950 HIP kernels reached through the C ABI in
``include/vis_hip.h`` (``csrc/libvis_hip.so``).  There is no CPU fallback.
"""
__version__ = "0.1.0"
