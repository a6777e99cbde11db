"""MI355X-native TTNet inference path (drop-in for the eval forward of
Anonymousijcai2024ttnet/scale_imagenet's TTNet ImageNet models).

Importing this package loads nothing heavy; ``scale_imagenet_amd.ttnet`` holds the
nn.Module mirror and ``scale_imagenet_amd._lib`` the ctypes binding of libttnet.so.
"""
__all__ = ["spec", "synth", "ttnet", "dist", "evaluate", "build"]
