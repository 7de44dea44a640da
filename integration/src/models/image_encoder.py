"""Drop-in for /root/reference/src/models/image_encoder.py (imported by mvd_unet.py:12)."""
from mvd_amd.mvd_unet import ImageEncoder  # noqa: F401
