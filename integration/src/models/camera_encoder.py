"""Drop-in for /root/reference/src/models/camera_encoder.py (imported by mvd_unet.py:11)."""
from mvd_amd.camera_encoder import CameraEncoder  # noqa: F401
