"""Drop-in for /root/reference/src/models/mvd_unet.py (imports at infer.py:1, val.py:24, train.py:17)."""
from mvd_amd.mvd_unet import MultiViewUNet, UNetOutput, create_mvd_pipeline  # noqa: F401
