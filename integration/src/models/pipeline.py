"""Drop-in for /root/reference/src/models/pipeline.py (MVDPipeline, used by mvd_unet.py:411 and training.py:329)."""
from mvd_amd.pipeline import MVDPipeline  # noqa: F401
