"""Drop-in for /root/reference/src/models/attention.py (imported by mvd_unet.py:10)."""
from mvd_amd.attention import ImageCrossAttentionProcessor, get_attention_processor_for_module  # noqa: F401
