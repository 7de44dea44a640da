"""Drop-in for /root/reference/src/training/scheduler.py (imported by mvd_unet.py:9)."""
from mvd_amd.scheduler import DDPMScheduler, ShiftSNRScheduler, SNR_to_betas, compute_snr  # noqa: F401
