"""Drop-in for /root/reference/src/utils.py (imported by infer.py:1, val.py)."""
from mvd_amd.utils import create_camera_matrix, create_output_dirs, load_image, log_debug  # noqa: F401
