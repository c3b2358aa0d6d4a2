from .base_icv_encoder import BaseICVEncoder, ICVEncoderOutput  # noqa: F401
from .global_icv_encoder import GlobalICVEncoder  # noqa: F401
