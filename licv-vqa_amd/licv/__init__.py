"""licv — MI355X-native engine of the L-ICV hot path (host side: Python over a C-ABI HIP library).

The compute lives in ``liblicv_hip.so`` (hand-written HIP for gfx950, sources in ``../csrc``); this
package holds the ctypes binding, thin tensor-level op wrappers, the native Idefics forward engine with
fused ICV injection, and the data-parallel trainer.  There is NO CPU fallback: using an op without the
library raises.
"""
from .config import IdeficsArch, idefics_arch  # noqa: F401

__version__ = "0.1.0"
