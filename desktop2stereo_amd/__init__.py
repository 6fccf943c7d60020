"""MI355X-native hot path of desktop2stereo (depth inference + stereo warp) behind the reference's call surface."""
import os as _os
import sys as _sys

# Kernel arguments in device memory: the batch-1 frame is a chain of ~85 dependent launches and the host-memory setting costs 7 % of it
# (DESIGN.md 3.7).  The HIP runtime reads the variable when it is loaded, so the default can only be set before torch is imported.
if "HIP_FORCE_DEV_KERNARG" not in _os.environ and "torch" not in _sys.modules:
    _os.environ["HIP_FORCE_DEV_KERNARG"] = "1"


def dev_kernarg() -> str:
    """HIP_FORCE_DEV_KERNARG as this process sees it ("unset" = the runtime's own default)."""
    return _os.environ.get("HIP_FORCE_DEV_KERNARG", "unset")
