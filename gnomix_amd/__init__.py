"""gnomix_amd — MI355X-native Gnomix inference hot path (base classifiers -> smoother -> labels).

Host side mirrors the reference's plugin interface (src/Base, src/Smooth, src/model.py); all compute
goes through the C ABI of libgnomix_hip.so (include/gnomix_hip.h).  No CPU fallback exists."""
from ._lib import GnxError, GnxLibraryError, Context, default_context, load as load_library  # noqa: F401
from .model import GnxModelData, DeviceModel  # noqa: F401
from .base import HipBase  # noqa: F401
from .smooth import HipSmoother  # noqa: F401
from .gnomix import HipGnomix  # noqa: F401

__all__ = ["GnxError", "GnxLibraryError", "Context", "default_context", "load_library", "GnxModelData",
           "DeviceModel", "HipBase", "HipSmoother", "HipGnomix"]
