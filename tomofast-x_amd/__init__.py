"""tomofast-x_amd: MI355X-native (gfx950) implementation of Tomofast-x's sensitivity-kernel hot path.

The directory name follows the project convention and is not a Python identifier; import it with
    import importlib; tfx = importlib.import_module("tomofast-x_amd")
Everything numerical runs in hand-written HIP kernels (csrc/ -> libtfx.so) behind the C ABI of include/tfx.h."""
from .lib import TfxError, SO_PATH, SYMBOLS, load          # noqa: F401
from .sensitivity import Context, get_load_balancing_nelements   # noqa: F401
from . import synthetic, inversion, distributed, sensit_io  # noqa: F401
