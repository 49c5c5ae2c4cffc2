"""3dgs.cpp_amd -- MI355X-native splat rasterizer behind the 3DGS.cpp API.

Product code lives in csrc/ (HIP kernels + C ABI + C++ host mirror).  The Python here is only the
ctypes harness (binding) and the synthetic-scene / PLY helpers (synth).  The directory name is not a
valid Python identifier; load it with __graft_entry__.load_package() (module name `gs3d_amd`).
"""
from . import synth  # noqa: F401
from .binding import *  # noqa: F401,F403
from . import binding  # noqa: F401
from . import dist  # noqa: F401
