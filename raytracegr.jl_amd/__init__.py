"""raytracegr.jl_amd — MI355X-native replacement for the hot path of eschnett/RayTraceGR.jl.

The directory name carries a dot (fixed by the project layout), so it is not importable by plain `import`;
`__graft_entry__.load_package()` (and tests/conftest.py) register it as module `raytracegr_jl_amd`.

Layout: csrc/ (HIP kernels + C ABI -> librtgr_hip.so), _abi.py (ctypes view of include/rtgr.h),
api.py (host mirror of the reference interface), sharded.py (row sharding over GPUs + RCCL gather),
build.py (hipcc driver), png.py (image output).
"""
from .api import *  # noqa: F401,F403
from . import _abi, api  # noqa: F401
