"""balm_amd: MI355X-native (HIP/gfx950) second-order bundle-adjustment hot path of BALM 2.0.

Only what the path needs: ``csrc/`` (HIP kernels + the C ABI of include/balm_hip.h), ``capi``
(ctypes binding of that ABI), ``voxhess`` (host-side mirror of the reference's VOX_HESS / BALM2
interface), ``scene`` (synthetic-scene generator = the reference's benchmark_virtual driver),
``dist`` (feature sharding across one-process-per-GPU ranks).
"""
__version__ = "0.1.0"
