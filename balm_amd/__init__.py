"""balm_amd: MI355X-native (HIP/gfx950) second-order bundle-adjustment hot path of BALM 2.0.

Only what the path needs: ``csrc/`` (HIP kernels + the C ABI of include/balm_hip.h; the C++ mirrors of the reference's
VOX_HESS / BALM2 interface are the headers include/balm_shim.hpp and include/balm_shim_virtual.hpp), ``capi`` (ctypes
binding of that ABI), ``scene`` (synthetic-scene generator = the reference's benchmark_virtual driver), ``dist``
(feature sharding across one-process-per-GPU ranks), ``virtual`` / ``realworld`` / ``consistency`` (the three drivers of
the reference, ROS-free, on the GPU path), ``sliding`` (a sliding-window BA on the device-resident voxel map: the
incremental use of the reference's octree that none of its shipped drivers runs).
"""
__version__ = "0.5.0"      # = balm_version() of the library, ABI revision 5
