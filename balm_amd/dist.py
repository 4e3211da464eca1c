"""Feature sharding across one-process-per-GPU ranks.

Hess, JacT and the residual are plain sums over features (the reference already sums thread-private
copies, bavoxel.hpp:1049-1056), so each rank owns a contiguous feature shard and the only exchange
is one all-reduce (sum, f64) of the packed payload [upper tiles of Gt Gt^T | per-pose block-diagonal
and gradient sums | residual] per Hessian evaluation, plus one scalar per residual-only evaluation.
The damped solve is replicated (identical inputs -> identical steps), so no broadcast is needed.
torch.distributed is plumbing here: backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU tests.
"""
import ctypes as C
import os

import numpy as np


def partition_features(n_obs, world_size):
    """Contiguous split of features balanced by the per-feature SYRK cost n_a (n_a + 1) / 2.

    n_obs: [F] number of observing poses per feature.  Returns [(start, end)] * world_size with
    every rank non-empty when F >= world_size."""
    n_obs = np.asarray(n_obs, dtype=np.float64)
    F = n_obs.shape[0]
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    if F < world_size:
        raise ValueError("fewer features (%d) than ranks (%d)" % (F, world_size))
    cost = n_obs * (n_obs + 1.0) * 0.5 + 1.0
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    bounds = [0]
    for r in range(1, world_size):
        target = cum[-1] * r / world_size
        k = int(np.searchsorted(cum, target))
        k = max(k, bounds[-1] + 1)            # non-empty
        k = min(k, F - (world_size - r))      # leave one feature for each remaining rank
        bounds.append(k)
    bounds.append(F)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None):
    """One process per GPU; rendezvous from MASTER_ADDR/MASTER_PORT (use 127.0.0.1 on one node)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class _DevMem:
    """Zero-copy view of a device buffer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class DeviceAllReduce:
    """all-reduce(sum) of a raw device buffer of f64 through torch.distributed (RCCL).

    Preferred path wraps the library's buffer zero-copy; if this torch build cannot import the CUDA
    array interface, falls back to staging through a torch tensor with two device-to-device copies."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.zero_copy = True
        self._stage = None
        self._hip = None
        self._views = {}          # (ptr, n) -> tensor view: the library reuses its payload buffers

    def __call__(self, ptr, n):
        torch, dist = self.torch, self.dist
        if self.zero_copy:
            t = self._views.get((ptr, n))
            if t is None:
                try:      # only the VIEW may fall back: a failed collective must surface, not trigger a second one
                    t = torch.as_tensor(_DevMem(ptr, n), device="cuda")
                    if t.data_ptr() != ptr:
                        raise RuntimeError("as_tensor copied")
                except Exception:
                    self.zero_copy = False
                    t = None
                if t is not None:
                    if len(self._views) > 64:
                        self._views.clear()
                    self._views[(ptr, n)] = t
            if t is not None:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                torch.cuda.synchronize()
                return
        if self._hip is None:
            self._hip = C.CDLL("libamdhip64.so")
            self._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        if self._stage is None or self._stage.numel() < n:
            self._stage = torch.empty(n, dtype=torch.float64, device="cuda")
        st = self._stage[:n]
        torch.cuda.synchronize()
        if self._hip.hipMemcpy(st.data_ptr(), ptr, n * 8, 3) != 0:      # hipMemcpyDeviceToDevice
            raise RuntimeError("hipMemcpy D2D (in) failed")
        dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group)
        torch.cuda.synchronize()
        if self._hip.hipMemcpy(ptr, st.data_ptr(), n * 8, 3) != 0:
            raise RuntimeError("hipMemcpy D2D (out) failed")


def install_allreduce(ctx, group=None):
    """Make `ctx` (balm_amd.capi.Context holding this rank's feature shard) sum its evaluations
    across all ranks of `group`."""
    hook = DeviceAllReduce(group)
    ctx.set_allreduce(hook)
    return hook


def install_rccl(ctx, group=None):
    """The library's own RCCL transport for this one-process-per-GPU job: rank 0 draws the communicator id, the
    process group carries its 128 bytes to every rank, and every rank joins with its context.  After this the
    all-reduces are issued inside libbalm_hip.so on its own stream -- no hook, no host synchronisation."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    ctx.comm_init_rank(world, rank, box[0])
    torch.cuda.synchronize()
    return "rccl-in-library"


def allreduce_host_payload(H, g, r, group=None):
    """CPU/gloo analogue of the device payload exchange: sums (H, g, r) across ranks.  Used by the
    world_size-2 gloo tests that cover the sharding logic without a GPU."""
    import torch
    import torch.distributed as dist
    n = g.shape[0]
    buf = torch.from_numpy(np.concatenate([np.asarray(H).reshape(-1), np.asarray(g), [r]]))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    out = buf.numpy()
    return out[:n * n].reshape(n, n).copy(), out[n * n:n * n + n].copy(), float(out[-1])
