"""RCCL communicator of the data-parallel train step, through the C ABI (include/dlsg.h: dlsg_comm_*, dlsg_allreduce_bucket*).

Replaces the gradient exchange of the reference's DistributedDataParallel wrapper over NCCL (run_gun.py:63-64,
train_debug.py:20).  One process per GPU; the 128-byte RCCL id travels from rank 0 to the other ranks over whatever
`torch.distributed` group the caller already has (any backend -- it is a 128-byte broadcast, not the data path); the
all-reduces themselves are RCCL kernels enqueued on a HIP stream the trainer chooses, so they can be captured into the
step's hipGraph.
"""
import ctypes as C

import torch

from .hip import load_library


def _agree_min(value, process_group=None, negate=False):
    """the minimum (negate: the maximum) of an int over the ranks of a torch.distributed group, on the device its backend moves"""
    import torch.distributed as dist
    dev = 'cuda' if dist.get_backend(process_group) == 'nccl' else 'cpu'
    t = torch.tensor([-value if negate else value], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=process_group)
    return -int(t.item()) if negate else int(t.item())


class RcclComm(object):
    def __init__(self, world, rank, process_group=None, lib=None):
        self.lib = lib or load_library()
        self.world, self.rank = int(world), int(rank)
        self._h = C.c_void_p()
        uid = (C.c_char * 128)()
        rc = 0
        if self.rank == 0:
            rc = int(self.lib.dlsg_comm_unique_id(uid))
        if self.world > 1:
            import torch.distributed as dist
            # rank 0 ships its STATUS with the id: a rank 0 whose librccl did not load must not leave its peers waiting in this
            # broadcast -- they receive the code and raise with it (the group's own timeout bounds a rank 0 that died earlier)
            box = [(rc, bytes(uid))]
            dist.broadcast_object_list(box, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                                       group=process_group)
            rc, raw = box[0]
            uid = (C.c_char * 128).from_buffer_copy(raw)
        if rc != 0:
            raise RuntimeError('dlsg_comm_unique_id failed on rank 0 with code %d (librccl not loadable?)' % rc)
        rc = int(self.lib.dlsg_comm_init(C.byref(self._h), uid, self.world, self.rank))
        if self.world > 1:
            # every rank learns whether EVERY rank has a communicator: one rank raising alone would leave the others in the first
            # all-reduce of the step
            worst = _agree_min(0 if rc == 0 else 1, process_group, negate=True)
            if worst != 0 and rc == 0:
                self.close()
                raise RuntimeError('dlsg_comm_init failed on another rank')
        if rc != 0:
            raise RuntimeError('dlsg_comm_init(world=%d, rank=%d) failed with code %d' % (self.world, self.rank, rc))
        v = C.c_int32()
        self.lib.dlsg_comm_info(self._h, None, None, C.byref(v))
        self.rccl_version_code = int(v.value)

    @property
    def rccl_version(self):
        v = self.rccl_version_code
        return '%d.%d.%d' % (v // 10000, (v // 100) % 100, v % 100) if v >= 10000 else str(v)

    def allreduce(self, views, stream):
        """in-place sum over all ranks of 1-d fp32 device views, enqueued on `stream` (a torch.cuda.Stream); several views =
        one RCCL group (one fused launch)"""
        views = [v for v in views if v.numel() > 0]
        if not views:
            return
        for v in views:
            assert v.is_cuda and v.dtype == torch.float32 and v.is_contiguous()
        st = C.c_void_p(stream.cuda_stream)
        if len(views) == 1:
            rc = self.lib.dlsg_allreduce_bucket(self._h, C.c_void_p(views[0].data_ptr()), views[0].numel(), st)
        else:
            ptrs = (C.c_void_p * len(views))(*[v.data_ptr() for v in views])
            cnts = (C.c_int64 * len(views))(*[v.numel() for v in views])
            rc = self.lib.dlsg_allreduce_buckets(self._h, ptrs, cnts, len(views), st)
        if rc != 0:
            raise RuntimeError('dlsg_allreduce_bucket failed with code %d' % rc)

    def allreduce_max_word(self, word, stream):
        """in-place max over all ranks of an int32 device tensor, enqueued on `stream`"""
        assert word.is_cuda and word.dtype == torch.int32 and word.is_contiguous()
        rc = self.lib.dlsg_allreduce_max_i32(self._h, C.c_void_p(word.data_ptr()), word.numel(), C.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError('dlsg_allreduce_max_i32 failed with code %d' % rc)

    def async_error(self):
        """ncclCommGetAsyncError of the communicator (0 = none); None when it cannot be read"""
        if not self._h:
            return None
        code = C.c_int32(0)
        rc = self.lib.dlsg_comm_async_error(self._h, C.byref(code))
        return int(code.value) if rc == 0 else None

    def close(self):
        if self._h:
            h, self._h = self._h, C.c_void_p()
            self.lib.dlsg_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
