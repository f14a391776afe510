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


class RcclComm(object):
    def __init__(self, world, rank, process_group=None, lib=None):
        self.lib = lib or load_library()
        self.world, self.rank = int(world), int(rank)
        self._h = C.c_void_p()
        uid = (C.c_char * 128)()
        if self.rank == 0:
            rc = self.lib.dlsg_comm_unique_id(uid)
            if rc != 0:
                raise RuntimeError('dlsg_comm_unique_id failed with code %d (librccl not loadable?)' % rc)
        if self.world > 1:
            import torch.distributed as dist
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                                       group=process_group)
            uid = (C.c_char * 128).from_buffer_copy(box[0])
        rc = self.lib.dlsg_comm_init(C.byref(self._h), uid, self.world, self.rank)
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

    def close(self):
        if self._h:
            h, self._h = self._h, C.c_void_p()
            self.lib.dlsg_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
