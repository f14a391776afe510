"""feature I/O and the train / eval loaders (reference utils/data.py:13-147, utils/opt.py:118-134).

File formats the reference reads, kept as they are:
  * frame features   HDF5 dataset `feats`  (N, 26, A+M) float32         (utils/opt.py:125-126)
  * region features  HDF5 datasets `vfeats` (N, 26, 36, 2048) float32, `sfeats` (N, 26, 36, 5) (unused by the model)
  * captions         pickle of the tuple (captions, pos_tags, lengths, video_ids)   (utils/data.py:19)
HDF5 is read through the HDF5 C library itself (libhdf5, ctypes) -- the library h5py wraps; h5py is not in this image.

MI355X-first loader: at 4-5 k clips/s a train step consumes 4.1 MB/clip = 17-21 GB/s of features, a third of what PCIe
Gen5 x16 delivers at best, and MSVD's whole feature set (1970 clips, regions cut to num_obj = 16: 8.1 GB) is 3 % of one
GPU's 288 GB.  So the default is `ResidentFeatures`: both arrays are uploaded ONCE (chunked through a pinned staging buffer)
and a batch is a device-side row gather (the `gather_rows` HIP kernel) -- no host work and no H2D traffic per step.
`StreamedFeatures` covers sets that should not live in HBM: a reader thread fills pinned buffers two batches ahead and
the H2D copies run on their own HIP stream.
Sampler and collate semantics are the reference's: DistributedSampler partition (utils/data.py:122-124), reshuffled by
`set_epoch`; a batch is ordered by its LAST tuple element, the video id (utils/data.py:90,104 -- descending for training,
ascending for evaluation; not by caption length), regions are cut to `num_obj` (run_gun.py:158).
"""
import ctypes as C
import ctypes.util
import glob
import math
import os
import pickle
import queue
import threading

import numpy as np
import torch

# ================================================================================================ HDF5 through libhdf5
_H5 = None


def _libhdf5():
    """dlopen libhdf5 once.  Search order: $DLSG_LIBHDF5, the loader path, conda's lib directory."""
    global _H5
    if _H5 is not None:
        return _H5
    cands = [os.environ.get('DLSG_LIBHDF5'), ctypes.util.find_library('hdf5')]
    cands += sorted(glob.glob('/opt/conda/lib/libhdf5.so*')) + sorted(glob.glob('/usr/lib/x86_64-linux-gnu/libhdf5*.so*'))
    err = None
    for c in cands:
        if not c:
            continue
        try:
            lib = C.CDLL(c)
            lib.H5open()
            break
        except OSError as e:
            err = e
    else:
        raise RuntimeError('libhdf5 not found (set DLSG_LIBHDF5 to its path): %s' % err)
    hid, hsz, vp = C.c_int64, C.POINTER(C.c_uint64), C.c_void_p
    sig = {'H5Fopen': (hid, [C.c_char_p, C.c_uint, hid]), 'H5Fcreate': (hid, [C.c_char_p, C.c_uint, hid, hid]),
           'H5Fclose': (C.c_int, [hid]), 'H5Dopen2': (hid, [hid, C.c_char_p, hid]), 'H5Dclose': (C.c_int, [hid]),
           'H5Dget_space': (hid, [hid]), 'H5Dget_type': (hid, [hid]), 'H5Sclose': (C.c_int, [hid]), 'H5Tclose': (C.c_int, [hid]),
           'H5Sget_simple_extent_ndims': (C.c_int, [hid]), 'H5Sget_simple_extent_dims': (C.c_int, [hid, hsz, hsz]),
           'H5Tget_class': (C.c_int, [hid]), 'H5Tget_size': (C.c_size_t, [hid]),
           'H5Sselect_hyperslab': (C.c_int, [hid, C.c_int, hsz, hsz, hsz, hsz]),
           'H5Screate_simple': (hid, [C.c_int, hsz, hsz]), 'H5Dread': (C.c_int, [hid, hid, hid, hid, hid, vp]),
           'H5Dwrite': (C.c_int, [hid, hid, hid, hid, hid, vp]),
           'H5Dcreate2': (hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), 'H5Lexists': (C.c_int, [hid, C.c_char_p, hid]),
           'H5Dget_offset': (C.c_uint64, [hid]), 'H5Dget_create_plist': (hid, [hid]), 'H5Pget_layout': (C.c_int, [hid]),
           'H5Pclose': (C.c_int, [hid]), 'H5Tget_order': (C.c_int, [hid])}
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.native = {np.dtype(np.float32): hid.in_dll(lib, 'H5T_NATIVE_FLOAT_g').value,
                  np.dtype(np.float64): hid.in_dll(lib, 'H5T_NATIVE_DOUBLE_g').value,
                  np.dtype(np.int64): hid.in_dll(lib, 'H5T_NATIVE_INT64_g').value,
                  np.dtype(np.int32): hid.in_dll(lib, 'H5T_NATIVE_INT32_g').value}
    _H5 = lib
    return lib


def _dims(a):
    return (C.c_uint64 * len(a))(*[int(x) for x in a])


class H5Dataset(object):
    """One HDF5 dataset, read like the reference reads it (`h5[name][video_id]`, utils/data.py:62-63): rows along the first
    axis, converted to `dtype` (float32) by the library."""

    def __init__(self, lib, did, name):
        self.lib, self.did, self.name = lib, did, name
        sp = lib.H5Dget_space(did)
        nd = lib.H5Sget_simple_extent_ndims(sp)
        d = (C.c_uint64 * nd)()
        lib.H5Sget_simple_extent_dims(sp, d, None)
        lib.H5Sclose(sp)
        self.shape = tuple(int(x) for x in d)
        tp = lib.H5Dget_type(did)
        self.is_float = lib.H5Tget_class(tp) == 1
        self.itemsize = int(lib.H5Tget_size(tp))
        self.little_endian = lib.H5Tget_order(tp) == 0
        lib.H5Tclose(tp)
        pl = lib.H5Dget_create_plist(did)
        self.contiguous = lib.H5Pget_layout(pl) == 1            # H5D_CONTIGUOUS (what create_dataset(data=...) writes)
        lib.H5Pclose(pl)
        self.path = None

    def memmap(self):
        """Zero-copy view of a contiguous little-endian float32 dataset: the file region libhdf5 reports (H5Dget_offset)
        mapped read-only.  None when the layout is chunked / compact or the type needs conversion (then read_rows is used)."""
        if not (self.contiguous and self.is_float and self.itemsize == 4 and self.little_endian and self.path):
            return None
        off = int(self.lib.H5Dget_offset(self.did))
        if off == 0 or off == 0xFFFFFFFFFFFFFFFF:              # HADDR_UNDEF: no storage allocated
            return None
        return np.memmap(self.path, dtype=np.float32, mode='r', offset=off, shape=self.shape)

    def __len__(self):
        return self.shape[0]

    def read_rows(self, start, count, out=None, dtype=np.float32):
        """rows [start, start+count) -> ndarray (count, *shape[1:]); `out` may be a (pinned) buffer to fill"""
        lib = self.lib
        if start < 0 or start + count > self.shape[0]:
            raise IndexError('rows %d..%d of %s with %d rows' % (start, start + count, self.name, self.shape[0]))
        shp = (count,) + self.shape[1:]
        if out is None:
            out = np.empty(shp, dtype=dtype)
        assert out.shape == shp and out.flags['C_CONTIGUOUS'] and out.dtype in lib.native
        if count == 0:
            return out
        fs = lib.H5Dget_space(self.did)
        lib.H5Sselect_hyperslab(fs, 0, _dims((start,) + (0,) * (len(shp) - 1)), None, _dims(shp), None)
        ms = lib.H5Screate_simple(len(shp), _dims(shp), None)
        rc = lib.H5Dread(self.did, lib.native[out.dtype], ms, fs, 0, out.ctypes.data_as(C.c_void_p))
        lib.H5Sclose(ms); lib.H5Sclose(fs)
        if rc < 0:
            raise IOError('H5Dread failed on %s' % self.name)
        return out

    def __getitem__(self, i):
        if isinstance(i, slice):
            s, e, st = i.indices(self.shape[0])
            assert st == 1
            return self.read_rows(s, max(0, e - s))
        i = int(i)
        if i < 0:
            i += self.shape[0]
        return self.read_rows(i, 1)[0]


class H5File(object):
    """Minimal h5py.File stand-in over libhdf5: `H5File(path)['feats']`, `H5File.create(path).write(name, array)`."""

    def __init__(self, path, mode='r'):
        self.lib = _libhdf5()
        if mode == 'r':
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            self.fid = self.lib.H5Fopen(path.encode(), 0, 0)           # H5F_ACC_RDONLY, H5P_DEFAULT
        else:
            self.fid = self.lib.H5Fcreate(path.encode(), 2, 0, 0)      # H5F_ACC_TRUNC
        if self.fid < 0:
            raise IOError('cannot open %s as HDF5' % path)
        self.path, self._open = path, {}

    @classmethod
    def create(cls, path):
        return cls(path, 'w')

    def __contains__(self, name):
        return self.lib.H5Lexists(self.fid, name.encode(), 0) > 0

    def __getitem__(self, name):
        if name not in self._open:
            if name not in self:
                raise KeyError('%s has no dataset %r' % (self.path, name))
            did = self.lib.H5Dopen2(self.fid, name.encode(), 0)
            if did < 0:
                raise KeyError(name)
            self._open[name] = H5Dataset(self.lib, did, name)
            self._open[name].path = self.path
        return self._open[name]

    def write(self, name, arr):
        arr = np.ascontiguousarray(arr)
        lib = self.lib
        sp = lib.H5Screate_simple(arr.ndim, _dims(arr.shape), None)
        did = lib.H5Dcreate2(self.fid, name.encode(), lib.native[arr.dtype], sp, 0, 0, 0)
        rc = lib.H5Dwrite(did, lib.native[arr.dtype], 0, 0, 0, arr.ctypes.data_as(C.c_void_p))
        lib.H5Dclose(did); lib.H5Sclose(sp)
        if did < 0 or rc < 0:
            raise IOError('cannot write %s to %s' % (name, self.path))
        return self

    def close(self):
        for d in self._open.values():
            self.lib.H5Dclose(d.did)
        self._open = {}
        if self.fid >= 0:
            self.lib.H5Fclose(self.fid)
            self.fid = -1

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ================================================================================================ captions
def _as_long(x):
    return x.to(torch.int64) if torch.is_tensor(x) else torch.as_tensor(np.asarray(x), dtype=torch.int64)


class CaptionSet(object):
    """The caption pickle of utils/data.py:19: (captions, pos_tags, lengths, video_ids), one entry per (video, sentence)."""

    def __init__(self, cap_pkl):
        with open(cap_pkl, 'rb') as f:
            captions, pos_tags, lengths, video_ids = pickle.load(f)
        self.captions = torch.stack([_as_long(c) for c in captions], 0)
        self.pos_tags = torch.stack([_as_long(p) for p in pos_tags], 0)
        self.lengths = [int(x) for x in lengths]
        self.video_ids = [int(v) for v in video_ids]
        assert len(self.lengths) == len(self.video_ids) == self.captions.shape[0]

    def __len__(self):
        return len(self.video_ids)


def _h2d(values, dtype, device):
    """small host arrays (gather indices, caption rows) to the device without stalling the host (hip.host_to_device)"""
    from .hip import host_to_device
    return host_to_device(values, dtype, device)


# ================================================================================================ feature stores
class _Features(object):
    """(frames `feats`, regions `vfeats`) of one dataset; regions are cut to the first `num_obj` objects (run_gun.py:158)."""

    def __init__(self, frame_h5, region_h5, num_obj, feats='feats', vfeats='vfeats'):
        self.h5f, self.h5r = H5File(frame_h5), H5File(region_h5)
        self.frames, self.regions = self.h5f[feats], self.h5r[vfeats]
        if len(self.regions.shape) != 4 or len(self.frames.shape) != 3:
            raise ValueError('expected feats (N,T,F) and vfeats (N,T,O,R), got %s / %s' % (self.frames.shape, self.regions.shape))
        self.n = min(self.frames.shape[0], self.regions.shape[0])
        self.num_obj = min(num_obj, self.regions.shape[2])
        self.frame_shape = self.frames.shape[1:]
        self.region_shape = (self.regions.shape[1], self.num_obj, self.regions.shape[3])

    def host_rows(self, start, count, fout=None, rout=None):
        f = self.frames.read_rows(start, count, out=fout)
        r = self.regions.read_rows(start, count)
        if rout is None:
            return f, np.ascontiguousarray(r[:, :, :self.num_obj])
        rout[...] = r[:, :, :self.num_obj]
        return f, rout

    def host_batch(self, video_ids):
        """reference path (utils/data.py:62-63, 92-95): one HDF5 row per video, stacked"""
        f = np.stack([self.frames[v] for v in video_ids], 0)
        r = np.stack([self.regions[v][:, :self.num_obj] for v in video_ids], 0)
        return f, r


class ResidentFeatures(_Features):
    """Whole feature set in HBM; `batch(video_ids)` is a device gather (HIP `gather_rows`), no per-step host work."""

    def __init__(self, frame_h5, region_h5, num_obj, device, ops=None, chunk=64, **kw):
        super().__init__(frame_h5, region_h5, num_obj, **kw)
        self.device = torch.device(device)
        self.ops = ops
        fw, rw = int(np.prod(self.frame_shape)), int(np.prod(self.region_shape))
        self.dev_frames = torch.empty(self.n, fw, dtype=torch.float32, device=self.device)
        self.dev_regions = torch.empty(self.n, rw, dtype=torch.float32, device=self.device)
        pin = self.device.type == 'cuda'
        sf = torch.empty(chunk, fw, dtype=torch.float32, pin_memory=pin)
        sr = torch.empty(chunk, rw, dtype=torch.float32, pin_memory=pin)
        for s in range(0, self.n, chunk):
            c = min(chunk, self.n - s)
            self.host_rows(s, c, fout=sf[:c].view(c, *self.frame_shape).numpy(), rout=sr[:c].view(c, *self.region_shape).numpy())
            self.dev_frames[s:s + c].copy_(sf[:c], non_blocking=False)
            self.dev_regions[s:s + c].copy_(sr[:c], non_blocking=False)
        self.bytes = (self.dev_frames.numel() + self.dev_regions.numel()) * 4

    @classmethod
    def from_arrays(cls, frames, regions, num_obj, device, ops=None):
        """A resident store over arrays that are already in memory (synthetic feature sets, tests, bench.py's `sustained` leg):
        frames (N, T, F), regions (N, T, O, R) as torch tensors or numpy arrays; regions are cut to the first `num_obj` objects as
        `_Features` does for a file.  No HDF5 involved; `batch()` is the same device gather."""
        self = cls.__new__(cls)
        frames, regions = torch.as_tensor(frames), torch.as_tensor(regions)
        if frames.dim() != 3 or regions.dim() != 4:
            raise ValueError('expected feats (N,T,F) and vfeats (N,T,O,R), got %s / %s' % (tuple(frames.shape), tuple(regions.shape)))
        self.h5f = self.h5r = self.frames = self.regions = None
        self.n = min(frames.shape[0], regions.shape[0])
        self.num_obj = min(num_obj, regions.shape[2])
        self.frame_shape = tuple(frames.shape[1:])
        self.region_shape = (regions.shape[1], self.num_obj, regions.shape[3])
        self.device, self.ops = torch.device(device), ops
        self.dev_frames = frames[:self.n].reshape(self.n, -1).to(device=self.device, dtype=torch.float32).contiguous()
        self.dev_regions = regions[:self.n, :, :self.num_obj].reshape(self.n, -1).to(device=self.device, dtype=torch.float32).contiguous()
        self.bytes = (self.dev_frames.numel() + self.dev_regions.numel()) * 4
        return self

    def batch(self, video_ids, out=None):
        """out: optional (frames, regions) device tensors of the batch shape to gather into (e.g. `Trainer.static_inputs()`)"""
        idx = _h2d(list(video_ids), torch.int64, self.device)
        B = idx.numel()
        if out is not None:
            f, r = out[0].view(B, -1), out[1].view(B, -1)
            assert f.is_contiguous() and r.is_contiguous() and f.shape[1] == self.dev_frames.shape[1] \
                and r.shape[1] == self.dev_regions.shape[1]
        else:
            f = torch.empty(B, self.dev_frames.shape[1], dtype=torch.float32, device=self.device)
            r = torch.empty(B, self.dev_regions.shape[1], dtype=torch.float32, device=self.device)
        if self.ops is not None:
            self.ops.gather_rows(self.dev_frames, idx, f)
            self.ops.gather_rows(self.dev_regions, idx, r)
        else:                                   # CPU build of the host logic (tests without a GPU)
            f.copy_(self.dev_frames[idx]); r.copy_(self.dev_regions[idx])
        return f.view(B, *self.frame_shape), r.view(B, *self.region_shape)


class StreamedFeatures(_Features):
    """Features stay on the host; `prefetch(list of id lists)` yields device batches `depth` batches ahead of the consumer.
    Batches are assembled straight into a ring of PINNED buffers by `workers` threads (numpy's copies release the GIL), from
    memory-mapped views of the HDF5 datasets when they are stored contiguously (no libhdf5 call, no intermediate array,
    only the first num_obj objects of a region row are touched), else through hyperslab reads by one thread (libhdf5 is
    not thread-safe); the H2D copies run on a side HIP stream."""

    def __init__(self, frame_h5, region_h5, num_obj, device, depth=2, workers=4, **kw):
        super().__init__(frame_h5, region_h5, num_obj, **kw)
        self.device, self.depth, self.workers = torch.device(device), depth, max(1, workers)
        self.mm_frames, self.mm_regions = self.frames.memmap(), self.regions.memmap()
        self.mapped = self.mm_frames is not None and self.mm_regions is not None

    def _fill(self, ids, tf, tr):
        """tf (B, T, F), tr (B, T, O, R) host tensors <- rows `ids`"""
        nf, nr = tf.numpy(), tr.numpy()
        if not self.mapped:
            for j, v in enumerate(ids):
                self.frames.read_rows(v, 1, out=nf[j:j + 1])
                nr[j] = self.regions.read_rows(v, 1)[0][:, :self.num_obj]
            return
        O = self.num_obj

        def work(lo, hi):
            for j in range(lo, hi):
                nf[j] = self.mm_frames[ids[j]]
                nr[j] = self.mm_regions[ids[j], :, :O]
        n = len(ids)
        k = min(self.workers, n)
        if k <= 1:
            work(0, n)
            return
        ts = [threading.Thread(target=work, args=(n * i // k, n * (i + 1) // k)) for i in range(k)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    def prefetch(self, id_batches):
        id_batches = list(id_batches)
        if not id_batches:
            return
        cuda = self.device.type == 'cuda'
        bmax = max(len(b) for b in id_batches)
        nbuf = self.depth + 2                               # being filled + queued (depth) + being copied
        ring = [(torch.empty(bmax, *self.frame_shape, dtype=torch.float32, pin_memory=cuda),
                 torch.empty(bmax, *self.region_shape, dtype=torch.float32, pin_memory=cuda)) for _ in range(nbuf)]
        free = queue.Queue()
        for i in range(nbuf):
            free.put(i)
        q = queue.Queue(maxsize=self.depth)

        stop = threading.Event()                            # set when the consumer abandons the generator early

        def take(src):
            while not stop.is_set():
                try:
                    return src.get(timeout=0.2)
                except queue.Empty:
                    pass
            return None

        def give(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.2)
                    return
                except queue.Full:
                    pass

        err = []

        def reader():
            try:
                for ids in id_batches:
                    i = take(free)
                    if i is None:
                        return
                    tf, tr = ring[i][0][:len(ids)], ring[i][1][:len(ids)]
                    self._fill(ids, tf, tr)
                    give((ids, i, tf, tr))
            except BaseException as e:                      # surfaces in the consumer, not in a dead thread
                err.append(e)
            finally:
                give(None)
        threading.Thread(target=reader, daemon=True).start()
        side = torch.cuda.Stream(device=self.device) if cuda else None
        pending = None                                      # (ring index, event): its H2D copy may still be reading the buffer
        try:
            yield from self._consume(q, free, side, cuda, pending)
            if err:
                raise err[0]
        finally:
            stop.set()                                      # a `break` in the consumer: the reader thread ends, the ring is freed

    def _consume(self, q, free, side, cuda, pending):
        while True:
            item = q.get()
            if item is None:
                return
            ids, i, tf, tr = item
            if cuda:
                with torch.cuda.stream(side):
                    df, dr = tf.to(self.device, non_blocking=True), tr.to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(side)
                torch.cuda.current_stream().wait_stream(side)
                df.record_stream(torch.cuda.current_stream()); dr.record_stream(torch.cuda.current_stream())
                if pending is not None:                     # the previous batch's copy has certainly been issued before this one
                    pending[1].synchronize()
                    free.put(pending[0])
                pending = (i, ev)
            else:
                df, dr = tf.clone(), tr.clone()
                free.put(i)
            yield ids, df, dr
        # (the last pending buffer is dropped with the ring)


# ================================================================================================ sampler + loaders
def distributed_indices(n, world, rank, epoch, shuffle=True, seed=0):
    """torch.utils.data.distributed.DistributedSampler's index list (what utils/data.py:122-124 builds): a permutation
    seeded by seed + epoch, padded by wrapping to a multiple of `world`, then every world-th index from `rank`."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    if world <= 1:
        return idx
    total = int(math.ceil(n / world)) * world
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * int(math.ceil(pad / len(idx))))[:pad]
    return idx[rank:total:world]


STREAMED_MAX_RANKS_PER_HOST = 2     # beyond this the streamed store cannot feed the ranks (see TrainLoader)


class TrainLoader(object):
    """get_train_loader (utils/data.py:115-131): yields (frames, regions, None, captions, pos_tags, cap_lens, video_ids), the
    tuple run_gun.py:147 unpacks (`spatials` is read by the reference and never used; it is not loaded here).
    frames / regions are device tensors, regions already cut to num_obj; captions / pos_tags are (B, max_words) int64 on the
    device; cap_lens / video_ids are tuples of ints ordered like the batch rows (video id descending, utils/data.py:90)."""

    def __init__(self, captions, features, batch_size, world_size=1, rank=0, shuffle=True, seed=None, drop_last=False):
        """seed None (the default, = the reference): one process shuffles as `DataLoader(shuffle=True)` does -- every pass over
        the data draws two words from torch's global generator (the DataLoader iterator's base seed, then the RandomSampler's
        seed: torch/utils/data/dataloader.py, sampler.py), so after the same `torch.manual_seed` (train_debug.py:34-36) the
        batches are the reference's (tests/golden/loader_ref.npz); several ranks use DistributedSampler's seed 0.  An int:
        a permutation seeded by seed + epoch, independent of the global generator."""
        self.caps = captions if isinstance(captions, CaptionSet) else CaptionSet(captions)
        self.features, self.batch_size = features, batch_size
        self.world, self.rank, self.shuffle, self.seed, self.drop_last = world_size, rank, shuffle, seed, drop_last
        self.epoch = 0
        if isinstance(features, StreamedFeatures) and world_size > STREAMED_MAX_RANKS_PER_HOST:
            # measured (tools/loader_bench.py): one process streams 3.9-6.6 k clips/s (16-27 GB/s of host reads + H2D), at or
            # below what ONE GPU's train step consumes; N ranks on one host would need N times that out of the same DRAM / PCIe
            # root -- 8 ranks ~ 35 k clips/s ~ 145 GB/s.  Only the HBM-resident store scales with the GPUs.
            import warnings
            warnings.warn('StreamedFeatures with %d ranks per host: the host-side stream (about 4-6 k clips/s per process, '
                          'shared DRAM / PCIe) will bound the step rate; use ResidentFeatures (the whole MSVD / MSR-VTT feature '
                          'set fits in one GPU\'s HBM) for multi-GPU training' % world_size, RuntimeWarning, stacklevel=2)

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _batches(self):
        n = len(self.caps)
        if self.world > 1:
            idx = distributed_indices(n, self.world, self.rank, self.epoch, True, self.seed or 0)
        elif self.shuffle and self.seed is None:        # single process: DataLoader(shuffle=True) on torch's global generator
            torch.empty((), dtype=torch.int64).random_()                       # the iterator's base seed (drawn, unused here)
            g = torch.Generator()
            g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
            idx = torch.randperm(n, generator=g).tolist()
        else:
            idx = distributed_indices(n, 1, 0, self.epoch, self.shuffle, self.seed or 0)
        out = []
        for s in range(0, len(idx), self.batch_size):
            b = idx[s:s + self.batch_size]
            if self.drop_last and len(b) < self.batch_size:
                break
            b = sorted(b, key=lambda i: self.caps.video_ids[i], reverse=True)      # stable, like list.sort in the collate
            out.append(b)
        return out

    def __len__(self):
        n = len(self.caps) if self.world <= 1 else int(math.ceil(len(self.caps) / self.world))
        return n // self.batch_size if self.drop_last else int(math.ceil(n / self.batch_size))

    def __iter__(self):
        batches = self._batches()
        dev = self.features.device
        caps = self.caps

        def pack(b, f, r):
            ib = torch.as_tensor(b, dtype=torch.int64)
            return (f, r, None, _h2d(caps.captions[ib], torch.int64, dev), _h2d(caps.pos_tags[ib], torch.int64, dev),
                    tuple(caps.lengths[i] for i in b), tuple(caps.video_ids[i] for i in b))
        if isinstance(self.features, StreamedFeatures):
            vids = [[caps.video_ids[i] for i in b] for b in batches]
            for b, (_, f, r) in zip(batches, self.features.prefetch(vids)):
                yield pack(b, f, r)
        else:
            for b in batches:
                f, r = self.features.batch([caps.video_ids[i] for i in b])
                yield pack(b, f, r)


class EvalLoader(object):
    """get_eval_loader (utils/data.py:134-147): videos of `eval_range` (utils/opt.py:81,89), yields
    (frames, regions, None, video_ids) with the batch in ascending video id (utils/data.py:104)."""

    def __init__(self, eval_range, features, batch_size, world_size=1, rank=0):
        self.ids = list(range(*eval_range))
        self.features, self.batch_size, self.world, self.rank = features, batch_size, world_size, rank

    def _mine(self):
        if self.world <= 1:
            return self.ids
        return [self.ids[i] for i in distributed_indices(len(self.ids), self.world, self.rank, 0, True, 0)]

    def __len__(self):
        return int(math.ceil(len(self._mine()) / self.batch_size))

    def __iter__(self):
        mine = self._mine()
        batches = [sorted(mine[s:s + self.batch_size]) for s in range(0, len(mine), self.batch_size)]
        if isinstance(self.features, StreamedFeatures):
            for b, (_, f, r) in zip(batches, self.features.prefetch(batches)):
                yield f, r, None, tuple(b)
        else:
            for b in batches:
                f, r = self.features.batch(b)
                yield f, r, None, tuple(b)
