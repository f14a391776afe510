"""feature I/O and loaders (dlsg_amd/data.py) against the reference's semantics (utils/data.py:13-147).
The HDF5 fixtures are written at test time by the HDF5 C library itself (no h5py in the image) in the reference's layout:
`feats` (N,26,A+M), `vfeats` (N,26,36,R), plus the caption pickle tuple (captions, pos_tags, lengths, video_ids)."""
import pickle

import numpy as np
import pytest
import torch

from dlsg_amd import data as D

try:
    D._libhdf5()
    HAVE_H5 = True
except RuntimeError:
    HAVE_H5 = False
needs_h5 = pytest.mark.skipif(not HAVE_H5, reason='libhdf5 not available')


def make_dataset(tmp_path, N=23, T=26, F=12, O=36, R=8, ncap=57, L=26, seed=0):
    rng = np.random.RandomState(seed)
    feats = rng.randn(N, T, F).astype(np.float32)
    vfeats = rng.randn(N, T, O, R).astype(np.float32)
    sfeats = rng.rand(N, T, O, 5).astype(np.float32)
    fp, rp = str(tmp_path / 'msvd_features.h5'), str(tmp_path / 'msvd_region_feature.h5')
    D.H5File.create(fp).write('feats', feats).close()
    D.H5File.create(rp).write('vfeats', vfeats).write('sfeats', sfeats).close()
    vids = rng.randint(0, N, size=ncap).tolist()
    lens = rng.randint(3, L + 1, size=ncap).tolist()
    caps = [torch.from_numpy(np.pad(rng.randint(4, 50, size=n), (0, L - n))).long() for n in lens]
    tags = [torch.from_numpy(np.pad(rng.randint(1, 9, size=n), (0, L - n))).long() for n in lens]
    cp = str(tmp_path / 'msvd_captions_train.pkl')
    with open(cp, 'wb') as f:
        pickle.dump((caps, tags, lens, vids), f)
    return fp, rp, cp, feats, vfeats, caps, tags, lens, vids


@needs_h5
def test_hdf5_rows_and_conversion(tmp_path):
    fp, rp, cp, feats, vfeats, *_ = make_dataset(tmp_path)
    with D.H5File(fp) as h:
        ds = h['feats']
        assert ds.shape == feats.shape and ds.is_float and ds.itemsize == 4 and len(ds) == feats.shape[0]
        assert np.array_equal(ds[5], feats[5]) and np.array_equal(ds[-1], feats[-1])
        assert np.array_equal(ds.read_rows(3, 7), feats[3:10]) and np.array_equal(ds[2:4], feats[2:4])
        with pytest.raises(IndexError):
            ds.read_rows(20, 9)
        with pytest.raises(KeyError):
            h['nope']
    p64 = str(tmp_path / 'd.h5')
    D.H5File.create(p64).write('x', feats.astype(np.float64)).write('i', np.arange(12, dtype=np.int64).reshape(3, 4)).close()
    with D.H5File(p64) as h:
        assert h['x'].itemsize == 8 and np.array_equal(h['x'][1], feats[1])           # the library converts to float32
        assert not h['i'].is_float and np.array_equal(h['i'].read_rows(0, 3, dtype=np.int64), np.arange(12).reshape(3, 4))
    with pytest.raises(FileNotFoundError):
        D.H5File(str(tmp_path / 'missing.h5'))


@pytest.mark.parametrize('n,world', [(57, 4), (64, 8), (5, 4), (1301, 2)])
def test_sampler_is_torchs_distributed_sampler(n, world):
    from torch.utils.data.distributed import DistributedSampler
    for epoch in (0, 3):
        for rank in range(world):
            s = DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=True, seed=0)
            s.set_epoch(epoch)
            assert list(iter(s)) == D.distributed_indices(n, world, rank, epoch, True, 0)
    assert D.distributed_indices(n, 1, 0, 0, shuffle=False) == list(range(n))


@needs_h5
@pytest.mark.parametrize('store', ['resident', 'streamed'])
def test_train_loader_batches_follow_the_reference_collate(tmp_path, store):
    fp, rp, cp, feats, vfeats, caps, tags, lens, vids = make_dataset(tmp_path)
    num_obj = 16
    fs = D.ResidentFeatures(fp, rp, num_obj, 'cpu', chunk=5) if store == 'resident' else D.StreamedFeatures(fp, rp, num_obj, 'cpu')
    for world, rank in ((1, 0), (2, 1)):
        ld = D.TrainLoader(cp, fs, batch_size=8, world_size=world, rank=rank, seed=0)
        ld.set_epoch(2)
        want_idx = D.distributed_indices(len(vids), world, rank, 2, True, 0)
        seen = []
        nb = 0
        for frames, regions, spatials, captions, pos_tags, cap_lens, video_ids in ld:
            nb += 1
            B = frames.shape[0]
            assert spatials is None and regions.shape == (B, 26, num_obj, 8) and frames.shape == (B, 26, 12)
            assert list(video_ids) == sorted(video_ids, reverse=True)                  # utils/data.py:90 sorts by x[-1] = video id
            for j, v in enumerate(video_ids):
                assert np.array_equal(frames[j].numpy(), feats[v])
                assert np.array_equal(regions[j].numpy(), vfeats[v][:, :num_obj])      # run_gun.py:158
            # caption rows travel with their video: find the sample each row came from
            chunk = want_idx[len(seen):len(seen) + B]
            chunk = sorted(chunk, key=lambda i: vids[i], reverse=True)
            for j, i in enumerate(chunk):
                assert vids[i] == video_ids[j] and lens[i] == cap_lens[j]
                assert torch.equal(captions[j], caps[i]) and torch.equal(pos_tags[j], tags[i])
            seen += chunk
        assert nb == len(ld) and sorted(seen) == sorted(want_idx)
    a = [b[-1] for b in D.TrainLoader(cp, fs, 8, seed=0)]
    ld2 = D.TrainLoader(cp, fs, 8, seed=0)
    ld2.set_epoch(1)
    assert a != [b[-1] for b in ld2]                                                    # reshuffled per epoch


@needs_h5
@pytest.mark.parametrize('mapped,workers', [(True, 1), (True, 3), (False, 1)])
def test_streamed_store_paths_agree(tmp_path, mapped, workers):
    """memory-mapped rows (contiguous datasets) and libhdf5 hyperslab reads fill the pinned ring with the same bytes; ragged
    last batch, more batches than ring slots"""
    fp, rp, cp, feats, vfeats, *_ = make_dataset(tmp_path)
    st = D.StreamedFeatures(fp, rp, 16, 'cpu', depth=2, workers=workers)
    assert st.mapped                                              # H5File.write stores contiguously, as h5py's create_dataset(data=) does
    off = st.frames.lib.H5Dget_offset(st.frames.did)
    with open(fp, 'rb') as f:
        f.seek(off)
        assert np.array_equal(np.frombuffer(f.read(feats[0].nbytes), np.float32).reshape(feats[0].shape), feats[0])
    st.mapped = mapped
    rng = np.random.RandomState(1)
    batches = [rng.randint(0, feats.shape[0], size=n).tolist() for n in (5, 5, 5, 5, 5, 5, 5, 2)]
    got = list(st.prefetch(batches))
    assert [g[0] for g in got] == batches
    for ids, f, r in got:
        assert np.array_equal(f.numpy(), feats[ids]) and np.array_equal(r.numpy(), vfeats[ids][:, :, :16])
    assert list(st.prefetch([])) == []


@needs_h5
def test_resident_batch_gathers_into_given_buffers(tmp_path):
    fp, rp, cp, feats, vfeats, *_ = make_dataset(tmp_path)
    fs = D.ResidentFeatures(fp, rp, 16, 'cpu')
    f = torch.full((3, 26, 12), -1.0); r = torch.full((3, 26, 16, 8), -1.0)
    f2, r2 = fs.batch([5, 0, 22], out=(f, r))
    assert f2.data_ptr() == f.data_ptr() and r2.data_ptr() == r.data_ptr()
    assert np.array_equal(f.numpy(), feats[[5, 0, 22]]) and np.array_equal(r.numpy(), vfeats[[5, 0, 22]][:, :, :16])


@needs_h5
def test_resident_store_from_arrays_equals_the_file_backed_one(tmp_path):
    """ResidentFeatures.from_arrays (bench.py's `sustained` leg, synthetic stores): same rows, same object cut as the HDF5 store"""
    fp, rp, cp, feats, vfeats, *_ = make_dataset(tmp_path)
    a = D.ResidentFeatures(fp, rp, 16, 'cpu')
    b = D.ResidentFeatures.from_arrays(feats, vfeats, 16, 'cpu')
    assert b.n == a.n and b.frame_shape == tuple(a.frame_shape) and b.region_shape == tuple(a.region_shape) and b.bytes == a.bytes
    for ids in ([5, 0, 22], [1], list(range(a.n))[::-1]):
        fa, ra = a.batch(ids)
        fb, rb = b.batch(ids)
        assert torch.equal(fa, fb) and torch.equal(ra, rb)
    with pytest.raises(ValueError):
        D.ResidentFeatures.from_arrays(feats[0], vfeats, 16, 'cpu')


@needs_h5
def test_eval_loader_range_and_order(tmp_path):
    fp, rp, cp, feats, vfeats, *_ = make_dataset(tmp_path)
    fs = D.ResidentFeatures(fp, rp, 36, 'cpu')
    got = []
    for frames, regions, spatials, video_ids in D.EvalLoader((13, 23), fs, batch_size=4):
        assert list(video_ids) == sorted(video_ids) and regions.shape[2] == 36
        for j, v in enumerate(video_ids):
            assert np.array_equal(frames[j].numpy(), feats[v])
        got += list(video_ids)
    assert got == list(range(13, 23))
    parts = [sum((list(b[-1]) for b in D.EvalLoader((13, 23), fs, 4, world_size=2, rank=r)), []) for r in range(2)]
    assert sorted(parts[0] + parts[1]) == list(range(13, 23))


@needs_h5
@pytest.mark.gpu
def test_resident_features_gather_on_device(tmp_path):
    """the whole feature set in HBM, a batch = the HIP row gather: bit-identical to the host path and to H2D streaming"""
    from dlsg_amd.hip import HipOps
    fp, rp, cp, feats, vfeats, caps, tags, lens, vids = make_dataset(tmp_path, N=40, F=6144, R=2048, O=36, ncap=90)
    ops = HipOps()
    res = D.ResidentFeatures(fp, rp, 16, 'cuda', ops=ops, chunk=7)
    assert res.bytes == 40 * (26 * 6144 + 26 * 16 * 2048) * 4
    st = D.StreamedFeatures(fp, rp, 16, 'cuda')
    a = list(D.TrainLoader(cp, res, 16, seed=3))
    b = list(D.TrainLoader(cp, st, 16, seed=3))
    assert len(a) == len(b) == 6
    for x, y in zip(a, b):
        assert x[-1] == y[-1] and x[-2] == y[-2]
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and torch.equal(x[3], y[3])
        for j, v in enumerate(x[-1]):
            assert np.array_equal(x[0][j].cpu().numpy(), feats[v])
            assert np.array_equal(x[1][j].cpu().numpy(), vfeats[v][:, :16])


@needs_h5
def test_streamed_prefetch_can_be_abandoned_and_reports_reader_errors(tmp_path):
    import threading
    import time
    fp, rp, cp, feats, vfeats, *_ = make_dataset(tmp_path)
    st = D.StreamedFeatures(fp, rp, 16, 'cpu', depth=1, workers=2)
    before = threading.active_count()
    gen = st.prefetch([[0, 1], [2, 3], [4, 5], [6, 7], [8, 9]])
    next(gen)
    gen.close()                                              # the consumer leaves early: the reader thread must end
    for _ in range(50):
        if threading.active_count() <= before:
            break
        time.sleep(0.1)
    assert threading.active_count() <= before
    with pytest.raises(IndexError):                          # a failing read surfaces in the consumer
        list(st.prefetch([[0, 1], [10 ** 6]]))


# ------------------------------------------------------------------ against the reference's own loaders (tests/golden/loader_ref.npz)
import os  # noqa: E402

REF_FX = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'loader_ref.npz')


def _check_train(fx, tag, loader, feats, vfeats, caps, tags, num_obj, dev='cpu'):
    nb = int(fx[tag + '.nbatch'])
    got = list(loader)
    assert len(got) == nb == len(loader), (tag, len(got), nb)
    for bi, (frames, regions, spatials, captions, pos_tags, cap_lens, video_ids) in enumerate(got):
        pre = '%s.b%d.' % (tag, bi)
        assert list(video_ids) == fx[pre + 'video_ids'].tolist(), pre
        assert list(cap_lens) == fx[pre + 'lengths'].tolist(), pre
        idx = fx[pre + 'index'].tolist()
        for j, i in enumerate(idx):                                   # the rows are the reference's rows, in its order
            assert torch.equal(captions[j].cpu(), caps[i]) and torch.equal(pos_tags[j].cpu(), tags[i]), (pre, j)
        fs = frames.double().sum((1, 2)).cpu().numpy()
        rs = regions.double().sum((1, 2, 3)).cpu().numpy()
        assert regions.shape[2] == num_obj
        assert np.allclose(fs, fx[pre + 'frames_sum'], rtol=0, atol=1e-3) and np.allclose(rs, fx[pre + 'regions_sum'], rtol=0, atol=1e-3), pre
        assert float(captions.double().sum()) == float(fx[pre + 'captions_sum']) and float(pos_tags.double().sum()) == float(fx[pre + 'pos_tags_sum'])


def _check_eval(fx, tag, loader):
    nb = int(fx[tag + '.nbatch'])
    got = list(loader)
    assert len(got) == nb == len(loader), (tag, len(got), nb)
    for bi, (frames, regions, spatials, video_ids) in enumerate(got):
        pre = '%s.b%d.' % (tag, bi)
        assert list(video_ids) == fx[pre + 'video_ids'].tolist(), pre
        assert np.allclose(frames.double().sum((1, 2)).cpu().numpy(), fx[pre + 'frames_sum'], rtol=0, atol=1e-3)
        assert np.allclose(regions.double().sum((1, 2, 3)).cpu().numpy(), fx[pre + 'regions_sum'], rtol=0, atol=1e-3)


def _reference_loader_cases(tmp_path, make_store):
    """every batch the REFERENCE's get_train_loader / get_eval_loader / DistributedSampler delivered on this data set (the
    fixture was written by tests/golden/make_goldens_r5.py from the imported utils/data.py): same rows, same order"""
    fx = np.load(REF_FX)
    N, num_obj, bs = int(fx['meta.N']), int(fx['meta.num_obj']), int(fx['meta.batch_size'])
    fp, rp, cp, feats, vfeats, caps, tags, lens, vids = make_dataset(tmp_path, N=N)
    store = make_store(fp, rp, num_obj)
    # one process, DataLoader(shuffle=True) after torch.manual_seed: two passes
    torch.manual_seed(int(fx['meta.torch_seed']))
    ld = D.TrainLoader(cp, store, batch_size=bs)
    _check_train(fx, 'train1.e0', ld, feats, vfeats, caps, tags, num_obj)
    _check_train(fx, 'train1.e1', ld, feats, vfeats, caps, tags, num_obj)
    # two ranks: DistributedSampler, epochs 0 and 1
    for rank in range(2):
        ld = D.TrainLoader(cp, store, batch_size=bs, world_size=2, rank=rank)
        for epoch in range(2):
            ld.set_epoch(epoch)
            _check_train(fx, 'train2.r%d.e%d' % (rank, epoch), ld, feats, vfeats, caps, tags, num_obj)
    rng = tuple(int(x) for x in fx['meta.eval_range'])
    _check_eval(fx, 'eval1', D.EvalLoader(rng, store, batch_size=5))
    for rank in range(2):
        _check_eval(fx, 'eval2.r%d' % rank, D.EvalLoader(rng, store, batch_size=5, world_size=2, rank=rank))


@needs_h5
@pytest.mark.parametrize('kind', ['resident', 'streamed'])
def test_loaders_deliver_the_reference_loaders_batches(tmp_path, kind):
    mk = (lambda fp, rp, no: D.ResidentFeatures(fp, rp, no, 'cpu', chunk=5)) if kind == 'resident' else \
         (lambda fp, rp, no: D.StreamedFeatures(fp, rp, no, 'cpu'))
    _reference_loader_cases(tmp_path, mk)


@needs_h5
@pytest.mark.gpu
def test_resident_store_on_the_gpu_delivers_the_reference_loaders_batches(tmp_path):
    from dlsg_amd.hip import HipOps
    ops = HipOps()
    _reference_loader_cases(tmp_path, lambda fp, rp, no: D.ResidentFeatures(fp, rp, no, 'cuda', ops=ops))
