"""GPU (-m gpu): the loop of INTEGRATION.md run for real on a tiny synthetic corpus -- HDF5 features + caption pickle written
in the reference's layout (utils/data.py:13-50), `TrainLoader` -> hipGraph-replayed `Trainer.step` until the model has
memorised the captions, `EvalLoader` -> greedy / beam decode -> `evaluate` (evaluate.py:56-98) scoring them against the
reference text file; then `GanTrainer.iteration` (run_gun.py:147-234) over loader batches, ragged last batch included."""
import math
import pickle
import random

import numpy as np
import pytest
import torch

import dlsg_amd
from dlsg_amd import data as D
from helpers import small_args, gan_args

pytestmark = pytest.mark.gpu

V, N, L = 40, 24, 26


def corpus(tmp_path, seed=0, make_args=small_args):
    rng = np.random.RandomState(seed)
    args = make_args()
    F = args.a_feature_size + args.m_feature_size
    feats = rng.randn(N, L, F).astype(np.float32)
    vfeats = rng.randn(N, L, 36, args.region_feature_size).astype(np.float32)
    fp, rp, cp, tp = [str(tmp_path / n) for n in ('f.h5', 'r.h5', 'c.pkl', 'ref.txt')]
    D.H5File.create(fp).write('feats', feats).close()
    D.H5File.create(rp).write('vfeats', vfeats).close()
    vocab = dlsg_amd.make_vocab(V)
    caps, lens, vids, lines = [], [], [], []
    for v in range(N):                                   # one sentence per clip: 3..8 words then <end>
        n = int(rng.randint(3, 9))
        ids = rng.randint(4, V, size=n).tolist()
        row = torch.zeros(L, dtype=torch.long)
        row[:n] = torch.tensor(ids)
        row[n] = vocab('<end>')
        caps.append(row); lens.append(n + 1); vids.append(v)
        lines.append('%d\t%s' % (v, ' '.join(vocab.idx2word[i] for i in ids)))
    with open(cp, 'wb') as f:
        pickle.dump((caps, [torch.zeros(L, dtype=torch.long)] * N, lens, vids), f)
    with open(tp, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    return args, vocab, fp, rp, cp, tp


def test_caption_training_memorises_a_tiny_corpus_and_scores_it(tmp_path):
    args, vocab, fp, rp, cp, tp = corpus(tmp_path)
    torch.manual_seed(0)
    random.seed(12)
    model = dlsg_amd.CapGnnModel(args, vocab).cuda().train()
    feats = D.ResidentFeatures(fp, rp, args.num_obj, 'cuda', ops=model.ops)
    loader = D.TrainLoader(cp, feats, 8, seed=0)
    tr = dlsg_amd.Trainer(model, lr=2e-3, use_graphs=True)
    first = last = None
    for epoch in range(300):
        loader.set_epoch(epoch)
        tot = 0.0
        for frames, regions, _, captions, _, cap_lens, _ in loader:
            tot += float(tr.step(frames, regions, captions, cap_lens, dlsg_amd.ss_epsilon(epoch)))
        first = tot if first is None else first
        last = tot
    assert math.isfinite(last) and last < 0.15 * first, (first, last)
    assert tr._graphs is not None                                        # the steps were hipGraph replays
    model.eval()
    ref = dlsg_amd.convert_data_to_coco_scorer_format(tp)
    for beam in (1, 3):
        model.update_beam_size(beam)
        scores, result = dlsg_amd.evaluate(model, D.EvalLoader((0, N), feats, 7), ref)
        assert len(result) == N
        exact = sum(result[v] == ref[str(v)][0]['caption'] for v in range(N))
        assert exact >= N - 4, (beam, exact)
        assert scores['Bleu_4'] > 0.8 and scores['ROUGE_L'] > 0.85 and scores['CIDEr'] > 5.0, scores


def test_gathering_into_the_graphs_static_inputs_equals_passing_batches(tmp_path):
    """bench.py's `sustained` loop (run_gun.py:147-160 with an HBM-resident store): every step the next batch is gathered straight
    into the replayed graph's static input buffers (`ResidentFeatures.batch(ids, out=...)`), captions and lengths go up through
    the pinned, asynchronous copies -- the weights after six steps are those of a trainer that is handed fresh batch tensors, bit
    for bit, and the host never has to wait for the device in between."""
    args, vocab, fp, rp, cp, tp = corpus(tmp_path)
    feats_np = D.H5File(fp)['feats'].read_rows(0, N)
    vfeats_np = D.H5File(rp)['vfeats'].read_rows(0, N)
    caps_t, _, lens, vids = pickle.load(open(cp, 'rb'))
    caps_t = torch.stack(caps_t)
    order = [list(np.random.RandomState(3 + k).permutation(N)[:8]) for k in range(6)]
    flats = []
    for static in (False, True):
        torch.manual_seed(0)
        random.seed(12)
        model = dlsg_amd.CapGnnModel(args, vocab).cuda().train()
        store = D.ResidentFeatures.from_arrays(feats_np, vfeats_np, args.num_obj, 'cuda', ops=model.ops)
        tr = dlsg_amd.Trainer(model, use_graphs=True)
        for k, ids in enumerate(order):
            ids = sorted(ids, reverse=True)
            cb, lb = caps_t[ids], [lens[i] for i in ids]
            st = tr.static_inputs() if static else None
            if st is not None:
                f, r = store.batch(ids, out=(st[0], st[1]))
                assert f.data_ptr() == st[0].data_ptr() and r.data_ptr() == st[1].data_ptr()
            else:
                f, r = store.batch(ids)
            tr.step(f, r, cb.cuda(), lb, 0.9)
        tr.check()
        flats.append(model._flat.clone())
    assert torch.equal(flats[0], flats[1])


def test_gan_loop_over_loader_batches(tmp_path):
    args, vocab, fp, rp, cp, tp = corpus(tmp_path, seed=1, make_args=lambda: gan_args(use_visual_gan=True))
    torch.manual_seed(0)
    random.seed(12)
    model = dlsg_amd.CapGnnModel(args, vocab).cuda().train()
    critic = dlsg_amd.DiscV2(args, len(vocab)).cuda()
    feats = D.StreamedFeatures(fp, rp, args.num_obj, 'cuda', workers=2)
    loader = D.TrainLoader(cp, feats, 10, seed=0)                          # 24 clips: batches of 10, 10, 4
    gan = dlsg_amd.GanTrainer(model, critic, lr=1e-3, num_D=2, gan_lambda=0.01, total_step=len(loader))
    before = [p.detach().clone() for p in critic.parameters()]
    caps = []
    for epoch in range(4):
        loader.set_epoch(epoch)
        for i, (frames, regions, _, captions, _, cap_lens, _) in enumerate(loader, start=1):
            out = gan.iteration(frames, regions, captions, cap_lens, 1.0, epoch, i)
            assert all(math.isfinite(out[k]) for k in ('cap_loss', 'loss_G', 'total_loss', 'loss_D', 'wasserstein')), out
            caps.append(out['cap_loss'])
    assert np.mean(caps[-3:]) < 0.93 * np.mean(caps[:3]), caps                    # 3.90 -> 3.42 in 12 iterations
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, critic.parameters()))
    path = str(tmp_path / 'last.pt')
    dlsg_amd.save_checkpoint(path, 3, gan)
    model2 = dlsg_amd.CapGnnModel(args, vocab).cuda()
    critic2 = dlsg_amd.DiscV2(args, len(vocab)).cuda()
    gan2 = dlsg_amd.GanTrainer(model2, critic2, lr=1e-3, num_D=2, gan_lambda=0.01, total_step=len(loader))
    assert dlsg_amd.load_checkpoint(path, gan2) == 3
    for (k, p), (_, q) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(p, q), k
