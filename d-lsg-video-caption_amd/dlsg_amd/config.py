"""Configuration namespace and vocabulary protocol of the hot path.

Field names are the reference's flat argparse namespace (utils/opt.py:14-93); only the fields the model
constructors read are kept.  `apply_dataset_overrides` restates what the reference trainer does to
`args` before building the model (run_gun.py:31-40).
"""
from argparse import Namespace


def make_args(**kw):
    """Defaults == utils/opt.py defaults for the fields the model reads."""
    a = dict(
        dataset='msvd', train_batch_size=128, test_batch_size=128, beam_size=5, use_glove=False,
        dropout=0.3, use_visual_gan=True,
        visual_hidden_size=1024, region_projected_size=1024, num_proposals=8, num_obj=16,
        word_size=300, query_hidden_size=1024, decode_hidden_size=1536,
        max_frames=26, max_words=26,
        a_feature_size=1536, m_feature_size=1024, region_feature_size=2048,
        learning_rate=1.6e-4, ss_factor=20,
        num_topk=3, num_D_visual=5, lambda_D_visual=0.01,          # DiscV2 / WGAN-GP (utils/opt.py:36-37,47)
    )
    a.update(kw)
    return Namespace(**a)


def apply_dataset_overrides(args):
    """run_gun.py:31-40"""
    if args.dataset == 'msvd':
        args.decode_hidden_size = 1024
        args.num_proposals = 8
        args.num_obj = 16
        args.num_topk = 3
    else:
        args.decode_hidden_size = 1536
        args.num_proposals = 5
        args.num_obj = 36
        args.num_topk = 5
    return args


def msvd_shaped(**kw):
    """BASELINE.json configs[0..1]: 26 frames, 2048-d 2D + 4096-d 3D, 16 regions, vocab 1k."""
    return apply_dataset_overrides(make_args(dataset='msvd', a_feature_size=2048, m_feature_size=4096, **kw))


def msrvtt_shaped(**kw):
    """BASELINE.json configs[2]: MSR-VTT-shaped, 36 regions, 5 proposals, D=1536, vocab 10k."""
    return apply_dataset_overrides(make_args(dataset='msr-vtt', a_feature_size=2048, m_feature_size=4096, **kw))


class Vocabulary(object):
    """Same protocol as utils/utils.py:12-43: __call__(word)->id (unknown -> <unk>), __len__, idx2word."""

    def __init__(self, extra_words=0):
        self.word2idx = {}
        self.idx2word = []
        for w in ('<pad>', '<start>', '<end>', '<unk>'):
            self.add_word(w)
        for i in range(extra_words):
            self.add_word('w%d' % i)

    @property
    def nwords(self):
        return len(self.idx2word)

    def add_word(self, w):
        if w not in self.word2idx:
            self.word2idx[w] = len(self.idx2word)
            self.idx2word.append(w)

    def __call__(self, w):
        return self.word2idx.get(w, self.word2idx['<unk>'])

    def __len__(self):
        return len(self.idx2word)


def make_vocab(size):
    return Vocabulary(size - 4)
