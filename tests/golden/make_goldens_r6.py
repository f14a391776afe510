"""Round-6 fixture, generated from the imported reference (/root/reference; authoring container only).

`small_decbeam.npz` -- models/layer.py:449-460: the reference's `Decoder.forward(cnn_feats, None, ...)` called ON ITS OWN with
beam_size != 1 runs `BeamSearch.search` itself and returns the best beam per clip.  Weights and inputs of small_msvd.npz, the
proposals from the reference's own encoder; beams 3 and 5, with the proposals' means as the global feature and with the
`step_feats` of small_stepfeats.npz in their place (layer.py:404-405).  Arrays only.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, HERE)
REF = '/root/reference'

for name in ('allennlp', 'allennlp.common', 'allennlp.common.checks'):
    sys.modules[name] = types.ModuleType(name)
sys.modules['allennlp.common.checks'].ConfigurationError = type('ConfigurationError', (Exception,), {})
sys.path.insert(0, REF)


def decoder_beam_case():
    import models.model as ref_model
    from dlsg_amd.config import make_vocab
    from make_goldens import small_args
    fx = np.load(os.path.join(HERE, 'small_msvd.npz'))
    step = torch.from_numpy(np.load(os.path.join(HERE, 'small_stepfeats.npz'))['step_feats'])
    args = small_args()
    vocab = make_vocab(int(fx['meta.V']))
    torch.manual_seed(0)
    net = ref_model.CapGnnModel(args, vocab).eval()
    net.load_state_dict({k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith('w.')})
    frames, regions = torch.from_numpy(fx['frames']), torch.from_numpy(fx['regions'])
    out = {}
    with torch.no_grad():
        obj, mot = net.encoder(frames, regions)
        for k in (3, 5):
            net.update_beam_size(k)
            ids, alphas = net.decoder(obj, None, None, 1.0, cnn_feats_2=mot)
            assert alphas == []
            out['beam%d_ids' % k] = ids.numpy()
            ids_s, _ = net.decoder(obj, None, None, 1.0, cnn_feats_2=mot, step_feats=step)
            out['beam%d_ids_stepfeats' % k] = ids_s.numpy()
            # the model's own beam search is this call on the encoder's proposals (models/model.py:37-42)
            assert np.array_equal(net(frames, regions, None)[0].numpy(), out['beam%d_ids' % k])
    assert np.array_equal(out['beam5_ids'], fx['beam5_ids'])
    np.savez_compressed(os.path.join(HERE, 'small_decbeam.npz'), **out)
    print('small_decbeam:', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    decoder_beam_case()
