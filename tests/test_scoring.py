"""BLEU / ROUGE_L / CIDEr-D of dlsg_amd/scoring.py against the reference's own scorer classes
(tests/golden/scoring.json, made by tests/golden/make_goldens_r2.py from caption-eval/pycocoevalcap) and the evaluate.py
driver functions."""
import json
import os

import numpy as np
import torch

import dlsg_amd
from dlsg_amd import scoring as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'scoring.json')


def test_metrics_match_reference_scorers():
    for case in json.load(open(GOLD)):
        gts, res = case['gts'], case['res']
        b, bp = S.bleu(gts, res, 4)
        assert np.allclose(b, case['bleu'], rtol=1e-12, atol=1e-15)
        assert np.allclose(bp, case['bleu_per'], rtol=1e-12, atol=1e-15)
        r, rp = S.rouge_l(gts, res)
        assert abs(r - case['rouge']) <= 1e-12 and np.allclose(rp, case['rouge_per'], atol=1e-12)
        c, cp = S.cider(gts, res)
        assert abs(c - case['cider']) <= 1e-9 and np.allclose(cp, case['cider_per'], atol=1e-9)


def test_tokenizer_drops_punctuation_and_case():
    assert S.tokenize('A man is playing the guitar.') == 'a man is playing the guitar'
    assert S.tokenize("Two dogs -- don't run, they WALK!") == "two dogs don't run they walk"
    assert S.tokenize('a woman is slicing an onion') == 'a woman is slicing an onion'


def test_reference_file_and_prediction_formats(tmp_path):
    p = tmp_path / 'ref.txt'
    p.write_text('vid1\tA man plays.\nvid1\tA män is playing guitar\nvid2\ta dog runs\n', encoding='utf-8')
    ref = S.convert_data_to_coco_scorer_format(str(p))
    assert [c['cap_id'] for c in ref['vid1']] == [0, 1] and ref['vid1'][1]['caption'] == 'A mn is playing guitar'
    pred = S.convert_prediction({'vid1': 'a man plays guitar', 'vid2': 'a dog runs'})
    scores, _ = S.CaptionScorer().score(ref, pred, pred.keys())
    assert set(scores) == {'Bleu_1', 'Bleu_2', 'Bleu_3', 'Bleu_4', 'ROUGE_L', 'CIDEr'}
    assert scores['Bleu_1'] > 0.7 and 0 < scores['ROUGE_L'] <= 1


def test_evaluate_driver_decodes_and_scores():
    """evaluate.py:56-98 with a stub network: ids -> words through decoder.decode_tokens, then the scorer"""
    vocab = dlsg_amd.make_vocab(4)
    for w in ('a', 'man', 'plays', 'guitar', 'dog', 'runs'):
        vocab.add_word(w)

    class Dec(object):
        def __init__(self):
            self.vocab = vocab
        decode_tokens = dlsg_amd.model.Decoder.decode_tokens

    class Net(object):
        decoder = Dec()

        def __call__(self, frames, regions, caption):
            ids = {7: ['a', 'man', 'plays', 'guitar', '<end>', 'dog'], 9: ['a', 'dog', 'runs', '<end>', 'a', 'a']}
            return torch.tensor([[vocab(w) for w in ids[int(f[0, 0])]] for f in frames]), 0, 0, []
    loader = [(torch.tensor([[[7.0]], [[9.0]]]), None, None, (7, 9))]
    ref = {'7': [{'caption': 'a man plays a guitar'}], '9': [{'caption': 'the dog runs'}]}
    scores, result = S.evaluate(Net(), loader, ref)
    assert result[7] == 'a man plays guitar' and result[9] == 'a dog runs'
    assert scores['Bleu_1'] > 0.6
