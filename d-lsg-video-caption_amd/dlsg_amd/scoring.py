"""caption scoring and the evaluation driver (reference evaluate.py:16-98 and the pure-Python
metrics of caption-eval/pycocoevalcap: BLEU-1..4, ROUGE_L, CIDEr-D).  CPU-side, off the GPU critical path.

Not reproduced: METEOR and the Stanford PTB tokenizer (caption-eval/pycocoevalcap/{meteor,tokenizer}) shell out to Java with
jars that are not in the reference tree; `tokenize` below is a regular-expression stand-in that lower-cases, splits
punctuation off and drops the tokenizer's punctuation list (ptbtokenizer.py:21-22), which is exact for the model's own
output (space-joined vocabulary words) and close for the raw reference sentences.

The metric definitions follow the published coco-caption algorithms:
  BLEU    corpus-level modified n-gram precision, brevity penalty against the CLOSEST reference length (bleu.py:40)
  ROUGE_L F-measure (beta = 1.2) of the best LCS precision and the best LCS recall over the references (rouge.py:43-71)
  CIDEr-D mean over n = 1..4 of clipped tf-idf cosine similarity with a Gaussian length penalty (sigma = 6), x10,
          idf from the evaluated reference set itself (cider_scorer.py:106-180)
pinned by tests/golden/scoring.json, produced by the reference's own scorer classes.
"""
import collections
import math
import re

PUNCTUATIONS = frozenset(["''", "'", "``", "`", "-LRB-", "-RRB-", "-LCB-", "-RCB-", ".", "?", "!", ",", ":", "-", "--", "...", ";"])
_TOKEN = re.compile(r"\.\.\.|--|``|''|[A-Za-z0-9]+(?:'[a-z]+)?|[^\sA-Za-z0-9]")


def tokenize(sentence):
    """lower-cased tokens without punctuation, joined by single spaces"""
    return ' '.join(t for t in _TOKEN.findall(sentence.lower().replace('\n', ' ')) if t not in PUNCTUATIONS)


def _ngrams(words, n):
    c = collections.Counter()
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            c[tuple(words[i:i + k])] += 1
    return c


# ------------------------------------------------------------------------------------------------ BLEU
def bleu(gts, res, n=4):
    """gts: id -> list of tokenized references; res: id -> [tokenized hypothesis].  Returns ([BLEU_1..n], per-id lists)."""
    small, tiny = 1e-9, 1e-15
    ids = sorted(gts.keys())
    tot_guess, tot_correct = [0] * n, [0] * n
    tot_test = tot_ref = 0
    per = [[] for _ in range(n)]
    for i in ids:
        hyp = res[i][0].split()
        refs = [r.split() for r in gts[i]]
        hc = _ngrams(hyp, n)
        mx = collections.Counter()
        for r in refs:
            for g, c in _ngrams(r, n).items():
                if c > mx[g]:
                    mx[g] = c
        guess = [max(0, len(hyp) - k) for k in range(n)]
        correct = [0] * n
        for g, c in hc.items():
            correct[len(g) - 1] += min(c, mx.get(g, 0))
        reflen = min((abs(len(r) - len(hyp)), len(r)) for r in refs)[1]            # closest length, shorter on ties
        tot_test += len(hyp); tot_ref += reflen
        b = 1.0
        ratio = (len(hyp) + tiny) / (reflen + small)
        for k in range(n):
            tot_guess[k] += guess[k]; tot_correct[k] += correct[k]
            b *= (correct[k] + tiny) / (guess[k] + small)
            s = b ** (1.0 / (k + 1))
            per[k].append(s * math.exp(1 - 1 / ratio) if ratio < 1 else s)
    out, b = [], 1.0
    ratio = (tot_test + tiny) / (tot_ref + small)
    for k in range(n):
        b *= (tot_correct[k] + tiny) / (tot_guess[k] + small)
        s = b ** (1.0 / (k + 1))
        out.append(s * math.exp(1 - 1 / ratio) if ratio < 1 else s)
    return out, per


# ------------------------------------------------------------------------------------------------ ROUGE_L
def _lcs(a, b):
    if len(a) < len(b):
        a, b = b, a
    prev = [0] * (len(b) + 1)
    for x in a:
        cur = [0]
        for j, y in enumerate(b, 1):
            cur.append(prev[j - 1] + 1 if x == y else max(prev[j], cur[j - 1]))
        prev = cur
    return prev[len(b)]


def rouge_l(gts, res, beta=1.2):
    ids = sorted(gts.keys())
    per = []
    for i in ids:
        hyp = res[i][0].split(' ')
        p = r = 0.0
        for ref in gts[i]:
            rt = ref.split(' ')
            l = _lcs(rt, hyp)
            p, r = max(p, l / float(len(hyp))), max(r, l / float(len(rt)))
        per.append((1 + beta ** 2) * p * r / (r + beta ** 2 * p) if p and r else 0.0)
    return sum(per) / len(per), per


# ------------------------------------------------------------------------------------------------ CIDEr-D
def cider(gts, res, n=4, sigma=6.0):
    ids = sorted(gts.keys())
    hyp = [_ngrams(res[i][0].split(), n) for i in ids]
    refs = [[_ngrams(r.split(), n) for r in gts[i]] for i in ids]
    df = collections.Counter()
    for rs in refs:
        for g in set(g for r in rs for g in r):
            df[g] += 1
    log_n = math.log(float(len(ids)))

    def vec(cnt):
        v = [dict() for _ in range(n)]
        norm = [0.0] * n
        length = 0
        for g, tf in cnt.items():
            k = len(g) - 1
            w = float(tf) * (log_n - math.log(max(1.0, df.get(g, 0.0))))
            v[k][g] = w
            norm[k] += w * w
            if k == 1:
                length += tf               # the reference scorer counts BIGRAMS here (cider_scorer.py:125-126)
        return v, [math.sqrt(x) for x in norm], length
    per = []
    for h, rs in zip(hyp, refs):
        vh, nh, lh = vec(h)
        tot = [0.0] * n
        for r in rs:
            vr, nr, lr = vec(r)
            pen = math.e ** (-(float(lh - lr) ** 2) / (2 * sigma ** 2))
            for k in range(n):
                s = sum(min(w, vr[k].get(g, 0.0)) * vr[k].get(g, 0.0) for g, w in vh[k].items())
                if nh[k] != 0 and nr[k] != 0:
                    s /= nh[k] * nr[k]
                tot[k] += s * pen
        per.append(sum(tot) / n / len(rs) * 10.0)
    return sum(per) / len(per), per


# ------------------------------------------------------------------------------------------------ driver (evaluate.py)
class CaptionScorer(object):
    """COCOScorer.score without the Java parts (caption-eval/cocoeval.py:52-103): GT / RES as built by
    convert_data_to_coco_scorer_format / convert_prediction.  Returns ({'Bleu_1'..'Bleu_4','ROUGE_L','CIDEr'}, None)."""

    def score(self, GT, RES, IDs):
        gts = {i: [tokenize(c['caption']) for c in GT[i]] for i in IDs}
        res = {i: [tokenize(c['caption']) for c in RES[i]] for i in IDs}
        out = {}
        b, _ = bleu(gts, res, 4)
        for k in range(4):
            out['Bleu_%d' % (k + 1)] = b[k]
        out['ROUGE_L'] = rouge_l(gts, res)[0]
        out['CIDEr'] = cider(gts, res)[0]
        self.eval = out
        return out, None


def convert_data_to_coco_scorer_format(reference):
    """evaluate.py:16-39: `vid<TAB>sentence` lines -> {vid: [{'video_id', 'cap_id', 'caption'}]}, non-ASCII characters dropped"""
    ref = {}
    with open(reference, 'r') as f:
        for line in f:
            parts = line.split('\t')
            vid, sent = parts[0], parts[1].strip().encode('ascii', 'ignore').decode('ascii')
            ref.setdefault(vid, [])
            ref[vid].append({u'video_id': vid, u'cap_id': len(ref[vid]), u'caption': sent})
    return ref


def convert_prediction(prediction):
    """evaluate.py:50-54"""
    return {str(k): [{u'video_id': str(k), u'caption': v}] for k, v in prediction.items()}


def gather_results(net, eval_loader):
    """evaluate.py:62-78 / 101-117: greedy or beam inference over the eval loader -> OrderedDict video id -> sentence"""
    import torch
    result = collections.OrderedDict()
    dec = (net.module if hasattr(net, 'module') else net).decoder
    with torch.no_grad():
        for frames, regions, spatials, video_ids in eval_loader:
            outputs = net(frames, regions, None)[0]
            ids = outputs.cpu()                                    # host synchronisation: the time-out word is final too
            chk = getattr(getattr(net.module if hasattr(net, 'module') else net, 'ops', None), 'check_persistent', None)
            if chk is not None:
                chk()                                              # a timed-out persistent launch must not pass as captions
            for tokens, vid in zip(ids, video_ids):
                result[vid] = dec.decode_tokens(tokens)
    return result


def merge_rank_results(result, process_group=None):
    """run_gun.py:270-276: every rank decodes its partition of the test clips (`EvalLoader(world_size, rank)`), the per-rank
    caption dicts are exchanged with `all_gather_object` and merged in rank order.  The reference hard-codes four ranks
    (`[None for _ in range(4)]`, `{**r[0], **r[1], **r[2], **r[3]}`); here it is the group's size.  Every rank gets the
    merged dict (the reference scores it on rank 0 only)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return result
    parts = [None] * dist.get_world_size(process_group)
    dist.all_gather_object(parts, result, group=process_group)
    merged = collections.OrderedDict()
    for part in parts:
        merged.update(part)
    return merged


def evaluate(net, eval_loader, reference, process_group=None, gather=True):
    """evaluate.py:56-98 -> (scores, result).  `reference`: dict from convert_data_to_coco_scorer_format.  Inside an
    initialised process group (several GPUs, each with its own partition of the clips in `eval_loader`) the ranks' results are
    merged first (`evaluate_multi_gpu` of evaluate.py:120-134 on the gathered dict, run_gun.py:268-281), so the scores cover
    the whole test set on every rank; gather=False scores this rank's partition only."""
    result = gather_results(net, eval_loader)
    if gather:
        result = merge_rank_results(result, process_group)
    pred = convert_prediction(result)
    scores, _ = CaptionScorer().score(reference, pred, list(pred.keys()))
    return scores, result
