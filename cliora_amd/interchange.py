"""On-disk formats either side of the chart path and the span bookkeeping of evaluation (row f2).

* Checkpoints.  The reference saves ``{'state_dict': net.state_dict()}`` with the embedding table optionally left out
  (``Trainer.save_model``, cliora/net/trainer.py:383-397) and loads it leniently (``Trainer.load_model``, :399-435):
  ``module.`` prefixes of DistributedDataParallel are stripped, keys the net does not have are dropped, missing keys keep
  the net's own initialisation, ``*_vis`` parameters fall back to their text twin, and the embedding table is only taken
  from the file when ``origin_emb`` is set.  ``cliora_amd.harness.Net`` has the reference's parameter names, so a file
  written by either side loads on the other (tests/test_interchange.py, checkpoints written by the reference).
* Trees.  Nested tuples of token positions, as ``ParsePredictor`` / ``cliora_cky_decode`` produce them.  ``tree_spans`` is
  the span list the reference gets from ``get_spans(get_actions(str(tree)))`` (cliora/analysis/utils.py:3-49): one
  ``(first, last)`` pair per internal node in reduce order, the root last.  ``SpanF1`` accumulates the corpus- and
  sentence-level F1 of cliora/scripts/parse.py:215-234, 286-290; ``parse_record`` is one line of ``parse.jsonl`` (:270-280).
"""
import collections
import json

import torch

PUNCTUATION = frozenset(w.lower() for w in ('.', ',', ':', '-LRB-', '-RRB-', "''", '``', '--', ';', '-', '?', '!', '...', '-LCB-', '-RCB-'))


# ---- checkpoints -------------------------------------------------------------------------------------------------
def _single(net):
    return net.module if isinstance(net, torch.nn.parallel.DistributedDataParallel) else net


def save_model(net, save_emb, path):
    state = net.state_dict()
    if not save_emb:
        state = collections.OrderedDict((k, v) for k, v in state.items() if 'embeddings' not in k)
    torch.save({'state_dict': state}, path)


def load_model(origin_emb, net, path, verbose=False):
    """Returns (taken, kept): the parameter names filled from the file and those left at the net's own values."""
    target = _single(net)
    have = target.state_dict()
    saved = torch.load(path, map_location='cpu')['state_dict']
    loaded = collections.OrderedDict()
    for k, v in saved.items():
        name = k[len('module.'):] if k.startswith('module.') else k
        if name in have:
            loaded[name] = v
        elif verbose:
            print('deleting {}'.format(name))
    in_file = set(loaded)
    taken, kept = [], []
    for k in have:
        if 'embeddings' in k and not origin_emb:
            loaded[k] = have[k]
            kept.append(k)
        elif k not in in_file:
            twin = k.replace('_vis', '')
            if '_vis' in k and 'img_encoder' not in k and twin in loaded:
                loaded[k] = loaded[twin]
                taken.append(k)
            else:
                if verbose:
                    print('Not initialize {}'.format(k))
                loaded[k] = have[k]
                kept.append(k)
        else:
            taken.append(k)
    target.load_state_dict(loaded)
    return taken, kept


# ---- trees ---------------------------------------------------------------------------------------------------------
def _is_leaf(node):
    return not isinstance(node, (list, tuple))


def flatten_tree(tree):
    out, stack = [], [tree]
    while stack:
        node = stack.pop()
        if _is_leaf(node):
            out.append(node)
        else:
            stack.extend(reversed(node))
    return out


def tree_spans(tree):
    """(first, last) leaf positions of every internal node, children before parents, left before right."""
    spans = []

    def walk(node, start):
        if _is_leaf(node):
            return 1
        width = 0
        for child in node:
            width += walk(child, start + width)
        spans.append((start, start + width - 1))
        return width

    walk(tree, 0)
    return spans


def replace_leaves(tree, leaves):
    it = iter(leaves)

    def walk(node):
        return next(it) if _is_leaf(node) else [walk(child) for child in node]

    return walk(tree)


def drop_leaves(tree, keep):
    """The tree without the leaves whose `keep` flag is False; chains of single children collapse."""
    pos = [0]

    def walk(node):
        if _is_leaf(node):
            k = keep[pos[0]]
            pos[0] += 1
            return node if k else None
        kids = [c for c in (walk(child) for child in node) if c is not None]
        if not kids:
            return None
        return kids[0] if len(kids) == 1 else kids

    return walk(tree)


def postprocess(tree, tokens=None):
    """Re-attach a sentence-final punctuation mark at the top: (rest, '.')   (parse.py:63-79)."""
    tokens = flatten_tree(tree) if tokens is None else tokens
    if tokens[-1].lower() not in PUNCTUATION:
        return tree
    rest = drop_leaves(tree, [True] * (len(tokens) - 1) + [False])
    assert rest is not None, 'No tokens left. Original = {}'.format(tokens)
    return (rest, tokens[-1])


class SpanF1(object):
    """Corpus-level (pooled tp/fp/fn) and mean sentence-level F1 over unlabeled spans, the root span excluded."""

    def __init__(self):
        self.tp = self.fp = self.fn = 0
        self.sentence = []

    def add(self, tree, gold_spans):
        pred = set(tree_spans(tree)[:-1])
        gold = set(gold_spans)
        hit = len(pred & gold)
        self.tp += hit
        self.fp += len(pred) - hit
        self.fn += len(gold) - hit
        prec = float(hit) / (len(pred) + 1e-8)
        reca = float(hit) / (len(gold) + 1e-8)
        if not gold:
            reca = 1.
            if not pred:
                prec = 1.
        self.sentence.append(2 * prec * reca / (prec + reca + 1e-8))
        return pred

    @property
    def corpus_f1(self):
        prec = self.tp / (self.tp + self.fp) if self.tp + self.fp else 0.
        reca = self.tp / (self.tp + self.fn) if self.tp + self.fn else 0.
        return 2 * prec * reca / (prec + reca) if prec + reca > 0 else 0.

    @property
    def sentence_f1(self):
        return sum(self.sentence) / len(self.sentence) if self.sentence else 0.


def parse_record(example_id, tree, sentence, gold_spans=(), pred_spans=(), pred_boxes=(), post=False):
    """One line of parse.jsonl: same keys, order and value shapes as the reference's writer."""
    words = replace_leaves(tree, sentence)
    if post:
        words = postprocess(words, sentence)
    rec = collections.OrderedDict(example_id=str(example_id), tree=words, tree_index_conll=tree, sentence=list(sentence),
                                  gold_spans=list(gold_spans), pred_spans=list(pred_spans), pred_boxes=list(pred_boxes))
    return json.dumps(rec)
