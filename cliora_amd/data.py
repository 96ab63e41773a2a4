"""Length-bucketed batching and rank sharding: the callers' side of the chart path.

The chart kernels need every sentence of a batch to have the same length; the reference gets that
from ``FixedLengthBatchSampler`` (cliora/data/dataloader.py:11-113: group example ids by exact
length, shuffle inside each group, emit whole batches in a shuffled order of lengths, optionally
the partial remainders, batch size optionally decreasing with length) and shards a batch over the
GPUs by ``torch.chunk`` on every field (cliora/data/batch_iterator.py:53-66, 134-136).  This module
restates both for synthetic / in-memory corpora; given the same ``numpy.random.RandomState`` it
yields the same batches as the reference (tests/golden/sampler_batches.npz).
"""
import numpy as np
import torch


class LengthBucketSampler(object):
    def __init__(self, lengths, batch_size, include_partial=False, rng=None, maxlen=None, length_to_size=None):
        self.lengths = [int(n) for n in lengths]
        self.batch_size = batch_size
        self.include_partial = include_partial
        self.rng = rng if rng is not None else np.random.RandomState(seed=11)
        self.maxlen = maxlen
        self.length_to_size = length_to_size
        self._plan_epoch()

    def batch_size_for(self, length):
        """Batch size in force at `length`: the last threshold <= length wins (dataloader.py:27-38)."""
        if not self.length_to_size:
            return self.batch_size
        size = self.batch_size
        for n in sorted(self.length_to_size):
            if n <= length:
                size = self.length_to_size[n]
        return size

    def _plan_epoch(self):
        buckets = {}                                   # insertion order = first appearance, as in the reference
        for i, n in enumerate(self.lengths):
            if self.maxlen is not None and self.maxlen > 0 and n > self.maxlen:
                continue
            buckets.setdefault(n, []).append(i)
        for n in buckets:
            self.rng.shuffle(buckets[n])
        order, partial = [], []
        for n, ids in buckets.items():
            bs = self.batch_size_for(n)
            full = len(ids) // bs
            order += [n] * full
            if full * bs < len(ids):
                partial.append(n)
        if self.include_partial:
            order += partial
        self.rng.shuffle(order)
        self._buckets, self._order = buckets, order

    def __len__(self):
        return len(self._order)

    def __iter__(self):
        self._plan_epoch() if getattr(self, '_used', False) else None
        self._used = True
        taken = {n: 0 for n in self._buckets}
        for n in self._order:
            bs = self.batch_size_for(n)
            start = taken[n] * bs
            taken[n] += 1
            yield self._buckets[n][start:start + bs]


def partition(value, rank, world):
    """One rank's share of a batch field: tensors by torch.chunk, sequences by the same index split,
    dicts field by field (batch_iterator.py:53-66)."""
    if value is None:
        return None
    if isinstance(value, dict):
        return {k: partition(v, rank, world) for k, v in value.items()}
    if isinstance(value, torch.Tensor):
        return torch.chunk(value, world, dim=0)[rank]
    idx = torch.chunk(torch.arange(len(value)), world, dim=0)[rank]
    return [value[int(i)] for i in idx]


def synthetic_batches(vocab, lengths, batch_size, seed=1234, k_neg=100, device='cpu', **sampler_kw):
    """Batch maps with the keys Trainer.run_net consumes (trainer.py:437-448) for a synthetic corpus."""
    gen = torch.Generator().manual_seed(seed)
    corpus = [torch.randint(0, vocab, (int(n),), generator=gen) for n in lengths]
    sampler = LengthBucketSampler(lengths, batch_size, rng=np.random.RandomState(seed), **sampler_kw)
    for ids in sampler:
        sents = torch.stack([corpus[i] for i in ids]).to(device)
        neg = torch.randperm(vocab, generator=gen)[:k_neg].to(device)      # without replacement: negative_sampler.py:37
        yield dict(example_ids=list(ids), sentences=sents, neg_samples=neg, batch_size=len(ids), length=sents.shape[1])
