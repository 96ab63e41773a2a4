"""Bucketed batching and rank partition against batches captured from the reference's
FixedLengthBatchSampler / BatchIterator.partition (tests/golden/sampler_batches.npz)."""
import numpy as np
import torch

from conftest import load_golden
from cliora_amd.data import LengthBucketSampler, partition, synthetic_batches


def test_sampler_reproduces_reference_batches():
    g = load_golden('sampler_batches.npz')
    lengths = g['lengths']
    for ci, c in enumerate(g['meta']['cases']):
        l2s = {int(k): v for k, v in c['length_to_size'].items()} if c['length_to_size'] else None
        s = LengthBucketSampler(lengths, c['batch_size'], include_partial=c['include_partial'],
                                rng=np.random.RandomState(c['seed']), maxlen=c['maxlen'], length_to_size=l2s)
        flat, sizes = [], []
        for _epoch in range(2):
            for b in s:
                assert len({int(lengths[i]) for i in b}) == 1          # every batch has ONE length: no padding anywhere
                flat += list(b)
                sizes.append(len(b))
        assert sizes == g['case%d_sizes' % ci].tolist(), ci
        assert flat == g['case%d_flat' % ci].tolist(), ci


def test_partition_matches_reference():
    g = load_golden('sampler_batches.npz')
    t = torch.arange(22).view(11, 2)
    lst = list(range(100, 111))
    for world in (2, 4):
        for rank in range(world):
            assert np.array_equal(partition(t, rank, world).numpy(), g['part_t_%d_%d' % (world, rank)])
            assert partition(lst, rank, world) == g['part_l_%d_%d' % (world, rank)].tolist()
    d = partition(dict(a=t, b=lst, c=None), 1, 2)
    assert d['c'] is None and d['b'] == lst[6:] and torch.equal(d['a'], t[6:])


def test_synthetic_batches_shapes():
    n = 0
    for bm in synthetic_batches(100, [5] * 9 + [7] * 8, 4, seed=1, k_neg=10):
        assert bm['sentences'].shape == (4, bm['length']) and bm['neg_samples'].shape == (10,)
        assert len(set(bm['neg_samples'].tolist())) == 10
        n += 1
    assert n == 4
