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


def test_device_feed_prefetches_in_order_and_shards():
    """DeviceFeed (the staged version of batch_iterator.py:97-184's .cuda() calls): same batches, same order, this rank's share;
    errors of the producer surface in the consumer.  On a CPU device it is a prefetching pass-through."""
    import torch
    from cliora_amd.data import DeviceFeed, partition, synthetic_batches
    lengths = [5] * 9 + [7] * 6 + [3] * 4
    want = list(synthetic_batches(50, lengths, 3, seed=5, k_neg=7))
    got = list(DeviceFeed(synthetic_batches(50, lengths, 3, seed=5, k_neg=7), 'cpu', depth=2))
    assert len(got) == len(want) and len(want) > 3
    for a, b in zip(got, want):
        assert a['example_ids'] == b['example_ids'] and torch.equal(a['sentences'], b['sentences']) and torch.equal(a['neg_samples'], b['neg_samples'])
    for rank in range(2):
        got = list(DeviceFeed(synthetic_batches(50, lengths, 4, seed=5, k_neg=7), 'cpu', depth=3, rank=rank, world=2))
        for a, b in zip(got, synthetic_batches(50, lengths, 4, seed=5, k_neg=7)):
            assert torch.equal(a['sentences'], partition(b['sentences'], rank, 2))
            assert a['example_ids'] == partition(b['example_ids'], rank, 2)

    def broken():
        yield want[0]
        raise RuntimeError('reader failed')
    feed = DeviceFeed(broken(), 'cpu')
    next(feed)
    try:
        next(feed)
        assert False, 'the producer error must surface'
    except RuntimeError as e:
        assert 'reader failed' in str(e)


def test_device_feed_closes_when_the_consumer_stops_early():
    """A consumer that leaves the loop early (max_step, an exception in the step) must not leave the producer blocked on a full
    queue holding staged batches: close() (or the context manager) stops and joins it."""
    import threading
    from cliora_amd.data import DeviceFeed, synthetic_batches
    lengths = np.random.RandomState(1).randint(3, 9, size=400)
    before = threading.active_count()
    with DeviceFeed(synthetic_batches(50, lengths, 4, seed=5, k_neg=7), 'cpu', depth=2) as feed:
        first = next(feed)
        assert 'sentences' in first
    assert not feed._thread.is_alive()
    assert threading.active_count() <= before
    # nested dicts are walked when the buffers are tied to the consumer's stream (CPU: a no-op that must not raise)
    DeviceFeed._record({'a': torch.zeros(2), 'b': {'c': torch.zeros(1)}}, None)
