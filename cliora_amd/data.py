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


class DeviceFeed(object):
    """Batches staged on the device ahead of the step that consumes them.

    The reference collates on DataLoader workers with ``pin_memory`` and moves each field with ``.cuda()`` inside the step
    loop (cliora/data/batch_iterator.py:97-184): the copy is synchronous with the step.  Here a background thread takes batch
    maps from any iterator, keeps this rank's share (``partition``, batch_iterator.py:134-136), pins the tensors and
    issues the host-to-device copies on a copy stream of its own, ``depth`` batches ahead; ``__next__`` only makes the
    caller's stream wait for the batch's copy event.  The region features of CLIORA (36 x 2048 floats per sentence, 18.9 MB
    per 64-sentence batch) are what makes this worth it; for text-only DIORA a batch is a few kB.

    Without a GPU (``device`` a CPU device) the feed degrades to a prefetching pass-through, which is what the CPU tests run."""

    _END = object()

    def __init__(self, batches, device, depth=2, rank=0, world=1):
        import queue
        import threading
        self.device = torch.device(device)
        self.rank, self.world = rank, world
        self._q = queue.Queue(maxsize=max(1, depth))
        self._cuda = self.device.type == 'cuda'
        self._copy_stream = torch.cuda.Stream(self.device) if self._cuda else None
        self._stop = threading.Event()
        self._queue_mod = queue
        self._thread = threading.Thread(target=self._produce, args=(iter(batches),), daemon=True)
        self._thread.start()

    def _put(self, item):
        """Blocking put that gives up when the consumer has closed the feed (a `break` at max_step, an exception in the step)."""
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except self._queue_mod.Full:
                continue
        return False

    def close(self):
        """Stop the producer and drop the staged batches (each holds pinned and device memory): call when leaving the loop early."""
        self._stop.set()
        try:
            while True:
                self._q.get_nowait()
        except self._queue_mod.Empty:
            pass
        if self._thread.is_alive() and self._thread is not __import__('threading').current_thread():
            self._thread.join(timeout=5.0)

    def __del__(self):
        try:
            self._stop.set()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _stage(self, v):
        if isinstance(v, dict):
            return {k: self._stage(x) for k, x in v.items()}
        if isinstance(v, torch.Tensor):
            if self._cuda:
                return v.pin_memory().to(self.device, non_blocking=True)
            return v.to(self.device)
        return v

    def _produce(self, it):
        try:
            for batch in it:
                if self.world > 1:     # every per-example field; the negatives are drawn after the split in the reference and stay whole
                    batch = {k: (partition(v, self.rank, self.world) if k != 'neg_samples' and isinstance(v, (torch.Tensor, list, tuple, dict)) else v)
                             for k, v in batch.items()}
                    if 'batch_size' in batch and isinstance(batch.get('sentences'), torch.Tensor):
                        batch['batch_size'] = int(batch['sentences'].shape[0])
                if self._cuda:
                    with torch.cuda.device(self.device), torch.cuda.stream(self._copy_stream):
                        staged = self._stage(batch)
                        ev = torch.cuda.Event()
                        ev.record(self._copy_stream)
                else:
                    staged, ev = self._stage(batch), None
                if not self._put((staged, ev)):
                    return
            self._put((self._END, None))
        except BaseException as e:                   # surfaces in the consumer
            self._put((e, None))

    def __iter__(self):
        return self

    def __next__(self):
        staged, ev = self._q.get()
        if staged is self._END:
            raise StopIteration
        if isinstance(staged, BaseException):
            raise staged
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            self._record(staged, cur)                # the allocator must not reuse the buffers while `cur` still reads them
        return staged

    @staticmethod
    def _record(v, stream):
        """record_stream on every device tensor of a staged batch, nested dicts included (mirrors _stage)."""
        if isinstance(v, dict):
            for x in v.values():
                DeviceFeed._record(x, stream)
        elif isinstance(v, torch.Tensor) and v.is_cuda:
            v.record_stream(stream)
