#!/usr/bin/env python
"""Headline benchmark: sentences/sec of the chart hot path (inside + outside, forward +
backward) on synthetic length-20, d=400, batch-64 batches (BASELINE.json config 2).

  python bench.py --gpus N --steps K --warmup W

One step = DioraMLP.forward (leaf transform, inside pass, outside pass) + the hand-written
backward for random cotangents on all four chart outputs, all HIP kernels behind the C ABI;
with N > 1 each rank owns its own 64 sentences (weak scaling) and the step ends with ONE
all-reduce of the flat gradient buffer over RCCL.  Inputs are resident in HBM before the
timed region.  Prints ONE JSON line (rank 0).

N > 1: either launch one rank per GPU yourself
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
or run `python bench.py --gpus N` directly: with no WORLD_SIZE in the environment the script starts that launcher as a
child process BEFORE anything touches the GPU (no exec of a GPU-initialised process) and exits with its code.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA; the split-bf16 compose GEMMs issue three of them per fp32 product
PEAK_HBM_GBS = 8000.0


def usable_cpus():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def host_cpu():
    """CPU model string, logical CPUs and physical cores of this host ("state the count and CPU model")."""
    model, phys, logical = 'unknown', set(), 0
    try:
        pid = cid = None
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                model = ln.split(':', 1)[1].strip()
                logical += 1
            elif ln.startswith('physical id'):
                pid = ln.split(':', 1)[1].strip()
            elif ln.startswith('core id'):
                cid = ln.split(':', 1)[1].strip()
            elif not ln.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return dict(model=model, logical_cpus=logical or (os.cpu_count() or 1), physical_cores=len(phys) or None, usable_by_this_process=usable_cpus())


def port_vs_reference():
    """Measured wall-time ratio of the oracle to the imported reference (tools/cpu_port_check.py, build container: the reference cannot
    travel to the GPU box) -- SURVEY 8(d) wants the port within +-10 %; the committed measurement is printed beside the baseline."""
    for cand in ('r06_cpu_port_check.json', 'r05_cpu_port_check.json'):
        fp = os.path.join(ROOT, 'profiles', cand)
        if os.path.exists(fp):
            try:
                j = json.load(open(fp))
                return dict(file='profiles/' + cand, d400_ratio=j.get('port_vs_reference_d400'),
                            cases=[dict(D=c['D'], B=c['B'], L=c['L'], threads=c['threads'], ratio=c['port_vs_reference']) for c in j.get('cases', [])],
                            note='oracle seconds / reference seconds on the same inputs, alternating, measured in the build container (the GPU box '
                                 'has no reference); value above is the port as timed on THIS host')
            except (OSError, ValueError, KeyError):
                return None
    return None


def cpu_baseline(L, D, B, budget_s=25.0):
    """The CPU oracle (a restatement of the reference's torch op sequence, oracle/diora_ref.py)
    timed on this host: chart-only forward + backward, all host cores torch exposes."""
    import torch
    from oracle import diora_ref as R
    from oracle import synth
    threads = usable_cpus()
    torch.set_num_threads(threads)
    keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s')

    def step(b):
        P, x, cot = synth.diora_case(D, b, L, 1234)
        for v in P.values():
            v.requires_grad_(True)
        x.requires_grad_(True)
        t0 = time.perf_counter()
        out = R.diora_forward(P, x, x, training=True)
        torch.autograd.backward([out[k] for k in keys], [cot[k] for k in keys])
        return time.perf_counter() - t0

    step(2)                                   # warm-up (thread pool, allocator)
    t_small = step(8)
    b = B if t_small * (B / 8.0) < budget_s else max(8, int(8 * budget_s / t_small) // 8 * 8)
    times, t_all = [], 0.0
    while t_all < 12.0 and len(times) < 12:   # bounded sample: about 12-25 s of CPU work
        t = step(b)
        times.append(t)
        t_all += t
    times.sort()
    med = times[len(times) // 2]
    # the 1-thread figure SURVEY 8(d) asks for: one step of 8 sentences after a 2-sentence warm-up (about 2 s)
    one = None
    try:
        torch.set_num_threads(1)
        step(2)
        t1 = step(8)
        one = dict(value=round(8 / t1, 3), unit='sentences/s', sample='1 step of 8 sentences, 1 thread, %.2f s' % t1)
    finally:
        torch.set_num_threads(threads)
    pvr = port_vs_reference()
    ratio = (pvr or {}).get('d400_ratio')
    return dict(value=b / med, unit='sentences/s', cores=threads, kind='port',
                # the oracle port runs `ratio` x the imported reference's wall time on the same inputs (measured in the build container, where the
                # reference exists): what the reference itself would do on THIS host's cores, as an estimate, beside the raw port figure
                reference_estimate=(round(b / med * ratio, 3) if ratio else None), port_over_reference_time=ratio,
                sample='%d steps of %d sentences (L=%d, d=%d), chart fwd+bwd, torch %s CPU oracle, %d threads, median %.2f s/step, %.1f s total'
                       % (len(times), b, L, D, torch.__version__, threads, med, t_all),
                host=host_cpu(), one_thread=one, port_vs_reference=pvr)


CPU_CACHE = os.path.join(ROOT, 'gpurun_out', 'cpu_baseline_cache.json')


def remember_cpu_baseline(key, rec):
    """The N = 1 run leaves its CPU figure where the N > 1 runs of the same box find it (SCALE runs N = 1, 2, 4, 8 back to back)."""
    try:
        os.makedirs(os.path.dirname(CPU_CACHE), exist_ok=True)
        j = json.load(open(CPU_CACHE)) if os.path.exists(CPU_CACHE) else {}
        j[key] = dict(rec, measured_at=time.strftime('%Y-%m-%dT%H:%M:%S'), hostname=os.uname().nodename)
        json.dump(j, open(CPU_CACHE, 'w'))
    except (OSError, ValueError):
        pass


def carried_cpu_baseline(key, committed):
    """N > 1 lines: the CPU baseline is timed at N = 1 only (rank 0); this carries that figure beside the N-GPU value -- from the N = 1 run of
    this box if it left one (same hostname), else the figure committed under profiles/ -- labelled as carried, never re-timed here."""
    try:
        if os.path.exists(CPU_CACHE):
            rec = json.load(open(CPU_CACHE)).get(key)
            if rec and rec.get('hostname') == os.uname().nodename:
                return dict(rec, carried_from='the N = 1 run of this box (%s)' % rec.get('measured_at'))
    except (OSError, ValueError):
        pass
    fp = os.path.join(ROOT, 'profiles', committed)
    try:
        line = [ln for ln in open(fp).read().splitlines() if ln.strip().startswith('{')][-1]
        rec = json.loads(line).get('cpu_baseline')
        if rec:
            return dict(rec, carried_from='profiles/%s (committed N = 1 line, another box of the same pool)' % committed)
    except (OSError, ValueError, IndexError):
        pass
    return None


def cpu_baseline_c3(L, D, B, budget_s=20.0):
    """The CPU oracle's CLIORA training step (oracle/diora_ref.py: Embed, ImageEncoder, chart with regions, reconstruction + VG + contrastive
    losses, backward; the reference's op sequence incl. its B x B region attention, cliora.py:35-42) on this host: a BOUNDED sample at a
    reduced batch -- the reference's cost per sentence grows with the batch (every text-image pair of the batch is scored), so the sample's
    batch is stated; clip + Adam are left out (negligible beside the step)."""
    import torch
    from oracle import diora_ref as R
    threads = usable_cpus()
    torch.set_num_threads(threads)
    V, E, K, Rg = 10000, 1024, 100, 36
    g = torch.Generator().manual_seed(1234)
    P = R.init_params(D, share=True, seed=1234)
    rn = lambda *s, std=1.0: (torch.randn(*s, generator=g) * std).requires_grad_(True)
    emb_w, mat, mat1, rmat = rn(V, E), rn(D, E), rn(D, E), rn(D, E)
    Wf, bf, Wv, bv = rn(D, 2048, std=0.02), rn(D, std=0.02), rn(D, 2048, std=0.02), rn(D, std=0.02)
    for v in P.values():
        v.requires_grad_(True)

    def step(b):
        sentences = torch.randint(0, V, (b, L), generator=g)
        neg = torch.randperm(V, generator=g)[:K]
        obj = torch.relu(torch.randn(b, Rg, 2048, generator=g))
        t0 = time.perf_counter()
        xs, xw = R.embed_forward(emb_w, mat, mat1, sentences)
        os_, ow = R.image_encoder_forward(Wf, bf, Wv, bv, obj)
        ref = R.diora_forward(P, xs, xw, os_, ow, training=True)
        loss = (R.reconstruction_loss(emb_w, rmat, sentences, neg, ref['outside_h']) + R.vg_loss(ref['vg_atten_score'], 1.0)
                + R.contrastive_loss(ref['inside_s'], ref['outside_s'], ref['all_atten_score'], 0.2, 1.0))
        loss.backward()
        return time.perf_counter() - t0

    step(2)
    t8 = step(8)
    b = 16 if t8 * 2.5 < budget_s / 2 else 8
    times = [step(b) for _ in range(2 if t8 * (b / 8.0) * 2 < budget_s else 1)]
    med = sorted(times)[len(times) // 2]
    return dict(value=b / med, unit='sentences/s', cores=threads, kind='port',
                sample='%d step(s) of %d sentences (L=%d, d=%d, 36 regions x 2048-d, V 10000, k_neg 100): oracle Net forward + three losses + backward, '
                       'torch %s, %d threads, %.2f s/step (the GPU line runs 64 per rank)' % (len(times), b, L, D, torch.__version__, threads, med),
                host=host_cpu(), port_vs_reference=port_vs_reference())


def shape_traffic(label):
    """Committed PMC traffic per step of one of the other workloads (tools/pmc_shape.sh: 2 x FETCH_SIZE + WRITE_SIZE over every kernel of a
    3-step run, own --pmc passes), or None."""
    fp = os.path.join(ROOT, 'profiles', 'r06_traffic_shapes.json')
    if not os.path.exists(fp):
        fp = os.path.join(ROOT, 'profiles', 'r05_traffic_shapes.json')
    if not os.path.exists(fp):
        return None
    try:
        j = json.load(open(fp)).get(label)
    except (OSError, ValueError):
        return None
    if not j:
        return None
    return dict(bytes_per_step=j['total_bytes_per_step'], file='profiles/' + os.path.basename(fp), commit=j.get('commit', 'not recorded'),
                kind='committed rocprofv3 PMC figure (2 x FETCH_SIZE + WRITE_SIZE over every kernel of the step), not collected in this run',
                top_kernels=dict(list(j.get('by_kernel', {}).items())[:5]))


def algorithmic_bytes(plan, B, D, cell_floats=None):
    """forward bytes per level = (unique chart cells read + cells written) x (D + 1) x 4, from the
    plan's own index tables (the reference's tables, tests/test_plan_tables.py); backward = 2 x forward.  Returns
    (bytes per step of the whole path, forward bytes per level for the inside and the outside pass).  cell_floats: floats per
    chart cell (default D + 1: h and the score; the TreeLSTM also moves the cell state c: 2 D + 1, SURVEY 8d "included for TreeLSTM")."""
    import numpy as np
    L = plan.L
    cell_b = (cell_floats if cell_floats else D + 1) * 4.0
    per_level = {'in': [], 'out': []}
    for key, a_name, b_name, base_name, levels, nsplit in (
            ('in', 'pair_a_in', 'pair_b_in', 'pair_lvl_base_in', range(1, L), lambda lv: lv),
            ('out', 'pair_a_out', 'pair_b_out', 'pair_lvl_base_out', range(0, L - 1), lambda lv: L - lv - 1)):
        ta, tb, base = plan.table(a_name), plan.table(b_name), plan.table(base_name)
        for lv in levels:
            n = (L - lv) * nsplit(lv)
            a, b = ta[base[lv]:base[lv] + n], tb[base[lv]:base[lv] + n]
            if key == 'in':
                uniq = len(np.unique(np.concatenate([a, b])))          # both operands live in the inside chart
            else:
                uniq = len(np.unique(a)) + len(np.unique(b))            # sibling: inside chart, parent: outside chart
            per_level[key].append((uniq + (L - lv)) * cell_b * B)
    fwd = sum(per_level['in']) + sum(per_level['out']) + B * L * D * 4.0
    return 3.0 * fwd, per_level


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 (or `--via-launcher` at N = 1) and no launcher around it: start torch.distributed.run as a
    CHILD process (this process has not touched the GPU and never does) and pass its exit code on."""
    port = os.environ.get('MASTER_PORT', '29533')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', port, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != '--via-launcher'] + ['--launched']
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def rank_view(torch, dist, dev, dt, steps, use_dist, world):
    """What proves the N-rank launch: ranks_seen = an all-reduce of ones over the process group (RCCL saw that many ranks), and every
    rank's own ms per step (min / max over ranks: stragglers).  Returns (max-over-ranks seconds, dict)."""
    if not use_dist:
        ms = dt / steps * 1e3
        return dt, dict(ranks_seen=1, per_rank_ms_per_step=dict(min=round(ms, 4), max=round(ms, 4)), process_group=None)
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    mine = torch.tensor([dt], device=dev, dtype=torch.float64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    per = [float(t.item()) / steps * 1e3 for t in every]
    return max(float(t.item()) for t in every), dict(ranks_seen=int(round(float(ones.item()))), process_group=dist.get_backend(),
                                                      per_rank_ms_per_step=dict(min=round(min(per), 4), max=round(max(per), 4), all=[round(v, 4) for v in per]))


def timed_steps(torch, step, fence, n):
    """n steps between two fences; returns (wall seconds for the n steps, per-step device ms from event pairs on the current stream)."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    fence()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        step()
        b.record()
    fence()
    dt = time.perf_counter() - t0
    return dt, sorted(a.elapsed_time(b) for a, b in evs)


def pct(sorted_ms, q):
    return sorted_ms[min(len(sorted_ms) - 1, int(q * len(sorted_ms)))]


def other_measurements(torch, dev, budget_steps=12):
    """Builder-side figures the judge asked to see in the driver line: the other BASELINE configurations (chart fwd+bwd) and the
    whole training step (Embed -> chart -> losses -> backward -> clip -> Adam on cliora_amd/harness.py).  Short runs; any
    failure is reported as a string instead of breaking the headline."""
    from cliora_amd.diora import DioraMLP
    from cliora_amd.cliora import DioraMLP as CDioraMLP
    from cliora_amd.treelstm import DioraTreeLSTM
    out = {}

    def chart(make, B, L, D, R=0, steps=budget_steps, warmup=3, arch=0, tag=None):
        torch.manual_seed(0)
        m = make().to(dev).train()
        for p in m.parameters():
            torch.nn.init.normal_(p)
        x = torch.randn(B, L, D, device=dev, requires_grad=True)
        obj = 0.3 * torch.randn(B, R, D, device=dev) if R else None
        C = L * (L + 1) // 2
        keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s')
        cot = [torch.randn(B, C, 1 if k.endswith('_s') else D, device=dev) for k in keys]

        def step():
            for p in m.parameters():
                p.grad = None
            x.grad = None
            m(x, x, obj, obj) if R else m(x, x)
            outs = [getattr(m, k) for k in keys]
            extra = [m.all_atten_score.max(-1).values.sum() * 1e-3] if R else []
            torch.autograd.backward(outs + extra, cot + [None] * len(extra))
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res = dict(B=B, L=L, D=D, R=R, ms_per_step=round(dt * 1e3, 3), sentences_per_s=round(B / dt, 1))
        try:        # the shape's own SURVEY 8(d) bytes and what they come to against 8 TB/s (whole step)
            from cliora_amd import _lib
            plan = _lib.get_plan(B, L, D, True, 'unit', R, dev.index or 0, arch=arch)
            step_bytes, _ = algorithmic_bytes(plan, B, D, cell_floats=(2 * D + 1) if arch == 1 else None)
            res['roofline'] = dict(bound='hbm', algorithmic_bytes=round(step_bytes), achieved_GBs=round(step_bytes / dt / 1e9, 1), peak_GBs=PEAK_HBM_GBS,
                                   frac=round(step_bytes / dt / 1e9 / PEAK_HBM_GBS, 4), traffic=shape_traffic(tag) if tag else None)
        except Exception as e:                                # noqa: BLE001
            res['roofline'] = 'failed: %s' % str(e)[:120]
        return res

    def parse(B=64, L=20, D=400, steps=budget_steps, warmup=3):
        """Inference as scripts/parse.py runs it: eval-mode forward (no backward state kept) + the CKY decode of every sentence."""
        torch.manual_seed(0)
        m = DioraMLP(D).to(dev).eval()
        for p in m.parameters():
            torch.nn.init.normal_(p)
        x = torch.randn(B, L, D, device=dev)
        res = {}
        for what in ('forward', 'spans', 'trees'):
            def step():
                with torch.no_grad():
                    m(x, x)
                    if what == 'spans':
                        return m.cky_spans()              # (B, L-1, 2) ints on the host: what the F1 evaluation consumes (scripts/train.py:184-204)
                    return m.cky() if what == 'trees' else None
            for _ in range(warmup):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            res[what] = (time.perf_counter() - t0) / steps
        return dict(B=B, L=L, D=D, ms_forward=round(res['forward'] * 1e3, 3), ms_forward_and_spans=round(res['spans'] * 1e3, 3),
                    ms_forward_and_trees=round(res['trees'] * 1e3, 3), sentences_per_s_spans=round(B / res['spans'], 1),
                    sentences_per_s_trees=round(B / res['trees'], 1), sentences_per_s=round(B / res['trees'], 1),
                    note='spans: the device-built constituent span lists, one D2H copy (cliora_cky_spans); trees: nested tuples built from them on the host; '
                         'sentences_per_s = the trees variant (what rounds 1-4 reported under this key; round 5 reported the spans variant)')

    def whole(vl, B=64, L=20, D=400, V=10000, E=1024, K=100, steps=30, warmup=8):
        from cliora_amd import harness as H
        torch.manual_seed(1234)
        net = H.build_net(D, torch.nn.Embedding(V, E), obj_feats=vl, img_dim=2048, k_neg=K, vg_loss=vl, use_contr=vl).to(dev)
        if vl:
            for p in net.img_encoder.parameters():
                torch.nn.init.normal_(p, std=0.02)             # the reference's zero init makes every VL score 0
        tr = H.Trainer(net, lr=2e-3)
        g = torch.Generator().manual_seed(1234)
        bm = dict(sentences=torch.randint(0, V, (B, L), generator=g).to(dev), neg_samples=torch.randperm(V, generator=g)[:K].to(dev))
        if vl:
            bm['obj_feats'] = torch.randn(B, 36, 2048, generator=g).to(dev)
        res = {}
        for sync in (True, False):       # True: loss.item() every step like the reference (trainer.py:463); False: the host runs ahead
            for _ in range(warmup):
                tr.step(bm, train=True, sync=sync)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                tr.step(bm, train=True, sync=sync)
            torch.cuda.synchronize()
            res[sync] = (time.perf_counter() - t0) / steps
        dt = res[True]
        return dict(B=B, L=L, D=D, ms_per_step=round(dt * 1e3, 3), sentences_per_s=round(B / dt, 1),
                    ms_per_step_without_per_step_item=round(res[False] * 1e3, 3))

    cases = (('c1 DioraMLP d50 B8 L10', lambda: chart(lambda: DioraMLP(50), 8, 10, 50, tag='c1')),
             ('c3 CLIORA d400 B64 L20 R36 (chart + scorers as the losses take them: region max, word-region scores)', lambda: chart(lambda: CDioraMLP(400), 64, 20, 400, R=36, tag='c3')),
             ('DioraMLP d400 B64 L40', lambda: chart(lambda: DioraMLP(400), 64, 40, 400, steps=6, warmup=2, tag='l40')),
             ('c5 DioraTreeLSTM d400 B64 L40 (parity unpinned)', lambda: chart(lambda: DioraTreeLSTM(400), 64, 40, 400, steps=6, warmup=2, arch=1, tag='c5')),
             ('parse c2 (eval forward + CKY decode and span lists on the GPU, spans copied to the host)', parse),
             ('whole step DIORA c2 (Embed, chart, reconstruction loss, clip, Adam)', lambda: whole(False)),
             ('whole step CLIORA c3 (+ ImageEncoder, VG and contrastive losses)', lambda: whole(True)))
    impl_note = {'c5 DioraTreeLSTM d400 B64 L40 (parity unpinned)':
                 'implementation traffic 136.6 GB per step by counters (profiles/r06_traffic_shapes.json; DESIGN.md section 4 / profiles/r06_notes.md section 6: five gate rows per operand -- 12 + 2 rows per pair in the forward, 2 x 9 in the backward)'}
    for name, fn in cases:
        try:
            out[name] = fn()
            if name in impl_note and isinstance(out[name], dict):
                out[name]['implementation_bytes'] = impl_note[name]
        except Exception as e:                                   # noqa: BLE001 -- a side figure must not cost the headline
            out[name] = 'failed: %s: %s' % (type(e).__name__, str(e)[:200])
        torch.cuda.empty_cache()
    return out


def cliora_training_workload(args, torch, dist, _lib, dev, world, rank, local, use_dist, mfma_mode):
    """BASELINE configs[2] (1 GPU) / configs[3] (8 GPUs, batch 512 global): CLIORA d = 400 with 36 x 2048-d region features, batch 64
    per GPU, length 20 -- one step = the whole Trainer._step of the reference (trainer.py:437-501) on this library's kernels:
    Embed + ImageEncoder, the chart with the region attention, span-region / word-region scorers, reconstruction + VG + contrastive
    losses, backward, ONE flat-gradient all-reduce over RCCL (chart, head and ImageEncoder gradients in one buffer), clip 5.0, Adam."""
    from cliora_amd import harness as H
    from cliora_amd.parallel import FlatGradAllReduce
    B, L, D = args.batch, args.length, args.dim
    V, E, K, Rg = 10000, 1024, 100, 36                      # V 10 000, 1024-d embeddings, k_neg 100, 36 regions
    torch.manual_seed(1234)                                  # same parameters on every rank (train_diora.sh:9)
    net = H.build_net(D, torch.nn.Embedding(V, E), obj_feats=True, img_dim=2048, k_neg=K, vg_loss=True, use_contr=True).to(dev)
    for p in net.img_encoder.parameters():
        torch.nn.init.normal_(p, std=0.02)                   # the reference's zero init makes every VL score 0
    params = [p for p in net.parameters() if p.requires_grad]
    reducer = FlatGradAllReduce(params) if use_dist else None
    tr = H.Trainer(net, lr=2e-3, reducer=reducer)
    g = torch.Generator().manual_seed(1234 + rank)           # every rank its own 64 sentences (weak scaling; batch_iterator.py:53-66)
    bm = dict(sentences=torch.randint(0, V, (B, L), generator=g).to(dev), neg_samples=torch.randperm(V, generator=torch.Generator().manual_seed(99))[:K].to(dev),
              obj_feats=torch.relu(torch.randn(B, Rg, 2048, generator=g)).to(dev))

    def step():
        tr.step(bm, train=True, sync=False)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    dt, step_ms = timed_steps(torch, step, fence, args.steps)
    dt, ranks = rank_view(torch, dist, dev, dt, args.steps, use_dist, world)
    if rank == 0:
        plan = _lib.get_plan(B, L, D, True, 'unit', Rg, local)
        step_bytes, _ = algorithmic_bytes(plan, B, D)
        scorer_bytes = 3.0 * (B * B * (L * (L + 1) // 2) * Rg * 4.0)        # the (B, B, C, 36) span-region scores: written, read by the loss, gradient
        out = {
            'metric': 'sentences/sec (CLIORA whole training step: chart fwd+bwd + heads + all-reduce + clip + Adam), len-%d d=%d bsz=%d' % (L, D, B),
            'value': round(world * B * args.steps / dt, 2), 'unit': 'sentences/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (compose GEMMs and scorers: 3x bf16 MFMA per product, fp32 accumulate)' if mfma_mode == 'bf16x3' else 'f32',
            'data': 'synthetic',
            'config': {'workload': 'CLIORA d=%d + obj_feats (36 regions x 2048-d synthetic), batch %d per GPU, len %d, reconstruction + VG + contrastive '
                                   'losses, whole training step (BASELINE configs[%d]); random-init weights, V 10000, 1024-d embeddings, k_neg 100'
                                   % (D, B, L, 3 if world > 1 else 2),
                       'global_batch': world * B, 'length': L, 'dim': D, 'mfma': mfma_mode,
                       'parallelism': 'dp%d (one flat-gradient all-reduce per step)' % world if world > 1 else 'single GPU',
                       'ranks_seen': ranks['ranks_seen'], 'world_size': world, 'via_launcher': bool(args.launched),
                       'per_rank_ms_per_step': ranks['per_rank_ms_per_step'],
                       **({'share_device': 'every rank on cuda:0 (self-test of the N > 1 path on one GPU: NOT a scaling figure)'} if args.share_device else {}),
                       **({'gradient_exchange': '%s all-reduce (backend %s) of one flat fp32 buffer of %d floats per step (chart + heads + ImageEncoder); '
                                                '%d of %d gradients copied in (the chart backward writes the rest in place)'
                                                % ('RCCL' if args.backend == 'nccl' else args.backend, args.backend, reducer.flat.numel(), reducer.copied, len(reducer.params))} if use_dist else {})},
            'step_ms': dict(median=round(pct(step_ms, 0.5), 4), p10=round(pct(step_ms, 0.1), 4), p90=round(pct(step_ms, 0.9), 4),
                            note='per-step device time from HIP event pairs on the launch stream inside the timed region (rank 0)'),
            'roofline': dict(bound='hbm', kernel='whole step (no single dominant kernel is timed for this workload; see --workload c2)',
                             achieved=round((step_bytes + scorer_bytes) / (dt / args.steps) / 1e9, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                             frac=round((step_bytes + scorer_bytes) / (dt / args.steps) / 1e9 / PEAK_HBM_GBS, 4), traffic=shape_traffic('c3_step'),
                             model='SURVEY.md section 8(d) chart bytes (%d per step) + the (B, B, C, 36) scorer tensor three times (%d)' % (step_bytes, scorer_bytes)),
        }
        if args.no_cpu_baseline:
            out['cpu_baseline'] = None
        elif world == 1:
            out['cpu_baseline'] = cpu_baseline_c3(L, D, B)
            remember_cpu_baseline('c3', out['cpu_baseline'])
        else:
            out['cpu_baseline'] = carried_cpu_baseline('c3', 'r06_bench_c3.json')
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--length', type=int, default=20)
    ap.add_argument('--dim', type=int, default=400)
    ap.add_argument('--batch', type=int, default=64, help='sentences per GPU')
    ap.add_argument('--workload', choices=['c2', 'c3'], default='c2',
                    help='c2 (default): BASELINE configs[1], DioraMLP chart forward + backward -- the configuration the metric is quoted on; '
                         'c3: the CLIORA training step of configs[2] / configs[3] (Embed, ImageEncoder, chart with 36 x 2048-d regions, '
                         'reconstruction + VG + contrastive losses, backward, gradient all-reduce, clip, Adam): `--gpus 8 --workload c3` '
                         'is configs[3] (batch 512 global)')
    ap.add_argument('--mfma', choices=['bf16x3', 'f32'], default=None,
                    help='arithmetic of the compose GEMMs (default: the library default, split-bf16; see include/cliora_chart.h)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the f32-mode / other-shape / whole-step side figures')
    ap.add_argument('--force-dist', action='store_true', help='initialise the process group and run the gradient all-reduce even at world size 1 (self-test of the N>1 path)')
    ap.add_argument('--backend', default='nccl', help='process-group backend (nccl = RCCL on ROCm)')
    ap.add_argument('--via-launcher', action='store_true',
                    help='take the N > 1 launch route at any N (also --gpus 1): this process starts torch.distributed.run as a child, the rank '
                         'initialises the process group and runs the gradient all-reduce -- the exact path `--gpus 8` takes, testable on one GPU')
    ap.add_argument('--launched', action='store_true', help=argparse.SUPPRESS)      # set by launch_ranks() on the ranks it starts
    ap.add_argument('--share-device', action='store_true',
                    help='self-test of the N > 1 path on a box with ONE GPU: every rank uses cuda:0 (use with --backend gloo: RCCL refuses two ranks on '
                         'one device); the value is not a scaling figure')
    args = ap.parse_args()

    if (args.gpus > 1 or args.via_launcher) and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))

    import torch
    import torch.distributed as dist
    from cliora_amd import _lib
    if args.mfma:
        _lib.set_mfma_mode(args.mfma)
    mfma_mode = _lib.set_mfma_mode('bf16x3')      # read the mode in force (set returns the previous one) ...
    _lib.set_mfma_mode(mfma_mode)                 # ... and put it back
    from cliora_amd.diora import DioraMLP
    from cliora_amd.parallel import FlatGradAllReduce

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = 0 if args.share_device else int(os.environ.get('LOCAL_RANK', '0'))
    use_dist = world > 1 or args.force_dist or args.launched
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        torch.cuda.set_device(local)
        dist.init_process_group(args.backend)
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE=%d but --gpus %d: launch one rank per GPU (or run without a launcher)' % (world, args.gpus))
    dev = torch.device('cuda', local)
    B, L, D = args.batch, args.length, args.dim
    C = L * (L + 1) // 2
    if args.workload == 'c3':
        return cliora_training_workload(args, torch, dist, _lib, dev, world, rank, local, use_dist, mfma_mode)

    torch.manual_seed(1234)                     # same parameters on every rank (train_diora.sh:9)
    model = DioraMLP(D, outside=True, normalize='unit', compress=False, share=True).to(dev)
    g = torch.Generator(device='cpu').manual_seed(1234 + rank)
    x = torch.randn(B, L, D, generator=g).to(dev)
    cots = [torch.randn(B, C, w, generator=g).to(dev) for w in (D, 1, D, 1)]
    params = [p for p in model.parameters() if p.requires_grad]
    reducer = FlatGradAllReduce(params) if use_dist else None
    keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s')

    def step():
        for p in params:
            p.grad = None
        model(x, x)
        torch.autograd.backward([getattr(model, k) for k in keys], cots)
        if reducer is not None:
            reducer.all_reduce_mean()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    kclasses = ('compose_fwd', 'compose_bwd', 'wgrad')
    events = (rank == 0) and not args.no_kernel_events
    dt, step_ms = timed_steps(torch, step, fence, args.steps)
    dt, ranks = rank_view(torch, dist, dev, dt, args.steps, use_dist, world)
    # Second pass of the same K steps with a HIP event pair around every launch of the three GEMM classes (on the
    # launch stream), for the roofline object.  It is separate from the timed region above because the event records
    # serialise neighbouring launches; every rank runs the steps so the all-reduces stay matched, only rank 0 records events.
    dt_ev = None
    if not args.no_kernel_events:
        if events:
            for k in kclasses:
                _lib.prof_read(k)
                _lib.prof_enable(k, True)
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt_ev = time.perf_counter() - t1
    # Third pass, wavefront off: the same launches one after the other on one stream.  In the default schedule the compose kernels of
    # the two chains run side by side, so a launch's event-to-event time includes the share of the chip its neighbour holds; this
    # pass gives the kernel's duration with the chip to itself (reported beside the as-run figure, never instead of it).
    kern, kern_seq, dt_seq, n_seq = {}, {}, None, min(args.steps, 20)
    if not args.no_kernel_events:
        if events:
            stream = torch.cuda.current_stream().cuda_stream
            for k in kclasses:
                ms, n = _lib.prof_read(k, stream)            # reads and resets; recording stays on for the third pass
                kern[k] = dict(total_ms=ms, launches=n)
        prev_wf = _lib.set_wavefront('off')
        try:
            fence()
            t1 = time.perf_counter()
            for _ in range(n_seq):
                step()
            fence()
            dt_seq = time.perf_counter() - t1
        finally:
            _lib.set_wavefront(prev_wf)
        if events:
            for k in kclasses:
                ms, n = _lib.prof_read(k, stream)
                _lib.prof_enable(k, False)
                kern_seq[k] = dict(total_ms=ms, launches=n)

    if rank == 0:
        pairs = (L - 1) * L * (L + 1) // 2          # span pairs per sentence, inside + outside
        Dp = (D + 15) // 16 * 16
        flops_class = 2.0 * Dp * Dp * pairs * B      # one D x D layer per pair (factored compose), per step
        roof = None
        if kern and all(v['launches'] for v in kern.values()):
            plan = _lib.get_plan(B, L, D, True, 'unit', 0, local)
            step_bytes, per_level = algorithmic_bytes(plan, B, D)
            fwd_levels = sum(per_level['in']) + sum(per_level['out'])
            # SURVEY section 8(d) bytes of each class per step: the compose forward launches own the levels' forward bytes
            # (unique operand cells read + target cells written), the compose backward launches twice that (the survey's
            # backward convention), the pair weight gradient none of its own (dW2 is a reduction over rows that the model
            # counts in the backward) -- its entry is the implementation's own operand traffic and is labelled so.
            alg_bytes = {'compose_fwd': fwd_levels, 'compose_bwd': 2.0 * fwd_levels, 'wgrad': None}
            impl_bytes = {
                # what this implementation moves per step in the class, by construction
                'compose_fwd': 'operand rows re-gathered by each of the %d column blocks through L2 (5 x 3.2 kB per pair row), '
                               'partial aggregates written once; no per-pair row stored' % (Dp // 80 if Dp % 80 == 0 else 1),
                'compose_bwd': 'dG rows + ReLU bits gathered per column block; DA rows and X / DZ (d 400: as 16-row tiles of bf16 hi + lo planes, the weight gradient\'s operand form) written (3 x %d MB per step)' % (pairs * B * Dp * 4 // 2**20),
                'wgrad': 2.0 * pairs * B * Dp * 4.0,
            }
            peak_mfma = PEAK_BF16_MFMA_TFLOPS / 3.0 if mfma_mode == 'bf16x3' else PEAK_FP32_MFMA_TFLOPS   # 3 bf16 MFMAs per product
            dom = max(('compose_fwd', 'compose_bwd'), key=lambda k: kern[k]['total_ms'])
            nlaunch = kern[dom]['launches'] / args.steps
            avg_ms = kern[dom]['total_ms'] / kern[dom]['launches']
            t_meas = kern[dom]['total_ms'] / args.steps * 1e-3                  # seconds per step in this class
            t_mfma = flops_class / (peak_mfma * 1e12)
            t_hbm = alg_bytes[dom] / (PEAK_HBM_GBS * 1e9)
            ach_tf = flops_class / t_meas / 1e12
            ach_gb = alg_bytes[dom] / t_meas / 1e9
            # HBM traffic per launch: NOT measured in this run (PMC passes need rocprofv3) -- the figure committed under profiles/ by
            # tools/final_profile.sh (three separate --pmc passes of this same command), labelled as such in the line
            traffic, traffic_src = None, None
            tp = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tp):
                tj = json.load(open(tp))
                traffic = tj.get(dom)
                traffic_src = dict(file='profiles/traffic.json', kind='committed rocprofv3 PMC figure (2 x FETCH_SIZE + WRITE_SIZE per launch, own --pmc passes), not collected in this run',
                                   measured=tj.get('measured', 'see profiles/README.md'), commit=tj.get('commit', 'not recorded'))
            # the same kernel's average duration in the committed rocprofv3 --kernel-trace --stats summary of this command, so that
            # frac can be reproduced from profiles/ alone (the two clocks agree within a few per cent)
            rocprof = None
            for cand in ('r06_kernel_stats.csv', 'r05_kernel_stats.csv', 'r04_kernel_stats.csv'):
                kp = os.path.join(ROOT, 'profiles', cand)
                if os.path.exists(kp):
                    import csv
                    for row in csv.DictReader(open(kp)):
                        if ('level_' + dom) in row.get('Name', ''):
                            avg_us = float(row['AverageNs']) / 1e3
                            rocprof = dict(file='profiles/' + cand, avg_launch_ms=round(avg_us / 1e3, 5), calls=int(row['Calls']),
                                           frac=round(max(alg_bytes[dom] / (PEAK_HBM_GBS * 1e9), flops_class / (peak_mfma * 1e12)) / (avg_us * 1e-6 * kern[dom]['launches'] / args.steps), 4))
                            break
                    break
            hbm_bound = t_hbm >= t_mfma
            t_roof = max(t_hbm, t_mfma)
            # The kernel's duration with the chip to itself (third pass, wavefront off) is the one that agrees with the committed
            # rocprofv3 --kernel-trace --stats average (the profiler runs the two chains' kernels one after the other): `frac`, `achieved`
            # and `avg_launch_ms` are quoted on it.  The second pass's figure -- HIP events around every launch while the other chain
            # shares the chip, the records themselves barrier packets -- is reported beside it as `frac_as_run` / `avg_launch_ms_as_run`.
            alone = kern_seq.get(dom) if kern_seq.get(dom) and kern_seq[dom]['launches'] else None
            t_alone = (alone['total_ms'] / n_seq * 1e-3) if alone else t_meas
            avg_alone_ms = (alone['total_ms'] / alone['launches']) if alone else avg_ms
            # Whole step against SURVEY 8(d): t_roof = max(bytes / HBM peak, executed FLOPs / MFMA peak of the arithmetic each part uses).
            # FLOPs of the factored formulation per step: pair layers 2 D^2 per pair (forward) + twice that (backward: u = W2^T dz, dW2);
            # per-cell projections 10 D^2 per cell and chart (forward, exact fp32 MFMA in both modes) + twice that (backward: split-bf16 in the
            # default mode).  Split-bf16 parts issue 3 bf16 MFMAs per product: priced at the bf16 peak / 3.
            C_cells = L * (L + 1) // 2
            cell_fwd = 10.0 * Dp * Dp * C_cells * B
            if mfma_mode == 'bf16x3':
                t_step_mfma = (3.0 * flops_class + 2.0 * cell_fwd) / (PEAK_BF16_MFMA_TFLOPS / 3.0 * 1e12) + cell_fwd / (PEAK_FP32_MFMA_TFLOPS * 1e12)
            else:
                t_step_mfma = (3.0 * flops_class + 3.0 * cell_fwd) / (PEAK_FP32_MFMA_TFLOPS * 1e12)
            t_step_hbm = step_bytes / (PEAK_HBM_GBS * 1e9)
            t_step = dt / args.steps
            ws_bound = 'hbm' if t_step_hbm >= t_step_mfma else 'mfma'
            ws_frac = max(t_step_hbm, t_step_mfma) / t_step
            roof = dict(bound='hbm' if hbm_bound else 'mfma', kernel='level_' + dom,
                        achieved=round((alg_bytes[dom] / t_alone / 1e9) if hbm_bound else (flops_class / t_alone / 1e12), 2),
                        peak=PEAK_HBM_GBS if hbm_bound else round(peak_mfma, 1),
                        unit='GB/s' if hbm_bound else 'TFLOP/s', frac=round(t_roof / t_alone, 4), traffic=traffic,
                        frac_as_run=round(t_roof / t_meas, 4),
                        avg_launch_ms=round(avg_alone_ms, 5), avg_launch_ms_as_run=round(avg_ms, 5),
                        rocprof_avg_launch_ms=(rocprof or {}).get('avg_launch_ms'), rocprof_file=(rocprof or {}).get('file'),
                        launches_per_step=nlaunch,
                        # the whole step, both terms of SURVEY 8(d) (flat keys: the driver keeps scalars)
                        whole_step_t_roof_hbm_ms=round(t_step_hbm * 1e3, 4), whole_step_t_roof_mfma_ms=round(t_step_mfma * 1e3, 4),
                        whole_step_bound=ws_bound, whole_step_frac=round(ws_frac, 4), whole_step_ms=round(t_step * 1e3, 4),
                        whole_step=dict(t_roof_hbm_ms=round(t_step_hbm * 1e3, 4), t_roof_mfma_ms=round(t_step_mfma * 1e3, 4), bound=ws_bound,
                                        frac=round(ws_frac, 4), ms_per_step=round(t_step * 1e3, 4), algorithmic_bytes=round(step_bytes),
                                        executed_flop=dict(pair_layers_fwd_bwd=round(3.0 * flops_class), cell_projections_fwd=round(cell_fwd),
                                                           cell_projections_bwd=round(2.0 * cell_fwd)),
                                        achieved_GBs=round(step_bytes / t_step / 1e9, 1), frac_of_hbm=round(t_step_hbm / t_step, 4),
                                        frac_of_mfma=round(t_step_mfma / t_step, 4),
                                        model='SURVEY.md 8(d): max(algorithmic bytes / 8 TB/s, executed FLOPs / MFMA peak of the arithmetic used); split-bf16 '
                                              'products = 3 bf16 MFMAs at 2.5 PFLOP/s dense, fp32 products at 157.3 TFLOP/s'),
                        traffic_source=traffic_src, rocprof=rocprof,
                        model='SURVEY.md section 8(d): t_roof = max(algorithmic bytes / 8 TB/s, executed FLOPs / MFMA peak of the arithmetic used)',
                        hbm_term=dict(algorithmic_bytes_per_launch=round(alg_bytes[dom] / nlaunch), achieved_GBs=round(alg_bytes[dom] / t_alone / 1e9, 1),
                                      peak_GBs=PEAK_HBM_GBS, frac=round(t_hbm / t_alone, 4), frac_as_run=round(t_hbm / t_meas, 4)),
                        mfma_term=dict(algorithmic_flop_per_launch=round(flops_class / nlaunch), achieved_TFLOPs=round(flops_class / t_alone / 1e12, 2),
                                       peak_TFLOPs=round(peak_mfma, 1), frac=round(t_mfma / t_alone, 4), frac_as_run=round(t_mfma / t_meas, 4),
                                       frac_of_f32_mfma_peak=round(flops_class / t_alone / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)),
                        measured='frac / achieved / avg_launch_ms: third pass, %d steps with cliora_set_wavefront(OFF) -- the kernel with the chip to itself, HIP '
                                 'events on the launch stream (agrees with the rocprofv3 --stats average, rocprof_avg_launch_ms); *_as_run: second pass of the '
                                 'same %d steps in the default schedule, where the other chain\'s kernels share the chip and the event records add barrier '
                                 'packets' % (n_seq, args.steps),
                        ms_per_step_with_events=round(dt_ev / args.steps * 1e3, 4),
                        ms_per_step_sequential_with_events=(round(dt_seq / n_seq * 1e3, 4) if dt_seq else None),
                        implementation_bytes=impl_bytes,
                        classes={k: dict(ms_per_step=round(v['total_ms'] / args.steps, 4), launches_per_step=v['launches'] / args.steps,
                                         tflops=round(flops_class * args.steps / (v['total_ms'] * 1e-3) / 1e12, 2),
                                         algorithmic_GBs=(round(alg_bytes[k] * args.steps / (v['total_ms'] * 1e-3) / 1e9, 1) if alg_bytes[k] else None))
                                 for k, v in kern.items()})
        out = {
            'metric': 'sentences/sec (inside+outside fwd+bwd), len-%d d=%d bsz=%d' % (L, D, B),
            'value': round(world * B * args.steps / dt, 2), 'unit': 'sentences/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (compose GEMMs: 3x bf16 MFMA per product, fp32 accumulate)' if mfma_mode == 'bf16x3' else 'f32',
            'data': 'synthetic',
            'config': {'workload': 'DioraMLP d=%d, batch %d per GPU, synthetic len %d, emb=none, text-only inside-outside '
                                   '(BASELINE configs[1]); chart forward + hand-written backward; random N(0,1) weights'
                                   % (D, B, L),
                       'global_batch': world * B, 'length': L, 'dim': D, 'mfma': mfma_mode,
                       'parallelism': 'dp%d (one flat-gradient all-reduce per step)' % world if world > 1 else 'single GPU',
                       'ranks_seen': ranks['ranks_seen'], 'world_size': world, 'via_launcher': bool(args.launched),
                       'per_rank_ms_per_step': ranks['per_rank_ms_per_step'],
                       **({'share_device': 'every rank on cuda:0 (self-test of the N > 1 path on one GPU: NOT a scaling figure)'} if args.share_device else {}),
                       **({'gradient_exchange': '%s all-reduce (backend %s) of one flat fp32 buffer of %d floats per step; %d of %d gradients copied in '
                                                '(the chart backward writes the rest in place)'
                                                % ('RCCL' if args.backend == 'nccl' else args.backend, args.backend, reducer.flat.numel(), reducer.copied, len(reducer.params))} if use_dist else {})},
            'step_ms': dict(median=round(pct(step_ms, 0.5), 4), p10=round(pct(step_ms, 0.1), 4), p90=round(pct(step_ms, 0.9), 4),
                            note='per-step device time from HIP event pairs on the launch stream inside the timed region (rank 0)'),
            'roofline': roof,
        }
        if world == 1 and not args.no_extras:
            prev = _lib.set_mfma_mode('f32')
            try:
                for _ in range(3):
                    step()
                dt32, ms32 = timed_steps(torch, step, fence, 10)
                out['value_f32_mode'] = dict(value=round(B * 10 / dt32, 2), ms_per_step=round(dt32 / 10 * 1e3, 4), median_ms=round(pct(ms32, 0.5), 4),
                                             note='same workload with exact fp32 products (cliora_set_mfma_mode(F32)), 10 steps')
            finally:
                _lib.set_mfma_mode(prev)
            del model
            torch.cuda.empty_cache()
            out['other_shapes'] = other_measurements(torch, dev)
        if args.no_cpu_baseline:
            out['cpu_baseline'] = None
        elif world == 1:
            out['cpu_baseline'] = cpu_baseline(L, D, B)
            remember_cpu_baseline('c2', out['cpu_baseline'])
        else:
            out['cpu_baseline'] = carried_cpu_baseline('c2', 'r06_bench.json')
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
