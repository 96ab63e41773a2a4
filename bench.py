#!/usr/bin/env python
"""Headline benchmark: sentences/sec of the chart hot path (inside + outside, forward +
backward) on synthetic length-20, d=400, batch-64 batches (BASELINE.json config 2).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = DioraMLP.forward (leaf transform, inside pass, outside pass) + the hand-written
backward for random cotangents on all four chart outputs, all HIP kernels behind the C ABI;
with N > 1 each rank owns its own 64 sentences (weak scaling) and the step ends with ONE
all-reduce of the flat gradient buffer over RCCL.  Inputs are resident in HBM before the
timed region.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
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
    return dict(value=b / med, unit='sentences/s', cores=threads, kind='port',
                sample='%d steps of %d sentences (L=%d, d=%d), chart fwd+bwd, torch %s CPU oracle, %d threads, median %.2f s/step, %.1f s total'
                       % (len(times), b, L, D, torch.__version__, threads, med, t_all))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--length', type=int, default=20)
    ap.add_argument('--dim', type=int, default=400)
    ap.add_argument('--batch', type=int, default=64, help='sentences per GPU')
    ap.add_argument('--mfma', choices=['bf16x3', 'f32'], default=None,
                    help='arithmetic of the compose GEMMs (default: the library default, split-bf16; see include/cliora_chart.h)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--force-dist', action='store_true', help='initialise the process group and run the gradient all-reduce even at world size 1 (self-test of the N>1 path)')
    args = ap.parse_args()

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
    local = int(os.environ.get('LOCAL_RANK', '0'))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl')
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    dev = torch.device('cuda', local)
    B, L, D = args.batch, args.length, args.dim
    C = L * (L + 1) // 2

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
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # Second pass of the same K steps with a HIP event pair around every launch of the three GEMM classes (on the
    # launch stream), for the roofline object.  It is separate from the timed region above because the ~230 event
    # records per step serialise neighbouring launches and cost ~8 % of a step (4.8 -> 5.3 ms on MI355X); every rank
    # runs the steps so the all-reduces stay matched, only rank 0 records events.
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

    out = None
    if rank == 0:
        pairs = (L - 1) * L * (L + 1) // 2          # span pairs per sentence, inside + outside
        Dp = (D + 15) // 16 * 16
        flops_class = 2.0 * Dp * Dp * pairs * B      # one D x D layer per pair (factored compose), per step
        kern = {}
        if events:
            stream = torch.cuda.current_stream().cuda_stream
            for k in kclasses:
                ms, n = _lib.prof_read(k, stream)
                _lib.prof_enable(k, False)
                kern[k] = dict(total_ms=ms, launches=n)
        roof = None
        if kern and all(v['launches'] for v in kern.values()):
            dom = max(kern, key=lambda k: kern[k]['total_ms'])
            nlaunch = kern[dom]['launches'] / args.steps
            avg_ms = kern[dom]['total_ms'] / kern[dom]['launches']
            # t_roof = max(algorithmic bytes / HBM bandwidth, executed FLOPs / MFMA peak of the dtype
            # used); the larger term names the bound.  Per step and class (one D x D layer per pair row, factored compose):
            #   FLOPs  2 Dp^2 per pair row
            #   bytes  fwd: x and y rows written + the operand cells read (each touched cell row once per level, at most 2
            #          per pair row);  bwd: y, x read and DA, DZ written per pair row + one dG row per target cell;
            #          weight gradient: x and dz rows read once
            row_b = 4.0 * Dp
            lv_in = [((L - l) * l * B, (L - l) * B) for l in range(1, L)]                 # (pair rows, target cells) per level
            lv_out = [((L - l) * (L - 1 - l) * B, (L - l) * B) for l in range(0, L - 1)]
            levels = lv_in + lv_out
            bytes_class = {
                'compose_fwd': sum(2 * r * row_b + min(2 * r, 2 * B * C) * row_b for r, _ in levels),
                'compose_bwd': sum(4 * r * row_b + c * row_b for r, c in levels),
                'wgrad': 2.0 * pairs * B * row_b,
            }
            peak_mfma = PEAK_BF16_MFMA_TFLOPS / 3.0 if mfma_mode == 'bf16x3' else PEAK_FP32_MFMA_TFLOPS   # 3 bf16 MFMAs per product
            t_meas = kern[dom]['total_ms'] / args.steps * 1e-3                  # seconds per step in this class
            t_mfma = flops_class / (peak_mfma * 1e12)
            t_hbm = bytes_class[dom] / (PEAK_HBM_GBS * 1e9)
            ach_tf = flops_class / t_meas / 1e12
            ach_gb = bytes_class[dom] / t_meas / 1e9
            traffic = None
            tp = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tp):
                traffic = json.load(open(tp)).get(dom)
            hbm_bound = t_hbm >= t_mfma
            roof = dict(bound='hbm' if hbm_bound else 'mfma', kernel=dom,
                        achieved=round(ach_gb if hbm_bound else ach_tf, 2), peak=PEAK_HBM_GBS if hbm_bound else round(peak_mfma, 1),
                        unit='GB/s' if hbm_bound else 'TFLOP/s', frac=round(max(t_hbm, t_mfma) / t_meas, 4), traffic=traffic,
                        hbm_term=dict(algorithmic_bytes_per_launch=round(bytes_class[dom] / nlaunch), achieved_GBs=round(ach_gb, 1),
                                      peak_GBs=PEAK_HBM_GBS, frac=round(t_hbm / t_meas, 4)),
                        mfma_term=dict(algorithmic_flop_per_launch=round(flops_class / nlaunch), achieved_TFLOPs=round(ach_tf, 2),
                                       peak_TFLOPs=round(peak_mfma, 1), frac=round(t_mfma / t_meas, 4),
                                       frac_of_f32_mfma_peak=round(ach_tf / PEAK_FP32_MFMA_TFLOPS, 4)),
                        avg_launch_ms=round(avg_ms, 5), launches_per_step=nlaunch,
                        measured='second pass of the same %d steps with per-launch HIP events' % args.steps,
                        ms_per_step_with_events=round(dt_ev / args.steps * 1e3, 4),
                        classes={k: dict(ms_per_step=round(v['total_ms'] / args.steps, 4),
                                         tflops=round(flops_class * args.steps / (v['total_ms'] * 1e-3) / 1e12, 2),
                                         algorithmic_GBs=round(bytes_class[k] * args.steps / (v['total_ms'] * 1e-3) / 1e9, 1))
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
                       'parallelism': 'dp%d (one flat-gradient all-reduce per step)' % world if world > 1 else 'single GPU'},
            'roofline': roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(L, D, B)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
