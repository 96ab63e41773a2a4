#!/usr/bin/env python
"""BUILD CONTAINER ONLY (imports /root/reference): wall time of the CPU oracle (oracle/diora_ref.py, what bench.py's `cpu_baseline`
times as kind "port") beside the imported reference's own DioraMLP (cliora/net/diora.py:295-450) on the same inputs, chart forward +
backward, alternating, same thread count -- SURVEY.md section 8(d) asks the port to stay within +-10 % of the reference.

  python tools/cpu_port_check.py [--out profiles/r05_cpu_port_check.json]

bench.py reads the committed JSON and prints the measured ratio as `cpu_baseline.port_vs_reference`.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
sys.modules.setdefault('cv2', types.ModuleType('cv2'))

import torch                                        # noqa: E402
from cliora.net import diora as ref_diora            # noqa: E402
from oracle import diora_ref as R                    # noqa: E402
from oracle import synth                             # noqa: E402

KEYS = ('inside_h', 'inside_s', 'outside_h', 'outside_s')


def ref_step(m, x, cot):
    for p in m.parameters():
        p.grad = None
    xg = x.clone().requires_grad_(True)
    t0 = time.perf_counter()
    m(xg, xg)
    torch.autograd.backward([getattr(m, k) for k in KEYS], [cot[k] for k in KEYS])
    return time.perf_counter() - t0


def port_step(P, x, cot):
    for v in P.values():
        v.grad = None
    xg = x.clone().requires_grad_(True)
    t0 = time.perf_counter()
    out = R.diora_forward(P, xg, xg, training=True)
    torch.autograd.backward([out[k] for k in KEYS], [cot[k] for k in KEYS])
    return time.perf_counter() - t0


def case(D, B, L, threads, reps):
    torch.set_num_threads(threads)
    P, x, cot = synth.diora_case(D, B, L, 1234)
    for v in P.values():
        v.requires_grad_(True)
    m = ref_diora.DioraMLP(D, outside=True, normalize='unit', compress=False, share=True)
    sd = m.state_dict()
    for k in sd:
        sd[k] = P[k if k in P else 'inside_' + k[len('outside_'):]].detach().clone()
    m.load_state_dict(sd)
    m.train()
    ref_step(m, x, cot); port_step(P, x, cot)         # warm-up (index caches, thread pool)
    tr, tp = [], []
    for _ in range(reps):                              # alternating: the box's noise hits both alike
        tr.append(ref_step(m, x, cot))
        tp.append(port_step(P, x, cot))
    tr.sort(); tp.sort()
    mr, mp = tr[len(tr) // 2], tp[len(tp) // 2]
    # same numbers?  (the pin itself is tests/test_oracle_golden.py; this is a sanity line)
    out = R.diora_forward(P, x, x, training=True)
    m(x, x)
    err = max(float((getattr(m, k).detach() - out[k].detach()).abs().max()) for k in KEYS)
    return dict(D=D, B=B, L=L, threads=threads, reps=reps, reference_s=round(mr, 4), port_s=round(mp, 4), port_vs_reference=round(mp / mr, 3),
                reference_min_s=round(tr[0], 4), port_min_s=round(tp[0], 4), max_abs_output_diff=err)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r05_cpu_port_check.json'))
    ap.add_argument('--threads', type=int, default=len(os.sched_getaffinity(0)))
    a = ap.parse_args()
    cases = [case(50, 8, 10, a.threads, 15), case(400, 16, 20, a.threads, 5), case(400, 32, 20, a.threads, 5), case(400, 16, 20, 1, 3)]
    cpu = [ln.split(':', 1)[1].strip() for ln in open('/proc/cpuinfo') if ln.startswith('model name')]
    rec = dict(what='CPU oracle (oracle/diora_ref.py: the "port" timed by bench.py cpu_baseline) vs the imported reference cliora.net.diora.DioraMLP, '
                    'chart forward + backward on the same inputs, alternating, median of reps',
               host=dict(cpu=cpu[0] if cpu else 'unknown', logical_cpus=len(cpu), torch=torch.__version__), cases=cases,
               port_vs_reference_d400=round(sum(c['port_vs_reference'] for c in cases if c['D'] == 400 and c['threads'] > 1) /
                                            max(1, sum(1 for c in cases if c['D'] == 400 and c['threads'] > 1)), 3))
    json.dump(rec, open(a.out, 'w'), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    main()
