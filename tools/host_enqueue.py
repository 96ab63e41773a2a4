"""Host time to ENQUEUE one chart step (forward + backward through the module) against the device time of the step:
how far the host runs ahead of the GPU.  python tools/host_enqueue.py [B L D]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from cliora_amd.diora import DioraMLP  # noqa: E402

B, L, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 20, 400)
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = DioraMLP(D).to(dev).train()
for p in m.parameters():
    torch.nn.init.normal_(p)
x = torch.randn(B, L, D, device=dev, requires_grad=True)
C = L * (L + 1) // 2
keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s')
cot = [torch.randn(B, C, 1 if k.endswith('_s') else D, device=dev) for k in keys]


def step():
    for p in m.parameters():
        p.grad = None
    x.grad = None
    m(x, x)
    torch.autograd.backward([getattr(m, k) for k in keys], cot)


for _ in range(5):
    step()
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for _ in range(8):
    t0 = time.perf_counter()
    step()
    host.append((time.perf_counter() - t0) * 1e3)
t_enq = (time.perf_counter() - t_all) * 1e3
torch.cuda.synchronize()
t_tot = (time.perf_counter() - t_all) * 1e3
print('B %d L %d D %d: host enqueue per step (ms): %s; 8 steps enqueued in %.2f ms, done in %.2f ms'
      % (B, L, D, ' '.join('%.2f' % h for h in host), t_enq, t_tot))
