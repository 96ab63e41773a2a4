import sys, time, torch
sys.path.insert(0, '.')
from cliora_amd.diora import DioraMLP
torch.manual_seed(0)
m = DioraMLP(400).cuda().eval()
for p in m.parameters(): torch.nn.init.normal_(p)
x = torch.randn(64, 20, 400, device='cuda')
def run(what, n=30):
    def step():
        with torch.no_grad():
            m(x, x)
            if what == 'spans': return m.cky_spans()
            if what == 'trees': return m.cky()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print({w: round(run(w), 3) for w in ('forward', 'spans', 'trees')})
