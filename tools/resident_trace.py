"""Phase times inside one sentence-resident forward launch (CLIORA_RES_TRACE=1: wall-clock stamps of workgroup 0, wave 0): per inside
level the cell routine (scores, softmax, compose, norm), the projection and the wait at the workgroup barrier.
python tools/resident_trace.py   (on the MI355X box)"""
import os, sys
os.environ['CLIORA_RES_TRACE']='1'; os.environ['CLIORA_RESIDENT']='1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, ctypes as C
from cliora_amd import _lib
from cliora_amd.diora import DioraMLP
D,B,L=50,8,10
torch.manual_seed(1)
m=DioraMLP(D).cuda(); x=torch.randn(B,L,D,device='cuda')
for _ in range(3):
    with torch.no_grad(): m(x,x)
torch.cuda.synchronize()
plan=_lib.get_plan(B,L,D,True,'unit',0,torch.cuda.current_device())
buf=np.zeros(256+8*L,dtype=np.uint64)
_lib.check(_lib.lib().cliora_resident_trace(plan.handle, buf.ctypes.data_as(C.c_void_p), buf.size, None),'trace')
f=buf[256:].reshape(L,8).astype(np.float64)/100.0
t=buf[:4*L].reshape(L,4).astype(np.float64)/100.0
for lv in range(2,L):
    print('step %d (inside level %d beside outside level %d): cell %.2f us, proj %.2f, barrier %.2f | step total %.2f' % (lv, lv, L-lv, t[lv,1]-t[lv,0], t[lv,2]-t[lv,1], t[lv,3]-t[lv,2], t[lv,3]-t[lv,0]))
for lv in range(2,L):
    print('  level %d cell: start->tables+issue %.2f, scores %.2f, softmax %.2f, compose %.2f, norm %.2f' % (lv, f[lv,0]-t[lv,0], f[lv,1]-f[lv,0], f[lv,2]-f[lv,1], f[lv,3]-f[lv,2], f[lv,4]-f[lv,3]))
print('forward steps 2..%d total %.1f us' % (L-1, t[L-1,3]-t[2,0]))

# ---- backward: per step the first outside cell (level j) and the first inside cell (level L-1-j)
keys=('inside_h','inside_s','outside_h','outside_s')
C_=L*(L+1)//2
cots=[torch.randn(B,C_,w,device='cuda') for w in (D,1,D,1)]
for _ in range(3):
    m(x,x); torch.autograd.backward([getattr(m,k) for k in keys], cots)
torch.cuda.synchronize()
buf=np.zeros(512+16*L,dtype=np.uint64)
_lib.check(_lib.lib().cliora_resident_trace(plan.handle, buf.ctypes.data_as(C.c_void_p), buf.size, None),'trace')
g=buf[512:].reshape(L,16).astype(np.float64)/100.0
print('backward, per step j: outside cell (level j) | inside cell (level L-1-j): gather, project, dnorm, pairs [us]; step span')
for j in range(L):
    o=g[j,:8]; i=g[j,8:]
    fo='%.2f %.2f %.2f %.2f' % (o[1]-o[0], o[2]-o[1], o[3]-o[2], o[4]-o[3]) if j < L-1 else '%.2f (root)' % (o[1]-o[0])
    fi='%.2f %.2f %.2f %.2f' % (i[1]-i[0], i[2]-i[1], i[3]-i[2], i[4]-i[3]) if j < L-1 else '%.2f %.2f (leaf)' % (i[1]-i[0], i[2]-i[1])
    print('  step %d: out %s | in %s | span %.2f' % (j, fo, fi, g[j,7]-min(o[0],i[0])))
print('backward chain total %.1f us' % (g[L-1,7]-g[0,0]))
