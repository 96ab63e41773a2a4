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
_lib.check(_lib.lib().cliora_persistent_trace(plan.handle, buf.ctypes.data_as(C.c_void_p), buf.size, None),'trace')
f=buf[256:].reshape(L,8).astype(np.float64)/100.0
t=buf[:4*L].reshape(L,4).astype(np.float64)/100.0
for lv in range(1,L):
    print('level %d: cell %.2f us, proj %.2f, barrier %.2f | level total %.2f' % (lv, t[lv,1]-t[lv,0], t[lv,2]-t[lv,1], t[lv,3]-t[lv,2], t[lv,3]-t[lv,0]))
for lv in range(1,L):
    print('  level %d cell: start->tables+issue %.2f, scores %.2f, softmax %.2f, compose %.2f, norm %.2f' % (lv, f[lv,0]-t[lv,0], f[lv,1]-f[lv,0], f[lv,2]-f[lv,1], f[lv,3]-f[lv,2], f[lv,4]-f[lv,3]))
print('inside pass total %.1f us' % (t[L-1,3]-t[1,0]))
