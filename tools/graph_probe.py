"""Dev tool: which C-ABI entry points survive HIP-graph capture?  Each probe runs in its own process."""
import subprocess
import sys

PROBES = {
    "gemm": "a=torch.randn(256,128,device=d).bfloat16(); b=torch.randn(256,128,device=d).bfloat16(); f=lambda: ops.gemm_nt(a,128,b,128,256,256,128)",
    "vocab": "h=torch.randn(256,64,device=d,requires_grad=True); w=torch.randn(500,64,device=d,requires_grad=True); bb=torch.zeros(500,device=d,requires_grad=True); t=torch.randint(1,500,(256,),device=d)\ndef f():\n    n,_=ops.VocabNLL.apply(h,w,bb,t,1.0,0,None,None); n.sum().backward()",
    "gru_fwd": "x=torch.randn(64,5,32,device=d); ws=[torch.randn(192,32,device=d)*.1,torch.randn(192,64,device=d)*.1,torch.zeros(192,device=d),torch.zeros(192,device=d)]\ndef f():\n    with torch.no_grad(): ops.GRULayer.apply(x,None,*ws,False)",
    "gru_bwd": "x=torch.randn(64,5,32,device=d,requires_grad=True); ws=[(torch.randn(192,32,device=d)*.1).requires_grad_(),(torch.randn(192,64,device=d)*.1).requires_grad_(),torch.zeros(192,device=d,requires_grad=True),torch.zeros(192,device=d,requires_grad=True)]\ndef f():\n    o,h,_=ops.GRULayer.apply(x,None,*ws,True); o.sum().backward()",
    "emb": "w=torch.randn(100,16,device=d,requires_grad=True); i=torch.randint(0,100,(8,9),device=d)\ndef f():\n    ops.embedding(w,i,0).sum().backward()",
    "linear": "x=torch.randn(64,32,device=d,requires_grad=True); w=torch.randn(16,32,device=d,requires_grad=True); b=torch.zeros(16,device=d,requires_grad=True)\ndef f():\n    ops.linear(x,w,b).sum().backward()",
    "reparam": "m=[torch.randn(8,4,device=d,requires_grad=True) for _ in range(4)]; e=torch.randn(8,4,device=d)\ndef f():\n    z,k=ops.reparam_kl(m[0],m[1],m[2],m[3],e); (z.sum()+k.sum()).backward()",
    "msum": "x=torch.randn(1000,device=d,requires_grad=True)\ndef f():\n    ops.sum(x).backward()",
    "adam": "from pevae_amd.optim import FlatAdam; p=torch.nn.Parameter(torch.randn(1000,device=d)); o=FlatAdam([p],1e-3,1.0); p.grad.fill_(0.1); f=o.step",
}
TEMPLATE = """
import sys, torch
sys.path.insert(0, '.')
from pevae_amd import ops
d = torch.device('cuda:0')
%s
f(); f(); torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    f()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    f()
g.replay(); g.replay(); torch.cuda.synchronize()
print('OK')
"""
for name, body in PROBES.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    r = subprocess.run([sys.executable, "-c", TEMPLATE % body], capture_output=True, text=True, timeout=120)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "Fatal" in l][-2:]
    print("%-8s rc=%d %s %s" % (name, r.returncode, tail, " | ".join(err)), flush=True)
