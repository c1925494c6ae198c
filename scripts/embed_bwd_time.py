import sys, torch
sys.path.insert(0, '/root/repo')
from idvs.morec_amd import ops
dev='cuda'
n_seq,T,H,V=2688,30,768,30522
ids=torch.randint(1000,V,(n_seq*T,),device=dev,dtype=torch.int32)
ids.view(n_seq,T)[:,0]=101; ids.view(n_seq,T)[:,10:]=0
dz=torch.randn(n_seq*T,H,device=dev).to(torch.bfloat16)
dword,dpos,dtyp=torch.zeros(V,H,device=dev),torch.zeros(512,H,device=dev),torch.zeros(H,device=dev)
order=torch.argsort(ids,stable=True).to(torch.int32)
for _ in range(3): ops.bert_embed_bwd_(ids,dz,dword,dpos,dtyp,0,T,order)
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.bert_embed_bwd_(ids,dz,dword,dpos,dtyp,0,T,order)
e1.record(); torch.cuda.synchronize()
print(f"embed bwd (scatter + pos/type): {e0.elapsed_time(e1)/20*1e3:.1f} us per call")
