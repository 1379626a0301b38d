import os, sys, time, torch
dev=torch.device('cuda:0')
def bench(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
M=int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for K,N in ((1344,1344),(1344,2688),(2688,2688),(2688,5376)):
    x=torch.randn(M,K,device=dev,dtype=torch.bfloat16); w=torch.randn(N,K,device=dev,dtype=torch.bfloat16); b=torch.randn(N,device=dev,dtype=torch.bfloat16)
    t1=bench(lambda: torch.nn.functional.linear(x,w,b))
    t2=bench(lambda: torch._addmm_activation(b,x,w.t(),use_gelu=False))
    wt=w.t().contiguous()
    t3=bench(lambda: torch._addmm_activation(b,x,wt,use_gelu=False))
    fl=2*M*K*N
    print(f"M{M} K{K} N{N}: linear {t1:.1f}us ({fl/t1/1e6:.0f} TF)  addmm_act {t2:.1f}us  addmm_act(NN layout) {t3:.1f}us ({fl/t3/1e6:.0f} TF)")
print('tunable', os.environ.get('PYTORCH_TUNABLEOP_ENABLED'))
