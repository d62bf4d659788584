import torch
dev=torch.device("cuda"); M,d=96*8192,1056
a=torch.randn(M,d,device=dev); b=torch.empty_like(a)
h=torch.randn(M,d,device=dev).bfloat16(); h2=torch.empty_like(h)
for name,fn,nbytes in (("fp32 copy",lambda: b.copy_(a),M*d*8.0),("bf16 copy",lambda: h2.copy_(h),M*d*4.0),("fp32 add inplace",lambda: a.add_(1.0),M*d*8.0)):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize(); t=e0.elapsed_time(e1)/10
    print(f"{name}: {t*1e3:.1f} us {nbytes/t/1e9:.2f} TB/s")
