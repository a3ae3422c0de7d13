import torch
dev='cuda:0'
for n in (4096, 8192):
    a=torch.randn(n,n,device=dev,dtype=torch.bfloat16); b=torch.randn(n,n,device=dev,dtype=torch.bfloat16)
    for _ in range(3): (a@b)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c=a@b
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(f"hipBLASLt bf16 {n}^3: {ms:.3f} ms  {2*n**3/ms/1e9:.0f} TFLOP/s")
