import torch, time
A = torch.randn(100000, 10000, device='cuda', dtype=torch.float32)
torch.backends.cuda.matmul.allow_tf32 = False
for _ in range(2):
    torch.cuda.synchronize(); t=time.time()
    G = A.t() @ A
    torch.cuda.synchronize(); dt=time.time()-t
    print("torch A^T A full: %.1f ms, %.1f TFLOP/s (full 2mn^2)" % (dt*1e3, 2*100000*10000*10000/dt/1e12))
