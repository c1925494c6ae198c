import torch
a = torch.randn(4096, 8192, device="cuda").bfloat16(); b = torch.randn(4096, 8192, device="cuda").bfloat16()
for _ in range(5):
    c = torch.matmul(a, b.t())
a2 = torch.randn(51200, 768, device="cuda").bfloat16(); b2 = torch.randn(2304, 768, device="cuda").bfloat16()
for _ in range(5):
    c2 = torch.matmul(a2, b2.t())
a3 = torch.randn(51200, 3072, device="cuda").bfloat16(); b3 = torch.randn(768, 3072, device="cuda").bfloat16()
for _ in range(5):
    c3 = torch.matmul(a3, b3.t())
torch.cuda.synchronize()
