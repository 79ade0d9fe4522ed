import torch
x = torch.empty(1 << 28, dtype=torch.int32, device="cuda")
for _ in range(5): x.fill_(7)
for _ in range(5): x.zero_()
y = torch.empty_like(x)
for _ in range(5): y.copy_(x)
torch.cuda.synchronize()
