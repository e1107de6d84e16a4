import torch, time
x = torch.empty(1<<28, dtype=torch.float32, device='cuda')  # 1 GiB
y = torch.empty_like(x)
x.normal_()
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): y.copy_(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/10
print(f'torch copy 1 GiB: {ms:.3f} ms -> {2*x.numel()*4/ms/1e9:.2f} TB/s (read+write)')
e0.record()
for _ in range(10): x.mul_(1.0001)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/10
print(f'torch inplace mul 1 GiB: {ms:.3f} ms -> {2*x.numel()*4/ms/1e9:.2f} TB/s')
e0.record()
for _ in range(10): s = x.sum()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/10
print(f'torch sum 1 GiB: {ms:.3f} ms -> {x.numel()*4/ms/1e9:.2f} TB/s (read only)')
