import torch, time
x = torch.empty(16384*16384, dtype=torch.float64, device='cuda')
for fn, name in ((lambda: x.fill_(1.5), 'fill_'), (lambda: x.zero_(), 'zero_')):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt=(time.time()-t)/20
    print(name, x.numel()*8/dt/1e9, 'GB/s')
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize(); t=time.time()
for _ in range(20): y.copy_(x)
torch.cuda.synchronize(); dt=(time.time()-t)/20
print('copy (r+w)', 2*x.numel()*8/dt/1e9, 'GB/s')
