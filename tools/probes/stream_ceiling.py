"""What the memory system gives simple streaming kernels at the byte mix of the 1x1 layers (ATen's elementwise kernels):
copy 1:1, read 1 : write 2 (x -> [x ; x] along channels, the forward's mix), write only, read only (sum)."""
import torch
dev = 'cuda'
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B, C, P = 32, 32, 131072
x = torch.randn(B, C, P, device=dev); y = torch.empty_like(x); z = torch.empty(B, 2 * C, P, device=dev)
nb = x.numel() * 4
t = timeit(lambda: y.copy_(x)); print(f'copy 1:1        {t:7.1f} us  {2 * nb / t / 1e6:6.2f} TB/s')
t = timeit(lambda: torch.cat([x, x], 1, out=z)); print(f'read 1 write 2  {t:7.1f} us  {3 * nb / t / 1e6:6.2f} TB/s (counting x once)')
t = timeit(lambda: torch.mul(x, 2.0, out=y)); print(f'mul 1:1         {t:7.1f} us  {2 * nb / t / 1e6:6.2f} TB/s')
t = timeit(lambda: z.fill_(1.0)); print(f'write only      {t:7.1f} us  {2 * nb / t / 1e6:6.2f} TB/s')
t = timeit(lambda: x.sum()); print(f'read only (sum) {t:7.1f} us  {nb / t / 1e6:6.2f} TB/s')
t = timeit(lambda: torch.add(z[:, :C], z[:, C:], out=y)); print(f'read 2 write 1  {t:7.1f} us  {3 * nb / t / 1e6:6.2f} TB/s')
