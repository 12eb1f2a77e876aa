import torch, time
x = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
d = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for n in (1 << 30, 1 << 26, 1 << 22, 1 << 20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = max(1, (1 << 31) // n)
    for i in range(reps): d[:n].copy_(x[:n], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"H2D pinned, {n >> 20} MiB pieces: {reps * n / dt / 1e9:.1f} GB/s")
s = [torch.cuda.Stream() for _ in range(4)]
n = 1 << 26
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(32):
    with torch.cuda.stream(s[i % 4]): d[(i % 16) * n:(i % 16 + 1) * n].copy_(x[(i % 16) * n:(i % 16 + 1) * n], non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"H2D pinned, 64 MiB pieces on 4 streams: {32 * n / dt / 1e9:.1f} GB/s")
