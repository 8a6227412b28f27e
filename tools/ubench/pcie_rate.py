"""Host<->device copy rates of the box: pageable vs pinned, and host memcpy into pinned memory by thread count."""
import time, threading, numpy as np, torch
n = 64 << 20
dev = torch.device("cuda:0")
d = torch.empty(n, dtype=torch.uint8, device=dev)
pg = torch.ones(n, dtype=torch.uint8)
pn = torch.ones(n, dtype=torch.uint8).pin_memory()
def t(f, k=5):
    best = 1e9
    for _ in range(k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
print("H2D pageable GB/s", round(n / t(lambda: d.copy_(pg)) / 1e9, 1))
print("H2D pinned   GB/s", round(n / t(lambda: d.copy_(pn, non_blocking=True)) / 1e9, 1))
print("D2H pageable GB/s", round(n / t(lambda: pg.copy_(d)) / 1e9, 1))
print("D2H pinned   GB/s", round(n / t(lambda: pn.copy_(d, non_blocking=True)) / 1e9, 1))
a = pg.numpy(); b = pn.numpy()
for th in (1, 2, 4, 8):
    def run():
        ts = [threading.Thread(target=lambda i=i: np.copyto(b[i * n // th:(i + 1) * n // th], a[i * n // th:(i + 1) * n // th])) for i in range(th)]
        [x.start() for x in ts]; [x.join() for x in ts]
    best = min((lambda: (lambda t0: (run(), time.perf_counter() - t0)[1])(time.perf_counter()))() for _ in range(5))
    print(f"host memcpy pageable->pinned {th} threads GB/s", round(n / best / 1e9, 1))
