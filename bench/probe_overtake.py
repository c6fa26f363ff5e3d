import torch, time
dev = torch.device("cuda", 0)
x = torch.zeros(1 << 20, device=dev)
prod = torch.cuda.Stream(device=dev)
filled = torch.cuda.Event()
streams = [torch.cuda.Stream(device=dev) for _ in range(10)]
evs = [torch.cuda.Event() for _ in streams]
torch.cuda.synchronize()
with torch.cuda.stream(prod):
    torch.cuda._sleep(40_000_000)
    filled.record(prod)
for s, e in zip(streams, evs):
    with torch.cuda.stream(s):
        x.add_(1)
        e.record(s)
t0 = time.time()
over = []
while not filled.query():
    pass
over = [e.query() for e in evs]
print("sleep took %.1f ms after enqueue; streams done when the producer finished:" % ((time.time() - t0) * 1e3), over)
torch.cuda.synchronize()
