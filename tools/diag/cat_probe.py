import threading, time, torch
dev = "cuda"
a = torch.randn(8, 32, 2048, device=dev, dtype=torch.float32)
b = torch.randn(8, 48, 2048, device=dev, dtype=torch.bfloat16)
m1 = torch.ones(8, 32, dtype=torch.long, device=dev); ids = torch.randint(0, 100, (8, 48), device=dev)
def body(tag, n=200, autocast=True):
    ts = []
    for i in range(n):
        t0 = time.perf_counter()
        if autocast:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                e = torch.cat([a.to(b.dtype), b], dim=1)
                m = torch.cat([m1, (ids != 0).long()], dim=1)
        else:
            e = torch.cat([a.to(b.dtype), b], dim=1)
            m = torch.cat([m1, (ids != 0).long()], dim=1)
        ts.append(time.perf_counter() - t0)
        if i % 50 == 49: torch.cuda.synchronize()
    ts = sorted(ts[20:])
    print(tag, "median %.1f us" % (ts[len(ts)//2] * 1e6), "p90 %.1f us" % (ts[int(len(ts)*.9)] * 1e6))
body("main autocast"); body("main plain", autocast=False)
t = threading.Thread(target=body, args=("thread autocast",)); t.start(); t.join()
# queue deep: many kernels already enqueued (launch queue back-pressure)
x = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
for _ in range(300): y = x @ x
body("main autocast, busy device")
torch.cuda.synchronize()
