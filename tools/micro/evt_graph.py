import torch, time
x = torch.randn(4096, 4096, device="cuda")
s = torch.cuda.Stream()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y = x @ x
s.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        e0.record()
        y = x @ x
        e1.record()
        z = y + 1
        e2.record()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("elapsed matmul us", e0.elapsed_time(e1) * 1e3, "add us", e1.elapsed_time(e2) * 1e3)
except Exception as ex:
    print("FAILED:", type(ex).__name__, ex)
