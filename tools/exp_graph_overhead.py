"""What a HIP-graph replay costs on this stack: K tiny kernels (x.add_(1)) eagerly vs recorded in a torch.cuda.CUDAGraph, per kernel."""
import sys, time, torch
dev = torch.device('cuda:0')
x = torch.zeros(1 << 16, device=dev)
for K in (8, 45, 200):
    def body():
        for _ in range(K):
            x.add_(1.0)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for name, fn in (('eager', body), ('graph', g.replay)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print('K=%d %s: %.1f us per iteration (host issue %.1f us), %.2f us per kernel' % (K, name, t / 50 * 1e6, t_host / 50 * 1e6, t / 50 / K * 1e6))
