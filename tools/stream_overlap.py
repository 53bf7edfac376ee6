"""Do two hipGraphs launched on two streams of the PyTorch process run concurrently?  Each graph is a chain of small
dependent kernels (a few workgroups each, far from filling the chip).  Also: ONE graph with two parallel branches."""
import torch, time
dev = torch.device("cuda:0")
def chain(x, n):
    for _ in range(n):
        x = x * 1.0001 + 0.5
    return x
xs = [torch.ones(65536, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
graphs = []
for x, st in zip(xs, streams):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        chain(x, 3)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=st):
        y = chain(x, 400)
    graphs.append(g)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def one():
    with torch.cuda.stream(streams[0]): graphs[0].replay()
def two():
    with torch.cuda.stream(streams[0]): graphs[0].replay()
    with torch.cuda.stream(streams[1]): graphs[1].replay()
print(f"one graph of 800 small kernels: {timed(one):.3f} ms; two such graphs on two streams: {timed(two):.3f} ms")
# one graph, two branches
g2 = torch.cuda.CUDAGraph()
s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.graph(g2, stream=s_main):
    s_side.wait_stream(s_main)
    with torch.cuda.stream(s_side):
        a = chain(xs[0], 400)
    b = chain(xs[1], 400)
    s_main.wait_stream(s_side)
def forked():
    with torch.cuda.stream(s_main): g2.replay()
print(f"one graph with two parallel branches of 800 kernels each: {timed(forked):.3f} ms")
