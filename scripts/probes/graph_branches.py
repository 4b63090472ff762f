import torch, time
dev = "cuda:0"
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 20_000_000
def body(main):
    ev = torch.cuda.Event(); ev.record(main)
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        torch.cuda._sleep(N)
        ev2 = torch.cuda.Event(); ev2.record(s2)
    torch.cuda._sleep(N)
    main.wait_event(ev2)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
with torch.cuda.stream(s1):
    print("one sleep: %.2f ms" % timeit(lambda: torch.cuda._sleep(N)))
    print("eager two streams: %.2f ms" % timeit(lambda: body(s1)))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s1, capture_error_mode="thread_local"):
        body(s1)
    print("graph replay: %.2f ms" % timeit(lambda: g.replay()))
