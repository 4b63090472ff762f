"""Does a replayed HIP graph run a side-stream branch concurrently with the main chain?  (a) one fork / join, (b) the training step's
pattern: the main chain forks a leaf to the side stream after every node, the side stream joins once at the end."""
import time
import torch
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 2_000_000


def fork_join(main, k):
    for _ in range(k):
        torch.cuda._sleep(N)
        ev = torch.cuda.Event(); ev.record(main)
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            torch.cuda._sleep(N)
    ev2 = torch.cuda.Event(); ev2.record(s2)
    main.wait_event(ev2)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


with torch.cuda.stream(s1):
    one = timeit(lambda: torch.cuda._sleep(N))
    for k in (1, 20):
        eager = timeit(lambda: fork_join(s1, k))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s1, capture_error_mode="thread_local"):
            fork_join(s1, k)
        rep = timeit(lambda: g.replay())
        print(f"{k} main nodes + {k} side leaves of {one:.2f} ms each: eager {eager:.2f} ms, graph replay {rep:.2f} ms  (serial would be {2 * k * one:.2f}, concurrent {(k + 1) * one:.2f})")
