"""Dependency latency between kernels of a captured hipGraph: the same chain of N tiny dependent kernels captured (a) on one
stream and (b) alternating between two streams with an event edge per hop (what a forked step tail would pay per edge).
Prints us per hop for both; the difference is the cross-queue signalling cost of this runtime.

    python tools/xqueue_latency.py [hops]
"""
import sys
import time

import torch


def capture(n, two_streams):
    x = torch.zeros(64, device='cuda')
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        x.add_(1.0)                 # (code object loaded outside the capture)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s1):
        cur = s1
        for i in range(n):
            if two_streams:
                nxt = s2 if cur is s1 else s1
                ev = torch.cuda.Event()
                ev.record(cur)
                nxt.wait_event(ev)
                cur = nxt
            with torch.cuda.stream(cur):
                x.add_(1.0)
        if cur is not s1:
            ev = torch.cuda.Event()
            ev.record(cur)
            s1.wait_event(ev)
    return g, x


def capture_diamonds(n):
    """n x [A on s1 -> (B on s2 || C on s1) -> D on s1 after both]: a real fork, two queues busy at once."""
    x, y = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s1):
        x.add_(1.0)
        y.add_(1.0)
    torch.cuda.synchronize()
    x.zero_()
    y.zero_()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s1):
        for i in range(n):
            x.add_(1.0)                                   # A
            ev = torch.cuda.Event()
            ev.record(s1)
            s2.wait_event(ev)
            with torch.cuda.stream(s2):
                y.add_(1.0)                               # B (side)
                ev2 = torch.cuda.Event()
                ev2.record(s2)
            x.add_(1.0)                                   # C (main)
            s1.wait_event(ev2)
            x.add_(1.0)                                   # D
    return g, x, y


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for two in (False, True):
        g, x = capture(n, two)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        x.zero_()
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert float(x[0]) == n * reps, (float(x[0]), n * reps)          # (every hop ran, in order)
        print('%-28s %6.2f us per hop (%d hops x %d replays)' % ('two streams, event per hop:' if two else 'one stream:', dt / (n * reps) * 1e6, n, reps))


def diamonds():
    n = 100
    g, x, y = capture_diamonds(n)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert float(y[0]) == n * (reps + 5) and float(x[0]) == 3 * n * (reps + 5)
    print('%-28s %6.2f us per diamond A -> (B || C) -> D  (serial A, B, C, D would be 4 hops; ideal fork 3)' % ('fork / join:', dt / (n * reps) * 1e6))


if __name__ == '__main__':
    main()
    diamonds()
