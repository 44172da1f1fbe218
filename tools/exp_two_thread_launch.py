"""Experiment (GPU): do two hipGraphs launched from two host threads get their packets submitted side by side?  Graph A = a chain
of 60 short kernels (the step chain of a 20-step launch), graph B = a chain of 30 (its extraction chain), each starting with a
wall-clock mark.  Sequential launches from one thread (A then B) against the same two launches from two threads."""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
mk = C.CDLL(os.path.join(ROOT, 'tools', 'ubench', 'libclock_mark.so'))
mk.clock_mark.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device('cuda', 0)
buf = torch.zeros(64, dtype=torch.int64, device=dev)
x = torch.zeros(1 << 16, device=dev)
s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()


def chain(n, slot0):
    mk.clock_mark(C.c_void_p(buf.data_ptr()), slot0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    for _ in range(n):
        torch.cuda._sleep(20000)          # ~10 us each
    mk.clock_mark(C.c_void_p(buf.data_ptr()), slot0 + 1, C.c_void_p(torch.cuda.current_stream().cuda_stream))


ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
chain(2, 0)
torch.cuda.synchronize()
with torch.cuda.graph(ga):
    chain(60, 0)
with torch.cuda.graph(gb):
    chain(30, 2)
for _ in range(2):
    ga.replay(); gb.replay()
torch.cuda.synchronize()


def seq():
    with torch.cuda.stream(s_a):
        ga.replay()
    with torch.cuda.stream(s_b):
        gb.replay()


def par():
    def tb():
        torch.cuda.set_device(0)
        with torch.cuda.stream(s_b):
            gb.replay()
    th = threading.Thread(target=tb)
    th.start()
    with torch.cuda.stream(s_a):
        ga.replay()
    th.join()


for name, fn in (('one thread, A then B', seq), ('two threads', par), ('one thread, A then B', seq), ('two threads', par)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t = buf[:4].cpu().numpy().astype('int64')
    print('%-22s host %.0f us | A: start 0, end +%.0f us | B: start +%.0f us, end +%.0f us'
          % (name, (t1 - t0) * 1e6, (t[1] - t[0]) / 100.0, (t[2] - t[0]) / 100.0, (t[3] - t[0]) / 100.0))
