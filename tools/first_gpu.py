"""First GPU contact: engine-level timing of the hot path on an ml_1m-like synthetic batch
(diagnostic script, not the benchmark)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from igmc_amd import engine, _lib, preprocessing
import scipy.sparse as sp

small = '--small' in sys.argv
t0 = time.time()
if small:
    nu, ni, nnz, hist = 943, 1682, 100000, preprocessing.ML_HIST['ml_100k'][3]
else:
    nu, ni, nnz, hist = preprocessing.ML_HIST['ml_1m']
u, v, r = preprocessing.synth_ml(nu, ni, nnz, hist, seed=0)
print('synth %.1fs' % (time.time() - t0), flush=True)
perm = np.random.default_rng(1).permutation(len(u))
ntest = int(np.ceil(0.1 * len(u)))
tr = perm[ntest:]
A = sp.csr_matrix((r[tr].astype(np.float32), (u[tr], v[tr])), shape=(nu, ni))
lib = _lib.load()
g = engine.Graph(A, 0, lib)
B, mnph = 50, (200 if small else 100)
b = engine.Batch(g, B, 1, mnph)
print('graph bytes', g.hbm_bytes(), 'node_cap', b.node_capacity, 'edge_cap', b.edge_capacity, flush=True)
lu = torch.from_numpy(u[tr].astype(np.int32)).cuda(); lv = torch.from_numpy(v[tr].astype(np.int32)).cuda()
ly = torch.from_numpy(r[tr].astype(np.float32)).cuda()
idx = torch.randperm(len(tr), dtype=torch.int32).cuda()
ws = engine.ModelWorkspace(lib, 0, 5, 4, 4, 0, b.node_capacity, b.edge_capacity, B)
torch.manual_seed(1)
P = (torch.rand(ws.n_params, device='cuda') - 0.5) * 0.3
G = torch.zeros_like(P); M1 = torch.zeros_like(P); M2 = torch.zeros_like(P)
out = torch.zeros(B, device='cuda'); loss = torch.zeros(2, device='cuda')
st = torch.cuda.current_stream().cuda_stream

def step(i, train=True, drop=False):
    b.extract(lu.data_ptr(), lv.data_ptr(), ly.data_ptr(), idx.data_ptr(), i * B, B, 1.0, 1, 0, st)
    if drop:
        b.edge_dropout(0.2, False, 1, i, st)
    if train:
        ws.loss_grad(P.data_ptr(), b, out.data_ptr(), G.data_ptr(), loss.data_ptr(), use_edge_flags=drop, seed=1, step=i, ARR=0.001, stream=st)
        ws.adam_step(P.data_ptr(), G.data_ptr(), M1.data_ptr(), M2.data_ptr(), i + 1, 1e-3, stream=st)
    else:
        ws.forward(P.data_ptr(), b, out.data_ptr(), stream=st)

for i in range(5):
    step(i)
torch.cuda.synchronize()
info = b.info()
print('N', info.num_nodes, 'E', info.num_edges, 'overflow', info.overflow, 'loss', loss.cpu().numpy(), flush=True)
for name, fn in (('train', lambda i: step(i)), ('train+dropout', lambda i: step(i, drop=True)), ('eval', lambda i: step(i, train=False))):
    torch.cuda.synchronize(); t = time.time(); K = 200
    for i in range(K):
        fn(i + 5)
    torch.cuda.synchronize(); dt = (time.time() - t) / K
    print('%s: %.1f us/step  %.0f subgraphs/s' % (name, dt * 1e6, B / dt), flush=True)
engine.profile_enable(lib, True)
for i in range(20):
    step(i + 300, drop=True)
torch.cuda.synchronize()
rows = engine.profile_fetch(lib, 64)
engine.profile_enable(lib, False)
tot = sum(ms for _, ms, _ in rows)
for name, ms, calls in sorted(rows, key=lambda x: -x[1]):
    print('%-22s %8.2f us/call x%3d  %5.1f%%' % (name, ms / calls * 1e3, calls // 20, 100 * ms / tot))
print('sum of kernels per step: %.1f us' % (tot / 20 * 1e3))
print('loss', loss.cpu().numpy())
