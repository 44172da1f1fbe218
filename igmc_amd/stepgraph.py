"""hipGraph replay of one optimisation step.

At batch 50 the step is ~35 short kernels: launch-bound.  The per-step scalars (offset into the
link permutation, epoch, step counter for the dropout hashes, Adam bias corrections) live in a small HBM control
block advanced by the step's own last kernel (``igmc_step_finish``), so the launch sequence is identical every step and
is captured ONCE into a hipGraph (``torch.cuda.CUDAGraph`` capturing the stream the C ABI launches on):

    extract (3 kernels) -> [edge dropout] -> forward / backward / finalize -> [all-reduce] -> step_finish
    (step_finish = Adam + loss + epoch total + control-block advance in one kernel)

Under data parallelism the graph ends before the gradient all-reduce (RCCL runs eagerly on the same stream),
followed by the control-block Adam launch.
"""
import ctypes as C
import os
import struct

import numpy as np
import torch

from . import _lib, engine, parallel


def _ctrl_words(step, first, epoch, adam_t, batch, lr, beta1, beta2, eps, wd):
    """Control block describing the NEXT step to run (the step's last kernel advances it)."""
    w = np.zeros(_lib.CTRL['WORDS'], dtype=np.int64)
    w[_lib.CTRL['STEP']], w[_lib.CTRL['FIRST']], w[_lib.CTRL['EPOCH']] = step, first, epoch
    w[_lib.CTRL['ADAM_T']], w[_lib.CTRL['BATCH']] = adam_t, batch
    step_size = float(lr) / (1.0 - float(beta1) ** adam_t)
    inv_sqrt_bc2 = 1.0 / (1.0 - float(beta2) ** adam_t) ** 0.5
    for key, val in (('LR', lr), ('BETA1', beta1), ('BETA2', beta2), ('EPS', eps), ('WD', wd),
                     ('STEP_SIZE', step_size), ('INV_SQRT_BC2', inv_sqrt_bc2)):
        w[_lib.CTRL[key]] = struct.unpack('<q', struct.pack('<d', float(val)))[0]
    return w


class StepGraph(object):
    """Runs training steps of ``batch_size`` links of ``dataset`` through the fused path, replaying a hipGraph."""

    def __init__(self, model, optimizer, dataset, batch_size, ARR, use_graph=None):
        self.model, self.opt, self.ds = model, optimizer, dataset
        self.B = int(batch_size)
        self.ARR = float(ARR)
        self.lib = _lib.load()
        self.world = parallel.world_size()
        flat = model.flat_parameters()
        self.dev = flat.device
        self.ctrl = torch.zeros(_lib.CTRL['WORDS'], dtype=torch.int64, device=self.dev)
        self.perm = torch.zeros(max(len(dataset), 1) + self.B, dtype=torch.int32, device=self.dev)
        self.arena = dataset.arena(self.B, slot='stepgraph')
        from .util_functions import DeviceBatch
        self._db = DeviceBatch(dataset, self.arena, self.B, self.perm, 0)
        self.ws = model._workspace(self._db)
        self.out = torch.empty(self.B, dtype=torch.float32, device=self.dev)
        self.loss = torch.zeros(2, dtype=torch.float32, device=self.dev)
        self.total = torch.zeros(1, dtype=torch.float64, device=self.dev)
        if use_graph is None:
            use_graph = os.environ.get('IGMC_NO_GRAPH', '0') != '1'
        self.use_graph = use_graph
        self.graph = None
        self._attached = False
        self.steps_done = 0

    # ------------------------------------------------------------------ control block
    def _attach(self):
        self.lib.call('igmc_batch_set_ctrl', self.arena.handle, C.c_void_p(self.ctrl.data_ptr()))
        self.lib.call('igmc_model_set_ctrl', self.ws.handle, C.c_void_p(self.ctrl.data_ptr()))
        self._attached = True

    def detach(self):
        if self._attached:
            self.lib.call('igmc_batch_set_ctrl', self.arena.handle, None)
            self.lib.call('igmc_model_set_ctrl', self.ws.handle, None)
            self._attached = False

    def begin_epoch(self, perm, epoch):
        """``perm``: this rank's link positions for the epoch (1-D int tensor, any device)."""
        n = len(perm)
        self.perm[:n].copy_(perm.to(dtype=torch.int32), non_blocking=False)
        g = self.opt.param_groups[0]
        w = _ctrl_words(self.model._step + 1, 0, epoch if self.ds.dynamic else 0, self.opt.t + 1, self.B, g['lr'],
                        g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'])
        self.ctrl.copy_(torch.from_numpy(w))
        self.total.zero_()
        self.n_links = n
        if not self._attached:
            self._attach()
        if self.graph is not None and abs(self._graph_lr - g['lr']) > 0:
            pass          # lr lives in the control block: no re-capture needed
        self._graph_lr = g['lr']

    # ------------------------------------------------------------------ one step
    def _enqueue(self, B, upto_grad_only=False):
        m, st = self.model, torch.cuda.current_stream().cuda_stream
        flat, grad = m.flat_parameters(), m.flat_grad()
        self.arena.extract(self.ds.link_u.data_ptr(), self.ds.link_v.data_ptr(), self.ds.link_y.data_ptr(),
                           self.perm.data_ptr(), 0, B, self.ds.sample_ratio, self.ds.seed, 0, st)
        use_flags = m.adj_dropout > 0
        if use_flags:
            self.arena.edge_dropout(m.adj_dropout, m.force_undirected, m.seed, 0, st)
        self.ws.loss_grad(flat.data_ptr(), self.arena, self.out.data_ptr(), grad.data_ptr(), None,
                          use_edge_flags=use_flags, seed=m.seed, step=0, multiply_by=float(m.multiply_by),
                          ARR=self.ARR, grad_scale=1.0 / (B * self.world), arr_scale=1.0 / self.world, stream=st)
        if upto_grad_only:
            return
        self._finish(B)

    def _finish(self, B):
        m, st = self.model, torch.cuda.current_stream().cuda_stream
        flat, grad = m.flat_parameters(), m.flat_grad()
        if self.world > 1:
            parallel.all_reduce_sum_(grad)
        # Adam + loss + epoch total + control-block advance in ONE launch (the step's last kernel)
        g = self.opt.param_groups[0]
        self.lib.call('igmc_step_finish', self.ws.handle, self.arena.handle, C.c_void_p(flat.data_ptr()),
                      C.c_void_p(grad.data_ptr()), C.c_void_p(self.opt.exp_avg.data_ptr()),
                      C.c_void_p(self.opt.exp_avg_sq.data_ptr()), self.ARR, C.c_void_p(self.loss.data_ptr()),
                      C.c_void_p(self.total.data_ptr()), C.c_void_p(self.ctrl.data_ptr()), 1, g['lr'],
                      g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], C.c_void_p(st))

    def _capture(self):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._enqueue(self.B, upto_grad_only=self.world > 1)
        self.graph = g

    def step(self, B=None):
        """One optimisation step on the next ``B`` links of the epoch permutation."""
        B = self.B if B is None else int(B)
        full = B == self.B
        if full and self.use_graph and self.graph is None and self.steps_done >= 3:
            # the captured step is NOT executed by the capture, so nothing is skipped or repeated
            self._capture()
        if full and self.graph is not None:
            self.graph.replay()
            if self.world > 1:
                self._finish(B)
        else:
            self._enqueue(B)
        self.steps_done += 1
        self.model._step += 1
        self.opt.t += 1

    def run_epoch(self, perm, epoch):
        """All batches of one epoch; returns (sum over batches of loss*B as a device float64 tensor, #links)."""
        self.begin_epoch(perm, epoch)
        n = self.n_links
        for first in range(0, n, self.B):
            self.step(min(self.B, n - first))
        self.detach()         # captured launches keep their own copy of the control pointer
        return self.total, n
