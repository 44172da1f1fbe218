"""hipGraph replay of the optimisation step, with the NEXT batch's extraction overlapped.

At batch 50 the step is ~25 short kernels and every serialized kernel costs >= 4.7 us on MI355X even when
trivial (rocprofv3, profiles/): the step is latency-bound, the chip is mostly idle.  Two measures:

* the per-step scalars (offset into the link permutation, epoch, step counter for the dropout hashes, Adam bias
  corrections) live in a small HBM control block advanced by the step's own last kernel (``igmc_step_finish``),
  so the launch sequence is identical every step and is captured ONCE into a hipGraph
  (``torch.cuda.CUDAGraph`` capturing the streams the C ABI launches on);
* enclosing-subgraph extraction depends only on the link indices, so the extraction of batch t+1 runs on a
  second stream (into the other of two arenas) while forward/backward of batch t run on the first:

      main :  forward / backward / finalize (arena t%2) ------------------+-> [all-reduce] -> step_finish
      side :  extract batch t+1 (arena (t+1)%2) [+ its edge dropout] -----+      (Adam + loss + tick)

Under data parallelism the graphs end before the gradient all-reduce (RCCL runs eagerly on the same stream).
"""
import ctypes as C
import os
import struct

import numpy as np
import torch

from . import _lib, parallel


def _ctrl_words(step, first, epoch, adam_t, batch, lr, beta1, beta2, eps, wd, free_run=False):
    """Control block describing the NEXT step to run (the step's last kernel advances it)."""
    w = np.zeros(_lib.CTRL['WORDS'], dtype=np.int64)
    w[_lib.CTRL['FREE_RUN']] = 1 if free_run else 0
    w[_lib.CTRL['READY']] = w[_lib.CTRL['READY_ODD']] = -1      # no batch sits in either arena yet
    w[_lib.CTRL['STEP']], w[_lib.CTRL['FIRST']], w[_lib.CTRL['EPOCH']] = step, first, epoch
    w[_lib.CTRL['FIRST_ODD']], w[_lib.CTRL['K']] = first + batch, 0     # batch k starts at slot[k & 1]
    w[_lib.CTRL['ADAM_T']], w[_lib.CTRL['BATCH']] = adam_t, batch
    step_size = float(lr) / (1.0 - float(beta1) ** adam_t)
    inv_sqrt_bc2 = 1.0 / (1.0 - float(beta2) ** adam_t) ** 0.5
    for key, val in (('LR', lr), ('BETA1', beta1), ('BETA2', beta2), ('EPS', eps), ('WD', wd),
                     ('STEP_SIZE', step_size), ('INV_SQRT_BC2', inv_sqrt_bc2)):
        w[_lib.CTRL[key]] = struct.unpack('<q', struct.pack('<d', float(val)))[0]
    return w


class StepGraph(object):
    """Runs training steps of ``batch_size`` links of ``dataset`` through the fused path."""

    def __init__(self, model, optimizer, dataset, batch_size, ARR, use_graph=None, overlap=None):
        self.model, self.opt, self.ds = model, optimizer, dataset
        self.B = int(batch_size)
        self.ARR = float(ARR)
        self.lib = _lib.load()
        self.world = parallel.world_size()
        flat = model.flat_parameters()
        self.dev = flat.device
        self.ctrl = torch.zeros(_lib.CTRL['WORDS'], dtype=torch.int64, device=self.dev)
        # permutation buffer, padded so that the (discarded) prefetch after the last batch stays in range
        self.perm = torch.zeros(max(len(dataset), 1) + 2 * self.B, dtype=torch.int32, device=self.dev)
        self.arenas = [dataset.arena(self.B, slot='stepgraph0'), dataset.arena(self.B, slot='stepgraph1')]
        from .util_functions import DeviceBatch
        self.ws = model._workspace(DeviceBatch(dataset, self.arenas[0], self.B, self.perm, 0))
        # the matrix-core subgraph kernel reads the dense induced blocks only: where it takes the step, the extraction
        # branch skips the CSR emission (it is produced on demand for inspection)
        if os.environ.get('IGMC_NO_LEAN', '0') != '1':
            for a in self.arenas:
                a.set_lean(self.ws.dense_path(a, self.B) or a.dense_layers(self.ws))
        self.out = torch.empty(self.B, dtype=torch.float32, device=self.dev)
        self.loss = torch.zeros(2, dtype=torch.float32, device=self.dev)
        self.total = torch.zeros(1, dtype=torch.float64, device=self.dev)
        if use_graph is None:
            use_graph = os.environ.get('IGMC_NO_GRAPH', '0') != '1'
        if overlap is None:
            overlap = os.environ.get('IGMC_NO_OVERLAP', '0') != '1'
        self.use_graph, self.overlap = use_graph, overlap
        # the multi-GPU launch structure (graph up to the gradients, eager all-reduce + step_finish) can be forced on
        # one GPU to test it: IGMC_FORCE_DP_PATH=1
        self.dp_path = self.world > 1 or os.environ.get('IGMC_FORCE_DP_PATH', '0') == '1'
        # IGMC_DP_CAPTURE_ALLREDUCE=1: under data parallelism the flat RCCL all-reduce and the Adam launch are captured
        # INTO the step graph (and 8 steps into one launch, like on one GPU) instead of being enqueued eagerly after every
        # replay.  Opt-in: a captured collective could not be validated on more than one GPU where this was written.
        self.dp_capture = os.environ.get('IGMC_DP_CAPTURE_ALLREDUCE', '0') == '1'
        self.side = torch.cuda.Stream(device=self.dev) if overlap else None
        # Free-running prefetch (igmc_hip.h, device-side step control): inside a multi-step graph the model chain and the
        # extraction chain are forked ONCE and joined ONCE; per step they hand-shake through the control block (the fused
        # step's last kernel waits for ready[next parity], a gate kernel in front of each extraction waits for the cursor
        # of its arena to move).  Paths whose fused step ends in k_finalize_ts: the subgraph kernel and the dense per-layer
        # kernels.  A cross-stream dependency of a hipGraph resolves ~9 us after its producer has finished
        # (profiles/r02_step_timeline.txt): per step that was the whole gap budget of the main chain.
        # OPT-IN (IGMC_FREE_RUN=1): +4 % at the headline, parameters bit-identical to the fork / join structure in 6 of 7
        # two-epoch comparisons on the GPU -- but ONE run of the edge-dropout variant diverged in the last GPU seconds of
        # round 2 and its cause is not found yet, so fork + join per step stays the default.
        self.free_run = bool(
            self.side is not None and use_graph and not self.dp_path
            and os.environ.get('IGMC_FREE_RUN', '0') == '1'
            and os.environ.get('IGMC_FIN_MODE', '1') != '0'
            and os.environ.get('IGMC_MAIN_FIRST', '1') == '1'
            # (the fused step must END in k_finalize_ts, the kernel that holds the wait: readout width a multiple of 16)
            and int(getattr(getattr(model, 'lin1', None), 'in_features', 0)) % 16 == 0
            and all(self.ws.dense_path(a, self.B) or a.dense_layers(self.ws) for a in self.arenas))
        self.graphs = [None, None]
        # several steps in ONE graph launch: consecutive launches of a replayed graph are separated by a gap of tens of
        # microseconds on the device, a sizeable part of a ~200 us step (IGMC_GRAPH_STEPS, even, 0 disables)
        self.multi_n = int(os.environ.get('IGMC_GRAPH_STEPS', '8')) & ~1
        self.multi_base = self.multi_n
        self.multi = None
        self._attached = False
        self.k = 0                      # steps done in the current epoch (parity selects the arena)
        self.steps_done = 0

    @property
    def arena(self):                    # the arena holding the batch of the most recent step
        return self.arenas[(self.k - 1) % 2 if self.k else 0]

    # ------------------------------------------------------------------ control block
    def _attach(self):
        for a in self.arenas:
            self.lib.call('igmc_batch_set_ctrl', a.handle, C.c_void_p(self.ctrl.data_ptr()))
        self.lib.call('igmc_model_set_ctrl', self.ws.handle, C.c_void_p(self.ctrl.data_ptr()))
        self._attached = True

    def detach(self):
        if self._attached:
            for a in self.arenas:
                self.lib.call('igmc_batch_set_ctrl', a.handle, None)
            self.lib.call('igmc_model_set_ctrl', self.ws.handle, None)
            self._attached = False

    def _extract(self, arena, slot, B):
        """Extraction (+ edge dropout) of the batch whose offset is in control slot ``slot`` (even/odd) into
        ``arena`` on the current stream."""
        m, st = self.model, torch.cuda.current_stream().cuda_stream
        cache = getattr(self.ds, '_cache', None)
        if cache is not None:       # static dataset (reference MyDataset): node sets from the HBM-resident cache
            arena.extract_cached(cache, self.ds.link_y.data_ptr(), self.perm.data_ptr(), slot, B, st)
        else:
            arena.extract(self.ds.link_u.data_ptr(), self.ds.link_v.data_ptr(), self.ds.link_y.data_ptr(),
                          self.perm.data_ptr(), slot, B, self.ds.sample_ratio, self.ds.seed, 0, st)
        if m.adj_dropout > 0:
            arena.edge_dropout(m.adj_dropout, m.force_undirected, m.seed, slot, st)
        if self.free_run:
            self.lib.call('igmc_batch_mark_ready', arena.handle, int(slot) & 1, C.c_void_p(st))

    def begin_epoch(self, perm, epoch):
        """``perm``: this rank's link positions for the epoch (1-D int tensor, any device)."""
        n = len(perm)
        self.perm[:n].copy_(perm.to(dtype=torch.int32), non_blocking=False)
        self.perm[n:n + 2 * self.B].copy_(self.perm[:2 * self.B] if n >= 2 * self.B else self.perm[n - 1].expand(2 * self.B))
        g = self.opt.param_groups[0]
        w = _ctrl_words(self.model._step + 1, 0, epoch if self.ds.dynamic else 0, self.opt.t + 1, self.B, g['lr'],
                        g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], free_run=self.free_run)
        self.ctrl.copy_(torch.from_numpy(w))
        self.total.zero_()
        self.n_links = n
        self.k = 0
        if not self._attached:
            self._attach()
        # the first batch of the epoch has nobody to prefetch it
        self._extract(self.arenas[0], 0, min(self.B, n))

    # ------------------------------------------------------------------ one step
    def _model(self, arena, B):
        m, st = self.model, torch.cuda.current_stream().cuda_stream
        flat, grad = m.flat_parameters(), m.flat_grad()
        self.ws.loss_grad(flat.data_ptr(), arena, self.out.data_ptr(), grad.data_ptr(), None,
                          use_edge_flags=m.adj_dropout > 0, seed=m.seed, step=0, multiply_by=float(m.multiply_by),
                          ARR=self.ARR, grad_scale=1.0 / (B * self.world), arr_scale=1.0 / self.world, stream=st)

    def _finish(self, arena):
        m, st = self.model, torch.cuda.current_stream().cuda_stream
        flat, grad = m.flat_parameters(), m.flat_grad()
        if self.world > 1 or (self.dp_path and parallel.is_dist()):
            parallel.all_reduce_sum_(grad)
        # Adam + loss + epoch total + control-block advance in ONE launch (the step's last kernel)
        g = self.opt.param_groups[0]
        self.lib.call('igmc_step_finish', self.ws.handle, arena.handle, C.c_void_p(flat.data_ptr()),
                      C.c_void_p(grad.data_ptr()), C.c_void_p(self.opt.exp_avg.data_ptr()),
                      C.c_void_p(self.opt.exp_avg_sq.data_ptr()), self.ARR, C.c_void_p(self.loss.data_ptr()),
                      C.c_void_p(self.total.data_ptr()), C.c_void_p(self.ctrl.data_ptr()), 1, g['lr'],
                      g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], C.c_void_p(st))

    def _train_step(self, arena):
        """Single GPU: forward + loss + backward + Adam (+ loss / total / control-block advance) with the minimum
        number of launches (``igmc_train_step``)."""
        m, st = self.model, torch.cuda.current_stream().cuda_stream
        flat, grad = m.flat_parameters(), m.flat_grad()
        g = self.opt.param_groups[0]
        self.lib.call('igmc_train_step', self.ws.handle, C.c_void_p(flat.data_ptr()), arena.handle,
                      int(m.adj_dropout > 0), None, m.seed & (2 ** 64 - 1), 0, float(m.multiply_by), self.ARR,
                      C.c_void_p(self.out.data_ptr()), C.c_void_p(grad.data_ptr()),
                      C.c_void_p(self.opt.exp_avg.data_ptr()), C.c_void_p(self.opt.exp_avg_sq.data_ptr()),
                      C.c_void_p(self.loss.data_ptr()), C.c_void_p(self.total.data_ptr()),
                      C.c_void_p(self.ctrl.data_ptr()), 1, g['lr'], g['betas'][0], g['betas'][1], g['eps'],
                      g['weight_decay'], C.c_void_p(st))

    def _enqueue(self, parity, B, with_finish=True):
        """model(batch in arenas[parity]) || extract(next batch -> arenas[1-parity]); then finish."""
        cur, nxt = self.arenas[parity], self.arenas[1 - parity]
        main = torch.cuda.current_stream()
        fused = (not self.dp_path) and with_finish and B == self.B      # igmc_train_step: gradients + Adam, minimum launches
        if self.side is not None:
            self.side.wait_stream(main)
            # launch order inside the fork: the model kernels are enqueued BEFORE the extraction branch, so that the
            # subgraph kernel (one workgroup per CU, 224 of 256 CUs) is dispatched first and the extraction workgroups
            # fill what is left; the other order lets ~50 extraction workgroups take CUs first and the cluster
            # members that find no CU stall their whole cluster: 320 k -> 341 k subgraphs/s (IGMC_MAIN_FIRST=0: old order)
            main_first = os.environ.get('IGMC_MAIN_FIRST', '1') == '1'
            if not main_first:
                with torch.cuda.stream(self.side):
                    self._extract(nxt, 1 - parity, self.B)
            if fused:
                # the step's last kernel advances ONLY the control slot of its own parity; the prefetch reads the
                # other one, so the two branches never touch the same word
                self._train_step(cur)
            else:
                self._model(cur, B)
            if main_first:
                with torch.cuda.stream(self.side):
                    self._extract(nxt, 1 - parity, self.B)
            main.wait_stream(self.side)
        else:
            if fused:
                self._extract(nxt, 1 - parity, self.B)
                self._train_step(cur)
            else:
                self._model(cur, B)
                self._extract(nxt, 1 - parity, self.B)
        if with_finish and not fused:
            self._finish(cur)

    def _enqueue_free_running(self, M):
        """M fused steps (starting at an even step) and the M extractions they prefetch as two chains with ONE fork and ONE
        join: step i runs on arena i % 2, the side chain extracts batch i + 1 into arena (i + 1) % 2 behind a gate that
        waits for step i - 1 to have consumed that arena (free-running prefetch, igmc_hip.h)."""
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)
        for i in range(M):
            self._train_step(self.arenas[i % 2])
        with torch.cuda.stream(self.side):
            st = torch.cuda.current_stream().cuda_stream
            for i in range(M):
                p = (i + 1) % 2
                self.lib.call('igmc_batch_gate', self.arenas[p].handle, p, C.c_void_p(st))
                self._extract(self.arenas[p], p, self.B)
        main.wait_stream(self.side)

    def _capture(self, parity):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        if self.world <= 1 and not (self.dp_path and parallel.is_dist()):
            with torch.cuda.graph(g):
                self._enqueue(parity, self.B, with_finish=self._finish_in_graph())
            self.graphs[parity] = g
            return
        # several processes: the RCCL watchdog thread of torch.distributed polls events while this thread captures, so
        # the capture must only police THIS thread ('thread_local'); a refused capture is not fatal -- the same HIP
        # launches then run eagerly (capturing does not execute anything, so no step is lost)
        try:
            if self.dp_capture:
                try:
                    with torch.cuda.graph(g, capture_error_mode='thread_local'):
                        self._enqueue(parity, self.B, with_finish=True)      # all-reduce + Adam inside the graph
                    self.graphs[parity] = g
                    return
                except RuntimeError as e:
                    import sys
                    sys.stderr.write('igmc_amd: RCCL all-reduce not capturable (%s); all-reduce stays eager\n'
                                     % str(e).splitlines()[0])
                    self.dp_capture = False
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                self._enqueue(parity, self.B, with_finish=False)
            self.graphs[parity] = g
        except RuntimeError as e:
            import sys
            sys.stderr.write('igmc_amd: hipGraph capture refused under data parallelism (%s); launching eagerly\n'
                             % str(e).splitlines()[0])
            self.use_graph, self.graphs = False, [None, None]
            torch.cuda.synchronize()

    def _finish_in_graph(self):
        """True when a captured step holds its own weight update (single GPU, or data parallelism with the collective
        captured); False = the graph ends at the local gradients and all-reduce + Adam are enqueued after each replay."""
        return (not self.dp_path) or self.dp_capture

    def prepare(self, steps_hint=None, group=None):
        """Capture every hipGraph this object will replay (both single-step parities and the multi-step group) NOW.
        ``steps_hint``: the caller is about to run exactly that many steps -- the group size becomes the largest even
        divisor of it in [IGMC_GRAPH_STEPS, 2 * IGMC_GRAPH_STEPS] (if any), so that the run is whole groups only (a step
        replayed on its own pays a graph-launch gap of its own).  ``group``: that many steps per graph launch.
        Capturing executes nothing, so no step is skipped or repeated; callers that time a region (bench.py) call this
        after their warm-up so that no capture (milliseconds each) falls inside the timed steps.  Needs at least one
        eagerly executed step before it (first launches load code objects, which a capture cannot do)."""
        if not self.use_graph or self.steps_done < 1 or not self._attached:
            return False
        if group is not None and self.multi_n >= 2 and int(group) >= 2:
            group = int(group) & ~1                              # explicit group size (steps per graph launch)
            if group != self.multi_n:
                self.multi_n, self.multi = group, None
        elif steps_hint and self.multi_n >= 2:
            base = self.multi_base
            for m in range(2 * base, base - 1, -2):
                if int(steps_hint) % m == 0:
                    if m != self.multi_n:
                        self.multi_n, self.multi = m, None      # (a group captured during warm-up had the default size)
                    break
        for parity in (0, 1):
            if self.graphs[parity] is None and self.use_graph:
                self._capture(parity)
        if self.use_graph and self._finish_in_graph() and self.multi_n >= 2 and self.multi is None:
            self._capture_multi()
        return True

    def _capture_multi(self):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        if self.free_run:
            with torch.cuda.graph(g):
                self._enqueue_free_running(self.multi_n)
            self.multi = g
            return
        if self.world <= 1 and not (self.dp_path and parallel.is_dist()):
            with torch.cuda.graph(g):
                for i in range(self.multi_n):
                    self._enqueue(i % 2, self.B)
            self.multi = g
            return
        try:
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                for i in range(self.multi_n):
                    self._enqueue(i % 2, self.B)
            self.multi = g
        except RuntimeError as e:
            import sys
            sys.stderr.write('igmc_amd: multi-step capture refused under data parallelism (%s); one step per launch\n'
                             % str(e).splitlines()[0])
            self.multi_n = 0
            torch.cuda.synchronize()

    def step(self, B=None):
        """One optimisation step on the next ``B`` links of the epoch permutation."""
        B = self.B if B is None else int(B)
        parity = self.k % 2
        if B != self.B:
            # ragged last batch of the epoch: its prefetch assumed a full batch -> extract again, run eagerly
            self._extract(self.arenas[parity], parity, B)
            self._model(self.arenas[parity], B)
            self._finish(self.arenas[parity])
        else:
            if self.use_graph and self.graphs[parity] is None and self.steps_done >= 4:
                self._capture(parity)     # capturing does not execute: nothing is skipped or repeated
            if self.graphs[parity] is not None:
                self.graphs[parity].replay()
                if not self._finish_in_graph():
                    self._finish(self.arenas[parity])
            else:
                self._enqueue(parity, B)
        self.k += 1
        self.steps_done += 1
        self.model._step += 1
        self.opt.t += 1

    def steps(self, n):
        """``n`` full-batch optimisation steps; whole groups of ``multi_n`` steps replay one multi-step graph (the
        per-step control block is advanced on the device, so the launch sequence of a group is always the same)."""
        n = int(n)
        while n > 0:
            M = self.multi_n
            multi_ok = (self.use_graph and self._finish_in_graph() and M >= 2 and self.k % 2 == 0 and
                        (self.steps_done >= 4 or self.multi is not None))
            if multi_ok and self.multi is None:
                # captured as soon as it can be (capturing executes nothing), also when fewer than M steps are asked for
                # right now: the milliseconds a capture costs then fall into the caller's warm-up, not into its first
                # long run
                self._capture_multi()
                multi_ok = self.multi is not None
            if multi_ok and n >= M:
                self.multi.replay()
                self.k += M
                self.steps_done += M
                self.model._step += M
                self.opt.t += M
                n -= M
            else:
                self.step()
                n -= 1

    def run_epoch(self, perm, epoch):
        """All batches of one epoch; returns (sum over batches of loss*B as a device float64 tensor, #links)."""
        self.begin_epoch(perm, epoch)
        n = self.n_links
        self.steps(n // self.B)
        if n % self.B:
            self.step(n % self.B)
        self.detach()         # captured launches keep their own copy of the control pointer
        self.check()
        return self.total, n

    def check(self):
        """Raises if a bounded device-side wait of the step kernels timed out (synchronises the stream)."""
        self.lib.call('igmc_model_check', self.ws.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if self.free_run and int(self.ctrl[_lib.CTRL['SYNC_ERR']].item()) != 0:
            raise RuntimeError('a bounded wait of the free-running prefetch timed out (GPU shared with another job?): '
                               'results of the affected steps are invalid; set IGMC_FREE_RUN=0')
