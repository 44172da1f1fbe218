"""hipGraph replay of the optimisation step in GROUPS, with the next group's extraction overlapped.

At batch 50 a step is a handful of short kernels; what the launch structure around them costs decides the rate
(profiles/r02_step_timeline.txt: a cross-stream dependency of a hipGraph resolves ~9 us after its producer finished, a
graph-launch boundary costs 35-100 us).  Structure:

* the per-step scalars (offset into the link permutation, epoch, step counter for the dropout hashes, Adam bias
  corrections) live in a small HBM control block advanced by the step's own last kernel (``include/igmc_hip.h``,
  device-side step control), so the launch sequence of a step never changes and is captured ONCE;
* steps run in groups of M (``IGMC_GROUP_STEPS``, default 50): the M batches of a group sit in M arenas (one arena set per
  group parity, 288 GB of HBM make that free), and a group is

      main :  step(arena q,0) -> step(arena q,1) -> ... -> step(arena q,M-1) ---------+-> join
      side :  extract(batch 0 of the NEXT group -> arena 1-q,0) -> ... -> (1-q,M-1) --+

  with ONE fork and ONE join.  ONE graph launch is a PAIR of groups (parity 0, then parity 1: 2 M steps), so a single
  captured graph serves every launch.  Both chains are ordinary stream-ordered kernel sequences with real graph
  dependencies at both ends: no per-step cross-stream dependency (the 9 us), no device-side spin waits, nothing that needs
  two branches of a graph to be co-scheduled.  (Round 2's "free-running prefetch" hand-shook per step through the control
  block with spin-wait kernels; it depended on the runtime running both branches concurrently and diverged once -- it is
  gone.)  The extraction chain is shorter than the model chain, so the tail of every group runs with the chip to itself.
* the cursor of a group parity is advanced by the LAST step of that group only, the prefetch reads the other parity's
  cursor: no word is written while another chain may read it; every extraction stamps its arena with the position it
  resolved and the consuming step's tick compares (``sync_err``).

Steps that do not fill a group (the first step of a process -- code objects load on first launch, which a capture cannot
do --, the remainder of an epoch, the ragged last batch) are launched eagerly from the same functions.
Under data parallelism the gradient all-reduce (``igmc_allreduce_grads``: RCCL on the library's own communicator, enqueued on
the step's stream) sits between the gradient kernels and the Adam kernel of every step, INSIDE the captured group; if the
runtime refuses to capture it the same launches run eagerly.
"""
import ctypes as C
import os
import struct
import sys

import numpy as np
import torch

from . import _lib, parallel

MAX_GROUP = 50               # steps of a group (arenas of an arena set); 32 until round 6: a group's fixed costs -- fork / join, its first
                             # step, the first launches of its extraction chain -- are ~60 us, 64.4 -> 63.2 us/step from 25 to 50
SPARSE_CANDIDATES = 128.0    # mean (user degree + item degree) below which extraction launches go batch by batch (_chunk)
GROUP_EXTRACT_CHUNK = 2      # batches per extraction launch (igmc_extract_group; measured: profiles/r03_extract_chunk_sweep.txt); 0 / 1 = arena by arena


def _ctrl_words(step, epoch, adam_t, batch, group, lr, beta1, beta2, eps, wd):
    """Control block describing the NEXT step to run (the step's last kernel advances it): a group of ``group`` steps
    starts at link offset 0, the following one at ``group * batch``."""
    w = np.zeros(_lib.CTRL['WORDS'], dtype=np.int64)
    w[_lib.CTRL['STEP']], w[_lib.CTRL['EPOCH']], w[_lib.CTRL['K']] = step, epoch, 0
    w[_lib.CTRL['FIRST']], w[_lib.CTRL['FIRST_ODD']] = 0, group * batch
    w[_lib.CTRL['GROUP']], w[_lib.CTRL['GK']], w[_lib.CTRL['GQ']] = group, 0, 0
    w[_lib.CTRL['ADAM_T']], w[_lib.CTRL['BATCH']] = adam_t, batch
    step_size = float(lr) / (1.0 - float(beta1) ** adam_t)
    inv_sqrt_bc2 = 1.0 / (1.0 - float(beta2) ** adam_t) ** 0.5
    for key, val in (('LR', lr), ('BETA1', beta1), ('BETA2', beta2), ('EPS', eps), ('WD', wd),
                     ('STEP_SIZE', step_size), ('INV_SQRT_BC2', inv_sqrt_bc2)):
        w[_lib.CTRL[key]] = struct.unpack('<q', struct.pack('<d', float(val)))[0]
    return w


def _group_size(steps_hint, default):
    """Largest M <= MAX_GROUP with 2 M | ``steps_hint`` (a run of a known length is then whole graph launches of 2 M
    steps); ``default`` when only tiny groups divide.  A run of at most MAX_GROUP steps is ONE group (a single-group
    launch, ``GroupPipeline.steps``): two groups of ten cost 70.0 us/step where one launch of 2 x 20 costs 67.6 -- a group's
    fixed costs (its boundary, the first launches of its extraction chain) weigh on every step of it."""
    n = int(steps_hint)
    if 4 <= n <= MAX_GROUP:
        return n
    for m in range(MAX_GROUP, 3, -1):
        if n % (2 * m) == 0:
            return m
    return default


EVAL_MAX_GROUP = 32          # evaluation passes: groups of 50 were slower (1.29 M against 1.33 M subgraphs/s over 20 000 links, round 6)


def _group_size_for(n_steps, cap=EVAL_MAX_GROUP):
    """Group size M <= ``cap`` for a run of ``n_steps`` steps that is NOT a multiple of a convenient length (an evaluation pass
    over a test set): the M that leaves the fewest steps outside whole graph launches of 2 M steps -- those run eagerly,
    extraction and model step one after the other, at about twice the cost --, the larger M on a tie."""
    n = int(n_steps)
    cap = max(1, min(int(cap), MAX_GROUP, max(1, n // 2)))
    best, rest = cap, n % (2 * cap)
    for m in range(cap, min(cap, 7), -1):
        r = n % (2 * m)
        if r < rest:
            best, rest = m, r
    return best


_SIDE_STREAMS = {}


def _side_stream(dev):
    """The extraction chain's stream of a device: ONE stream of the process's own per device (hipStreamCreateWithFlags,
    non-blocking), never one of torch's pooled streams.  torch hands out 32 pooled streams round robin and ``torch.cuda.graph``
    captures on one of them; a process that has captured a few dozen two-branch step graphs -- a test session, a sweep -- gets a
    pooled stream back in the other role, and the HIP runtime then crashes in ``hip::Graph::UpdateStreams`` at the next replay
    (``profiles/r06_experiments/README.md``: found with the test files in another order; the round-5 tree has it too)."""
    if os.environ.get('IGMC_POOLED_SIDE_STREAM', '0') == '1':          # (A/B hook: torch's pooled stream, as before round 6)
        return torch.cuda.Stream(device=dev)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(key)
    if st is None:
        hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
        h = C.c_void_p()
        with torch.cuda.device(key):
            rc = hip.hipStreamCreateWithFlags(C.byref(h), 1)          # hipStreamNonBlocking
        if rc != 0 or not h.value:
            return torch.cuda.Stream(device=dev)                       # (no runtime handle: the pooled stream, as before)
        st = _SIDE_STREAMS[key] = torch.cuda.ExternalStream(h.value, device=dev)
    return st


class GroupPipeline(object):
    """Host logic of the grouped step pipeline -- which extraction / step / re-grouping is launched when --, independent of
    what launches them.  A backend supplies ``_arena(q, i)``, ``_extract(arena, sel, B)``, ``_enqueue_step(arena, B)``,
    ``_dev_regroup(first_cur, first_next)``, ``_fork()`` / ``_join()`` (two chains: model steps || extraction of the next
    group), ``_capture()`` (a replayable graph of ``_enqueue_pair`` or None) and ``_count(n)``.  :class:`StepGraph` is the
    HIP backend; the CPU tests drive the same logic with the emulation build of the kernels under ``gloo``."""

    def _init_pipeline(self, batch_size, group, use_graph):
        self.B = int(batch_size)
        self.M_default = max(1, min(MAX_GROUP, int(group)))
        self.M = self.M_default
        self.use_graph = bool(use_graph)
        self.graph = None
        self.graph1 = [None, None]      # single-group launches: the group of parity q alone (+ the next group's extraction)
        self.single_launch = False      # prepare(): the run it was told about is ONE single-group launch
        self.pacing_fallback = None     # '1' once the extraction chain's pacing gates kept timing out (StepGraph.check)
        self.n_links = 0
        self.k = 0                      # steps done in the current epoch
        self.gq, self.gk = 0, 0         # parity of the current group, steps done in it (mirrors of the control block)
        self.avail = 0                  # batches 0 .. avail-1 of the current group sit extracted in its arenas
        self.steps_done = 0
        self._last = None

    def _reset_epoch(self, n_links):
        self.n_links = int(n_links)
        self.k, self.gq, self.gk, self.avail = 0, 0, 0, 0

    def _extract_many(self, q, count):
        """Batches 0 .. count-1 of the group of parity ``q`` into its arenas: ONE launch per extraction stage for all of them
        where the backend can (extraction is a dependent chain per link -- its throughput is the number of links in flight),
        else arena by arena."""
        if count > 0 and not self._extract_group(q, count):
            for i in range(count):
                self._extract(self._arena(q, i), q | (i << 1), self.B)

    def _extract_group(self, q, count):
        return False

    def _extract_plan(self, q, count):
        """The extraction of batches 0 .. count-1 of the group of parity ``q`` as a list of launches (callables), in order."""
        return [lambda: self._extract_many(q, count)]

    def _fill_group(self):
        """Eager extraction of the batches of the current group that lie inside the epoch (group parity 0, from batch 0)."""
        assert self.gq == 0 and self.gk == 0
        cnt = min(self.M, max(0, self.n_links // self.B - self.k))
        self._extract_many(0, cnt)
        self.avail = cnt

    def _regroup(self):
        """A new group starts at the current position: cursors re-based on the device, its batches extracted eagerly."""
        self._dev_regroup(self.k * self.B, (self.k + self.M) * self.B)
        self.gq, self.gk = 0, 0
        self._fill_group()

    def _enqueue_group(self, q, standalone=False):
        """M steps on the arenas of parity ``q`` || extraction of the next group's M batches into the other set.
        ``standalone``: the group is a launch of its own (not the second half of a pair): nothing is known about what ran in
        front of its first step."""
        cur = [self._arena(q, i) for i in range(self.M)]
        for i in range(self.M):
            self._arena(1 - q, i)
        # PACED extraction (default where the subgraph kernel takes the steps): launch j of the next group's extraction is held
        # back until step j * per of this group has begun -- and 10 us longer.  The subgraph kernel's workgroups need whole CUs
        # (two waves per SIMD with 256 registers each): once its 200-224 workgroups are on the chip an extraction launch runs in
        # the CUs it leaves and costs the step ~3 us; dispatched TOGETHER with it (or just before it) the extraction's
        # workgroups take CUs the subgraph kernel then waits for: +14 us on that step, and as a launch of two batches lasts about
        # as long as a step the two stay in lock-step for several steps (tools/exp_group_timeline.py: in-graph wall-clock
        # marks; profiles/r05_experiments/group_timeline_*.txt).
        # HOW a launch is held back (IGMC_EXTRACT_PACED): 2 (default) = a one-wave GATE kernel in front of it on the extraction
        # chain that polls the control block's step counter and then waits IGMC_GATE_DELAY_US (igmc_ctrl_gate) -- no edge
        # leaves the step chain; 1 = an edge from the end of step j * per - 1 (the launch then starts with the step: 75.5 vs
        # 74.3 us/step in the 20-step form); 0 = not at all (76.1).  Elsewhere (dense-layer kernels, sort-pool family) the
        # extraction chain runs free beside the group's steps: the longer chain of the cap-200 arenas, confined to the CUs the
        # dense-layer launches leave, would reach the group's join late (125.8 -> 129.9 us, round 4).
        plan = self._extract_plan(1 - q, self.M)
        mode = os.environ.get('IGMC_EXTRACT_PACED', (self.pacing_fallback or '2') if self._paced_default() else '0')
        paced = mode != '0' and len(plan) > 1
        per = max(1, self.M // max(1, len(plan)))
        self._fork()
        nxt = 0
        for i in range(self.M):
            mark = None
            if paced and nxt < len(plan) and i == nxt * per and (i > 0 or mode == '2'):
                mark = self._mark() if mode == '1' else ('gate', q, i)
            if (q == 1 and not standalone) or i > 0:
                # inside a launch nothing but the previous step touches the parameters: that step left the weight
                # images of its updated parameters behind, this one starts with the subgraph kernel
                self._hint_unchanged()
            # (the model kernels are enqueued BEFORE the extraction launch that becomes ready with them: capturing the group's
            #  first extraction launch in front of its first step makes the launch start before the subgraph kernel is on the
            #  chip -- 78.0 vs 74.3 us/step in the 20-step form, profiles/r05_experiments)
            self._enqueue_step(cur[i], self.B)
            if paced and nxt < len(plan) and i == nxt * per:
                self._side_after(mark, plan[nxt])
                nxt += 1
        for fn in plan[nxt:]:
            self._side(fn)
        self._join()

    def _hint_unchanged(self):
        """Backends that keep weight images tell the library that the parameters are those of the previous step."""

    def _paced_default(self):
        return False

    def _mark(self):
        """A point of the main chain the extraction chain can wait for (backends with two chains)."""
        return None

    def _side_after(self, mark, fn):
        """``fn`` on the extraction chain, not before ``mark`` (None: no further dependency)."""
        self._side(fn)

    def _enqueue_pair(self):
        """One graph launch: the group of parity 0, then the group of parity 1 (2 M steps)."""
        self._enqueue_group(0)
        self._enqueue_group(1)

    def _advance(self, n):
        self.k += n
        self.steps_done += n
        self._count(n)

    def _single(self, B=None):
        """One eagerly launched step on the next batch (``B`` links: the ragged last batch of an epoch)."""
        full = B is None or int(B) == self.B
        B = self.B if B is None else int(B)
        arena = self._arena(self.gq, self.gk)
        if self.gk >= self.avail or not full:
            self._extract(arena, self.gq | (self.gk << 1), B)      # (a prefetch assumed a full batch)
            self.avail = self.gk + 1 if full else self.avail
        else:
            # prefetched -- possibly by a REPLAYED launch, which leaves no trace on the host: the arena's last extraction CALL
            # may be an older one of another size (the ragged last batch of the epoch before)
            self._assume_full(arena)
        self._enqueue_step(arena, B)
        self._last = arena
        self._advance(1)
        self.gk += 1
        if self.gk >= self.M:
            self.gq, self.gk, self.avail = self.gq ^ 1, 0, 0

    def _assume_full(self, arena):
        """Backends whose arenas remember the size of their last extraction call are told it is a full batch."""

    def step(self, B=None):
        """One optimisation step on the next ``B`` links of the epoch permutation (eager launches)."""
        self._single(B)

    def steps(self, n):
        """``n`` full-batch optimisation steps: whole pairs of groups replay one graph launch each (the control block is
        advanced on the device, so the launch sequence is always the same), the rest is launched eagerly."""
        n = int(n)
        while n > 0:
            M = self.M
            if n >= 2 * M and (self.steps_done >= 1 or not self.use_graph) and self.n_links // self.B - self.k >= 2 * M:
                if self.gq != 0 or self.gk != 0 or self.avail < M:
                    self._regroup()
                g = self._capture() if self.use_graph else None
                if g is not None:
                    g.replay()
                else:
                    self._enqueue_pair()
                self._last = self._arena(1, M - 1)
                self._advance(2 * M)
                self.gq, self.gk, self.avail = 0, 0, M
                n -= 2 * M
            elif n >= M and M >= 2 and (self.steps_done >= 1 or not self.use_graph) and self.gk == 0 and self.avail >= M and \
                    self.n_links // self.B - self.k >= M:
                # what is left is at least ONE group, extracted and waiting in its arenas: a single-group launch (the group of
                # the current parity + the extraction of the group behind it) instead of M eager steps
                q = self.gq
                g = self._capture_single(q) if self.use_graph else None
                if g is not None:
                    g.replay()
                else:
                    self._enqueue_group(q, standalone=True)
                self._last = self._arena(q, M - 1)
                self._advance(M)
                self.gq, self.gk, self.avail = q ^ 1, 0, M
                n -= M
            else:
                self._single()
                n -= 1

    def _capture_single(self, q):
        """A replayable graph of ``_enqueue_group(q, standalone=True)`` or None (backends without graphs)."""
        return None


class StepGraph(GroupPipeline):
    """Runs training steps of ``batch_size`` links of ``dataset`` through the fused path (HIP backend of the pipeline)."""
    SLOT = 'stepgraph'          # name of the arenas in the dataset's arena table
    TRAINING = True             # edge dropout drawn behind every extraction, gradient exchange under data parallelism

    def __init__(self, model, optimizer, dataset, batch_size, ARR, use_graph=None, overlap=None, group=None):
        self.model, self.opt, self.ds = model, optimizer, dataset
        self.ARR = float(ARR)
        self.lib = _lib.load()
        self.world = parallel.world_size()
        flat = model.flat_parameters()
        self.dev = flat.device
        env = os.environ.get('IGMC_GROUP_STEPS', os.environ.get('IGMC_GRAPH_STEPS', str(MAX_GROUP)))
        if use_graph is None:
            use_graph = os.environ.get('IGMC_NO_GRAPH', '0') != '1'
        if overlap is None:
            overlap = os.environ.get('IGMC_NO_OVERLAP', '0') != '1'
        self._init_pipeline(batch_size, group if group is not None else env, use_graph)
        self.overlap = overlap
        self.ctrl = torch.zeros(_lib.CTRL['WORDS'], dtype=torch.int64, device=self.dev)
        # permutation buffer, padded so that the (discarded) prefetch of the group after the last one stays in range: a
        # launch that starts at step k <= n/B - 2 M reads positions below (k + 3 M) B <= n + M B
        self.pad = 4 * MAX_GROUP * self.B          # (twice that: a launch started anywhere inside the epoch stays in range)
        self.perm = torch.zeros(max(len(dataset), 1) + self.pad, dtype=torch.int32, device=self.dev)
        self.sets = [[], []]                   # arenas of the even / odd groups (created on first use)
        self.ws = None
        self.sp = None                         # sort-pool readout family (DGCNN_RS): its own step kernels
        self._attached = False
        self._lean = True                      # every arena so far extracts lean (dense blocks only)
        self._sets = {}                        # engine.BatchSet per group parity (group extraction)
        self._arena(0, 0)
        self.out = torch.empty(self.B, dtype=torch.float32, device=self.dev)
        self.loss = torch.zeros(2, dtype=torch.float32, device=self.dev)
        self.total = torch.zeros(1, dtype=torch.float64, device=self.dev)
        # the multi-GPU step (gradient kernels -> all-reduce -> Adam launch) can be forced on one GPU to test it
        self.dp_path = self.TRAINING and (self.world > 1 or os.environ.get('IGMC_FORCE_DP_PATH', '0') == '1')
        self.side = _side_stream(self.dev) if overlap else None
        # gradient exchange: the library's own RCCL communicator (igmc_allreduce_grads), enqueued on the step's stream
        self.comm = parallel.grad_comm(self.lib, self.dev.index if self.dev.index is not None else 0) if self.dp_path else None
        if self.comm is not None and not getattr(self.comm, 'capturable', True):
            self.use_graph = False             # (an exchange that synchronises the stream: eager launches)

    # ------------------------------------------------------------------ arenas
    def _arena(self, q, i):
        """Arena i of the set of group parity q."""
        from .util_functions import DeviceBatch
        while len(self.sets[q]) <= i:
            a = self.ds.arena(self.B, slot='%s%d.%d' % (self.SLOT, q, len(self.sets[q])))
            # models the subgraph kernel does not take although the dense blocks exist (sort-pool readout, side features):
            # with the transposed blocks their conv layers run on the matrix cores (k_dl_*) instead of walking CSR rows
            if hasattr(self.model, '_sortpool') or getattr(self.model, 'side_features', False):
                a.want_transposed()
            self.sets[q].append(a)
            if self.ws is None:
                probe = DeviceBatch(self.ds, a, self.B, self.perm, 0)
                if hasattr(self.model, '_sortpool'):
                    self.sp = self.model._sortpool(probe)
                self.ws = self.model._workspace(probe)
            # the matrix-core subgraph kernel reads the dense induced blocks only: where it takes the step, the
            # extraction skips the CSR emission (it is produced on demand for inspection)
            lean = False
            if self.sp is None and os.environ.get('IGMC_NO_LEAN', '0') != '1':
                lean = bool(self.ws.dense_path(a, self.B) or a.dense_layers(self.ws))
                a.set_lean(lean)
            self._lean = self._lean and lean
            self.lib.call('igmc_batch_set_ctrl', a.handle, C.c_void_p(self.ctrl.data_ptr()) if self._attached else None)
        return self.sets[q][i]

    @property
    def arenas(self):                   # (the first arena of each set: what geometry / path queries look at)
        return [self._arena(0, 0), self._arena(1, 0)]

    @property
    def arena(self):                    # the arena holding the batch of the most recent step
        return self._last if self._last is not None else self.sets[0][0]

    # ------------------------------------------------------------------ control block
    def _attach(self):
        for s in self.sets:
            for a in s:
                self.lib.call('igmc_batch_set_ctrl', a.handle, C.c_void_p(self.ctrl.data_ptr()))
        self.lib.call('igmc_model_set_ctrl', self.ws.handle, C.c_void_p(self.ctrl.data_ptr()))
        self._attached = True

    def detach(self):
        if self._attached:
            for s in self.sets:
                for a in s:
                    self.lib.call('igmc_batch_set_ctrl', a.handle, None)
            self.lib.call('igmc_model_set_ctrl', self.ws.handle, None)
            self._attached = False

    def _extract(self, arena, sel, B):
        """Extraction (+ edge dropout) of the batch selected by ``sel`` = q | (i << 1) (batch i of the group of parity q,
        resolved on the device from the control block) into ``arena`` on the current stream."""
        m, st = self.model, torch.cuda.current_stream().cuda_stream
        cache = getattr(self.ds, '_cache', None)
        if cache is not None:       # static dataset (reference MyDataset): node sets from the HBM-resident cache
            arena.extract_cached(cache, self.ds.link_y.data_ptr(), self.perm.data_ptr(), sel, B, st)
        else:
            arena.extract(self.ds.link_u.data_ptr(), self.ds.link_v.data_ptr(), self.ds.link_y.data_ptr(),
                          self.perm.data_ptr(), sel, B, self.ds.sample_ratio, self.ds.seed, 0, st)
        if self.TRAINING and m.adj_dropout > 0:
            arena.edge_dropout(m.adj_dropout, m.force_undirected, m.seed, sel, st)

    def _chunk(self):
        """Batches per extraction launch: a whole group in ONE launch keeps the subgraph kernel's clusters off the chip while
        it runs (thousands of small workgroups), one batch per launch runs beside 60 % of the steps; chunks in between."""
        env = os.environ.get('IGMC_GROUP_EXTRACT_CHUNK')
        if env is not None:
            c = int(env)
        else:
            c = GROUP_EXTRACT_CHUNK
            # sparse rating graphs (the Monti sets: the two neighbourhoods of a link hold fewer than SPARSE_CANDIDATES nodes on
            # average): a batch's extraction is so short that one launch sequence per batch, beside every step, disturbs the
            # steps less than a launch of two beside every other step -- douban 63.5 -> 62.6, flixster 88.3 -> 87.0 us/step;
            # ml_100k (165 candidates) 103.0 -> 103.5, the MovieLens-1M shape (435) 64.3 -> 66.4 (round 6, same box)
            g = getattr(self.ds, 'graph', None)
            if self.TRAINING and g is not None and g.nnz * (1.0 / max(1, g.n_users) + 1.0 / max(1, g.n_items)) < SPARSE_CANDIDATES:
                c = 1
        return max(0, min(c, self.M))

    def _batch_sets(self, q, count=0):
        """The arena set of group parity ``q`` as engine.BatchSets of ``_chunk()`` arenas each (created outside any capture),
        or None where group extraction does not apply."""
        c = self._chunk()
        if c < 2 or getattr(self.ds, '_cache', None) is not None or getattr(self.ds, '_side', None) is not None or \
                not self._attached:
            return None
        n = max(count, self.M)
        arenas = [self._arena(q, i) for i in range(n)]
        if not self._lean:
            return None
        key = (q, c)
        sets = self._sets.get(key)
        if sets is None or sum(len(b.arenas) for b in sets) < n:
            from . import engine
            sets = self._sets[key] = [engine.BatchSet(arenas[i0:i0 + c]) for i0 in range(0, n, c)]
        return sets

    def _extract_group(self, q, count):
        """igmc_extract_group: chunks of the group in one launch per stage (dynamic datasets, lean arenas with dense blocks,
        no side features); False = not applicable here."""
        m = self.model
        sets = self._batch_sets(q, count)
        if sets is None:
            return False
        st = torch.cuda.current_stream().cuda_stream
        i0 = 0
        for bs in sets:
            n = min(len(bs.arenas), count - i0)
            if n <= 0:
                break
            bs.extract(n, self.ds.link_u.data_ptr(), self.ds.link_v.data_ptr(), self.ds.link_y.data_ptr(),
                       self.perm.data_ptr(), q | (i0 << 1), self.B, self.ds.sample_ratio, self.ds.seed,
                       drop_p=m.adj_dropout if (self.TRAINING and m.adj_dropout > 0) else 0.0, force_undirected=m.force_undirected,
                       drop_seed=m.seed, stream=st)
            i0 += n
        return True

    def _extract_plan(self, q, count):
        m = self.model
        sets = self._batch_sets(q, count)
        if sets is None:
            return [(lambda i=i: self._extract(self._arena(q, i), q | (i << 1), self.B)) for i in range(count)]
        plan, i0 = [], 0
        for bs in sets:
            n = min(len(bs.arenas), count - i0)
            if n <= 0:
                break

            def launch(bs=bs, n=n, i0=i0):
                bs.extract(n, self.ds.link_u.data_ptr(), self.ds.link_v.data_ptr(), self.ds.link_y.data_ptr(),
                           self.perm.data_ptr(), q | (i0 << 1), self.B, self.ds.sample_ratio, self.ds.seed,
                           drop_p=m.adj_dropout if (self.TRAINING and m.adj_dropout > 0) else 0.0,
                           force_undirected=m.force_undirected, drop_seed=m.seed,
                           stream=torch.cuda.current_stream().cuda_stream)
            plan.append(launch)
            i0 += n
        return plan

    def _step_form(self):
        if getattr(self, '_form', None) is None:
            self._form = 0 if self.sp is not None else self.ws.step_form(self._arena(0, 0), self.B)
        return self._form

    def _paced_default(self):
        # subgraph kernel, and the group-split dense-layer kernels (cap <= 128 arenas, two relation groups: an extraction
        # launch is shorter than a step there too, and running free it lands on the boundaries of the step's launches --
        # flixster 89.5 -> 88.2, ml_10m_lite 100.0 -> 98.0 us/step, round 6); the cap-200 arenas' longer chain runs free
        # (ml_100k: 102.8 free, 103.9 .. 106.6 gated)
        return self.pacing_policy(self._step_form(), self.TRAINING)[0]

    @classmethod
    def pacing_policy(cls, form, training):
        """(extraction launches held back by gates?, microseconds into the step at which a gate opens) for the kernels a step
        is made of (``igmc_model_step_form``: 1 subgraph kernel, 2 dense-layer kernels, 3 their group-split form, 0 per-layer
        kernels).  Evaluation passes run the chain free: their forward-only launches are short, and since the extraction kernels got
        faster the gates cost more than they save there (1.333 M -> 1.367 M subgraphs/s over 20 000 links, round 6)."""
        if form == 1 and training:
            return True, cls.GATE_DELAY_US
        if form == 3 and training:
            return True, cls.GATE_DELAY_DL_US
        return False, cls.GATE_DELAY_US

    def _mark(self):
        if self.side is None:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    GATE_TIMEOUT_US = 2000.0      # a gate that has not seen its step after this long lets the extraction go (a hint only)
    GATE_DELAY_US = 10.0          # ... and it opens this long after its step began (IGMC_GATE_DELAY_US): the step's subgraph kernel
                                  # is on the chip by then, the extraction launch takes the CUs it leaves
    GATE_DELAY_DL_US = 40.0       # ... dense-layer kernels: once the step's SECOND launch (k_dl_bwd, ~31 us in) is on the chip -- a
                                  # gate that opens around that boundary costs most (delay 10: 98.1 .. 103.6 us/step on ml_10m_lite)

    def _side_after(self, mark, fn):
        if self.side is not None and mark is not None:
            if isinstance(mark, tuple):         # ('gate', q, steps of the group that must be done)
                # (the first launch of a group has no step to wait for: it is released at the group's start, together with
                #  the group's first step -- the delay applies to it unconditionally)
                delay = float(os.environ.get('IGMC_GATE_DELAY_US', self.pacing_policy(self._step_form(), self.TRAINING)[1]))
                self.lib.call('igmc_ctrl_gate', C.c_void_p(self.ctrl.data_ptr()), int(mark[1]), int(mark[2]), delay,
                              1 if int(mark[2]) == 0 else 0, self.GATE_TIMEOUT_US, C.c_void_p(self.side.cuda_stream))
            else:
                self.side.wait_event(mark)
        self._side(fn)

    def begin_epoch(self, perm, epoch):
        """``perm``: this rank's link positions for the epoch (1-D int tensor, any device)."""
        n = len(perm)
        self.perm[:n].copy_(perm.to(dtype=torch.int32), non_blocking=False)
        wrap = torch.arange(self.pad, device=self.dev) % max(n, 1)
        self.perm[n:n + self.pad].copy_(self.perm[:max(n, 1)].index_select(0, wrap))
        g = self.opt.param_groups[0] if self.opt is not None else dict(lr=0.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
        w = _ctrl_words(self.model._step + 1, epoch if self.ds.dynamic else 0, (self.opt.t if self.opt is not None else 0) + 1,
                        self.B, self.M, g['lr'], g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'])
        self.ctrl.copy_(torch.from_numpy(w))
        self.total.zero_()
        self._reset_epoch(n)
        if not self._attached:
            self._attach()
        self._fill_group()              # the first group of the epoch has nobody to prefetch it

    # ------------------------------------------------------------------ backend hooks of the pipeline
    def _dev_regroup(self, first_cur, first_next):
        self.lib.call('igmc_ctrl_regroup', C.c_void_p(self.ctrl.data_ptr()), self.M, first_cur, first_next,
                      C.c_void_p(torch.cuda.current_stream().cuda_stream))

    def _fork(self):
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())

    def _side(self, fn):
        if self.side is not None:
            with torch.cuda.stream(self.side):
                fn()
        else:
            fn()

    def _join(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def _count(self, n):
        self.model._step += n
        self.opt.t += n

    def _hint_unchanged(self):
        if self.sp is None:
            self.lib.call('igmc_model_weights_unchanged', self.ws.handle, 1)

    def _assume_full(self, arena):
        arena.assume_size(self.B)

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
        if self.comm is not None:
            self.comm.all_reduce_(grad, st)          # ONE flat all-reduce (RCCL), same stream: capturable
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

    def _train_step_dp(self, arena):
        """Data parallel: ``igmc_train_step_dp`` -- the single-GPU step's kernels with the exchange of the step's reduced
        gradient sources between their reduction and the gradient / Adam kernel (one grouped RCCL call on this stream)."""
        m, st = self.model, torch.cuda.current_stream().cuda_stream
        flat, grad = m.flat_parameters(), m.flat_grad()
        g = self.opt.param_groups[0]
        self.lib.call('igmc_train_step_dp', self.ws.handle, self.comm.handle if self.comm is not None else None,
                      C.c_void_p(flat.data_ptr()), arena.handle,
                      int(m.adj_dropout > 0), None, m.seed & (2 ** 64 - 1), 0, float(m.multiply_by), self.ARR,
                      C.c_void_p(self.out.data_ptr()), C.c_void_p(grad.data_ptr()),
                      C.c_void_p(self.opt.exp_avg.data_ptr()), C.c_void_p(self.opt.exp_avg_sq.data_ptr()),
                      C.c_void_p(self.loss.data_ptr()), C.c_void_p(self.total.data_ptr()),
                      C.c_void_p(self.ctrl.data_ptr()), 1, g['lr'], g['betas'][0], g['betas'][1], g['eps'],
                      g['weight_decay'], C.c_void_p(st))

    def _sortpool_step(self, arena, B):
        """DGCNN_RS (reference models.py:123-167): conv kernels + sort-pool readout forward / backward -> [all-reduce] ->
        Adam + loss + tick (igmc_sortpool_step_finish), every scalar from the control block: capturable like IGMC's."""
        m, st = self.model, torch.cuda.current_stream().cuda_stream
        flat, grad = m.flat_parameters(), m.flat_grad()
        self.sp.loss_grad(flat.data_ptr(), arena, self.out.data_ptr(), grad.data_ptr(), None,
                          use_edge_flags=m.adj_dropout > 0, seed=m.seed, step=0, ARR=self.ARR,
                          grad_scale=1.0 / (B * self.world), arr_scale=1.0 / self.world, stream=st)
        if self.comm is not None:
            self.comm.all_reduce_(grad, st)
        g = self.opt.param_groups[0]
        self.lib.call('igmc_sortpool_step_finish', self.sp.handle, arena.handle, C.c_void_p(flat.data_ptr()),
                      C.c_void_p(grad.data_ptr()), C.c_void_p(self.opt.exp_avg.data_ptr()),
                      C.c_void_p(self.opt.exp_avg_sq.data_ptr()), self.ARR, C.c_void_p(self.loss.data_ptr()),
                      C.c_void_p(self.total.data_ptr()), C.c_void_p(self.ctrl.data_ptr()), 1, g['lr'],
                      g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], C.c_void_p(st))

    def _enqueue_step(self, arena, B):
        """The kernels of one optimisation step on the batch in ``arena`` (current stream)."""
        if self.sp is not None:
            self._sortpool_step(arena, B)
        elif self.dp_path and os.environ.get('IGMC_DP_FLAT', '0') != '1':
            self._train_step_dp(arena)       # the exchange inside the step (IGMC_DP_FLAT=1: the three-call sequence below)
        elif not self.dp_path and B == self.B:
            self._train_step(arena)          # gradients + Adam in the minimum number of launches
        else:
            self._model(arena, B)
            self._finish(arena)

    @property
    def graphs(self):                   # (compatibility of older call sites: the captured graphs)
        return [self.graph] + list(self.graph1)

    def drop_graphs(self):
        """Forget every captured graph (they are captured again on their next use)."""
        self.graph, self.graph1 = None, [None, None]

    def _capture(self):
        if self.graph is not None or not self.use_graph:
            return self.graph
        self.graph = self._capture_fn(self._enqueue_pair)
        return self.graph

    def _capture_single(self, q):
        if self.graph1[q] is not None or not self.use_graph:
            return self.graph1[q]
        self.graph1[q] = self._capture_fn(lambda: self._enqueue_group(q, standalone=True))
        return self.graph1[q]

    def _capture_fn(self, enqueue):
        """The launches of ``enqueue()`` as a hipGraph, or None (capture refused under data parallelism: eager from then on)."""
        for qq in (0, 1):                   # (arenas and arena sets are created on first use: never inside a capture)
            self._arena(qq, self.M - 1)
            self._batch_sets(qq)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        dist = parallel.is_dist()
        try:
            # several processes: the watchdog thread of torch.distributed polls events while this thread captures, so the
            # capture must only police THIS thread; a refused capture is not fatal -- the same launches then run eagerly
            # (capturing executes nothing, so no step is lost)
            with torch.cuda.graph(g, **(dict(capture_error_mode='thread_local') if dist else {})):
                enqueue()
        except RuntimeError as e:
            if self.comm is None and not dist:
                raise
            sys.stderr.write('igmc_amd: hipGraph capture refused under data parallelism (%s); launching eagerly\n'
                             % str(e).splitlines()[0])
            self.use_graph, self.graph, self.graph1, g = False, None, [None, None], None
            torch.cuda.synchronize()
        # every rank is through its capture (or has given it up) before any rank replays: the first replay of a fast rank
        # would otherwise poll the exchange words of a rank that is still capturing (the polls are bounded by wall-clock time)
        if self.comm is not None:
            parallel.barrier()
        return g

    def prepare(self, steps_hint=None, group=None, prime=True):
        """Capture every hipGraph this object will replay NOW and align the groups with the current position.
        ``steps_hint``: the caller is about to run exactly that many steps -- the group size M is chosen with 2 M dividing
        it, so that the run is whole graph launches.  ``group``: M (a graph launch holds 2 M steps).
        Capturing executes nothing; aligning re-extracts the batches of the group that starts at the current position
        (the steady state of an epoch: every group trains on batches extracted while the previous one ran).  Callers that
        time a region (bench.py) call this after their warm-up so that neither falls inside the timed steps.  Needs at
        least one eagerly executed step before it (first launches load code objects, which a capture cannot do).
        ``prime``: a graph captured by this call is launched once with its effects undone (``_prime``)."""
        if not self.use_graph or self.steps_done < 1 or not self._attached:
            return False
        M = self.M
        if group is not None:
            M = max(1, min(MAX_GROUP, int(group)))
        elif steps_hint:
            M = _group_size(steps_hint, self.M_default)
        if M != self.M:
            self.M = M
            self.drop_graphs()
            self.avail = 0
        if self.gq != 0 or self.gk != 0 or self.avail < self.M:
            self._regroup()
        # a run shorter than a pair of groups is ONE single-group launch (steps()): that graph is the one to have ready
        single = steps_hint is not None and self.M <= int(steps_hint) < 2 * self.M
        self.single_launch = bool(single)
        # (single-group launches alternate between the two parities' graphs: both are captured, and primed in that order)
        have = (lambda: self.graph1[0] if self.graph1[1] is not None else None) if single else (lambda: self.graph)
        cap = (lambda: (self._capture_single(0), self._capture_single(1))) if single else self._capture
        order = (lambda: [self.graph1[0], self.graph1[1], self.graph1[0]]) if single else (lambda: [self.graph] * 3)
        fresh = have() is None
        cap()
        # (a launch reads link positions up to 3 M batches ahead of its first step: only where steps() would launch it too)
        if fresh and have() is not None and prime and self.avail >= self.M and \
                self.n_links // self.B - self.k >= 2 * self.M:
            self._prime(order())
            if have() is None and self.use_graph:           # (the primed graph's gates timed out: captured again, edge-paced)
                cap()
                if have() is not None:
                    self._prime(order())
        return self.use_graph

    def _prime(self, graphs=None):
        """First launch of the freshly instantiated graph with its effects undone.  The first launch of a hipGraph costs
        ~140 us more than the following ones (profiles/r02_callB_graph_first_replay.txt) -- a one-time cost like the code
        object load of a kernel's first launch.  The launch runs 2 M real steps; parameters, Adam moments, control block and
        epoch total are restored afterwards and the current group's batches are extracted again, so the training state is
        exactly what it was (host-side step counters are not touched)."""
        m = self.model
        keep = [t.clone() for t in (m.flat_parameters(), self.opt.exp_avg, self.opt.exp_avg_sq, self.ctrl, self.total, self.loss)]
        # (three launches: the first costs ~140 us more than a steady-state launch, the second still ~30 us)
        # Every launch moves the device cursors on by 2 M batches and prefetches one group further: launch L reads link
        # positions below (k + (2 L + 1) M) B, which must stay inside the padded permutation buffer.
        room = (self.n_links // self.B - self.k + self.pad // self.B) // self.M
        launches = min(max(1, int(os.environ.get('IGMC_PRIME_LAUNCHES', '3'))), (room - 1) // 2)
        graphs = graphs if graphs is not None else [self.graph] * 3
        for g in graphs[:launches]:         # (single-group launches: parity 0, parity 1, parity 0 -- the order steps() replays them in)
            g.replay()
            torch.cuda.synchronize()
        # The priming launches double as the PROBE of the extraction chain's pacing gates: where the graph's two
        # chains are not served concurrently (a profiler serialising dispatches, one hardware queue) every gate spins for its
        # whole time-out -- found here, before the first real group, instead of at the first check() an epoch later.
        if launches > 0 and self.pacing_fallback is None and self._paced_default() and \
                os.environ.get('IGMC_EXTRACT_PACED') is None:
            gave_up = int(self.ctrl[_lib.CTRL['GATE_TIMEOUTS']].item())
            if self.comm is not None and self.world > 1:
                gave_up = parallel.all_reduce_max_int(gave_up, self.dev.index)
            if gave_up >= 4:
                sys.stderr.write('igmc_amd: %d pacing gates timed out while the step graph was primed (streams not served '
                                 'concurrently?); pacing by graph edges\n' % gave_up)
                self.pacing_fallback = '1'
                self.drop_graphs()
        for t, k in zip((m.flat_parameters(), self.opt.exp_avg, self.opt.exp_avg_sq, self.ctrl, self.total, self.loss), keep):
            t.copy_(k)
        self._regroup()

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
        """Raises if a bounded device-side wait of the step kernels timed out, or if a step consumed an arena that did not
        hold the batch of its cursor (synchronises the stream)."""
        self.lib.call('igmc_model_check', self.ws.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if self.comm is not None and hasattr(self.comm, 'check'):       # (peer exchange: a bounded poll that ran out)
            self.comm.check(torch.cuda.current_stream().cuda_stream)
        words = self.ctrl.cpu()
        err = int(words[_lib.CTRL['SYNC_ERR']])
        # pacing gates that gave up (igmc_ctrl_gate): harmless one by one, but gates that KEEP timing out mean the two chains
        # are not served concurrently here (dispatches serialised by a profiler, both streams on one hardware queue) and every
        # gate costs its timeout: pace by graph edges from now on (the graph is captured again on its next use)
        gave_up = int(words[_lib.CTRL['GATE_TIMEOUTS']])
        if self.comm is not None and self.world > 1 and self.pacing_fallback is None and self._paced_default():
            # Under data parallelism the fallback is a decision of ALL ranks (the MAX of their counts): a rank that dropped
            # its graph alone would wait in _capture()'s barrier while the others replay theirs, whose in-graph exchange then
            # polls this rank's words until it times out.  Every rank reaches check() at the same points of a run.
            gave_up_all = parallel.all_reduce_max_int(gave_up, self.dev.index)
        else:
            gave_up_all = gave_up
        if gave_up_all >= 4 and self.pacing_fallback is None:
            sys.stderr.write('igmc_amd: %d pacing gates of the extraction chain timed out (streams not served concurrently?); '
                             'pacing by graph edges from here on\n' % gave_up_all)
            self.pacing_fallback = '1'
            self.drop_graphs()
        if gave_up:
            self.ctrl[_lib.CTRL['GATE_TIMEOUTS']] = 0
        if err:
            what = []
            if err & 2:
                what.append('a step trained on an arena whose stamp is not the batch of its cursor')
            if err & 4:
                what.append('a step trained on edge-dropout draws keyed by another batch')
            raise RuntimeError('device-side step control: %s (sync_err=%d); results of the affected steps are invalid'
                               % ('; '.join(what) or 'error', err))


class EvalGraph(StepGraph):
    """``eval_loss`` (reference ``train_eval.py:182-199``) through the same grouped pipeline: a step = forward of a batch +
    accumulation of its squared errors + tick; the next group's batches are extracted meanwhile (no edge dropout, no
    gradient exchange).  Same kernels on the same batches in the same order as the eager loop: the sums are bit-identical."""
    SLOT = 'evalgraph'
    TRAINING = False

    def __init__(self, model, dataset, batch_size, use_graph=None, overlap=None, group=None):
        StepGraph.__init__(self, model, None, dataset, batch_size, 0.0, use_graph=use_graph, overlap=overlap, group=group)
        if self.sp is not None:
            raise NotImplementedError('the sort-pool family evaluates through its own forward (train_eval.eval_loss)')
        self.acc = torch.zeros(2, dtype=torch.float64, device=self.dev)

    def _enqueue_step(self, arena, B):
        m, st = self.model, torch.cuda.current_stream().cuda_stream
        self.ws.forward(m.flat_parameters().data_ptr(), arena, self.out.data_ptr(), training=False,
                        multiply_by=float(m.multiply_by), stream=st)
        # (squared errors + the step's tick in ONE launch: an evaluation step is two launches)
        self.ws.sse_accumulate(self.out.data_ptr(), arena, self.acc.data_ptr(), stream=st, ctrl=self.ctrl.data_ptr())

    def _count(self, n):
        pass

    def run(self, perm, epoch):
        """Sum of squared errors and link count over ``perm`` (this rank's positions): device float64[2]."""
        self.acc.zero_()
        self.begin_epoch(perm, epoch)
        n = self.n_links
        self.steps(n // self.B)
        if n % self.B:
            self.step(n % self.B)
        self.detach()
        self.check()
        return self.acc
