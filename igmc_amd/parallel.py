"""Data parallelism over the GPUs of one node: one process per GPU, ``torch.distributed`` with the
``nccl`` backend (= RCCL over xGMI on ROCm); ``gloo`` for the CPU-side tests of the host logic.

The reference has no distributed code (single process, single device: ``train_eval.py:20``).  The hot path
shards by LINKS: the rating graph (~9 MB) and the 197 KB parameter / optimiser state are replicated, rank k
takes ``perm[k::G]`` of the (identical) epoch permutation, extracts and trains its own batches, and the only
exchange is ONE collective per step (latency-bound at ~200 KB, so never one call per tensor): the step's reduced
gradient sources -- or, where a step keeps none, the flat gradient -- summed over the ranks INSIDE the step
(``igmc_train_step_dp``), then the identical gradient / Adam kernel on every rank.  Evaluation all-reduces (sum of
squared errors, count) once.
"""
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def rank():
    return dist.get_rank() if is_dist() else 0


def world_size():
    return dist.get_world_size() if is_dist() else 1


def init_from_env(backend=None):
    """Initialise from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*); no-op for 1 process."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if ws <= 1 or is_dist():
        return rank(), world_size()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    local = int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0')))
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(local)
    dist.init_process_group(backend=backend, rank=int(os.environ['RANK']), world_size=ws)
    return rank(), world_size()


def shard_positions(perm, rank_, world, pad=True):
    """Rank ``rank_``'s share ``perm[rank_::world]`` of a permutation (any indexable 1-D tensor/array).
    ``pad``: wrap around so that every rank gets ceil(n/world) items (equal step counts -> no collective
    mismatch during training); evaluation uses ``pad=False`` so nothing is counted twice."""
    n = len(perm)
    if world <= 1:
        return perm
    mine = perm[rank_::world]
    if pad:
        per = (n + world - 1) // world
        if len(mine) < per:
            mine = torch.cat([mine, perm[:per - len(mine)]]) if torch.is_tensor(mine) else \
                type(perm)(list(mine) + list(perm[:per - len(mine)]))
    return mine


def all_reduce_sum_(t):
    """In-place sum over ranks of ONE flat tensor (the gradient buffer / the eval accumulators)."""
    # (IGMC_DP_ALLREDUCE_ALWAYS=1: also in a 1-rank group -- lets one GPU exercise the collective's enqueue / capture path)
    if is_dist() and (world_size() > 1 or os.environ.get('IGMC_DP_ALLREDUCE_ALWAYS', '0') == '1'):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class GradComm(object):
    """The library's own RCCL communicator for the flat-gradient all-reduce (``include/igmc_hip.h``, gradient exchange):
    the collective is enqueued through the C ABI on the step's stream, so it is captured into the step's hipGraph like
    any kernel.  The 128-byte RCCL id travels from rank 0 over whatever ``torch.distributed`` backend is up."""
    transport = 'rccl'

    def __init__(self, lib, device):
        import ctypes as C
        self.lib, self.C = lib, C
        r, w = rank(), world_size()
        buf = (C.c_uint8 * 128)()
        # Rank 0's id request is the fallible step (RCCL missing): EVERY rank takes part in the one broadcast of
        # (status, id) whatever happened there, so a failure on rank 0 cannot leave the others waiting for an id while
        # rank 0 has moved on to the next collective.
        status = 'ok'
        if r == 0:
            try:
                lib.call('igmc_comm_unique_id', C.cast(buf, C.c_void_p))
            except RuntimeError as e:
                status = str(e) or 'igmc_comm_unique_id failed'
        status, ident = broadcast_object((status, bytes(buf)), 0)
        if status != 'ok':
            raise RuntimeError('rank 0 could not create the RCCL id: ' + status)
        buf = (C.c_uint8 * 128).from_buffer_copy(ident)
        h = C.c_void_p()
        lib.call('igmc_comm_create', C.cast(buf, C.c_void_p), r, w, int(device), C.byref(h))
        self.handle = h

    def info(self):
        """(rank, world size) as RCCL sees them."""
        r, w = self.C.c_int(-1), self.C.c_int(-1)
        self.lib.call('igmc_comm_info', self.handle, self.C.byref(r), self.C.byref(w))
        return r.value, w.value

    def all_reduce_(self, t, stream, scale=1.0):
        self.lib.call('igmc_allreduce_grads', self.handle, self.C.c_void_p(t.data_ptr()), t.numel(), float(scale),
                      self.C.c_void_p(stream))
        return t

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.cdll.igmc_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PeerComm(object):
    """A communicator whose sum is a one-shot all-reduce over PEER-MAPPED buffers (``igmc_comm_peer_alloc`` /
    ``igmc_comm_peer_connect``): every rank publishes its span into its own device buffer, mapped by the other ranks of
    the node through HIP IPC, and reads every rank's words in rank order -- one launch, about one xGMI round trip instead
    of a ring's 2 (G - 1) hops, bit-identical replicas, capturable into the step's hipGraph.  The 64-byte IPC handles
    travel over whatever ``torch.distributed`` backend is up.  The constructor finishes with a SELF-TEST exchange (rank
    r contributes r + 1 everywhere): a node whose peer memory is not visible, or ranks that cannot be on their chips at
    once, fail here (bounded polls) instead of inside a training step."""
    transport = 'p2p'
    capturable = True

    def __init__(self, lib, device, max_floats=1 << 18):
        import ctypes as C
        self.lib, self.C = lib, C
        r, w = rank(), world_size()
        h = C.c_void_p()
        handle = (C.c_uint8 * 64)()
        # The allocation is the fallible step (hipMalloc, the IPC handle, world > IGMC_MAX_PEERS): EVERY rank takes part in
        # the one gather of (status, handle) whatever happened locally, and every rank raises after it when any rank failed
        # -- a rank that left early would be in grad_comm's next collective while the others wait in this one.
        status = 'ok'
        try:
            lib.call('igmc_comm_peer_alloc', r, w, int(device), int(max_floats), C.byref(h), C.cast(handle, C.c_void_p))
            self.handle = h
        except RuntimeError as e:
            status = str(e) or 'igmc_comm_peer_alloc failed'
        got = [(status, bytes(handle))]
        if is_dist() and w > 1:
            got = [None] * w
            dist.all_gather_object(got, (status, bytes(handle)))
        bad = ['rank %d: %s' % (k, s) for k, (s, _) in enumerate(got) if s != 'ok']
        if bad:
            self.close()
            raise RuntimeError('peer communicator: ' + '; '.join(bad))
        blob = (C.c_uint8 * (64 * w)).from_buffer_copy(b''.join(hh for _, hh in got))
        # (mapping the peers' buffers can fail too -- peer access refused: the outcome is agreed on the same way)
        status = 'ok'
        try:
            lib.call('igmc_comm_peer_connect', self.handle, C.cast(blob, C.c_void_p))
        except RuntimeError as e:
            status = str(e) or 'igmc_comm_peer_connect failed'
        if is_dist() and w > 1:
            got = [None] * w
            dist.all_gather_object(got, status)
            bad = ['rank %d: %s' % (k, s) for k, s in enumerate(got) if s != 'ok']
        else:
            bad = [] if status == 'ok' else [status]
        if bad:
            self.close()
            raise RuntimeError('peer communicator: ' + '; '.join(bad))
        self.fine_grained = lib.cdll.igmc_comm_kind(self.handle) == 4
        # SELF-TEST (also the first use of the mapped pointers) -- what ``auto`` keys on, and the cross-DEVICE proof of the
        # exchange where the ranks sit on different GPUs: eight launches (both slots four times over, slot reuse included)
        # of element-distinct values, rank r contributing (r + 1) (1 + i % 7) + k at element i of launch k -- a stale word
        # of an earlier launch, a word of the wrong slot or a torn read gives a wrong sum somewhere.  Bounded by a short
        # wall-clock limit of its own (IGMC_PEER_SELFTEST_TIMEOUT_S, 20 s): a node whose peer memory is not visible fails
        # HERE, within seconds and on every rank, and grad_comm moves on to RCCL.  Outcome raised after the last launch, so
        # that every rank issues the same launches whatever it sees.
        dev = torch.device('cuda', int(device))
        st = torch.cuda.current_stream(dev).cuda_stream
        keep = os.environ.get('IGMC_PEER_TIMEOUT_S')
        os.environ['IGMC_PEER_TIMEOUT_S'] = os.environ.get('IGMC_PEER_SELFTEST_TIMEOUT_S', '20')
        bad = None
        try:
            for k in range(8):
                n = 1000 + 4093 * k
                ramp = 1.0 + (torch.arange(n, device=dev) % 7).float()
                t = float(r + 1) * ramp + float(k)
                self.all_reduce_(t, st)
                want = (w * (w + 1) / 2.0) * ramp + float(w * k)
                if bad is None and not bool((t == want).all().item()):
                    i = int((t != want).nonzero()[0].item())
                    bad = 'launch %d, element %d: %g, expected %g' % (k, i, float(t[i]), float(want[i]))
            self.check(st)
        finally:
            if keep is None:
                os.environ.pop('IGMC_PEER_TIMEOUT_S', None)
            else:
                os.environ['IGMC_PEER_TIMEOUT_S'] = keep
        if bad:
            raise RuntimeError('peer all-reduce self-test: wrong sums (%s)' % bad)

    def check(self, stream):
        self.lib.call('igmc_comm_check', self.handle, self.C.c_void_p(stream))

    def info(self):
        r, w = self.C.c_int(-1), self.C.c_int(-1)
        self.lib.call('igmc_comm_info', self.handle, self.C.byref(r), self.C.byref(w))
        return r.value, w.value

    def all_reduce_(self, t, stream, scale=1.0):
        self.lib.call('igmc_allreduce_grads', self.handle, self.C.c_void_p(t.data_ptr()), t.numel(), float(scale),
                      self.C.c_void_p(stream))
        return t

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.cdll.igmc_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostComm(object):
    """A communicator whose sum is the caller's (``igmc_comm_create_host``): ``fn(ptr, n, stream)`` must leave the sum over
    the ranks in the ``n`` floats at ``ptr``.  For a host that already owns a process group; the CPU-side tests drive the
    library's data-parallel step over ``gloo`` with it."""

    def __init__(self, lib, fn, rank_, world):
        import ctypes as C
        from . import _lib
        self.lib, self.C = lib, C

        def _cb(user, ptr, n, stream):
            try:
                fn(ptr, n, stream)
                return 0
            except Exception:          # (an exception cannot cross the C frame: reported as the call's failure)
                import traceback
                traceback.print_exc()
                return 1
        self._cb = _lib.ALLREDUCE_FN(_cb)          # (kept alive as long as the communicator)
        h = C.c_void_p()
        lib.call('igmc_comm_create_host', C.cast(self._cb, C.c_void_p), None, int(rank_), int(world), C.byref(h))
        self.handle = h

    info = GradComm.info
    all_reduce_ = GradComm.all_reduce_
    close = GradComm.close

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_grad_comms = {}


class _DevSpan(object):
    """``n`` float32 values at a device address, for ``torch.as_tensor`` (no copy)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr='<f4', data=(int(ptr), False), version=2)


def process_group_comm(lib, device):
    """A communicator whose sum is the process group's ``torch.distributed`` already has (``igmc_comm_create_host``):
    ``IGMC_DP_HOST_COMM=1``.  With the ``nccl`` backend the all-reduce is enqueued by torch on the current stream; with
    ``gloo`` the span is staged through the host (the stream is synchronised: not capturable -- the caller launches its
    steps eagerly).  For hosts without RCCL, and for running the data-parallel step with several ranks on ONE GPU (RCCL
    refuses two ranks on a device), which is how the two-rank GPU test runs."""
    dev = torch.device('cuda', int(device))
    backend = dist.get_backend()

    def host_sum(ptr, n, stream):
        # the contract of igmc_allreduce_fn: the sum is ordered on ``stream`` (the step's stream), which need not be
        # torch's current one -- work is enqueued on / synchronised against THAT stream
        t = torch.as_tensor(_DevSpan(ptr, n), device=dev)
        s = torch.cuda.ExternalStream(int(stream), device=dev) if stream else torch.cuda.current_stream(dev)
        with torch.cuda.stream(s):
            if backend == 'gloo':
                s.synchronize()
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                t.copy_(h)
                s.synchronize()
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
    c = HostComm(lib, host_sum, rank(), world_size())
    c.capturable = backend != 'gloo'          # (the gloo route synchronises the stream: its steps cannot be captured)
    c.transport = 'host-callback:' + str(backend)
    return c


def _agree(ok, device):
    """True iff `ok` holds on EVERY rank (every rank must end up on the same transport)."""
    if not (is_dist() and world_size() > 1):
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend() == 'nccl':
        t = t.to(torch.device('cuda', int(device)))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def grad_comm(lib, device):
    """One communicator per (process, device); needed when there is more than one rank, or with
    ``IGMC_DP_ALLREDUCE_ALWAYS=1`` (a one-rank communicator: lets one GPU exercise the exchange's enqueue / capture).
    ``IGMC_DP_TRANSPORT`` picks the exchange: ``p2p`` (one-shot all-reduce over peer-mapped buffers, :class:`PeerComm`),
    ``rccl`` (the library's own RCCL communicator, :class:`GradComm`), ``host`` (the process group ``torch.distributed``
    already has, :func:`process_group_comm`; ``IGMC_DP_HOST_COMM=1`` is the older spelling).  Default ``auto``: the first of
    p2p -> rccl -> host that EVERY rank can set up -- p2p includes a self-test exchange, RCCL refuses e.g. two ranks on one
    device -- with a line on stderr saying why an earlier one was passed over."""
    if world_size() <= 1 and os.environ.get('IGMC_DP_ALLREDUCE_ALWAYS', '0') != '1':
        return None
    key = int(device)
    if key in _grad_comms:
        return _grad_comms[key]
    want = os.environ.get('IGMC_DP_TRANSPORT', 'host' if os.environ.get('IGMC_DP_HOST_COMM', '0') == '1' else 'auto')
    if want not in ('auto', 'p2p', 'rccl', 'host'):
        raise ValueError('IGMC_DP_TRANSPORT=%r (auto, p2p, rccl or host)' % want)
    import sys
    tried = []
    order = ('p2p', 'rccl', 'host') if want == 'auto' else (want,)
    if want == 'auto' and os.environ.get('IGMC_DP_NO_P2P', '0') == '1':
        order = ('rccl', 'host')
    for kind in order:
        comm, why = make_comm(lib, device, kind)
        if comm is not None:
            if tried:
                print('[igmc] gradient exchange over %s (%s)' % (kind, '; '.join(tried)), file=sys.stderr)
            if want == 'auto' and kind == 'p2p' and world_size() > 1 and os.environ.get('IGMC_DP_AUTO_MEASURE', '1') == '1':
                comm = _faster_of_p2p_and_rccl(lib, device, comm)
            _grad_comms[key] = comm
            return comm
        tried.append('%s could not be set up on every rank: %s' % (kind, why or 'another rank failed'))
    raise RuntimeError('no gradient exchange: ' + '; '.join(tried))


def _exchange_us(comm, device, n=61000, reps=20):
    """Microseconds per all-reduce of ``n`` floats (a step's exchange) on this rank's stream, MAX over the ranks."""
    dev = torch.device('cuda', int(device))
    st = torch.cuda.current_stream(dev).cuda_stream
    t = torch.zeros(n, dtype=torch.float32, device=dev)
    for _ in range(3):
        comm.all_reduce_(t, st)
    torch.cuda.synchronize(dev)
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        comm.all_reduce_(t, st)
    e1.record()
    torch.cuda.synchronize(dev)
    if hasattr(comm, 'check'):
        comm.check(st)
    us = torch.tensor([e0.elapsed_time(e1) / reps * 1e3], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(us, op=dist.ReduceOp.MAX)
    return float(us.item())


def _faster_of_p2p_and_rccl(lib, device, p2p):
    """``auto`` with more than one rank and a peer exchange that passed its self-test: the library's RCCL communicator is set up
    beside it, a step's exchange (61 k floats) is timed over both -- MAX over the ranks, so every rank sees the same two numbers --
    and the steps go over the faster one.  Whatever fails on the way (RCCL refusing this set of ranks, a timing that raises)
    leaves the peer exchange in place.  ``IGMC_DP_AUTO_MEASURE=0``: p2p without the comparison."""
    import sys
    rccl, why = make_comm(lib, device, 'rccl')
    if rccl is None:
        return p2p
    ok, us = True, {}
    try:
        us['p2p'] = _exchange_us(p2p, device)
        us['rccl'] = _exchange_us(rccl, device)
    except Exception as e:          # noqa: BLE001  (a diagnostic comparison: never the reason a run fails)
        ok = False
        print('[igmc] gradient exchange: timing p2p against rccl failed (%s); p2p' % e, file=sys.stderr)
    if not _agree(ok, device):
        rccl.close()
        return p2p
    pick = 'rccl' if us['rccl'] < us['p2p'] else 'p2p'
    if rank() == 0:
        print('[igmc] gradient exchange: p2p %.1f us, rccl %.1f us per step exchange -> %s' % (us['p2p'], us['rccl'], pick), file=sys.stderr)
    if pick == 'rccl':
        p2p.close()
        rccl.auto_us = us          # (what the choice was made on: bench.py's dp_check carries it)
        return rccl
    rccl.close()
    p2p.auto_us = us
    return p2p


def make_comm(lib, device, kind):
    """A communicator of ONE kind (``p2p`` / ``rccl`` / ``host``) on every rank, or ``(None, why)`` on every rank: the ranks
    agree on the outcome (a collective), so nobody is left holding a communicator the others could not set up.  Every rank
    must call this at the same point.  ``grad_comm`` walks its transports through it; ``bench.py`` uses it to time the
    transports it is NOT training over (``dp_check.allreduce_us``) and to read RCCL's own view of the world size."""
    comm, why = None, ''
    if kind == 'host':
        if not is_dist():
            why = 'no torch.distributed process group'
        else:
            comm = process_group_comm(lib, device)
    else:
        try:
            comm = PeerComm(lib, device) if kind == 'p2p' else GradComm(lib, device)
        except RuntimeError as e:          # (RCCL missing / refusing this set of ranks; IPC or the self-test failing)
            why = str(e)
    if _agree(comm is not None, device):
        return comm, ''
    if comm is not None:
        comm.close()
    return None, why or 'another rank failed'


def all_reduce_max_int(value, device=None):
    """MAX over the ranks of a small host-side integer (decisions every rank must take together)."""
    if not (is_dist() and world_size() > 1):
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64)
    if dist.get_backend() == 'nccl':
        t = t.to(torch.device('cuda', int(device) if device is not None else torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def barrier():
    if is_dist() and world_size() > 1:
        dist.barrier()


def broadcast_(t, src=0):
    if is_dist() and world_size() > 1:
        dist.broadcast(t, src=src)
    return t


def broadcast_object(obj, src=0):
    """A small picklable object from rank ``src`` to every rank (host-side decisions that all ranks must share)."""
    if is_dist() and world_size() > 1:
        box = [obj]
        dist.broadcast_object_list(box, src=src)
        return box[0]
    return obj
