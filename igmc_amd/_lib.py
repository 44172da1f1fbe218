"""ctypes binding of ``libigmc_hip.so`` (C ABI declared in ``include/igmc_hip.h``).

The product path requires the gfx950 library: :func:`load` raises ``RuntimeError`` when it is
missing or cannot be loaded -- there is NO CPU fallback.  (``bind()`` is generic over a
``ctypes.CDLL`` so that the kernel-logic tests can bind the host emulation build of the same
sources; nothing in this package does that.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libigmc_hip.so')

vp = C.c_void_p
i32, i64, u64, f32, f64 = C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double


class BatchInfo(C.Structure):
    _fields_ = [('num_graphs', C.c_int32), ('num_nodes', C.c_int32), ('num_edges', C.c_int32),
                ('overflow', C.c_int32), ('node_capacity', C.c_int32), ('edge_capacity', C.c_int32),
                ('num_labels', C.c_int32), ('hop', C.c_int32)]


# name -> (restype, argtypes)        (every symbol of include/igmc_hip.h)
SIGNATURES = {
    'igmc_last_error': (C.c_char_p, []),
    'igmc_version': (i32, []),
    'igmc_graph_create': (i32, [i32, i32, i64, vp, vp, vp, i32, C.POINTER(vp)]),
    'igmc_graph_destroy': (None, [vp]),
    'igmc_graph_hbm_bytes': (i64, [vp]),
    'igmc_batch_create': (i32, [vp, i32, i32, i32, C.POINTER(vp)]),
    'igmc_batch_destroy': (None, [vp]),
    'igmc_extract_batch': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, f64, u64, u64, vp]),
    'igmc_extract_batch_replay': (i32, [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    'igmc_extract_batch_cached': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    'igmc_batch_set_create': (i32, [vp, i32, C.POINTER(vp)]),
    'igmc_batch_set_destroy': (None, [vp]),
    'igmc_extract_group': (i32, [vp, vp, i32, vp, vp, vp, vp, i32, i32, f64, u64, f32, i32, u64, vp]),
    'igmc_batch_edge_dropout': (i32, [vp, f32, i32, u64, u64, vp]),
    'igmc_batch_set_edge_flags': (i32, [vp, vp, i64]),
    'igmc_batch_clear_edge_flags': (i32, [vp]),
    'igmc_batch_get_info': (i32, [vp, C.POINTER(BatchInfo), vp]),
    'igmc_batch_download': (i32, [vp] + [vp] * 11 + [vp]),
    'igmc_batch_device_ptr': (vp, [vp, i32]),
    'igmc_batch_set_side_features': (i32, [vp, vp, i32]),
    'igmc_batch_bind_side_source': (i32, [vp, vp, i32]),
    'igmc_batch_set_lean': (i32, [vp, i32]),
    'igmc_batch_assume_size': (i32, [vp, i32]),
    'igmc_batch_want_transposed': (i32, [vp]),
    'igmc_model_dense_path': (i32, [vp, vp, i32]),
    'igmc_model_step_form': (i32, [vp, vp, i32]),
    'igmc_model_reset_exchange': (i32, [vp, vp]),
    'igmc_model_create': (i32, [i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(vp)]),
    'igmc_model_destroy': (None, [vp]),
    'igmc_param_count': (i64, [vp]),
    'igmc_param_offset': (i64, [vp, i32, i32, C.POINTER(i64)]),
    'igmc_model_forward': (i32, [vp, vp, vp, i32, i32, vp, u64, u64, f32, vp, vp]),
    'igmc_model_backward': (i32, [vp, vp, vp, vp, f32, vp, vp]),
    'igmc_model_loss_grad': (i32, [vp, vp, vp, i32, vp, u64, u64, f32, f32, f32, f32, vp, vp, vp, vp]),
    'igmc_adam_step': (i32, [vp, vp, vp, vp, i64, i64, f32, f32, f32, f32, f32, vp]),
    'igmc_model_dense_layers': (i32, [vp, vp, i32]),
    'igmc_sortpool_create': (i32, [vp, i32, i32, vp]),
    'igmc_sortpool_destroy': (None, [vp]),
    'igmc_sortpool_layout': (i32, [vp, vp]),
    'igmc_sortpool_forward': (i32, [vp, vp, vp, i32, i32, vp, u64, u64, vp, vp]),
    'igmc_sortpool_loss_grad': (i32, [vp, vp, vp, i32, vp, u64, u64, f32, f32, f32, vp, vp, vp, vp]),
    'igmc_sortpool_step_finish': (i32, [vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp]),
    'igmc_sse_accumulate': (i32, [vp, vp, vp, vp]),
    'igmc_sse_accumulate_tick': (i32, [vp, vp, vp, vp, vp]),
    'igmc_comm_unique_id': (i32, [vp]),
    'igmc_comm_create': (i32, [vp, i32, i32, i32, C.POINTER(vp)]),
    'igmc_comm_destroy': (None, [vp]),
    'igmc_comm_info': (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
    'igmc_allreduce_grads': (i32, [vp, vp, i64, f32, vp]),
    'igmc_comm_create_host': (i32, [vp, vp, i32, i32, C.POINTER(vp)]),
    'igmc_comm_peer_alloc': (i32, [i32, i32, i32, i64, C.POINTER(vp), vp]),
    'igmc_comm_peer_connect': (i32, [vp, vp]),
    'igmc_comm_check': (i32, [vp, vp]),
    'igmc_comm_kind': (i32, [vp]),
    'igmc_train_step_dp': (i32, [vp, vp, vp, vp, i32, vp, u64, u64, f32, f32, vp, vp, vp, vp, vp, vp, vp, i64, f32, f32, f32,
                                 f32, f32, vp]),
    'igmc_ctrl_tick': (i32, [vp, vp]),
    'igmc_ctrl_regroup': (i32, [vp, i32, i64, i64, vp]),
    'igmc_ctrl_gate': (i32, [vp, i32, i32, f64, i32, f64, vp]),
    'igmc_batch_set_ctrl': (i32, [vp, vp]),
    'igmc_model_set_ctrl': (i32, [vp, vp]),
    'igmc_adam_step_ctrl': (i32, [vp, vp, vp, vp, i64, vp, vp]),
    'igmc_train_step': (i32, [vp, vp, vp, i32, vp, u64, u64, f32, f32, vp, vp, vp, vp, vp, vp, vp, i64, f32, f32, f32,
                              f32, f32, vp]),
    'igmc_step_finish': (i32, [vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp]),
    'igmc_model_weights_unchanged': (i32, [vp, i32]),
    'igmc_profile_enable': (i32, [i32]),
    'igmc_profile_fetch': (i32, [vp, vp, vp, i32]),
    'igmc_profile_gs_clock': (i32, [vp, C.POINTER(i64), C.POINTER(f64), i32]),
    'igmc_model_check': (i32, [vp, vp]),
}

BUF = dict(NODE_OFF=0, N_USERS=1, NODE_LABEL=2, NODE_GID=3, NODE_GRAPH=4, ROW_PTR=5, ECR=6, ECODE=7,
           EFLAG=8, Y=9, TOTALS=10)
CTRL = dict(STEP=0, FIRST=1, EPOCH=2, ADAM_T=3, BATCH=4, DONE=5, FIRST_ODD=6, K=7, LR=8, BETA1=9, BETA2=10, EPS=11, WD=12, STEP_SIZE=13,
            INV_SQRT_BC2=14, GROUP=15, GK=16, GQ=17, SYNC_ERR=18, GATE_TIMEOUTS=19, WORDS=24)
ALLREDUCE_FN = C.CFUNCTYPE(i32, vp, vp, i64, vp)      # igmc_allreduce_fn (igmc_comm_create_host)
P = dict(BASIS=0, ROOT=1, BIAS=2, ATT=3, LIN1_W=4, LIN1_B=5, LIN2_W=6, LIN2_B=7)


class Lib(object):
    """Thin checked wrapper: ``lib.call('igmc_xxx', ...)`` raises RuntimeError with the C-side message."""

    def __init__(self, cdll, path):
        self.cdll = cdll
        self.path = path
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(cdll, name)      # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args

    def call(self, name, *args):
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            msg = self.cdll.igmc_last_error()
            raise RuntimeError('%s failed: %s' % (name, msg.decode() if msg else rc))

    def __getattr__(self, name):
        return getattr(self.cdll, name)


def bind(cdll, path='<cdll>'):
    return Lib(cdll, path)


_cached = None


def load(path=None):
    """Load the gfx950 library (building nothing).  Fails loudly: no fallback exists."""
    global _cached
    if _cached is not None and path is None:
        return _cached
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            'igmc_amd: %s is missing -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  There is no CPU fallback.' % p)
    try:
        cdll = C.CDLL(p)
    except OSError as e:
        raise RuntimeError('igmc_amd: cannot load %s (%s).  A ROCm runtime with an MI355X (gfx950) device is '
                           'required; there is no CPU fallback.' % (p, e))
    lib = bind(cdll, p)
    if path is None:
        _cached = lib
    return lib
