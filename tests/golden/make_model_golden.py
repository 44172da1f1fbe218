"""Golden vectors of the MODEL half produced by the reference's OWN code (``models.py`` / ``train_eval.py``, UNMODIFIED).

Run in the build container (needs /root/reference; the GPU box does not have it):

    python tests/golden/make_model_golden.py            # all cases (the headline case takes a few minutes)
    python tests/golden/make_model_golden.py --small    # without the headline case

What runs is ``/root/reference/models.py`` (``IGMC`` ``:170-217``, ``DGCNN_RS`` ``:123-167`` on ``DGCNN`` ``:63-120``)
and ``/root/reference/train_eval.py`` (``train`` ``:149-179``, ``eval_loss`` / ``eval_rmse`` ``:182-205``,
``eval_loss_ensemble`` ``:208-239``), imported through the ``torch_geometric`` stand-in of ``oracle/ref_stub`` whose
three operators (``RGCNConv``, ``dropout_adj``, ``global_sort_pool``) and collate are ``oracle/pyg_ref``'s restatements
of PyG 1.4.2 (absent here: those stay unpinned).  The enclosing subgraphs come from the reference's own extractor
(``util_functions.subgraph_extraction_labeling`` + ``construct_pyg_graph``), the batches from the stub's ``DataLoader``.

Nothing in the reference is touched; what the script needs to see from outside is recorded by ordinary Python means:

* the edge-dropout keep mask: ``torch_geometric.utils.MASK_SOURCE`` (the stand-in draws through it);
* the MLP-dropout mask: ``torch.nn.functional.dropout`` is replaced by a recording twin for the duration of a run
  (the reference calls ``F.dropout(x, p=0.5, training=...)``, ``models.py:212``);
* per-step outputs: a forward hook on the model; per-step gradients: an ``Adam`` subclass that snapshots ``p.grad`` in
  ``step()`` (handed to the reference's ``train`` as its optimizer);
* the node lists of the headline case: the recording row-indexer proxy of ``make_golden.py``.

Cases (file ``tests/golden/model_golden.npz``, keys ``<case>/...``):

* ``igmc_r5``      synthetic MovieLens-shaped 300 x 200 graph, cap 15, R = 5, edge dropout 0.2, 3 batches of 8;
* ``igmc_side``    the same graph with side features (3 + 5 columns), ``multiply_by`` 20, ``force_undirected``;
* ``igmc_r10``     flixster (bundled, R = 10), 3 batches of 8, uncapped;
* ``dgcnn_rs``     flixster (bundled, R = 10), ``DGCNN_RS`` with ``k = 0.6`` (percentile form: ``models.py:69-72``), 2 batches of 8;
* ``headline``     BASELINE.json configs[2]: ml_1m-shaped graph (``preprocessing.create_trainvaltest_split('ml_1m', 1234,
  True)``), hop 1, cap 100, ONE batch of 50 -- node lists + outputs / loss / gradients with edge dropout 0 and 0.2.

For every case: the initial ``state_dict``; per batch the collated tensors (or, headline, the node lists they follow
from); eval-mode outputs; ONE epoch of the reference's ``train`` (ARR 0.001, ``Adam(lr=1e-3)``) with the drawn masks,
per-step outputs, FIRST-step gradients, the returned epoch loss and the parameters afterwards; ``eval_loss`` /
``eval_rmse`` of the trained parameters; ``eval_loss_ensemble`` over [initial, trained].
"""
import math
import os
import random
import sys
import tempfile
import warnings

import numpy as np
import scipy.sparse as ssp
import torch

warnings.simplefilter('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle', 'ref_stub'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import util_functions as REF_U  # noqa: E402  (unmodified reference modules)
import models as REF_M  # noqa: E402
import train_eval as REF_T  # noqa: E402
import torch_geometric.utils as TGU  # noqa: E402  (the stand-in)
from torch_geometric.data import DataLoader  # noqa: E402
from make_golden import _RecRows  # noqa: E402
from igmc_amd import preprocessing  # noqa: E402

CPU = torch.device('cpu')


class GraphList(list):
    """What the reference's model constructors read of a dataset: ``num_features`` and iteration."""
    num_features = 4


class Recorder(object):
    """Draws + records the two dropout masks of a run (seeded numpy generator: replayable)."""

    def __init__(self, seed, p_edge):
        self.rng = np.random.default_rng(seed)
        self.p_edge = p_edge
        self.edge_masks, self.lin_masks = [], []

    def edge(self, n):
        keep = self.rng.random(n) >= self.p_edge
        self.edge_masks.append(keep)
        return torch.from_numpy(keep)

    def dropout(self, x, p=0.5, training=True, inplace=False):
        if not training:
            return x
        keep = self.rng.random(tuple(x.shape)) >= p
        self.lin_masks.append(keep)
        return x * torch.from_numpy(keep).to(x.dtype) / (1.0 - p)

    def __enter__(self):
        self._saved = torch.nn.functional.dropout
        torch.nn.functional.dropout = self.dropout
        TGU.MASK_SOURCE = self.edge
        return self

    def __exit__(self, *a):
        torch.nn.functional.dropout = self._saved
        TGU.MASK_SOURCE = None


class RecAdam(torch.optim.Adam):
    """``torch.optim.Adam`` that snapshots the gradients it is about to apply."""

    def __init__(self, named, **kw):
        named = list(named)
        self.names = [n for n, _ in named]
        self.grads = []
        super().__init__([p for _, p in named], **kw)

    def step(self, closure=None):
        ps = self.param_groups[0]['params']
        self.grads.append({n: p.grad.detach().clone() for n, p in zip(self.names, ps)})
        return super().step(closure)


def extract(A, links, labels, class_values, h, mnph, seed, u_features=None, v_features=None, record=False):
    """Reference extractor + graph construction for every link -> GraphList of stub ``Data`` (+ node lists)."""
    Arow_real = REF_U.SparseRowIndexer(A)
    Arow = _RecRows(Arow_real) if record else Arow_real
    Acol = REF_U.SparseColIndexer(A.tocsc())
    random.seed(seed)
    graphs, lists = GraphList(), []
    for (i, j), lab in zip(links, labels):
        if record:
            Arow.rec.clear()
        tmp = REF_U.subgraph_extraction_labeling((i, j), Arow, Acol, h, 1.0, mnph, u_features, v_features,
                                                 class_values, lab)
        graphs.append(REF_U.construct_pyg_graph(*tmp))
        if record:
            lists.append((np.asarray(Arow.rec['u_nodes_last'], np.int32), np.asarray(Arow.rec['v_nodes'], np.int32)))
    return graphs, lists


def state_np(model):
    return {k: v.detach().cpu().numpy().astype(np.float32).copy() for k, v in model.state_dict().items()}


def put_state(out, prefix, sd):
    for k, v in sd.items():
        out[prefix + '/' + k] = np.asarray(v, np.float32)


def put_batches(out, case, loader):
    nb = 0
    for b, data in enumerate(loader):
        p = '%s/batch%d/' % (case, b)
        out[p + 'label'] = data.x.argmax(1).numpy().astype(np.uint8)
        assert torch.equal(data.x, torch.nn.functional.one_hot(data.x.argmax(1), data.x.shape[1]).float())
        out[p + 'edge_index'] = data.edge_index.numpy().astype(np.int32)
        out[p + 'edge_type'] = data.edge_type.numpy().astype(np.uint8)
        out[p + 'sizes'] = np.bincount(data.batch.numpy(), minlength=data.num_graphs).astype(np.int32)
        out[p + 'y'] = data.y.numpy().astype(np.float32)
        if hasattr(data, 'u_feature'):
            out[p + 'u_feature'] = data.u_feature.numpy().astype(np.float32)
            out[p + 'v_feature'] = data.v_feature.numpy().astype(np.float32)
        nb += 1
    out[case + '/n_batches'] = np.array(nb)


def run_reference(out, case, model, graphs, batch_size, p_edge, ARR=0.001, lr=1e-3, seed=7, full_grads=False):
    """eval outputs -> one epoch of the reference's ``train`` -> eval_loss / eval_rmse -> ensemble."""
    loader = DataLoader(graphs, batch_size, shuffle=False)
    init = state_np(model)
    put_state(out, case + '/init', init)
    # ---- eval-mode forward (models.py forward with training=False: no dropout of either kind)
    model.eval()
    with torch.no_grad():
        for b, data in enumerate(loader):
            out['%s/eval_out/%d' % (case, b)] = model(data).numpy().astype(np.float32)
    # ---- one epoch of train() (train_eval.py:149-179)
    outs = []
    hook = model.register_forward_hook(lambda m, i, o: outs.append(o.detach().numpy().astype(np.float32).copy()))
    opt = RecAdam(model.named_parameters(), lr=lr, weight_decay=0)
    with Recorder(seed, p_edge) as rec:
        epoch_loss = REF_T.train(model, opt, loader, CPU, regression=True, ARR=ARR)
    hook.remove()
    n_steps = len(opt.grads)
    assert n_steps == len(outs) == len(rec.lin_masks) and (p_edge == 0 or len(rec.edge_masks) == n_steps)
    out[case + '/train/epoch_loss'] = np.array(epoch_loss, np.float64)
    out[case + '/train/hyper'] = np.array([ARR, lr, p_edge], np.float64)
    for s in range(n_steps):
        out['%s/train/out/%d' % (case, s)] = outs[s]
        out['%s/train/lin_mask/%d' % (case, s)] = np.packbits(rec.lin_masks[s].reshape(-1))
        if p_edge > 0:
            out['%s/train/edge_mask/%d' % (case, s)] = np.packbits(rec.edge_masks[s])
            out['%s/train/edge_mask_n/%d' % (case, s)] = np.array(len(rec.edge_masks[s]))
        if s == 0 or full_grads:
            put_state(out, '%s/train/grad%d' % (case, s), {k: v.numpy() for k, v in opt.grads[s].items()})
    post = state_np(model)
    put_state(out, case + '/post', post)
    # ---- eval_loss / eval_rmse of the trained parameters (train_eval.py:182-205)
    out[case + '/eval_loss_post'] = np.array(REF_T.eval_loss(model, loader, CPU, regression=True), np.float64)
    out[case + '/eval_rmse_post'] = np.array(REF_T.eval_rmse(model, loader, CPU), np.float64)
    # ---- ensemble over [initial, trained] (train_eval.py:208-245)
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for n, sd in (('a', init), ('b', post)):
            p = os.path.join(td, n + '.pth')
            torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, p)
            paths.append(p)
        out[case + '/eval_loss_ensemble'] = np.array(
            REF_T.eval_loss_ensemble(model, paths, loader, CPU, regression=True), np.float64)
        out[case + '/eval_rmse_ensemble'] = np.array(REF_T.eval_rmse_ensemble(model, paths, loader, CPU), np.float64)
    return loader


def put_geometry(out, case, A, links, labels, cv, lists, u_features=None, v_features=None):
    """What an engine needs to rebuild the SAME subgraphs: links, labels and the node lists the reference chose."""
    out[case + '/graph_fingerprint'] = np.array(graph_fingerprint(A), np.uint64)
    out[case + '/links'] = np.asarray(links, np.int32).reshape(-1, 2)
    out[case + '/link_labels'] = np.asarray(labels, np.int32)
    out[case + '/class_values'] = np.asarray(cv, np.float64)
    out[case + '/u_nodes'] = np.concatenate([l[0] for l in lists])
    out[case + '/v_nodes'] = np.concatenate([l[1] for l in lists])
    out[case + '/u_off'] = np.cumsum([0] + [len(l[0]) for l in lists]).astype(np.int32)
    out[case + '/v_off'] = np.cumsum([0] + [len(l[1]) for l in lists]).astype(np.int32)
    if u_features is not None:
        out[case + '/u_features'], out[case + '/v_features'] = u_features, v_features


def perturb(model, seed):
    """Initial parameters = the reference's ``reset_parameters`` + noise, so that bias / att gradients are exercised."""
    torch.manual_seed(seed)
    model.reset_parameters()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))


def synth_graph():
    u, v, r = preprocessing.synth_ml(300, 200, 9000, preprocessing.ML_HIST['ml_100k'][3], seed=3)
    A = ssp.csr_matrix((r.astype(np.float32), (u, v)), shape=(300, 200))
    pick = np.random.default_rng(5).choice(len(u), 24, replace=False)
    return A, list(zip(u[pick].tolist(), v[pick].tolist())), (r[pick].astype(int) - 1).tolist()


def case_igmc_r5(out):
    A, links, labels = synth_graph()
    cv = np.array([1., 2., 3., 4., 5.])
    graphs, lists = extract(A, links, labels, cv, 1, 15, seed=11, record=True)
    put_geometry(out, 'igmc_r5', A, links, labels, cv, lists)
    model = REF_M.IGMC(graphs, latent_dim=[32, 32, 32, 32], num_relations=5, num_bases=4, regression=True,
                       adj_dropout=0.2, force_undirected=False, side_features=False, n_side_features=0, multiply_by=1)
    perturb(model, 1)
    put_batches(out, 'igmc_r5', run_reference(out, 'igmc_r5', model, graphs, 8, 0.2))
    out['igmc_r5/ctor'] = np.array([5, 0, 1, 0], np.float64)      # R, n_side, multiply_by, force_undirected


def case_igmc_side(out):
    A, links, labels = synth_graph()
    cv = np.array([20., 40., 60., 80., 100.])
    rng = np.random.default_rng(9)
    uf = rng.standard_normal((300, 3)).astype(np.float32)
    vf = (rng.random((200, 5)) < 0.4).astype(np.float32)
    graphs, lists = extract(A, links, labels, cv, 1, 15, seed=12, u_features=uf, v_features=vf, record=True)
    put_geometry(out, 'igmc_side', A, links, labels, cv, lists, uf, vf)
    model = REF_M.IGMC(graphs, latent_dim=[32, 32, 32, 32], num_relations=5, num_bases=4, regression=True,
                       adj_dropout=0.2, force_undirected=True, side_features=True, n_side_features=8, multiply_by=20)
    perturb(model, 2)
    put_batches(out, 'igmc_side', run_reference(out, 'igmc_side', model, graphs, 8, 0.2))
    out['igmc_side/ctor'] = np.array([5, 8, 20, 1], np.float64)


def case_igmc_r10(out):
    os.chdir(ROOT)
    (_, _, adj, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = preprocessing.load_data_monti('flixster', testing=True)
    links = list(zip(tr_u[:24].tolist(), tr_v[:24].tolist()))
    graphs, lists = extract(adj, links, tr_l[:24].tolist(), cv, 1, 10000, seed=1, record=True)
    put_geometry(out, 'igmc_r10', adj, links, tr_l[:24].tolist(), cv, lists)
    assert len(cv) == 10
    model = REF_M.IGMC(graphs, latent_dim=[32, 32, 32, 32], num_relations=10, num_bases=4, regression=True,
                       adj_dropout=0.2, force_undirected=False, side_features=False, n_side_features=0, multiply_by=1)
    perturb(model, 3)
    put_batches(out, 'igmc_r10', run_reference(out, 'igmc_r10', model, graphs, 8, 0.2))
    out['igmc_r10/ctor'] = np.array([10, 0, 1, 0], np.float64)


def case_dgcnn_rs(out):
    os.chdir(ROOT)
    (_, _, adj, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = preprocessing.load_data_monti('flixster', testing=True)
    links = list(zip(tr_u[40:56].tolist(), tr_v[40:56].tolist()))
    graphs, lists = extract(adj, links, tr_l[40:56].tolist(), cv, 1, 10000, seed=1, record=True)
    put_geometry(out, 'dgcnn_rs', adj, links, tr_l[40:56].tolist(), cv, lists)
    model = REF_M.DGCNN_RS(graphs, latent_dim=[32, 32, 32, 1], k=0.6, num_relations=10, num_bases=4, regression=True,
                           adj_dropout=0.2, force_undirected=False)
    perturb(model, 4)
    out['dgcnn_rs/k'] = np.array(model.k)
    out['dgcnn_rs/num_nodes'] = np.array([g.num_nodes for g in graphs], np.int32)
    put_batches(out, 'dgcnn_rs', run_reference(out, 'dgcnn_rs', model, graphs, 8, 0.2))
    out['dgcnn_rs/ctor'] = np.array([10, 0, 1, 0], np.float64)


def headline_links(n=50, seed=3):
    """The first ``n`` links of ``tests/test_gpu_headline.py::ml_case('ml_1m', 100, 250)``."""
    split = preprocessing.create_trainvaltest_split('ml_1m', 1234, True, verbose=False)
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = split
    pick = np.random.default_rng(seed).permutation(len(tr_u))[:250][:n]
    links = np.stack([tr_u[pick], tr_v[pick]], 1).astype(np.int64)
    return A, links, np.asarray(tr_l)[pick].astype(np.int64), np.asarray(cv, np.float64)


def graph_fingerprint(A):
    import hashlib
    A = ssp.csr_matrix(A)
    A.sort_indices()
    h = hashlib.sha256()
    for a in (A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float32)):
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest()[:8], np.uint64)[0]


def case_headline(out):
    os.chdir(ROOT)
    A, links, labels, cv = headline_links()
    graphs, lists = extract(A, [tuple(l) for l in links.tolist()], labels.tolist(), cv, 1, 100, seed=21, record=True)
    case = 'headline'
    put_geometry(out, case, A, links, labels, cv, lists)
    loader = DataLoader(graphs, 50, shuffle=False)
    data = next(iter(loader))
    out[case + '/N_E'] = np.array([data.x.shape[0], data.edge_index.shape[1]], np.int64)
    out[case + '/edge_checksum'] = np.array([int(data.edge_index[0].sum()), int(data.edge_index[1].sum()),
                                             int(data.edge_type.sum())], np.int64)
    for tag, p_edge in (('nodrop', 0.0), ('drop', 0.2)):
        model = REF_M.IGMC(graphs, latent_dim=[32, 32, 32, 32], num_relations=5, num_bases=4, regression=True,
                           adj_dropout=p_edge, force_undirected=False, side_features=False, n_side_features=0,
                           multiply_by=1)
        perturb(model, 1)
        run_reference(out, case + '_' + tag, model, graphs, 50, p_edge)
        out[case + '_' + tag + '/ctor'] = np.array([5, 0, 1, 0], np.float64)
        print(tag, 'epoch loss', float(out[case + '_' + tag + '/train/epoch_loss']))


def main():
    dst = os.path.join(HERE, 'model_golden.npz')
    out = {}
    small = '--small' in sys.argv
    if small and os.path.exists(dst):          # keep the headline arrays of an earlier full run
        z = np.load(dst)
        out.update({k: z[k] for k in z.files if k.startswith('headline')})
    for fn in (case_igmc_r5, case_igmc_side, case_igmc_r10, case_dgcnn_rs):
        fn(out)
        print(fn.__name__, 'done')
    if not small:
        case_headline(out)
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes;', len(out), 'arrays')


if __name__ == '__main__':
    main()
