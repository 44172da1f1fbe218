"""Drop-in mirror of the reference's ``util_functions.py`` (datasets + extraction), backed by the
gfx950 engine.  Same names, argument meaning and return shapes as the reference:

* ``SparseRowIndexer`` / ``SparseColIndexer``     reference ``util_functions.py:20-66``
* ``MyDataset`` / ``MyDynamicDataset``            reference ``:69-145``
* ``links2subgraphs``                             reference ``:148-205``
* ``subgraph_extraction_labeling``                reference ``:208-277``
* ``construct_pyg_graph``                         reference ``:280-297``

What differs by design: the rating graph lives in HBM once; a *batch* of links is
extracted, labelled and collated by HIP kernels (``igmc_amd/csrc/extract.hip``) -- there are no
DataLoader worker processes, no pickling and no H2D copy of graphs.  Per-hop sampling uses a
counter-based hash (uniform k-subsets like ``random.sample``, but reproducible): ``MyDynamicDataset``
re-samples every epoch; ``MyDataset`` ("static", pre-extracted in the reference) extracts every subgraph
once and keeps the node sets in a native packed cache (``<root>/processed/data.igmc.npz``, resident in
HBM; ~100x smaller than the reference's pickled ``data.pt``), from which batches are rebuilt on the GPU.

There is no CPU fallback: constructing a dataset without the gfx950 library / a GPU raises.
"""
import os

import numpy as np
import scipy.sparse as ssp
import torch

from . import engine


class Data(object):
    """PyG-``Data``-shaped view of ONE enclosing subgraph (``x, edge_index, edge_type, y`` [+ side features])."""

    def __init__(self, x=None, edge_index=None, **kwargs):
        self.x = x
        self.edge_index = edge_index
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        return self.x.shape[0]

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self


class SparseRowIndexer(object):
    """Holds the training rating matrix (CSR, values = label + 1).  The reference builds per-row object
    arrays here (``:20-41``); the engine only needs the CSR itself, uploaded once by the dataset."""

    def __init__(self, csr_matrix):
        self.matrix = ssp.csr_matrix(csr_matrix)
        self.shape = self.matrix.shape

    def __getitem__(self, row_selector):
        return self.matrix[row_selector]


class SparseColIndexer(object):
    def __init__(self, csc_matrix):
        self.matrix = ssp.csc_matrix(csc_matrix)
        self.shape = self.matrix.shape

    def __getitem__(self, col_selector):
        return self.matrix[:, col_selector]


_graph_cache = {}


def _graph_for(A, device):
    """One HBM-resident copy per (matrix object, device)."""
    key = (id(A), device)
    ent = _graph_cache.get(key)
    if ent is None or ent[0] is not A:
        ent = (A, engine.Graph(A, device=device))
        _graph_cache[key] = ent
    return ent[1]


def _batch_to_tuples(d, L):
    """Downloaded single/multi-graph batch -> list of reference-style 7-tuples."""
    out = []
    for g in range(d['B']):
        lo, hi, nu = int(d['node_off'][g]), int(d['node_off'][g + 1]), int(d['n_users'][g])
        us, vs, rs = [], [], []
        for i in range(lo, lo + nu):                       # user rows hold the item -> user entries
            for p in range(d['row_ptr'][i], d['row_ptr'][i + 1]):
                us.append(i - lo)
                vs.append(int(d['col'][p]) - lo)
                rs.append(int(d['erel'][p]))
        order = np.lexsort((np.asarray(vs, np.int64), np.asarray(us, np.int64))) if us else []
        u = np.asarray(us, np.int32)[order] if len(us) else np.zeros(0, np.int32)
        v = np.asarray(vs, np.int32)[order] if len(us) else np.zeros(0, np.int32)
        r = np.asarray(rs, np.float32)[order] if len(us) else np.zeros(0, np.float32)
        out.append((u, v, r, d['node_label'][lo:hi].astype(int).tolist(), L - 1, float(d['y'][g]),
                    d['node_gid'][lo:lo + nu].copy(), d['node_gid'][lo + nu:hi].copy()))
    return out


def subgraph_extraction_labeling(ind, Arow, Acol, h=1, sample_ratio=1.0, max_nodes_per_hop=None,
                                 u_features=None, v_features=None, class_values=None, y=1,
                                 seed=0, epoch=0, device=None):
    """h-hop enclosing subgraph around link ``ind`` (reference ``:208-277``), extracted on the GPU.

    Returns the reference's 7-tuple ``(u, v, r, node_labels, max_node_label, y, node_features)`` with the
    target user / item first on their side; the other nodes are in ascending id order (the reference's
    order is CPython set-iteration order; the model is invariant to it).  ``Acol`` is accepted for
    signature compatibility (the CSC orientation is derived on the device)."""
    A = Arow.matrix if isinstance(Arow, SparseRowIndexer) else ssp.csr_matrix(Arow)
    device = torch.cuda.current_device() if device is None else device
    g = _graph_for(A, device)
    b = engine.Batch(g, 1, h, max_nodes_per_hop)
    yv = float(class_values[y]) if class_values is not None else float(y)
    lu = torch.tensor([int(ind[0])], dtype=torch.int32, device='cuda:%d' % device)
    lv = torch.tensor([int(ind[1])], dtype=torch.int32, device='cuda:%d' % device)
    ly = torch.tensor([yv], dtype=torch.float32, device='cuda:%d' % device)
    st = torch.cuda.current_stream().cuda_stream
    b.extract(lu.data_ptr(), lv.data_ptr(), ly.data_ptr(), None, 0, 1, sample_ratio, seed, epoch, st)
    d = b.download(st)
    u, v, r, labels, max_label, _, un, vn = _batch_to_tuples(d, 2 * h + 2)[0]
    node_features = None
    if u_features is not None and v_features is not None:      # reference :250-253, :272-275
        node_features = [u_features[[int(un[0])]][0] if hasattr(u_features, 'toarray') else u_features[int(un[0])],
                         v_features[[int(vn[0])]][0] if hasattr(v_features, 'toarray') else v_features[int(vn[0])]]
    b.close()
    yout = class_values[y] if class_values is not None else y
    return u, v, r, labels, max_label, yout, node_features


def one_hot(idx, length):
    idx = np.asarray(idx)
    x = np.zeros([len(idx), length])
    x[np.arange(len(idx)), idx] = 1.0
    return x


def _dense_row(f):
    return np.asarray(f.todense()).ravel() if hasattr(f, 'todense') else np.asarray(f).ravel()


def construct_pyg_graph(u, v, r, node_labels, max_node_label, y, node_features):
    """reference ``:280-297``."""
    u, v = torch.as_tensor(np.asarray(u), dtype=torch.long), torch.as_tensor(np.asarray(v), dtype=torch.long)
    r = torch.as_tensor(np.asarray(r), dtype=torch.long)
    edge_index = torch.stack([torch.cat([u, v]), torch.cat([v, u])], 0)
    edge_type = torch.cat([r, r])
    x = torch.as_tensor(one_hot(node_labels, max_node_label + 1), dtype=torch.float32)
    y = torch.as_tensor([y], dtype=torch.float32)
    data = Data(x, edge_index, edge_type=edge_type, y=y)
    if node_features is not None:
        if type(node_features) == list:
            u_feature, v_feature = node_features
            data.u_feature = torch.as_tensor(_dense_row(u_feature), dtype=torch.float32).unsqueeze(0)
            data.v_feature = torch.as_tensor(_dense_row(v_feature), dtype=torch.float32).unsqueeze(0)
        else:
            data.x = torch.cat([data.x, torch.as_tensor(node_features, dtype=torch.float32)], 1)
    return data


class DeviceBatch(object):
    """One extracted + collated batch living in an engine arena (valid until the arena is reused).

    Carries what ``IGMC.forward`` / the train loop need (``num_graphs``, ``y``) and materialises the
    PyG-style tensors (``x, edge_index, edge_type, batch``) lazily on request (host round trip; only for
    inspection / compatibility -- the model consumes the arena directly)."""

    def __init__(self, dataset, arena, num_graphs, positions, first, side=None):
        self.dataset = dataset
        self.arena = arena
        self.num_graphs = int(num_graphs)
        self._positions, self._first = positions, int(first)
        self.side = side            # [B, n_side] device tensor or None
        self._pyg = None
        self._y = None

    @property
    def link_pos(self):
        """Dataset positions of the batch's links (device int64)."""
        if self._positions is None:
            return torch.arange(self._first, self._first + self.num_graphs, device=self.dataset.link_y.device)
        return self._positions[self._first:self._first + self.num_graphs].long()

    @property
    def y(self):
        """Rating values of the batch (lazy: the kernels read them straight from the dataset's link array)."""
        if self._y is None:
            self._y = self.dataset.link_y.index_select(0, self.link_pos)
        return self._y

    def to(self, device):
        return self

    def _materialise(self):
        if self._pyg is None:
            d = self.arena.download(torch.cuda.current_stream().cuda_stream)
            N = d['N']
            dst = np.repeat(np.arange(N, dtype=np.int64), np.diff(d['row_ptr']).astype(np.int64))
            x = np.zeros((N, self.arena.num_labels), np.float32)
            x[np.arange(N), d['node_label']] = 1.0
            self._pyg = dict(x=torch.from_numpy(x), edge_index=torch.from_numpy(np.stack([d['col'].astype(np.int64), dst], 0)),
                             edge_type=torch.from_numpy(d['erel'].astype(np.int64)),
                             batch=torch.from_numpy(d['node_graph'].astype(np.int64)), raw=d)
        return self._pyg

    x = property(lambda self: self._materialise()['x'])
    edge_index = property(lambda self: self._materialise()['edge_index'])
    edge_type = property(lambda self: self._materialise()['edge_type'])
    batch = property(lambda self: self._materialise()['batch'])


class _EngineDataset(object):
    """Shared machinery of MyDataset / MyDynamicDataset."""
    dynamic = True

    def _setup(self, root, A, links, labels, h, sample_ratio, max_nodes_per_hop, u_features, v_features,
               class_values, max_num, device=None, seed=0):
        self.root = root
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.A = ssp.csr_matrix(A)
        self.Arow = SparseRowIndexer(self.A)
        self.Acol = None            # CSC orientation is built on the device (reference builds it with A.tocsc())
        self.links = (np.asarray(links[0]), np.asarray(links[1]))
        self.labels = np.asarray(labels)
        self.h = int(h)
        self.sample_ratio = float(sample_ratio)
        self.max_nodes_per_hop = None if max_nodes_per_hop is None else int(max_nodes_per_hop)
        self.u_features, self.v_features = u_features, v_features
        self.class_values = np.asarray(class_values, dtype=np.float64)
        self.max_num = max_num
        self.seed = int(seed)
        if max_num is not None:                       # reference :84-90 / :127-133
            np.random.seed(123)
            num_links = len(self.links[0])
            perm = np.random.permutation(num_links)[:max_num]
            self.links = (self.links[0][perm], self.links[1][perm])
            self.labels = self.labels[perm]
        dev = 'cuda:%d' % self.device
        self.graph = _graph_for(self.A, self.device)
        self.link_u = torch.from_numpy(np.ascontiguousarray(self.links[0], dtype=np.int32)).to(dev)
        self.link_v = torch.from_numpy(np.ascontiguousarray(self.links[1], dtype=np.int32)).to(dev)
        self.link_y = torch.from_numpy(self.class_values[self.labels].astype(np.float32)).to(dev)   # reference :247
        self._arenas = {}
        self._side = None
        if u_features is not None and v_features is not None:
            # only the two target nodes' features are used (reference :272-275, models.py:208-209)
            uf = ssp.csr_matrix(u_features)[self.links[0]]
            vf = ssp.csr_matrix(v_features)[self.links[1]]
            side = np.asarray(ssp.hstack([uf, vf]).todense(), dtype=np.float32)
            self._side = torch.from_numpy(np.ascontiguousarray(side)).to(dev)
            self.n_side_features = side.shape[1]
        else:
            self.n_side_features = 0

    # ---- reference surface
    def __len__(self):
        return len(self.links[0])

    @property
    def num_features(self):
        return 2 * self.h + 2          # one-hot of the node label (reference :246, :285)

    def arena(self, max_graphs, slot=0):
        key = (int(max_graphs), slot)
        if key not in self._arenas:
            a = engine.Batch(self.graph, int(max_graphs), self.h, self.max_nodes_per_hop)
            if self._side is not None:
                # the target nodes' feature rows are gathered by the extraction launch itself (device-side, also
                # under hipGraph replay of the training step)
                a.bind_side_source(self._side.data_ptr(), self.n_side_features)
            self._arenas[key] = a
        return self._arenas[key]

    def extract(self, positions, first, B, epoch=0, slot=0, max_graphs=None, stream=None):
        """Extract links ``positions[first:first+B]`` (device int32 tensor, or None = identity) into an arena."""
        arena = self.arena(max_graphs or B, slot)
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        ep = epoch if self.dynamic else 0
        arena.extract(self.link_u.data_ptr(), self.link_v.data_ptr(), self.link_y.data_ptr(),
                      None if positions is None else positions.data_ptr(), first, B, self.sample_ratio,
                      self.seed, ep, st)
        side = None
        if self._side is not None:        # view of the rows the extraction launch gathered (for inspection / get())
            if positions is None:
                idx = torch.arange(first, first + B, device=self.link_y.device)
            else:
                idx = positions[first:first + B].long()
            side = self._side.index_select(0, idx)
        return DeviceBatch(self, arena, B, positions, first, side)

    def get(self, idx):
        """One subgraph as a PyG-style ``Data`` (reference ``MyDynamicDataset.get``, ``:138-145``)."""
        db = self.extract(None, int(idx), 1, epoch=getattr(self, '_epoch', 0), slot=-1)
        p = db._materialise()
        data = Data(p['x'], p['edge_index'], edge_type=p['edge_type'], y=db.y.detach().cpu().view(1))
        if db.side is not None:
            nu = self.u_features.shape[1]
            data.u_feature = db.side[:, :nu].cpu()
            data.v_feature = db.side[:, nu:].cpu()
        return data

    def __getitem__(self, idx):
        return self.get(idx)


class MyDynamicDataset(_EngineDataset):
    """reference ``:113-145``: enclosing subgraphs extracted on the fly (re-sampled every epoch)."""
    dynamic = True

    def __init__(self, root, A, links, labels, h, sample_ratio, max_nodes_per_hop, u_features, v_features,
                 class_values, max_num=None, device=None, seed=0):
        self._setup(root, A, links, labels, h, sample_ratio, max_nodes_per_hop, u_features, v_features,
                    class_values, max_num, device, seed)


class MyDataset(_EngineDataset):
    """reference ``:69-110``: the "static" dataset -- every enclosing subgraph is extracted ONCE (``process``) and
    cached under ``<root>/processed/`` (``data.pt`` / ``data_{max_num}.pt`` in the reference: a pickled PyG collate of
    every graph, ~12 GB for ml_100k).  The native cache (``data.igmc.npz`` / ``data_{max_num}.igmc.npz``) keeps what
    cannot be re-derived without the sampler -- per link the node sets with their hop distances, packed int32 / uint8
    arrays -- and is uploaded to HBM once; batches are rebuilt from it on the GPU (``igmc_extract_batch_cached``: induced
    edges, labels, collation).  A dataset without ``root`` (``links2subgraphs``) or with ``cache=False`` re-derives
    its subgraphs with an epoch-independent sampling key instead, which yields the same graphs.  ``parallel`` is accepted
    and ignored (the reference's ``mp.Pool`` extraction is the GPU kernels here)."""
    dynamic = False

    def __init__(self, root, A, links, labels, h, sample_ratio, max_nodes_per_hop, u_features, v_features,
                 class_values, max_num=None, parallel=True, device=None, seed=0, cache=None):
        self.parallel = parallel
        self._setup(root, A, links, labels, h, sample_ratio, max_nodes_per_hop, u_features, v_features,
                    class_values, max_num, device, seed)
        self._cache = None
        if cache is None:
            cache = root is not None and os.environ.get('IGMC_STATIC_CACHE', '1') != '0'
        if cache and root is not None:
            self.process()

    @property
    def processed_file_names(self):
        name = 'data.igmc.npz'
        if self.max_num is not None:
            name = 'data_{}.igmc.npz'.format(self.max_num)
        return [name]

    @property
    def processed_paths(self):
        return [os.path.join(self.root, 'processed', n) for n in self.processed_file_names]

    def _fingerprint(self):
        """What the cached node sets depend on (a stale cache is rebuilt, never silently reused)."""
        import hashlib
        h = hashlib.sha256()
        A = self.A
        for arr in (A.indptr, A.indices, np.asarray(A.data, np.float32), np.asarray(self.links[0], np.int64),
                    np.asarray(self.links[1], np.int64)):
            h.update(np.ascontiguousarray(arr).tobytes())
        h.update(repr((A.shape, self.h, self.sample_ratio, self.max_nodes_per_hop, self.seed)).encode())
        return h.hexdigest()

    def process(self, chunk=512):
        """reference ``MyDataset.process`` (``:101-110``): extract every subgraph once and store the cache; or load it."""
        from . import parallel
        path = self.processed_paths[0]
        fp = self._fingerprint()

        def load():
            if os.path.exists(path):
                try:
                    zz = np.load(path)
                    if str(zz['fingerprint']) == fp:
                        return zz
                except Exception:
                    pass
            return None
        # several ranks: rank 0 builds the cache (the others would extract the same subgraphs and race on the same file),
        # everybody loads it after the barrier
        builder = parallel.rank() == 0
        z = load() if builder else None
        failure = None
        if z is None and builder:
            try:            # (a failure here must reach the ranks waiting below, not leave them at a barrier: ADVICE r3)
                n = len(self)
                uoff, voff = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
                un, vn, ud, vd = [], [], [], []
                for first in range(0, n, chunk):
                    B = min(chunk, n - first)
                    arena = self.arena(chunk, slot='process')
                    st = torch.cuda.current_stream().cuda_stream
                    arena.extract(self.link_u.data_ptr(), self.link_v.data_ptr(), self.link_y.data_ptr(), None, first, B,
                                  self.sample_ratio, self.seed, 0, st)
                    d = arena.download(st)
                    for g in range(B):
                        lo, hi, nu = int(d['node_off'][g]), int(d['node_off'][g + 1]), int(d['n_users'][g])
                        un.append(d['node_gid'][lo:lo + nu])
                        vn.append(d['node_gid'][lo + nu:hi])
                        ud.append(d['node_label'][lo:lo + nu] // 2)
                        vd.append(d['node_label'][lo + nu:hi] // 2)
                        uoff[first + g + 1] = uoff[first + g] + nu
                        voff[first + g + 1] = voff[first + g] + (hi - lo - nu)
                self._arenas.pop((chunk, 'process'), None)
                cat = lambda xs, dt: np.ascontiguousarray(np.concatenate(xs).astype(dt)) if xs else np.zeros(0, dt)
                z = dict(uoff=uoff, voff=voff, unodes=cat(un, np.int32), vnodes=cat(vn, np.int32), udist=cat(ud, np.uint8),
                         vdist=cat(vd, np.uint8), fingerprint=np.array(fp))
                os.makedirs(os.path.dirname(path), exist_ok=True)
                tmp = '%s.%d.tmp.npz' % (path, os.getpid())
                np.savez(tmp, **z)
                os.replace(tmp, path)
            except Exception as e:
                if parallel.world_size() <= 1:
                    raise
                failure = '%s: %s' % (type(e).__name__, e)
        if parallel.world_size() > 1:
            # rank 0's verdict doubles as the barrier: every rank raises when the build failed
            failure = parallel.broadcast_object(failure, 0)
            if failure is not None:
                raise RuntimeError('rank 0 could not build the static-dataset cache %s (%s)' % (path, failure))
            if not builder:
                z = load()
                if z is None:
                    raise RuntimeError('rank %d: the static-dataset cache %s written by rank 0 is missing or stale '
                                       '(ranks must share the data directory)' % (parallel.rank(), path))
        dev = 'cuda:%d' % self.device
        self._cache_t = {k: torch.from_numpy(np.ascontiguousarray(z[k])).to(dev) for k in
                         ('uoff', 'voff', 'unodes', 'vnodes', 'udist', 'vdist')}
        self._cache = {k: t.data_ptr() for k, t in self._cache_t.items()}

    def extract(self, positions, first, B, epoch=0, slot=0, max_graphs=None, stream=None):
        if self._cache is None:
            return super().extract(positions, first, B, epoch=epoch, slot=slot, max_graphs=max_graphs, stream=stream)
        arena = self.arena(max_graphs or B, slot)
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        arena.extract_cached(self._cache, self.link_y.data_ptr(), None if positions is None else positions.data_ptr(),
                             first, B, st)
        side = None
        if self._side is not None:
            idx = (torch.arange(first, first + B, device=self.link_y.device) if positions is None
                   else positions[first:first + B].long())
            side = self._side.index_select(0, idx)
        return DeviceBatch(self, arena, B, positions, first, side)


def links2subgraphs(Arow, Acol, links, labels, h=1, sample_ratio=1.0, max_nodes_per_hop=None, u_features=None,
                    v_features=None, class_values=None, parallel=True, batch_size=512):
    """reference ``:148-205``: list of PyG-style graphs for all links (GPU extraction in chunks)."""
    A = Arow.matrix if isinstance(Arow, SparseRowIndexer) else ssp.csr_matrix(Arow)
    ds = MyDataset(None, A, links, labels, h, sample_ratio, max_nodes_per_hop, u_features, v_features, class_values)
    out = []
    n = len(ds)
    for first in range(0, n, batch_size):
        B = min(batch_size, n - first)
        db = ds.extract(None, first, B, max_graphs=batch_size)
        d = db._materialise()['raw']
        for k, (u, v, r, labs, ml, y, un, vn) in enumerate(_batch_to_tuples(d, 2 * h + 2)):
            nf = None
            if db.side is not None:
                nu = u_features.shape[1]
                s = db.side[k].cpu().numpy()
                nf = [s[:nu], s[nu:]]
            out.append(construct_pyg_graph(u, v, r, labs, ml, y, nf))
    return out
