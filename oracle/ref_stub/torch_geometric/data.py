"""``torch_geometric.data`` surface used by the reference (TEST INFRASTRUCTURE ONLY).

* ``Data(x, edge_index, **kw)`` keeps its arguments as attributes (what ``construct_pyg_graph`` relies on,
  ``/root/reference/util_functions.py:280-297``); ``num_nodes`` (``/root/reference/models.py:71``) and ``to``
  (``/root/reference/train_eval.py:159``) as PyG defines them.
* ``DataLoader(dataset, batch_size, shuffle, num_workers)`` (``/root/reference/train_eval.py:44-51,121``): batches of
  ``Batch.from_data_list`` in dataset order (``shuffle`` draws one ``torch.randperm`` per epoch like torch's
  ``RandomSampler``); ``num_workers`` is accepted and ignored (single process).
"""
import torch


class Data(object):
    def __init__(self, x=None, edge_index=None, **kwargs):
        self.x = x
        self.edge_index = edge_index
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        return int(self.x.shape[0])

    def to(self, device):
        return self


class Dataset(object):
    def __init__(self, root=None, *a, **k):
        self.root = root


class InMemoryDataset(Dataset):
    pass


class Batch(Data):
    """PyG ``Batch.from_data_list``: keys containing ``index`` get the cumulative node offset and are concatenated
    on the last dimension, every other tensor on dimension 0; ``batch`` = graph id of a node."""

    @staticmethod
    def from_data_list(data_list):
        b = Batch()
        keys = [k for k in vars(data_list[0]) if getattr(data_list[0], k) is not None]
        cols = {k: [] for k in keys}
        bvec, off = [], 0
        for g, d in enumerate(data_list):
            n = d.num_nodes
            for k in keys:
                v = getattr(d, k)
                cols[k].append(v + off if 'index' in k else v)
            bvec.append(torch.full((n,), g, dtype=torch.long))
            off += n
        for k in keys:
            setattr(b, k, torch.cat(cols[k], -1 if 'index' in k else 0))
        b.batch = torch.cat(bvec, 0)
        b.num_graphs = len(data_list)
        return b


class DataLoader(object):
    def __init__(self, dataset, batch_size=1, shuffle=False, num_workers=0, **kwargs):
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), bool(shuffle)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n).tolist() if self.shuffle else list(range(n))
        for s in range(0, n, self.batch_size):
            yield Batch.from_data_list([self.dataset[i] for i in order[s:s + self.batch_size]])


class DenseDataLoader(DataLoader):
    """Name only (imported at ``/root/reference/train_eval.py:12``, never used on the IGMC path)."""
