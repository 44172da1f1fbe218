"""Minimal ``torch_geometric.data`` surface used by the reference extractor.

Only attribute storage is provided: ``Data(x, edge_index, **kw)`` keeps its
arguments as attributes (what ``construct_pyg_graph`` relies on,
``/root/reference/util_functions.py:280-297``).
"""


class Data(object):
    def __init__(self, x=None, edge_index=None, **kwargs):
        self.x = x
        self.edge_index = edge_index
        for k, v in kwargs.items():
            setattr(self, k, v)


class Dataset(object):
    def __init__(self, root=None, *a, **k):
        self.root = root


class InMemoryDataset(Dataset):
    pass
