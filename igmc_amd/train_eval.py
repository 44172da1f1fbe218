"""Drop-in mirror of the reference's ``train_eval.py`` on the gfx950 engine: same function names,
arguments, return values, log strings and checkpoint formats --

* ``train_multiple_epochs``      reference ``train_eval.py:23-111``
* ``test_once``                  reference ``:114-139``
* ``train``                      reference ``:149-179``
* ``eval_loss`` / ``eval_rmse``  reference ``:182-205``
* ``eval_loss_ensemble`` / ``eval_rmse_ensemble``   reference ``:208-245``

What differs by design: the ``DataLoader`` yields device-resident batches extracted by HIP kernels (no
worker processes / pickling / H2D); one optimisation step = extract -> forward -> loss (+ARR) -> backward
-> [one flat RCCL all-reduce] -> fused Adam, all on the GPU with no host synchronisation inside the epoch
(the reference syncs every step for ``loss.item()`` and ``empty_cache()``, ``train_eval.py:176-178``).
"""
import math
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib, engine, parallel
from .models import IGMC
from .stepgraph import EvalGraph, StepGraph, _group_size_for
from .util_functions import DeviceBatch

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')


class DataLoader(object):
    """Replacement of the PyG ``DataLoader`` call sites (reference ``train_eval.py:44-51, 121``):
    iterates a dataset in batches of ``batch_size`` links (last one may be smaller, like the reference's
    ``drop_last=False``) and yields :class:`DeviceBatch` objects.  ``num_workers`` is accepted and ignored.
    Under data parallelism every rank walks its own shard of the same permutation."""

    def __init__(self, dataset, batch_size=1, shuffle=False, num_workers=0, pad_shards=None, **_ignored):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.epoch = 0
        self.pad_shards = shuffle if pad_shards is None else pad_shards
        self._base_seed = int(torch.initial_seed()) & 0x7FFFFFFF

    def _n_local(self):
        n, G = len(self.dataset), parallel.world_size()
        if G <= 1:
            return n
        return (n + G - 1) // G if self.pad_shards else len(range(parallel.rank(), n, G))

    def __len__(self):
        return (self._n_local() + self.batch_size - 1) // self.batch_size

    def epoch_positions(self):
        """Start a new epoch: this rank's link positions (1-D CPU int64 tensor), shuffled when requested."""
        self.epoch += 1
        n = len(self.dataset)
        if self.shuffle:
            gen = torch.Generator()
            gen.manual_seed(self._base_seed + 7919 * self.epoch)
            perm = torch.randperm(n, generator=gen)
        else:
            perm = torch.arange(n)
        return parallel.shard_positions(perm, parallel.rank(), parallel.world_size(), pad=self.pad_shards)

    def __iter__(self):
        perm = self.epoch_positions().to(device=self.dataset.link_y.device, dtype=torch.int32)
        nl = len(perm)
        for first in range(0, nl, self.batch_size):
            B = min(self.batch_size, nl - first)
            yield self.dataset.extract(perm, first, B, epoch=self.epoch, max_graphs=self.batch_size)


class FlatAdam(object):
    """``torch.optim.Adam`` semantics (betas (0.9,0.999), eps 1e-8, L2 ``weight_decay`` added to the gradient)
    as ONE fused kernel over the model's flat parameter buffer.  ``param_groups`` / ``state_dict()`` /
    ``load_state_dict()`` follow torch's Adam format so the reference's LR decay (``train_eval.py:94-96``)
    and optimiser checkpoints (``Main.py:36-45``, ``train_eval.py:60-62``) keep working."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        self.model = model
        flat = model.flat_parameters()
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.t = 0
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False,
                                  params=list(model.parameters()))]

    def zero_grad(self, set_to_none=False):
        self.model.flat_grad().zero_()

    def step(self, stream=None):
        g = self.param_groups[0]
        m = self.model
        flat, grad = m.flat_parameters(), m.flat_grad()
        if self.exp_avg.device != flat.device:
            self.exp_avg, self.exp_avg_sq = self.exp_avg.to(flat.device), self.exp_avg_sq.to(flat.device)
        self.t += 1
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        engine.adam_step(_lib.load(), flat.data_ptr(), grad.data_ptr(), self.exp_avg.data_ptr(),
                         self.exp_avg_sq.data_ptr(), flat.numel(), self.t, g['lr'], g['betas'][0], g['betas'][1],
                         g['eps'], g['weight_decay'], stream=st)

    def state_dict(self):
        state = {}
        names = [k for k, _ in self.model.named_parameters()]
        where = {k: (o, n, s) for (k, o, n, s) in self.model._views}
        for i, k in enumerate(names):
            o, n, s = where[k]
            state[i] = dict(step=torch.tensor(float(self.t)), exp_avg=self.exp_avg[o:o + n].view(s).clone(),
                            exp_avg_sq=self.exp_avg_sq[o:o + n].view(s).clone())
        g = dict(self.param_groups[0])
        g['params'] = list(range(len(names)))
        return dict(state=state if self.t > 0 else {}, param_groups=[g])

    def load_state_dict(self, sd):
        names = [k for k, _ in self.model.named_parameters()]
        where = {k: (o, n, s) for (k, o, n, s) in self.model._views}
        for i, k in enumerate(names):
            if i in sd['state']:
                o, n, s = where[k]
                st = sd['state'][i]
                self.exp_avg[o:o + n].copy_(st['exp_avg'].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
                self.t = int(float(st['step']))
        for key in ('lr', 'betas', 'eps', 'weight_decay'):
            if key in sd['param_groups'][0]:
                self.param_groups[0][key] = sd['param_groups'][0][key]


def Adam(params_or_model, lr=1e-3, weight_decay=0, **kw):
    """``Adam(model.parameters(), ...)`` call-site compatibility (reference ``train_eval.py:54``)."""
    if isinstance(params_or_model, IGMC):
        return FlatAdam(params_or_model, lr=lr, weight_decay=weight_decay, **kw)
    return torch.optim.Adam(params_or_model, lr=lr, weight_decay=weight_decay, **kw)


def train_multiple_epochs(train_dataset, test_dataset, model, epochs, batch_size, lr, lr_decay_factor,
                          lr_decay_step_size, weight_decay, ARR=0, test_freq=1, logger=None, continue_from=None,
                          res_dir=None):
    rmses = []
    # the reference sizes DataLoader workers by class name (train_eval.py:40,46); here extraction is on the GPU
    train_loader = DataLoader(train_dataset, batch_size, shuffle=True)
    test_loader = DataLoader(test_dataset, batch_size, shuffle=False)

    model.to(device).reset_parameters()
    if parallel.world_size() > 1:                 # identical replicas: rank 0's initial weights everywhere
        parallel.broadcast_(model.flat_parameters(), 0)
    optimizer = FlatAdam(model, lr=lr, weight_decay=weight_decay)
    start_epoch = 1
    if continue_from is not None:
        model.load_state_dict(torch.load(os.path.join(res_dir, 'model_checkpoint{}.pth'.format(continue_from)),
                                         map_location=device))
        optimizer.load_state_dict(torch.load(os.path.join(res_dir, 'optimizer_checkpoint{}.pth'.format(continue_from)),
                                             map_location=device))
        start_epoch = continue_from + 1
        epochs -= continue_from
        # the epoch shuffles, the dynamic sampling keys and the dropout hashes are keyed on the loader's epoch counter
        # and the model's step counter: a resumed run continues them instead of replaying epochs 1..
        train_loader.epoch = continue_from
        test_loader.epoch = continue_from
        model._step = continue_from * len(train_loader)

    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t_start = time.perf_counter()
    for epoch in range(start_epoch, epochs + start_epoch):
        train_loss = train(model, optimizer, train_loader, device, regression=True, ARR=ARR, epoch=epoch)
        if epoch % test_freq == 0:
            rmses.append(eval_rmse(model, test_loader, device))
        else:
            rmses.append(np.nan)
        eval_info = {'epoch': epoch, 'train_loss': train_loss, 'test_rmse': rmses[-1]}
        if parallel.rank() == 0:
            print('Epoch {}, train loss {:.6f}, test rmse {:.6f}'.format(*eval_info.values()))
        if epoch % lr_decay_step_size == 0:          # after the eval, before the logger saves (ref :94-99)
            for param_group in optimizer.param_groups:
                param_group['lr'] = lr_decay_factor * param_group['lr']
        if logger is not None and parallel.rank() == 0:
            logger(eval_info, model, optimizer)

    if torch.cuda.is_available():
        torch.cuda.synchronize()
    duration = time.perf_counter() - t_start
    if parallel.rank() == 0:
        print('Final Test RMSE: {:.6f}, Duration: {:.6f}'.format(rmses[-1], duration))
    return rmses[-1]


def test_once(test_dataset, model, batch_size, logger=None, ensemble=False, checkpoints=None):
    test_loader = DataLoader(test_dataset, batch_size, shuffle=False)
    model.to(device)
    t_start = time.perf_counter()
    if ensemble and checkpoints:
        rmse = eval_rmse_ensemble(model, checkpoints, test_loader, device, show_progress=True)
    else:
        rmse = eval_rmse(model, test_loader, device, show_progress=True)
    duration = time.perf_counter() - t_start
    if parallel.rank() == 0:
        print('Test Once RMSE: {:.6f}, Duration: {:.6f}'.format(rmse, duration))
    epoch_info = 'test_once' if not ensemble else 'ensemble'
    eval_info = {'epoch': epoch_info, 'train_loss': 0, 'test_rmse': rmse}
    if logger is not None and parallel.rank() == 0:
        logger(eval_info, None, None)
    return rmse


def num_graphs(data):
    return data.num_graphs


def train(model, optimizer, loader, device, regression=False, ARR=0, show_progress=False, epoch=None):
    """One epoch (reference ``train_eval.py:149-179``); returns mean over samples of (batch MSE + ARR term)."""
    model.train()
    if not regression:
        raise NotImplementedError('only the regression objective (what the reference runs) is implemented')
    G = parallel.world_size()
    n_total = 0
    if isinstance(optimizer, FlatAdam):
        # ---- fused path (IGMC and the sort-pool family DGCNN_RS): no autograd, no host sync inside the epoch; the steps
        #      are replayed as hipGraphs, groups at a time
        sg = getattr(loader, '_stepgraph', None)
        if sg is None or sg.model is not model or sg.opt is not optimizer or sg.ARR != float(ARR):
            if sg is not None:
                sg.detach()
            sg = StepGraph(model, optimizer, loader.dataset, loader.batch_size, ARR)
            loader._stepgraph = sg
        total, n_total = sg.run_epoch(loader.epoch_positions(), loader.epoch)
        if G > 1:
            cnt = torch.tensor([float(n_total)], dtype=torch.float64, device=total.device)
            tot = total.clone()
            parallel.all_reduce_sum_(tot)
            parallel.all_reduce_sum_(cnt)
            return float(tot.item() / cnt.item())
        return float(total.item()) / max(len(loader.dataset), 1)
    # ---- generic path: any torch optimiser through the differentiable forward (reference-style loop)
    total_loss = 0.0
    for data in loader:
        optimizer.zero_grad()
        out = model(data)
        loss = F.mse_loss(out, data.y.view(-1))
        if ARR != 0:
            for gconv in model.convs:
                w = torch.matmul(gconv.att, gconv.basis.view(gconv.num_bases, -1)).view(
                    gconv.num_relations, gconv.in_channels, gconv.out_channels)
                loss = loss + ARR * torch.sum((w[1:, :, :] - w[:-1, :, :]) ** 2)
        loss.backward()
        total_loss += loss.item() * num_graphs(data)
        optimizer.step()
    return total_loss / len(loader.dataset)


def eval_loss(model, loader, device, regression=False, show_progress=False):
    """Mean squared error over the loader's dataset (reference ``train_eval.py:182-199``)."""
    model.eval()
    if not regression:
        raise NotImplementedError('only the regression objective (what the reference runs) is implemented')
    flat = model.flat_parameters()
    nb = loader._n_local() // loader.batch_size
    if not hasattr(model, 'forward_into') and nb >= 8 and os.environ.get('IGMC_NO_EVAL_GRAPH', '0') != '1':
        # the same grouped pipeline as training: forward + squared-error accumulation per step, extraction of the next
        # group beside it, pairs of groups replayed from a hipGraph (bit-identical to the loop below)
        eg = getattr(loader, '_evalgraph', None)
        if eg is None or eg.model is not model:
            if eg is not None:
                eg.detach()
            # (group size: the one that leaves the fewest steps outside whole graph launches -- 100 batches as two launches
            #  of 50 steps instead of one of 64 and 36 eager steps, which took as long as the 64)
            eg = EvalGraph(model, loader.dataset, loader.batch_size, group=_group_size_for(nb))
            loader._evalgraph = eg
        acc = eg.run(loader.epoch_positions(), loader.epoch).clone()
        _check_workspaces(model)
        parallel.all_reduce_sum_(acc)
        sse, cnt = acc.tolist()
        return sse / max(cnt, 1.0)
    acc = torch.zeros(2, dtype=torch.float64, device=flat.device)
    out = None
    for data in loader:
        ws = model._workspace(data)
        st = torch.cuda.current_stream().cuda_stream
        B = data.num_graphs
        if out is None or out.numel() < B:
            out = torch.empty(max(B, loader.batch_size), dtype=torch.float32, device=flat.device)
        if hasattr(model, 'forward_into'):           # readout families other than IGMC's (DGCNN_RS)
            model.forward_into(data, out, training=False, stream=st)
        else:
            ws.forward(flat.data_ptr(), data.arena, out.data_ptr(), training=False,
                       multiply_by=float(model.multiply_by), stream=st)
        ws.sse_accumulate(out.data_ptr(), data.arena, acc.data_ptr(), stream=st)
    _check_workspaces(model)
    parallel.all_reduce_sum_(acc)
    sse, cnt = acc.tolist()
    return sse / max(cnt, 1.0)


def _check_workspaces(model):
    """A bounded device-side wait of the subgraph kernel's cluster exchange that timed out leaves partial
    activations: never report an RMSE computed from them (synchronises the stream)."""
    st = torch.cuda.current_stream().cuda_stream
    for ws in model._ws.values():
        ws.lib.call('igmc_model_check', ws.handle, engine._p(st))


def eval_rmse(model, loader, device, show_progress=False):
    return math.sqrt(eval_loss(model, loader, device, True, show_progress))


def eval_loss_ensemble(model, checkpoints, loader, device, regression=False, show_progress=False):
    """Mean of the predictions of several checkpoints, then MSE (reference ``train_eval.py:208-239``)."""
    Outs, ys = [], []
    for i, checkpoint in enumerate(checkpoints):
        model.load_state_dict(torch.load(checkpoint, map_location=model.flat_parameters().device))
        model.eval()
        outs = []
        for data in loader:
            if i == 0:
                ys.append(data.y.view(-1))
            with torch.no_grad():
                outs.append(model(data))
        Outs.append(torch.cat(outs, 0).view(-1, 1))
        _check_workspaces(model)
    ys = torch.cat(ys, 0)
    Outs = torch.cat(Outs, 1).mean(1)
    acc = torch.stack([F.mse_loss(Outs, ys, reduction='sum').double(),
                       torch.tensor(float(len(ys)), dtype=torch.float64, device=ys.device)])
    parallel.all_reduce_sum_(acc)
    sse, cnt = acc.tolist()
    return sse / max(cnt, 1.0)


def eval_rmse_ensemble(model, checkpoints, loader, device, show_progress=False):
    return math.sqrt(eval_loss_ensemble(model, checkpoints, loader, device, True, show_progress))


def visualize(*args, **kwargs):
    raise NotImplementedError('visualize (reference train_eval.py:248-322) is a plotting helper and out of scope')
