"""IGMC experiment driver on the MI355X engine -- same command line, result files and log format as the
reference's ``Main.py`` (flags: reference ``Main.py:49-136``; outputs ``results/<name><appendix>_<mode>/
{log.txt,cmd_input.txt,model_checkpointN.pth,optimizer_checkpointN.pth}``: ``Main.py:31-45,188-210``).

    python Main.py --data-name ml_1m --save-appendix _mnph100 --data-appendix _mnph100 --max-nodes-per-hop 100 \
        --testing --epochs 40 --save-interval 5 --adj-dropout 0 --lr-decay-step-size 20 --ensemble --dynamic-train

Multi-GPU (one process per GPU, RCCL gradient all-reduce):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 Main.py ...

Differences from the reference script, on purpose: the crash paths it has as shipped (``rmse`` undefined
without ``--ensemble``, ``args.epoch``: ``Main.py:471-472``) are fixed -- the value returned by
``train_multiple_epochs`` is used; ``--visualize`` is out of scope; MovieLens is read from ``raw_data/`` when an
operator provides it and otherwise replaced by the MovieLens-shaped synthetic generator (no network here).
"""
from __future__ import print_function

import argparse
import math
import os
import random
import sys
from shutil import rmtree

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from igmc_amd.hostcpu import limit_host_threads  # noqa: E402

limit_host_threads()          # before numpy / torch: pools sized for the container's CPU quota, not the visible CPUs

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(int(os.environ['OMP_NUM_THREADS']))

from igmc_amd import parallel  # noqa: E402
from igmc_amd.models import IGMC
from igmc_amd.preprocessing import create_trainvaltest_split, load_data_monti, load_official_trainvaltest_split
from igmc_amd.train_eval import test_once, train_multiple_epochs
from igmc_amd.util_functions import MyDataset, MyDynamicDataset


def build_parser():
    p = argparse.ArgumentParser(description='Inductive Graph-based Matrix Completion (MI355X engine)')
    # general settings
    p.add_argument('--testing', action='store_true', default=False,
                   help='if set, use testing mode which splits all ratings into train/test; otherwise train/val/test')
    p.add_argument('--no-train', action='store_true', default=False, help='skip training, only test')
    p.add_argument('--debug', action='store_true', default=False, help='use 1000 links per split')
    p.add_argument('--data-name', default='ml_100k', help='dataset name')
    p.add_argument('--data-appendix', default='', help='appendix of the data cache directory name')
    p.add_argument('--save-appendix', default='', help='appendix of the results directory name')
    p.add_argument('--max-train-num', type=int, default=None)
    p.add_argument('--max-val-num', type=int, default=None)
    p.add_argument('--max-test-num', type=int, default=None)
    p.add_argument('--seed', type=int, default=1, metavar='S')
    p.add_argument('--data-seed', type=int, default=1234, metavar='S')
    p.add_argument('--reprocess', action='store_true', default=False)
    p.add_argument('--dynamic-train', action='store_true', default=False,
                   help='re-sample enclosing subgraphs every epoch (MyDynamicDataset)')
    p.add_argument('--dynamic-test', action='store_true', default=False)
    p.add_argument('--dynamic-val', action='store_true', default=False)
    p.add_argument('--keep-old', action='store_true', default=False)
    p.add_argument('--save-interval', type=int, default=10)
    # subgraph extraction settings
    p.add_argument('--hop', default=1, metavar='S')
    p.add_argument('--sample-ratio', type=float, default=1.0)
    p.add_argument('--max-nodes-per-hop', default=10000)
    p.add_argument('--use-features', action='store_true', default=False)
    # edge dropout settings
    p.add_argument('--adj-dropout', type=float, default=0.2)
    p.add_argument('--force-undirected', action='store_true', default=False)
    # optimization settings
    p.add_argument('--continue-from', type=int, default=None)
    p.add_argument('--lr', type=float, default=1e-3, metavar='LR')
    p.add_argument('--lr-decay-step-size', type=int, default=50)
    p.add_argument('--lr-decay-factor', type=float, default=0.1)
    p.add_argument('--epochs', type=int, default=80, metavar='N')
    p.add_argument('--batch-size', type=int, default=50, metavar='N')
    p.add_argument('--test-freq', type=int, default=1, metavar='N')
    p.add_argument('--ARR', type=float, default=0.001)
    # transfer learning, ensemble, visualization
    p.add_argument('--dgcnn-rs', action='store_true', default=False,
                   help='train DGCNN_RS (sort-pool readout, reference models.py:123-167) instead of IGMC')
    p.add_argument('--k', type=float, default=0.6,
                   help='sort-pool size of DGCNN_RS: a percentile of the subgraph sizes if < 1 (reference Main.py:369)')
    p.add_argument('--transfer', default='')
    p.add_argument('--num-relations', type=int, default=5)
    p.add_argument('--multiply-by', type=int, default=1)
    p.add_argument('--visualize', action='store_true', default=False)
    p.add_argument('--ensemble', action='store_true', default=False)
    p.add_argument('--standard-rating', action='store_true', default=False)
    # sparsity experiment settings
    p.add_argument('--ratio', type=float, default=1.0)
    return p


def rating_maps(args):
    """reference ``Main.py:153-177``."""
    rating_map, post_rating_map = None, None
    if args.standard_rating:
        if args.data_name in ['flixster', 'ml_10m']:
            rating_map = {x: int(math.ceil(x)) for x in np.arange(0.5, 5.01, 0.5).tolist()}
        elif args.data_name == 'yahoo_music':
            rating_map = {x: (x - 1) // 20 + 1 for x in range(1, 101)}
    if args.transfer:
        if args.data_name in ['flixster', 'ml_10m']:
            levels = np.arange(0.5, 5.01, 0.5).tolist()
            post_rating_map = {x: int(i // (10 / args.num_relations)) for i, x in enumerate(levels)}
        elif args.data_name == 'yahoo_music':
            levels = np.arange(1, 101).tolist()
            post_rating_map = {x: int(i // (100 / args.num_relations)) for i, x in enumerate(levels)}
        else:
            levels = np.arange(1, 6).tolist()
            post_rating_map = {x: int(i // (5 / args.num_relations)) for i, x in enumerate(levels)}
    return rating_map, post_rating_map


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank, world = parallel.init_from_env()
    torch.manual_seed(args.seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(args.seed)
    if rank == 0:
        print(args)
    random.seed(args.seed)
    np.random.seed(args.seed)
    args.hop = int(args.hop)
    if args.max_nodes_per_hop is not None:
        args.max_nodes_per_hop = int(args.max_nodes_per_hop)
    rating_map, post_rating_map = rating_maps(args)

    args.file_dir = os.getcwd()          # the reference resolves the literal '__file__' against the cwd (Main.py:183)
    val_test_appendix = 'testmode' if args.testing else 'valmode'
    args.res_dir = os.path.join(args.file_dir, 'results/{}{}_{}'.format(args.data_name, args.save_appendix,
                                                                        val_test_appendix))
    src_dir = args.res_dir if args.transfer == '' else args.transfer
    args.model_pos = os.path.join(src_dir, 'model_checkpoint{}.pth'.format(args.epochs))
    if rank == 0:
        os.makedirs(args.res_dir, exist_ok=True)
        with open(os.path.join(args.res_dir, 'cmd_input.txt'), 'a') as f:
            f.write('python ' + ' '.join(sys.argv) + '\n')
    parallel.barrier()

    def logger(info, model, optimizer):
        """reference ``Main.py:31-45``."""
        epoch, train_loss, test_rmse = info['epoch'], info['train_loss'], info['test_rmse']
        with open(os.path.join(args.res_dir, 'log.txt'), 'a') as f:
            f.write('Epoch {}, train loss {:.4f}, test rmse {:.6f}\n'.format(epoch, train_loss, test_rmse))
        if type(epoch) == int and epoch % args.save_interval == 0:
            print('Saving model states...')
            if model is not None:
                torch.save(model.state_dict(), os.path.join(args.res_dir, 'model_checkpoint{}.pth'.format(epoch)))
            if optimizer is not None:
                torch.save(optimizer.state_dict(),
                           os.path.join(args.res_dir, 'optimizer_checkpoint{}.pth'.format(epoch)))

    # ---- data (reference Main.py:228-251)
    if args.data_name in ['flixster', 'douban', 'yahoo_music']:
        split = load_data_monti(args.data_name, args.testing, rating_map, post_rating_map)
    elif args.data_name == 'ml_100k':       # reference Main.py:236-243: the official u1.base / u1.test split
        if rank == 0:
            print('Using official MovieLens split u1.base/u1.test with 20% validation...')
        split = load_official_trainvaltest_split(args.data_name, args.testing, rating_map, post_rating_map, args.ratio,
                                                 verbose=(rank == 0))
    else:
        split = create_trainvaltest_split(args.data_name, args.data_seed, args.testing, None, True, rank == 0,
                                          rating_map, post_rating_map, args.ratio)
    (u_features, v_features, adj_train, train_labels, train_u, train_v, val_labels, val_u, val_v,
     test_labels, test_u, test_v, class_values) = split
    if rank == 0:
        print('All ratings are:')
        print(class_values)
    if args.use_features:
        if u_features is None or v_features is None:
            raise ValueError('--use-features: dataset %s has no side features here' % args.data_name)
        u_features, v_features = u_features.toarray(), v_features.toarray()
        n_features = u_features.shape[1] + v_features.shape[1]
    else:
        u_features, v_features, n_features = None, None, 0
    if args.debug:
        num_data = 1000
        train_u, train_v, train_labels = train_u[:num_data], train_v[:num_data], train_labels[:num_data]
        val_u, val_v, val_labels = val_u[:num_data], val_v[:num_data], val_labels[:num_data]
        test_u, test_v, test_labels = test_u[:num_data], test_v[:num_data], test_labels[:num_data]
    if rank == 0:
        print('#train: %d, #val: %d, #test: %d' % (len(train_u), len(val_u), len(test_u)))

    # ---- datasets (reference Main.py:297-350): GPU-resident; static datasets keep their node sets under
    #      data/<name>/<mode>/<part>/processed/ (MyDataset.process: built by rank 0, loaded by the others)
    data_combo = (args.data_name, args.data_appendix, val_test_appendix)
    if args.reprocess:
        if rank == 0:
            for part in ('train', 'val', 'test'):
                d = 'data/{}{}/{}/{}'.format(*(data_combo + (part,)))
                if os.path.isdir(d):
                    rmtree(d)
        parallel.barrier()      # nobody looks for a cache before rank 0 has removed the old one
    if args.dgcnn_rs and args.use_features:
        raise SystemExit('--dgcnn-rs takes no side features (the sort-pool readout has no place for them, '
                         'reference models.py:123-167): drop --use-features')

    def make(dynamic, part, idx, labels, max_num):
        cls = MyDynamicDataset if dynamic else MyDataset
        return cls('data/{}{}/{}/{}'.format(*(data_combo + (part,))), adj_train, idx, labels, args.hop,
                   args.sample_ratio, args.max_nodes_per_hop, u_features, v_features, class_values,
                   max_num=max_num, seed=args.seed)

    train_graphs = make(args.dynamic_train, 'train', (train_u, train_v), train_labels, args.max_train_num)
    test_graphs = make(args.dynamic_test, 'test', (test_u, test_v), test_labels, args.max_test_num)
    if not args.testing:
        test_graphs = make(args.dynamic_val, 'val', (val_u, val_v), val_labels, args.max_val_num)
    if rank == 0:
        print('Used #train graphs: %d, #test graphs: %d' % (len(train_graphs), len(test_graphs)))

    # ---- model (reference Main.py:381-402)
    if args.transfer:
        num_relations, multiply_by = args.num_relations, args.multiply_by
    else:
        num_relations, multiply_by = len(class_values), 1
    if args.dgcnn_rs:
        # the reference keeps this model behind `if False` (Main.py:364-380); --dgcnn-rs is the switch it never had
        from igmc_amd.models import DGCNN_RS
        model = DGCNN_RS(train_graphs, latent_dim=[32, 32, 32, 1], k=args.k, num_relations=len(class_values),
                         num_bases=4, regression=True, adj_dropout=args.adj_dropout,
                         force_undirected=args.force_undirected, seed=args.seed)
        if not args.transfer and rank == 0:     # record the k used in sortpooling (reference Main.py:376-380)
            with open(os.path.join(args.res_dir, 'cmd_input.txt'), 'a') as f:
                f.write(' --k ' + str(model.k) + '\n')
                print('k is saved.')
    else:
        model = IGMC(train_graphs, latent_dim=[32, 32, 32, 32], num_relations=num_relations, num_bases=4,
                     regression=True, adj_dropout=args.adj_dropout, force_undirected=args.force_undirected,
                     side_features=args.use_features, n_side_features=n_features, multiply_by=multiply_by,
                     seed=args.seed)
    if rank == 0:
        print('Total number of parameters is {}'.format(sum(p.numel() for p in model.parameters())))

    rmse = float('nan')
    if not args.no_train:
        rmse = train_multiple_epochs(train_graphs, test_graphs, model, args.epochs, args.batch_size, args.lr,
                                     lr_decay_factor=args.lr_decay_factor,
                                     lr_decay_step_size=args.lr_decay_step_size, weight_decay=0, ARR=args.ARR,
                                     test_freq=args.test_freq, logger=logger, continue_from=args.continue_from,
                                     res_dir=args.res_dir)
    if args.visualize:
        raise NotImplementedError('--visualize (networkx/matplotlib plots) is outside the accelerated path')
    # only rank 0 writes checkpoints (inside `logger`): nobody may look for them before it is done
    parallel.barrier()

    epoch_info = 'epoch {}'.format(args.epochs)
    if args.ensemble:
        if args.data_name == 'ml_1m':
            start_epoch, end_epoch, interval = args.epochs - 15, args.epochs, 5
        else:
            start_epoch, end_epoch, interval = args.epochs - 30, args.epochs, 10
        ckpt_dir = args.transfer if args.transfer else args.res_dir
        checkpoints = [os.path.join(ckpt_dir, 'model_checkpoint%d.pth' % x)
                       for x in range(start_epoch, end_epoch + 1, interval)]
        # rank 0 decides which of the scheduled checkpoints exist and every rank uses ITS list (a rank looking at
        # the file system on its own could ensemble a different set and all-reduce inconsistent squared errors)
        present = parallel.broadcast_object([os.path.exists(c) for c in checkpoints] if rank == 0 else None)
        missing = [c for c, ok in zip(checkpoints, present) if not ok]
        if missing:      # the reference's fixed schedule assumes its default epoch counts (Main.py:437-441)
            if rank == 0:
                print('ensemble: skipping %d missing checkpoint(s): %s' % (len(missing), ', '.join(map(os.path.basename, missing))))
            checkpoints = [c for c, ok in zip(checkpoints, present) if ok]
            if not checkpoints:
                raise FileNotFoundError('--ensemble: none of the scheduled checkpoints exist in %s' % ckpt_dir)
        epoch_info = ('transfer {}, '.format(args.transfer) if args.transfer else '') + \
            'ensemble of range({}, {}, {})'.format(start_epoch, end_epoch, interval)
        rmse = test_once(test_graphs, model, args.batch_size, logger=None, ensemble=True, checkpoints=checkpoints)
        if rank == 0:
            print('Ensemble test rmse is: {:.6f}'.format(rmse))
    elif args.transfer:
        model.load_state_dict(torch.load(args.model_pos, map_location='cpu'))
        rmse = test_once(test_graphs, model, args.batch_size, logger=None)
        epoch_info = 'transfer {}, epoch {}'.format(args.transfer, args.epochs)
    elif args.no_train:
        model.load_state_dict(torch.load(args.model_pos, map_location='cpu'))
        rmse = test_once(test_graphs, model, args.batch_size, logger=None)
    if rank == 0:
        print('Test rmse is: {:.6f}'.format(rmse))
        logger({'epoch': epoch_info, 'train_loss': 0, 'test_rmse': rmse}, None, None)
    return rmse


if __name__ == '__main__':
    main()
