"""Golden outputs of the REFERENCE loader ``preprocessing.load_data_monti`` (reference
``preprocessing.py:203-333``) on the bundled datasets.

Run in the build container with the interpreter that has h5py (the main one has not):

    cd /root/reference && /opt/conda/bin/python3.9 /root/repo/tests/golden/make_golden_loader.py

Writes ``tests/golden/loader_<dataset>.npz``.  ``tests/test_oracle_golden.py`` pins
``igmc_amd.preprocessing.load_data_monti`` against them (the GPU box has no /root/reference).
"""
import os
import sys
import warnings
import numpy as np

warnings.simplefilter('ignore')
sys.path.insert(0, '/root/reference')
os.chdir('/root/reference')
import preprocessing as P  # noqa: E402  (the unmodified reference module)

OUT = '/root/repo/tests/golden'
for name in ['flixster', 'douban', 'yahoo_music']:
    rec = {}
    for testing in (True, False):
        o = P.load_data_monti(name, testing=testing)
        A = o[2].tocoo()
        order = np.lexsort((A.col, A.row))
        tag = 'T' if testing else 'F'
        rec['adj_row_' + tag] = A.row[order].astype(np.int32)
        rec['adj_col_' + tag] = A.col[order].astype(np.int32)
        rec['adj_val_' + tag] = A.data[order].astype(np.uint8)
        for k, i in (('train_labels', 3), ('train_u', 4), ('train_v', 5), ('val_labels', 6), ('val_u', 7),
                     ('val_v', 8), ('test_labels', 9), ('test_u', 10), ('test_v', 11)):
            rec[k + '_' + tag] = np.asarray(o[i]).astype(np.int32)
        rec['class_values'] = np.asarray(o[12], dtype=np.float64)
        rec['u_feat_shape'] = np.array(o[0].shape)
        rec['v_feat_shape'] = np.array(o[1].shape)
        rec['u_feat_sum'] = np.array([o[0].sum()])
        rec['v_feat_sum'] = np.array([o[1].sum()])
    if name == 'flixster':   # transfer setting of Main.py:153-177 (rating map onto 5 levels)
        rm = {x: int(np.ceil(x)) for x in np.arange(0.5, 5.01, 0.5).tolist()}
        o = P.load_data_monti(name, testing=True, rating_map=rm)
        A = o[2].tocoo()
        order = np.lexsort((A.col, A.row))
        rec['adj_val_map'] = A.data[order].astype(np.uint8)
        rec['class_values_map'] = np.asarray(o[12], dtype=np.float64)
        rec['test_labels_map'] = np.asarray(o[9]).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, 'loader_%s.npz' % name), **rec)
    print(name, {k: v.shape for k, v in rec.items() if k.endswith('_T')})
