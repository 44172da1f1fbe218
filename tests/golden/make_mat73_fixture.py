"""A small MATLAB-v7.3-shaped HDF5 file for ``igmc_amd/mat73.py`` (the reader of the Monti ``.mat`` files), written with the
real h5py so that the reader is checked against the library the reference uses (``preprocessing.py:32-55``):

    /opt/conda/bin/python3.9 tests/golden/make_mat73_fixture.py

* 512-byte user block (MATLAB's header), default (earliest) file-format version = superblock 0, old-style groups;
* ``M``: chunked + deflate float64 (what MATLAB writes for the rating matrices), ragged edge chunks;
* ``S``: chunked + shuffle + deflate float32;  ``I``: contiguous int32;  ``U``: uint8 chunked without filters;
* ``W``: a MATLAB sparse matrix = group with ``data`` / ``ir`` / ``jc``;
* 40 small datasets more, so that the root group's symbol table spills over several B-tree leaves.
The expected arrays go to ``mat73_fixture.npz`` beside it."""
import os
import numpy as np
import h5py
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(11)
M = np.where(rng.random((67, 45)) < 0.2, rng.integers(1, 6, (67, 45)), 0).astype(np.float64)
S = rng.standard_normal((33, 20)).astype(np.float32)
I = rng.integers(-1000, 1000, (9, 14)).astype(np.int32)
U = rng.integers(0, 255, (25, 10)).astype(np.uint8)
Wd = sp.random(30, 40, density=0.1, random_state=3, format='csc', dtype=np.float64)
path = os.path.join(HERE, 'mat73_fixture.mat')
with h5py.File(path, 'w', userblock_size=512) as f:
    f.create_dataset('M', data=M, chunks=(16, 16), compression='gzip', compression_opts=3)
    f.create_dataset('S', data=S, chunks=(8, 20), compression='gzip', shuffle=True)
    f.create_dataset('I', data=I)
    f.create_dataset('U', data=U, chunks=(10, 4))
    g = f.create_group('W')
    g.create_dataset('data', data=Wd.data)
    g.create_dataset('ir', data=Wd.indices.astype(np.uint64))
    g.create_dataset('jc', data=Wd.indptr.astype(np.uint64))
    for k in range(40):
        f.create_dataset('x%02d' % k, data=np.arange(k + 1, dtype=np.float64))
with open(path, 'r+b') as f:
    f.write(b'MATLAB 7.3 MAT-file, fixture of tests/golden/make_mat73_fixture.py'.ljust(512, b' '))
np.savez_compressed(os.path.join(HERE, 'mat73_fixture.npz'), M=M, S=S, I=I, U=U, W_data=Wd.data, W_ir=Wd.indices.astype(np.uint64),
                    W_jc=Wd.indptr.astype(np.uint64))
print(path, os.path.getsize(path))
