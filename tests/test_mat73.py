"""``igmc_amd/mat73.py`` -- the reader of MATLAB -v7.3 (HDF5) files behind the Monti loaders (reference
``preprocessing.py:32-55`` uses h5py): against a fixture written by the real h5py (tests/golden/make_mat73_fixture.py) and,
where the reference's own files are at hand, against the bundled conversions of them."""
import os

import numpy as np
import pytest

from igmc_amd import mat73, preprocessing

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/raw_data'


def test_reader_against_the_h5py_written_fixture():
    f = mat73.File(os.path.join(HERE, 'golden', 'mat73_fixture.mat'))
    z = np.load(os.path.join(HERE, 'golden', 'mat73_fixture.npz'))
    assert len(list(f.keys())) == 45 and 'M' in f and 'nope' not in f
    for k in ('M', 'S', 'I', 'U'):          # chunked + deflate (ragged edge chunks), + shuffle, contiguous, chunked unfiltered
        a = f[k]
        assert a.dtype == z[k].dtype and np.array_equal(a, z[k]), k
    W = f['W']                               # MATLAB sparse matrix: a group with data / ir / jc
    assert isinstance(W, mat73.Group) and 'ir' in W.keys()
    for k in ('data', 'ir', 'jc'):
        assert np.array_equal(W[k], z['W_' + k])
    for k in range(40):                      # (the root symbol table spans several B-tree leaves)
        assert np.array_equal(f['x%02d' % k], np.arange(k + 1.0))
    # the reference's load_matlab_file convention: sparse -> csc float32, dense -> float32 transposed
    w = mat73.load_matlab_field(os.path.join(HERE, 'golden', 'mat73_fixture.mat'), 'W')
    assert w.shape == (30, 40) and w.dtype == np.float32 and w.nnz == len(z['W_data'])
    m = mat73.load_matlab_field(os.path.join(HERE, 'golden', 'mat73_fixture.mat'), 'M')
    assert m.dtype == np.float32 and np.array_equal(m, z['M'].T.astype(np.float32))


def test_not_an_hdf5_file(tmp_path):
    p = tmp_path / 'old.mat'
    p.write_bytes(b'MATLAB 5.0 MAT-file' + b'\0' * 600)
    with pytest.raises(ValueError):
        mat73.File(str(p))


@pytest.mark.parametrize('name', ['douban', 'flixster', 'yahoo_music'])
def test_reference_mat_files_read_like_their_bundled_conversions(name, tmp_path, monkeypatch):
    """The loader fed with the reference's ORIGINAL ``training_test_dataset.mat`` (through mat73) gives the arrays of the
    bundled ``.npz`` (converted once with h5py, tests/golden/convert_mat.py), bit for bit."""
    src = os.path.join(REF, name, 'training_test_dataset.mat')
    if not os.path.exists(src):
        pytest.skip('the reference checkout is not on this machine')
    want = preprocessing._load_monti_arrays(name)                    # bundled .npz
    os.makedirs(tmp_path / 'raw_data' / name)
    os.symlink(src, tmp_path / 'raw_data' / name / 'training_test_dataset.mat')
    monkeypatch.setattr(preprocessing, '_find_raw',
                        lambda ds, fn: str(tmp_path / 'raw_data' / ds / fn) if os.path.exists(tmp_path / 'raw_data' / ds / fn) else None)
    got = preprocessing._load_monti_arrays(name)                     # original .mat
    assert sorted(got) == sorted(want)
    for k in want:
        assert np.array_equal(got[k], want[k]), k
