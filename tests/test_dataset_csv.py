"""The trainer's real-data route (train/train.py:124-128 `img_path`, :423-430 `load_images`,
util/io.py:46-83 `load_csv`): per-epoch CSV lists + PNG frames in the RobotCar directory layout."""
import os

import numpy as np
import pytest

from soft_contrastive_learning_amd.train import dataset
from soft_contrastive_learning_amd.util import cv, io


def write_set(root, csv_file, n, size=(96, 128), date='2014-12-02-15-30-08', folder=1, seed=0, t0=1418000000000000):
    """n frames <root>/<date>_stereo_centre_<folder:02d>/<t>.png + the list; returns the frames."""
    rng = np.random.RandomState(seed)
    d = os.path.join(root, '{}_stereo_centre_{:02d}'.format(date, folder))
    os.makedirs(d, exist_ok=True)
    frames, cols = [], {k: [] for k in ('date', 'folder', 't', 'easting', 'northing', 'yaw')}
    for i in range(n):
        img = rng.randint(0, 256, size + (3,)).astype(np.uint8)
        t = t0 + 62500 * i
        io.save_img(img, os.path.join(d, '%d.png' % t))
        frames.append(img)
        for k, v in zip(cols, (date, folder, t, 620000.0 + 2.0 * i, 5735000.0 + 0.5 * i, 0.01 * i)):
            cols[k].append(v)
    io.save_csv(cols, csv_file)
    return frames


def test_csv_set_paths_and_geometry(tmp_path):
    root, lists = str(tmp_path / 'img'), tmp_path / 'lists'
    lists.mkdir()
    frames = write_set(root, str(lists / 'train_ref_000.csv'), 5, size=(96, 128))
    s = dataset.CsvImageSet(str(lists / 'train_ref_000.csv'), root)
    assert len(s) == 5 and s.xy.shape == (5, 2) and s.yaw.shape == (5,)
    assert s.path(2).endswith(os.path.join('2014-12-02-15-30-08_stereo_centre_01', '1418000000125000.png'))
    got = s.load_images([3, 0])
    assert got.dtype == np.float32 and got.shape == (2, 180, 240, 3)       # longer side -> 240
    assert np.array_equal(got[0], cv.resize_img(frames[3], 240).astype(np.float32))
    flat = dataset.CsvImageSet(str(lists / 'train_ref_000.csv'), root, vlad_cores=0)
    got = flat.load_images([1])
    assert got.shape == (1, 180, 240, 3)
    assert np.array_equal(got[0], cv.standard_size(frames[1], 180, 240).astype(np.float32))
    serial = dataset.CsvImageSet(str(lists / 'train_ref_000.csv'), root, loader_threads=1)
    pooled = dataset.CsvImageSet(str(lists / 'train_ref_000.csv'), root, loader_threads=4)
    assert np.array_equal(serial.load_images([4, 1, 1, 0, 2]), pooled.load_images([4, 1, 1, 0, 2]))
    procs = dataset.make_loader_pool(2)                     # (no device in this process: a real pool)
    if procs is not None:
        try:
            forked = dataset.CsvImageSet(str(lists / 'train_ref_000.csv'), root, pool=procs)
            assert np.array_equal(serial.load_images([4, 1, 1, 0, 2]), forked.load_images([4, 1, 1, 0, 2]))
        finally:
            procs.shutdown()
    assert dataset.make_loader_pool(0) is None
    (lists / 'bad.csv').write_text('date,folder\n1,2\n')
    with pytest.raises(ValueError):
        dataset.CsvImageSet(str(lists / 'bad.csv'), root)


def test_loader_pool_workers_run_before_the_first_load_and_are_never_replaced(tmp_path):
    """The workers must exist when make_loader_pool returns (the trainer initialises the GPU right
    after it: a process started later would be an exec from a process that holds the device), the
    first load_images() must start none, and a pool that lost a worker falls back to the threads
    instead of starting a replacement."""
    import signal
    root, lists = str(tmp_path / 'img'), tmp_path / 'lists'
    lists.mkdir()
    write_set(root, str(lists / 'train_ref_000.csv'), 4, size=(96, 128))
    procs = dataset.make_loader_pool(3)
    try:
        assert len(procs.pids) == 3
        for pid in procs.pids:
            os.kill(pid, 0)                                  # alive now, before any load_images()
        s = dataset.CsvImageSet(str(lists / 'train_ref_000.csv'), root, pool=procs)
        serial = dataset.CsvImageSet(str(lists / 'train_ref_000.csv'), root, loader_threads=1)
        want = serial.load_images([0, 1, 2, 3])
        assert np.array_equal(s.load_images([0, 1, 2, 3]), want)
        assert sorted(p.pid for p in procs._ex._processes.values()) == procs.pids   # the same three
        os.kill(procs.pids[0], signal.SIGKILL)               # a worker dies
        assert np.array_equal(s.load_images([0, 1, 2, 3]), want)                    # threads took over
        assert procs.broken
        assert np.array_equal(s.load_images([3, 2]), want[[3, 2]])
    finally:
        procs.shutdown()


def test_example_pictures_of_the_localisation_check(tmp_path):
    """train/train.py:400-420: query | retrieved | optimal, captioned, one file per chosen query."""
    from soft_contrastive_learning_amd.train import evaluate
    root, lists = str(tmp_path / 'img'), tmp_path / 'lists'
    lists.mkdir()
    write_set(root, str(lists / 'ref_000.csv'), 8, size=(48, 64), folder=1, seed=1)
    write_set(root, str(lists / 'qry_000.csv'), 6, size=(48, 64), folder=2, seed=2, t0=1500000000000000)
    refs = dataset.CsvImageSet(str(lists / 'ref_000.csv'), root)
    qrys = dataset.CsvImageSet(str(lists / 'qry_000.csv'), root)
    ref_idx, q_idx = np.arange(0, 8, 2), np.array([1, 3, 4])
    nearest = np.array([[0, 1], [3, 2], [1, 0]])
    folder = evaluate.save_example_pictures(str(tmp_path / 'out'), 'other', '00_checkpoint-4', qrys, q_idx,
                                            refs, ref_idx, nearest, rng=np.random.RandomState(0), count=2)
    files = sorted(os.listdir(folder))
    assert folder.endswith(os.path.join('out', 'other_00_checkpoint-4')) and len(files) == 2
    assert all(f in {os.path.basename(qrys.path(i)) for i in q_idx} for f in files)
    pic = io.load_img(os.path.join(folder, files[0]))
    assert pic.shape == (48, 3 * 64, 3)
    syn = dataset.SyntheticImageSet(12, 32, 40, seed=3)
    folder = evaluate.save_example_pictures(str(tmp_path / 'out'), 'local', 's', syn, [0, 5], syn, [1, 2, 3],
                                            np.array([[2], [0]]), rng=np.random.RandomState(1))
    assert sorted(os.listdir(folder)) == ['0.png', '5.png']


def test_localisation_plots(tmp_path):
    """train/train.py:368-396: <mode>_<out_name>_<rad>.pdf for rad in 50, 25, 10."""
    from soft_contrastive_learning_amd.train import evaluate
    rng = np.random.RandomState(4)
    g = rng.uniform(0, 60, size=(20, 5))
    paths = evaluate.save_localization_plots(str(tmp_path / 'run'), 'local', '00_checkpoint-100', g,
                                             rng.uniform(0, 5, size=(20, 1)))
    assert [os.path.basename(p) for p in paths] == ['local_00_checkpoint-100_%d.pdf' % r for r in (50, 25, 10)]
    assert all(open(p, 'rb').read(5) == b'%PDF-' for p in paths)
