"""Image geometry and file formats of the callers (util/cv.py:7-20, util/io.py of the reference) as
restated without OpenCV in soft_contrastive_learning_amd/util/.  OpenCV is not installed here: the
resampler is pinned by cases derivable by hand and against an independent float64 bilinear sampler
on the same coordinate rule."""
import numpy as np
import pytest

from soft_contrastive_learning_amd.util import cv, io


def _bilinear_f64(img, fx, fy):
    """Bilinear sampling at ((d + 0.5) / f - 0.5), edges clamped, float64, rounded half up."""
    sh, sw = img.shape[:2]
    dw, dh = int(np.rint(sw * fx)), int(np.rint(sh * fy))
    x = np.clip((np.arange(dw) + 0.5) / fx - 0.5, 0, sw - 1)
    y = np.clip((np.arange(dh) + 0.5) / fy - 0.5, 0, sh - 1)
    x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
    x1, y1 = np.minimum(x0 + 1, sw - 1), np.minimum(y0 + 1, sh - 1)
    ax, ay = (x - x0)[None, :, None], (y - y0)[:, None, None]
    s = img.astype(np.float64)
    top = s[y0][:, x0] * (1 - ax) + s[y0][:, x1] * ax
    bot = s[y1][:, x0] * (1 - ax) + s[y1][:, x1] * ax
    return top * (1 - ay) + bot * ay


def test_identity_and_halving_by_hand():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (12, 16, 3)).astype(np.uint8)
    assert np.array_equal(cv.resize_linear(img, 1.0, 1.0), img)
    # factor 1/2: the sample point of destination d is 2 d + 0.5 — the mean of a 2 x 2 block,
    # (a + b + c + d + 2) >> 2 in the fixed-point arithmetic
    half = cv.resize_linear(img, 0.5, 0.5)
    blocks = img.astype(np.int32).reshape(6, 2, 8, 2, 3).sum(axis=(1, 3))
    assert half.shape == (6, 8, 3)
    assert np.array_equal(half, ((blocks + 2) >> 2).astype(np.uint8))
    # factor 2: destination 0 lies left of the first centre (clamped), 1 and 2 are 1/4 : 3/4 mixes
    row = np.array([[0, 100, 200]], dtype=np.uint8)
    up = cv.resize_linear(row, 2.0, 1.0)
    assert up.tolist() == [[0, 25, 75, 125, 175, 200]]
    # a constant image stays constant at any factor, 255 does not overflow
    assert (cv.resize_linear(np.full((40, 30, 3), 255, np.uint8), 0.37, 0.37) == 255).all()


@pytest.mark.parametrize("shape,f", [((960, 1280, 3), 240 / 1280.0), ((97, 131, 3), 0.61), ((50, 40), 1.7),
                                     ((480, 640, 3), 240 / 640.0)])
def test_against_a_float64_sampler(shape, f):
    rng = np.random.RandomState(sum(shape))
    img = rng.randint(0, 256, shape).astype(np.uint8)
    got = cv.resize_linear(img, f, f)
    ref = _bilinear_f64(img if img.ndim == 3 else img[:, :, None], f, f)
    ref = ref if img.ndim == 3 else ref[:, :, 0]
    assert got.shape == ref.shape
    # 11-bit weights and the truncating shifts: within one grey level of the exact value
    assert np.abs(got.astype(np.float64) - ref).max() <= 1.0
    assert np.abs(got.astype(np.float64) - ref).mean() < 0.3


def test_no_low_pass_filter():
    """The reference's frames come down 5.33 x with plain bilinear sampling: a one-pixel checkerboard
    does NOT average to grey (what an antialiased resize, e.g. PIL's, returns)."""
    yy, xx = np.mgrid[0:960, 0:1280]
    board = (((yy + xx) & 1) * 255).astype(np.uint8)[:, :, None].repeat(3, axis=2)
    small = cv.resize_img(board, 240)
    assert small.shape == (180, 240, 3)
    from PIL import Image
    pil = np.asarray(Image.fromarray(board).resize((240, 180), Image.BILINEAR))
    assert small.std() > 30 and pil.std() < 3             # aliased pattern against PIL's flat grey


def test_loader_geometry():
    rng = np.random.RandomState(2)
    frame = rng.randint(0, 256, (960, 1280, 3)).astype(np.uint8)           # a RobotCar centre frame
    assert cv.resize_img(frame, 240).shape == (180, 240, 3)                # train/train.py:427
    assert cv.resize_img(frame.transpose(1, 0, 2).copy(), 240).shape == (240, 180, 3)
    wide = rng.randint(0, 256, (768, 1024, 3)).astype(np.uint8)
    assert cv.standard_size(wide, 180, 240).shape == (180, 240, 3)         # train/train.py:429
    tall = rng.randint(0, 256, (1000, 600, 3)).astype(np.uint8)
    out = cv.standard_size(tall, 180, 240)                                  # covers, then crops rows
    assert out.shape == (180, 240, 3)
    full = cv.resize_linear(tall, 240 / 600.0, 240 / 600.0)
    top = (full.shape[0] - 180) // 2
    assert np.array_equal(out, full[top:top + 180])
    with pytest.raises(ValueError):
        cv.resize_linear(frame.astype(np.float32), 0.5, 0.5)


def test_image_and_list_files(tmp_path):
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, (20, 30, 3)).astype(np.uint8)
    io.save_img(img, tmp_path / 'a.png')
    assert np.array_equal(io.load_img(tmp_path / 'a.png'), img)            # RGB in, RGB out
    grey = rng.randint(0, 256, (8, 9)).astype(np.uint8)
    from PIL import Image
    Image.fromarray(grey).save(tmp_path / 'g.png')
    got = io.load_img(tmp_path / 'g.png')                                   # imread: three channels
    assert got.shape == (8, 9, 3) and np.array_equal(got[:, :, 1], grey)
    cols = {'path': ['x/1.png', 'x/2.png'], 'easting': [1.5, 2.5]}
    io.save_csv(cols, tmp_path / 'l.csv')
    back = io.load_csv(tmp_path / 'l.csv')
    assert back == {'path': ['x/1.png', 'x/2.png'], 'easting': ['1.5', '2.5']}      # strings, like csv.reader
    (tmp_path / 'n.csv').write_text('a,b\n1,2\n')
    assert io.load_csv(tmp_path / 'n.csv', has_header=False, keys=['p', 'q']) == {'p': ['a', '1'], 'q': ['b', '2']}
    assert io.load_csv(tmp_path / 'n.csv', has_header=False) == {0: ['a', '1'], 1: ['b', '2']}
    (tmp_path / 'one.csv').write_text('a,b\n')
    assert io.load_csv(tmp_path / 'one.csv') == ['a', 'b']                  # util/io.py:80-83
    io.save_csv({'k': 3, 'm': 'z'}, tmp_path / 's.csv')
    assert io.load_csv(tmp_path / 's.csv') == {'k': ['3'], 'm': ['z']}
    io.save_pickle([np.arange(3)], tmp_path / 'p.pickle')
    assert np.array_equal(io.load_pickle(tmp_path / 'p.pickle')[0], np.arange(3))


def test_example_picture_helpers():
    """util/cv.py:23-34 (the localisation check's example pictures, train/train.py:400-420)."""
    rng = np.random.RandomState(6)
    left = rng.randint(0, 256, (180, 240, 3)).astype(np.uint8)
    right = rng.randint(0, 256, (90, 160, 3)).astype(np.uint8)
    both = cv.merge_images(left, right)
    assert both.shape == (180, 240 + 320, 3) and np.array_equal(both[:, :240], left)
    assert np.array_equal(both[:, 240:], cv.resize_linear(right, dsize=(320, 180)))
    assert np.array_equal(cv.resize_linear(right, dsize=(320, 180)), cv.resize_linear(right, 2.0, 2.0))
    noted = cv.put_text('Top 1: 3.2', left.copy())
    assert noted.shape == left.shape and (noted != left).any()
    changed = np.argwhere((noted != left).any(axis=2))
    assert changed[:, 0].max() <= 45 and changed[:, 1].min() >= 5       # around the anchor (10, 35)


def test_resampler_properties():
    """Size-independent properties of the restated resampler: outputs stay inside the range of the
    inputs they mix; mirroring commutes with resizing whenever the destination size is exact
    (W fx integral: the sample points are then symmetric about the image centre and the two
    fixed-point weights swap roles)."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None)
    @given(st.integers(1, 6), st.integers(1, 6), st.integers(1, 5), st.integers(1, 5), st.integers(0, 2 ** 31 - 1))
    def check(ph, pw, qh, qw, seed):
        rng = np.random.RandomState(seed)
        sh, sw = 4 * qh * rng.randint(1, 4), 4 * qw * rng.randint(1, 4)
        img = rng.randint(0, 256, (sh, sw, 3)).astype(np.uint8)
        fy, fx = ph / float(qh), pw / float(qw)                  # sh fy and sw fx are integers
        out = cv.resize_linear(img, fx, fy)
        assert out.shape == (sh * ph // qh, sw * pw // qw, 3)
        assert out.min() >= img.min() and out.max() <= img.max()
        assert np.array_equal(cv.resize_linear(img[:, ::-1].copy(), fx, fy), out[:, ::-1])
        assert np.array_equal(cv.resize_linear(img[::-1].copy(), fx, fy), out[::-1])
    check()


def test_sixteen_bit_png_loads_as_its_high_byte(tmp_path):
    """the 16-bit branch of load_img was untested (and Image.point with a function is
    not defined for 'I;16' in every Pillow version): a 16-bit grey PNG comes back as three equal
    8-bit channels holding the high byte."""
    from PIL import Image
    from soft_contrastive_learning_amd.util import io
    arr = (np.arange(6 * 8, dtype=np.uint32).reshape(6, 8) * 1367 % 65536).astype(np.uint16)
    path = tmp_path / 'deep.png'
    Image.fromarray(arr).save(str(path))
    with Image.open(str(path)) as im:
        assert im.mode.startswith('I')
    got = io.load_img(path)
    assert got.shape == (6, 8, 3) and got.dtype == np.uint8
    for c in range(3):
        assert np.array_equal(got[:, :, c], (arr >> 8).astype(np.uint8))


def test_resampler_against_opencv_when_it_is_there():
    """Optional (skipped here: no cv2 in the image): util.cv.resize_img / standard_size restate
    OpenCV's 8-bit INTER_LINEAR from memory; where cv2 exists the two must agree bit for bit on a
    RobotCar-sized frame — the maximum deviation is what a first cv2 user should record."""
    cv2 = pytest.importorskip('cv2')
    from soft_contrastive_learning_amd.util import cv
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(960, 1280, 3), dtype=np.uint8)
    for max_side in (240, 640):
        want = cv2.resize(img, (max_side, max_side * 960 // 1280), interpolation=cv2.INTER_LINEAR)
        got = cv.resize_img(img, max_side)
        assert got.shape == want.shape
        assert int(np.abs(got.astype(int) - want.astype(int)).max()) == 0
