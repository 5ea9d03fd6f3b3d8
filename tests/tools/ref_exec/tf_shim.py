"""A NumPy stand-in for the ~35 TensorFlow 1.x ops that ``/root/reference/model/losses.py`` calls
(and, further down, the eight graph-building calls of ``model/nets.py:7-131``) — losses.py
at lines 5-135 (wms / evil_triplet / ms / logratio), 139-185 (ms_det), 197-250 (evil_quadruplet,
worst_pos_distance, distance / huber_distance losses), 627-646 (pairwise_distance_loss) and
656-700 (helpers) — so that the reference's OWN source text can be executed in the build
container, where TensorFlow 1.10 does not exist, and its outputs frozen as fixtures
(tests/tools/ref_exec/make_golden_ref.py -> tests/golden/golden_ref_v1.json).

TEST TOOLING, BUILD CONTAINER ONLY.  Neither this file's consumer (the reference) nor anything it
produces at run time travels to the GPU box; only the JSON does.

What this does and does not buy.  By the task's rules a stand-in library pins nothing formally:
parity stays "unpinned".  What changes is the trust surface.  Before: "a 300-line hand
transcription of losses.py is faithful".  Now: "each op below means what TensorFlow 1.10 means
by it" — every rank, axis, broadcast, transpose and tile of the reference's text is executed as
written, by the reference's own statements, not re-read by the builder.  Eager evaluation in
float32, one NumPy call per TF op, TF's static-shape API (``get_shape()``) on every tensor, and
TF's STRICTER shape rules enforced where NumPy would silently broadcast (``tf.where`` operands of
one shape, ``tf.matmul`` without batch broadcasting, ``tf.tile`` multiples of the tensor's rank).

Known places where float32 results can differ from TensorFlow's in the last bits (none is a
semantic difference): ``matmul`` / ``reduce_*`` accumulation order (Eigen vs NumPy pairwise),
``exp`` / ``log`` (Eigen's vectorised polynomials vs NumPy's), ``rsqrt`` inside l2_normalize, and
``tanh`` — TF 1.10 evaluates float32 tanh with Eigen's rational approximation, which this shim can
imitate (``TANH = 'eigen'``, the polynomial recalled in oracle/losses_np.py) or replace by
``np.tanh`` (``TANH = 'numpy'``); the fixture generator records both.
"""
import builtins
import contextlib
import types

import numpy as np

F32 = np.float32
float32 = np.float32
float64 = np.float64
int32 = np.int32
int64 = np.int64
bool_ = np.bool_

TANH = 'eigen'            # 'eigen' | 'numpy' (see the module docstring)
CALLS = {}                # op name -> number of calls (the generator prints the ops a case used)


class Dimension(int):
    """tf.Dimension: usable as an int (the reference writes ``int(num_neg)``)."""
    @property
    def value(self):
        return int(self)


class TensorShape(tuple):
    def as_list(self):
        return [int(d) for d in self]

    def __getitem__(self, i):
        got = tuple.__getitem__(self, i)
        return TensorShape(got) if isinstance(i, builtins.slice) else Dimension(got)

    @property
    def ndims(self):
        return len(self)


class Tensor(np.ndarray):
    """An eager value with the static-shape API of tf.Tensor."""

    def get_shape(self):
        return TensorShape(Dimension(d) for d in self.shape)

    def __array_finalize__(self, obj):
        pass


def _t(x, dtype=None):
    """convert_to_tensor: Python floats become float32 (TF's default float), ints int32."""
    if isinstance(x, Tensor) and (dtype is None or x.dtype == dtype):
        return x
    a = np.asarray(x)
    if dtype is None:
        if a.dtype == np.float64 and not isinstance(x, np.ndarray):
            dtype = F32
        elif a.dtype == np.int64 and not isinstance(x, np.ndarray):
            dtype = np.int32
        else:
            dtype = a.dtype
    return np.asarray(a, dtype=dtype, order='C').view(Tensor)


def _count(name):
    CALLS[name] = CALLS.get(name, 0) + 1


def _same_float(a, b, op):
    """TF refuses mixed dtypes in a binary op; Python scalars take the tensor's dtype."""
    ta = isinstance(a, np.ndarray)
    tb = isinstance(b, np.ndarray)
    if ta and tb:
        if a.dtype != b.dtype:
            raise TypeError('%s: dtypes %s and %s do not match (TensorFlow does not promote)'
                            % (op, a.dtype, b.dtype))
        return _t(a), _t(b)
    if ta:
        return _t(a), _t(np.asarray(b, dtype=a.dtype))
    if tb:
        return _t(np.asarray(a, dtype=b.dtype)), _t(b)
    return _t(a), _t(b)


def _binary(name, fn):
    def op(x, y, name_=None):
        _count(name)
        a, b = _same_float(x, y, name)
        with np.errstate(over='ignore', divide='ignore', invalid='ignore'):
            return _t(fn(a.view(np.ndarray), b.view(np.ndarray)))
    op.__name__ = name
    return op


add = _binary('add', np.add)
subtract = _binary('subtract', np.subtract)
multiply = _binary('multiply', np.multiply)
divide = _binary('divide', np.divide)
div = _binary('div', np.divide)                    # float operands: true division
maximum = _binary('maximum', np.maximum)
minimum = _binary('minimum', np.minimum)
equal = _binary('equal', np.equal)
squared_difference = _binary('squared_difference', lambda a, b: (a - b) * (a - b))


def _unary(name, fn):
    def op(x, name_=None):
        _count(name)
        a = _t(x)
        with np.errstate(over='ignore', divide='ignore', invalid='ignore'):
            return _t(fn(a.view(np.ndarray)).astype(a.dtype, copy=False))
    op.__name__ = name
    return op


exp = _unary('exp', np.exp)
log = _unary('log', np.log)
logical_not = _unary('logical_not', np.logical_not)
zeros_like = _unary('zeros_like', np.zeros_like)
ones_like = _unary('ones_like', np.ones_like)


def _eigen_fast_tanh_f32(x):
    """Eigen 3.3.90 generic_fast_tanh_float (RECALLED — the same restatement as
    oracle/losses_np.py:eigen_fast_tanh_f32): clamp to +-9, 13th / 6th degree rational."""
    x = np.maximum(F32(-9.0), np.minimum(F32(9.0), x.astype(F32)))
    a = [F32(v) for v in (4.89352455891786e-03, 6.37261928875436e-04, 1.48572235717979e-05,
                          5.12229709037114e-08, -8.60467152213735e-11, 2.00018790482477e-13,
                          -2.76076847742355e-16)]
    b = [F32(v) for v in (4.89352518554385e-03, 2.26843463243900e-03, 1.18534705686654e-04,
                          1.19825839466702e-06)]
    x2 = x * x
    p = x2 * a[6] + a[5]
    for c in (a[4], a[3], a[2], a[1], a[0]):
        p = x2 * p + c
    p = x * p
    q = x2 * b[3] + b[2]
    q = x2 * q + b[1]
    q = x2 * q + b[0]
    return (p / q).astype(F32)


def tanh(x, name=None):
    _count('tanh')
    a = _t(x)
    if TANH == 'eigen' and a.dtype == F32:
        return _t(_eigen_fast_tanh_f32(a.view(np.ndarray)))
    return _t(np.tanh(a.view(np.ndarray)).astype(a.dtype))


def cast(x, dtype, name=None):
    _count('cast')
    return _t(np.asarray(x).astype(dtype))


def constant(value, dtype=None, shape=None, name=None):
    _count('constant')
    a = _t(value, dtype)
    return a if shape is None else _t(np.broadcast_to(a, shape).copy())


def eye(num_rows, num_columns=None, dtype=F32, name=None):
    _count('eye')
    return _t(np.eye(int(num_rows), None if num_columns is None else int(num_columns), dtype=dtype))


def zeros(shape, dtype=F32, name=None):
    _count('zeros')
    return _t(np.zeros([int(d) for d in shape], dtype=dtype))


def ones(shape, dtype=F32, name=None):
    _count('ones')
    return _t(np.ones([int(d) for d in shape], dtype=dtype))


def fill(dims, value, name=None):
    _count('fill')
    v = _t(value)
    return _t(np.full([int(d) for d in dims], v, dtype=v.dtype))


def where(condition, x=None, y=None, name=None):
    """TF 1.10 Select: x and y of ONE shape; condition of that shape (or a vector over dim 0)."""
    _count('where')
    if x is None or y is None:
        raise NotImplementedError('tf.where(condition) alone is not used by the reference')
    c, a, b = np.asarray(condition), _t(x), _t(y)
    if c.dtype != np.bool_:
        raise TypeError('where: condition must be bool')
    if a.shape != b.shape or a.dtype != b.dtype:
        raise ValueError('where: x %s %s and y %s %s must match' % (a.shape, a.dtype, b.shape, b.dtype))
    if c.shape != a.shape and not (c.ndim == 1 and c.shape[0] == a.shape[0]):
        raise ValueError('where: condition %s does not match x %s (Select does not broadcast)'
                         % (c.shape, a.shape))
    if c.shape != a.shape:
        c = c.reshape((-1,) + (1,) * (a.ndim - 1))
    return _t(np.where(c, a.view(np.ndarray), b.view(np.ndarray)))


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    """No batch broadcasting in TF 1.10: equal ranks >= 2 and equal leading dimensions."""
    _count('matmul')
    a, b = _same_float(a, b, 'matmul')
    if a.ndim < 2 or a.ndim != b.ndim or a.shape[:-2] != b.shape[:-2]:
        raise ValueError('matmul: shapes %s and %s' % (a.shape, b.shape))
    x = np.swapaxes(a.view(np.ndarray), -1, -2) if transpose_a else a.view(np.ndarray)
    y = np.swapaxes(b.view(np.ndarray), -1, -2) if transpose_b else b.view(np.ndarray)
    return _t(np.matmul(x, y))


def einsum(equation, *inputs):
    _count('einsum')
    return _t(np.einsum(equation, *[_t(i).view(np.ndarray) for i in inputs]))


def transpose(a, perm=None, name=None):
    """perm None = reverse ALL dimensions (the logratio quirk)."""
    _count('transpose')
    return _t(np.transpose(_t(a).view(np.ndarray), perm))


def tile(input, multiples, name=None):       # noqa: A002 (TensorFlow's argument name)
    _count('tile')
    a = _t(input)
    m = [int(v) for v in multiples]
    if len(m) != a.ndim:
        raise ValueError('tile: %d multiples for a rank-%d tensor' % (len(m), a.ndim))
    return _t(np.tile(a.view(np.ndarray), m))


def reshape(tensor, shape, name=None):
    _count('reshape')
    return _t(np.reshape(_t(tensor).view(np.ndarray), [int(d) for d in shape]))


def concat(values, axis, name=None):
    _count('concat')
    return _t(np.concatenate([_t(v).view(np.ndarray) for v in values], axis=int(axis)))


def _reduce(name, fn):
    def op(input_tensor, axis=None, keepdims=False, name_=None, keep_dims=None):
        _count(name)
        a = _t(input_tensor).view(np.ndarray)
        kd = builtins.bool(keepdims if keep_dims is None else keep_dims)
        if name in ('reduce_sum', 'reduce_mean'):
            return _t(np.asarray(fn(a, axis=axis, keepdims=kd, dtype=a.dtype)))
        return _t(np.asarray(fn(a, axis=axis, keepdims=kd)))
    op.__name__ = name
    return op


reduce_sum = _reduce('reduce_sum', np.sum)
reduce_mean = _reduce('reduce_mean', np.mean)
reduce_max = _reduce('reduce_max', np.max)
reduce_min = _reduce('reduce_min', np.min)


@contextlib.contextmanager
def name_scope(name, default_name=None, values=None):
    yield name


def _l2_normalize(x, axis=None, epsilon=1e-12, name=None, dim=None):
    """x * rsqrt(max(sum(x^2, axis), epsilon))   (python/ops/nn_impl.py)."""
    _count('nn.l2_normalize')
    a = _t(x).view(np.ndarray)
    ax = axis if axis is not None else dim
    sq = np.sum(a * a, axis=ax, keepdims=True, dtype=a.dtype)
    inv = (a.dtype.type(1.0) / np.sqrt(np.maximum(sq, a.dtype.type(epsilon)))).astype(a.dtype)
    return _t(a * inv)


# ---------------------------------------------------------------------------------------------
# The ops model/nets.py:7-131 calls (graph builders in TensorFlow; eager here).  Variables come
# from VARIABLES, keyed by their full TensorFlow name ('vgg16_netvlad_pca/conv1_1/kernel', ...):
# the generator fills it before it calls the reference's function, the way a restored checkpoint
# would (train/train.py:882-905).  A name the dictionary lacks is an error, never an initialiser.
VARIABLES = {}
CREATED = []              # full names in creation order (the generator records them)
_SCOPES = []


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    _count('variable_scope')
    _SCOPES.append(name)
    try:
        yield
    finally:
        _SCOPES.pop()


def _variable(full, shape, dtype):
    if full not in VARIABLES:
        raise KeyError('variable %r was not supplied (tf_shim.VARIABLES)' % full)
    v = np.asarray(VARIABLES[full])
    if shape is not None and tuple(v.shape) != tuple(shape):
        raise ValueError('variable %s: shape %s, the graph asks for %s' % (full, v.shape, tuple(shape)))
    if dtype is not None and v.dtype != np.dtype(dtype):
        raise TypeError('variable %s: dtype %s, the graph asks for %s' % (full, v.dtype, np.dtype(dtype)))
    CREATED.append(full)
    return _t(v)


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True):
    _count('get_variable')
    if isinstance(shape, (int, np.integer)):
        shape = (int(shape),)
    return _variable('/'.join(_SCOPES + [name]), shape, dtype)


def _conv2d_nhwc(x, w, stride, padding):
    """NHWC input, HWIO filter, cross-correlation (TensorFlow's conv2d); 'SAME' with stride 1 pads
    (k - 1) // 2 before and k // 2 after; products and sums in float64, one rounding to float32."""
    kh, kw, cin, cout = w.shape
    if x.shape[3] != cin:
        raise ValueError('conv2d: input has %d channels, filter %d' % (x.shape[3], cin))
    if stride != 1:
        raise NotImplementedError('conv2d stride %r' % (stride,))
    pad = padding.upper()
    if pad == 'SAME':
        x = np.pad(x, ((0, 0), ((kh - 1) // 2, kh // 2), ((kw - 1) // 2, kw // 2), (0, 0)))
    elif pad != 'VALID':
        raise ValueError('conv2d padding %r' % padding)
    b, h, wd, _ = x.shape
    ho, wo = h - kh + 1, wd - kw + 1
    out = np.zeros((b, ho, wo, cout), np.float64)
    w64 = w.astype(np.float64)
    x64 = x.astype(np.float64)
    for dy in range(kh):
        for dx in range(kw):
            out += x64[:, dy:dy + ho, dx:dx + wo, :] @ w64[dy, dx]
    return out


def _nn_conv2d(input, filter, strides, padding, name=None):          # noqa: A002
    _count('nn.conv2d')
    x = _t(input)
    if len(strides) != 4 or any(float(s) != 1.0 for s in strides):
        raise NotImplementedError('nn.conv2d strides %r' % (strides,))
    w = np.asarray(filter, dtype=x.dtype)            # a NumPy filter takes the input's dtype
    return _t(_conv2d_nhwc(x.view(np.ndarray), w, 1, padding).astype(x.dtype))


def _relu(features, name=None):
    _count('nn.relu')
    x = _t(features)
    return _t(np.maximum(x.view(np.ndarray), x.dtype.type(0)))


def _layers_conv2d(inputs, filters, kernel_size, strides=(1, 1), padding='valid', activation=None,
                   use_bias=True, name=None):
    """tf.layers.conv2d: variables '<scope>/<name>/kernel' [kh,kw,in,filters] and '<name>/bias'."""
    _count('layers.conv2d')
    x = _t(inputs)
    if name is None:
        raise NotImplementedError('layers.conv2d without a name (auto-numbered scopes)')
    kh, kw = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
    st = strides if isinstance(strides, int) else strides[0]
    scope = '/'.join(_SCOPES + [name])
    w = _variable(scope + '/kernel', (kh, kw, x.shape[3], filters), x.dtype)
    y = _conv2d_nhwc(x.view(np.ndarray), w.view(np.ndarray), st, padding)
    if use_bias:
        y = y + _variable(scope + '/bias', (filters,), x.dtype).view(np.ndarray).astype(np.float64)
    y = _t(y.astype(x.dtype))
    return activation(y) if activation is not None else y


def _layers_max_pooling2d(inputs, pool_size, strides, padding='valid', name=None):
    """'valid' (the default, and what nets.py:37 gets): windows that do not fit are dropped."""
    _count('layers.max_pooling2d')
    x = _t(inputs).view(np.ndarray)
    if padding.lower() != 'valid' or pool_size != strides or isinstance(pool_size, (tuple, list)):
        raise NotImplementedError('max_pooling2d(%r, %r, %r)' % (pool_size, strides, padding))
    k = int(pool_size)
    b, h, w, c = x.shape
    x = x[:, :h // k * k, :w // k * k, :].reshape(b, h // k, k, w // k, k, c)
    return _t(x.max(axis=(2, 4)))


nn = types.SimpleNamespace(l2_normalize=_l2_normalize, conv2d=_nn_conv2d, relu=_relu)
layers = types.SimpleNamespace(conv2d=_layers_conv2d, max_pooling2d=_layers_max_pooling2d)


class _Reduction:
    NONE = 'none'
    SUM = 'weighted_sum'
    MEAN = 'weighted_mean'
    SUM_BY_NONZERO_WEIGHTS = 'weighted_sum_by_nonzero_weights'


def _huber_loss(labels, predictions, weights=1.0, delta=1.0, scope=None, loss_collection=None,
                reduction=_Reduction.SUM_BY_NONZERO_WEIGHTS):
    """python/ops/losses/losses_impl.py (RECALLED): error = predictions - labels;
    quadratic = min(|error|, delta); linear = |error| - quadratic;
    0.5 * quadratic^2 + delta * linear; default reduction with weights 1.0 = the mean."""
    _count('losses.huber_loss')
    lab, pred = _same_float(labels, predictions, 'huber_loss')
    err = pred.view(np.ndarray) - lab.view(np.ndarray)
    abs_err = np.abs(err)
    quad = np.minimum(abs_err, err.dtype.type(delta))
    lin = abs_err - quad
    out = err.dtype.type(0.5) * quad * quad + err.dtype.type(delta) * lin
    if reduction == _Reduction.NONE:
        return _t(out)
    if reduction == _Reduction.SUM:
        return _t(np.asarray(np.sum(out, dtype=out.dtype)))
    return _t(np.asarray(np.sum(out, dtype=out.dtype) / out.dtype.type(out.size)))


losses = types.SimpleNamespace(huber_loss=_huber_loss, Reduction=_Reduction)


def _unsupported(name):
    def op(*a, **k):
        raise NotImplementedError('tf.%s is outside the hot path (SURVEY.md section 2: OUT OF SCOPE)' % name)
    return op


linalg = types.SimpleNamespace(svd=_unsupported('linalg.svd'), eigh=_unsupported('linalg.eigh'),
                               trace=_unsupported('linalg.trace'))
slice = _unsupported('slice')        # noqa: A001


# ---------------------------------------------------------------------------------------------
# What build_model() of train/train.py:585-879 calls around the losses.  Placeholders are EAGER:
# each takes the next entry of FEEDS whose dtype and static shape fit (None = any extent), so the
# function's own reshape / split / label statements run on real values as it builds its `ops`.
FEEDS = []


def placeholder(dtype, shape=None, name=None):
    _count('placeholder')
    want = np.dtype(dtype)
    for pos, val in enumerate(FEEDS):
        v = np.asarray(val)
        if v.dtype != want:
            continue
        if shape is not None:
            shp = tuple(shape)
            if len(shp) != v.ndim or any(d is not None and int(d) != e for d, e in zip(shp, v.shape)):
                continue
        del FEEDS[pos]
        return _t(v)
    raise LookupError('no fed value left for placeholder(%s, shape=%r)' % (want, shape))


def placeholder_with_default(input, shape, name=None):         # noqa: A002
    _count('placeholder_with_default')
    return _t(input)


def no_op(name=None):
    return None


def split(value, num_or_size_splits, axis=0, num=None, name=None):
    _count('split')
    v = _t(value)
    if isinstance(num_or_size_splits, (int, np.integer)):
        if v.shape[axis] % num_or_size_splits:
            raise ValueError('split: %d does not divide axis of %d' % (num_or_size_splits, v.shape[axis]))
        return [_t(p) for p in np.split(v.view(np.ndarray), num_or_size_splits, axis=axis)]
    sizes = [int(x) for x in num_or_size_splits]
    if builtins.sum(sizes) != v.shape[axis]:
        raise ValueError('split: sizes %r do not add up to %d' % (sizes, v.shape[axis]))
    return [_t(p) for p in np.split(v.view(np.ndarray), np.cumsum(sizes)[:-1], axis=axis)]


class Variable:
    def __init__(self, initial_value, trainable=True, name=None):
        self.value = initial_value


class _Optimizer:
    made = []

    def __init__(self, learning_rate, **kw):
        self.learning_rate, self.kw = learning_rate, kw
        _Optimizer.made.append(self)

    def minimize(self, loss, global_step=None, var_list=None):
        _count('train.minimize')
        return ('train_op', type(self).__name__)


class _Adam(_Optimizer):
    pass


class _Momentum(_Optimizer):
    pass


train = types.SimpleNamespace(AdamOptimizer=_Adam, MomentumOptimizer=_Momentum, made=_Optimizer.made)
SUMMARIES = {}
summary = types.SimpleNamespace(scalar=lambda name, tensor: SUMMARIES.__setitem__(name, tensor))
GraphKeys = types.SimpleNamespace(UPDATE_OPS='update_ops')


def get_collection(key, scope=None):
    return []


@contextlib.contextmanager
def control_dependencies(control_inputs):
    yield


def _layers_flatten(inputs, name=None):
    _count('layers.flatten')
    x = _t(inputs)
    return _t(x.view(np.ndarray).reshape(x.shape[0], -1))


layers.flatten = _layers_flatten
globals()['bool'] = np.bool_             # tf.bool (kept out of a plain assignment: the name is a builtin)


class _SummaryValues(list):
    def add(self, tag=None, simple_value=None):
        self.append((tag, simple_value))


class Summary:
    """tf.Summary as evaluate_localization_thread uses it (train/train.py:362, 380-385): a list of
    (tag, simple_value)."""

    def __init__(self):
        self.value = _SummaryValues()


class Session:
    """The reference's __main__ smoke prints through a session (model/losses.py:712-714)."""
    def run(self, fetches):
        return np.asarray(fetches)
