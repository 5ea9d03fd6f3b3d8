"""PCA whitening of the 32768-d descriptors on the device.

The reference fits ``sklearn.decomposition.PCA(whiten=True, n_components=d)`` on a PCA
feature set and transforms reference and query features with it before the kNN
(evaluation/top-n.py:74-77).  With n samples of E = 32768 dimensions (n << E) the exact
fit is an eigen-decomposition of the n x n Gram matrix of the centred data:

    Xc = X - mean,  Xc Xc^T = U diag(s^2) U^T,  components V = Xc^T U / s,
    explained_variance = s^2 / (n - 1),  whitened(x) = (x - mean) V / sqrt(explained_variance)

The Gram matrix and the eigen-solve run in float64 on the device (library GEMM / rocSOLVER
through torch — plumbing, not a kernel of this package); the projection matrix is stored in
float32 like scikit-learn does for float32 input.  Component signs follow scikit-learn's
``svd_flip(u_based_decision=False)``: the entry of largest magnitude of every component is
positive.  (scikit-learn picks a *randomized* solver for inputs this large, so the
reference's own output is only defined up to that solver's tolerance; L2 distances between
whitened vectors do not depend on the signs.)
"""
import numpy as np
import torch


class PCAWhitening:
    def __init__(self, n_components, device=None, chunk=4096):
        self.n_components = int(n_components)
        self.device = torch.device(device) if device is not None else None
        self.chunk = int(chunk)
        self.mean_ = None               # [E] f32
        self.components_ = None         # [d, E] f32 (rows = principal axes, sklearn layout)
        self.explained_variance_ = None  # [d] f64
        self._proj = None               # [E, d] f32 = V / sqrt(explained_variance)
        self._offset = None             # [d] f32 = mean @ _proj

    def _dev(self, x):
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        dev = self.device if self.device is not None else t.device
        return t.to(dev, torch.float32)

    def fit(self, x):
        x = self._dev(x)
        n, e = x.shape
        d = self.n_components
        if not 1 <= d <= min(n, e):
            raise ValueError('n_components=%d must be between 1 and min(n_samples, n_features)=%d'
                             % (d, min(n, e)))
        mean = torch.cat([x[:, c:c + self.chunk].double().mean(dim=0)
                          for c in range(0, e, self.chunk)])
        small_side_is_samples = n <= e
        m = n if small_side_is_samples else e
        gram = torch.zeros(m, m, dtype=torch.float64, device=x.device)
        if small_side_is_samples:
            for c in range(0, e, self.chunk):
                xc = x[:, c:c + self.chunk].double() - mean[c:c + self.chunk]
                gram.addmm_(xc, xc.t())
        else:
            for r in range(0, n, self.chunk):
                xc = x[r:r + self.chunk].double() - mean
                gram.addmm_(xc.t(), xc)
        evals, evecs = torch.linalg.eigh(gram)
        evals = evals.flip(0)[:d].clamp_min(0.0)
        evecs = evecs.flip(1)[:, :d]
        if small_side_is_samples:
            s = evals.sqrt()
            comp = torch.empty(e, d, dtype=torch.float64, device=x.device)
            scale = torch.where(s > 0, 1.0 / s, torch.zeros_like(s))
            for c in range(0, e, self.chunk):
                xc = x[:, c:c + self.chunk].double() - mean[c:c + self.chunk]
                comp[c:c + self.chunk] = (xc.t() @ evecs) * scale
        else:
            comp = evecs
        # svd_flip(u_based_decision=False)
        pick = comp.abs().argmax(dim=0)
        sign = torch.sign(comp[pick, torch.arange(d, device=x.device)])
        sign = torch.where(sign == 0, torch.ones_like(sign), sign)
        comp = comp * sign
        var = evals / max(n - 1, 1)
        inv_std = torch.where(var > 0, var.rsqrt(), torch.zeros_like(var))
        self.mean_ = mean.float()
        self.components_ = comp.t().contiguous().float()
        self.explained_variance_ = var
        self._proj = (comp * inv_std).float().contiguous()
        self._offset = (mean @ (comp * inv_std)).float()
        return self

    def transform(self, x, batch=16384):
        """[m, E] -> [m, d] float32 on the fit device."""
        if self._proj is None:
            raise RuntimeError('PCAWhitening.transform called before fit')
        host = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        if host.shape[1] != self._proj.shape[0]:
            raise ValueError('expected %d features, got %d' % (self._proj.shape[0], host.shape[1]))
        out = torch.empty(host.shape[0], self.n_components, dtype=torch.float32,
                          device=self._proj.device)
        for r in range(0, host.shape[0], batch):
            blk = host[r:r + batch].to(self._proj.device, torch.float32)
            out[r:r + batch] = blk @ self._proj - self._offset
        return out
