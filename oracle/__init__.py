"""CPU oracle for the soft-contrastive hot path.  TEST INFRASTRUCTURE ONLY.

This package is a literal float32 NumPy restatement (plus a float64 torch-autograd
twin for gradients) of the reference's hot-path arithmetic:

  * ``model/losses.py``  (wms_loss, ms_loss, logratio_loss, evil_* twins,
    _pairwise_squared_distances)                       -> ``oracle.losses_np``
  * ``netvlad_tf.layers.netVLAD`` + ``model/nets.py:66`` channel L2 norm
                                                        -> ``oracle.netvlad_np``
  * ``pointnetvlad_cls`` tuple losses                  -> ``oracle.losses_np``
  * ``evaluation/top-n.py:103-108`` exact L2 top-N      -> ``oracle.topn_np``
  * ``tf.train.AdamOptimizer`` / ``MomentumOptimizer`` (train/train.py:865-878)
                                                        -> ``oracle.adam_np``

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.  The
product package ``soft_contrastive_learning_amd`` never imports it and has no CPU
fallback: it raises if the HIP library is missing.

PARITY UNPINNED.  The reference ships no tests and no golden vectors for this path,
TensorFlow 1.10 / netvlad_tf / pointnetvlad are not importable in the build
container, and the two third-party modules are neither vendored nor version-pinned
by the reference (README.md:10-11).  The oracle is therefore pinned only by
  (a) the single hand-derivable smoke constant the reference holds
      (model/losses.py:708-711 -> KAT K1), and
  (b) pencil-derived known-answer tests K2..K8 (tests/test_oracle_kats.py).
Third-party algorithms restated from their published sources:
  netvlad_tf  (github.com/uzh-rpg/netvlad_tf_open, python/netvlad_tf/layers.py,
               no version pinned by the reference),
  pointnetvlad (github.com/mikacuy/pointnetvlad, pointnetvlad_cls.py, no version
               pinned by the reference),
  tensorflow   (==1.10.0, README.md:6: python/training/adam.py + the ApplyAdam /
               ApplyMomentum kernels of core/kernels/training_ops.cc).
"""
