"""In-training evaluation of the reference trainer: the loss on the other region's tuples
(``get_eval_loss``, train/train.py:1112-1149) and the localisation check
(``evaluate_localization`` :1156-1193 + ``evaluate_localization_thread`` :360-397): reference
and query descriptors, the 5 nearest references per query in descriptor space — the
reference builds ``KDTree(ref_features).query(query_features, k=5)``, here the exact HIP
top-n — and from their geographic distances the curves the reference plots:
``top_n[i, j]`` = best distance among the first j + 1 hits, % of queries below a tolerance,
``AUC@Top1`` over 25 tolerances in [0, rad] and ``%<rad m@Top1`` for rad in 50, 25, 10.
"""
import numpy as np
import torch

from ..evaluation import retrieval
from ..model import nets


def extract_features(model, image_set, indices, images_per_pass):
    """``extract_features`` (train/train.py:1196-1213): forward-only descriptors of the listed
    images in list order, in passes of ``images_per_pass`` (the list is padded with image 0 up
    to a multiple, like the reference's padding arrays) -> float32 [len(indices), E] on the
    model's device."""
    dev = next(model.parameters()).device
    idx = np.asarray(indices, dtype=int)
    pad = (-len(idx)) % images_per_pass
    padded = np.concatenate([idx, np.zeros(pad, dtype=int)])
    outs = []
    with torch.no_grad():
        for s in range(0, len(padded), images_per_pass):
            img = torch.from_numpy(image_set.load_images(padded[s:s + images_per_pass])).to(dev)
            outs.append(nets.full_out(img, model=model).float())
    return torch.cat(outs, 0)[:len(idx)]


def localization_metrics(top_g_dists, nearest_d_dist=None, radii=(50, 25, 10)):
    """The summary values of evaluate_localization_thread (train/train.py:363-385).
    ``top_g_dists`` [Q,k]: geographic distance of every retrieved reference."""
    import sklearn.metrics
    g = np.asarray(top_g_dists, dtype=np.float64)
    top_n = np.minimum.accumulate(g, axis=1)                 # best of the first j + 1 hits
    out = {}
    for rad in radii:
        xs = np.linspace(0, rad, num=25)
        for n in range(top_n.shape[1]):
            ys = [float(np.sum(top_n[:, n] < x)) / float(len(top_n)) * 100 for x in xs]
            if n == 0:
                out['%dm-auc@Top1' % rad] = float(sklearn.metrics.auc(xs, ys))
                out['%%<%dm@Top1' % rad] = ys[-1]
            out['%%<%dm@Top%d' % (rad, n + 1)] = ys[-1]
        if nearest_d_dist is not None:
            opt = np.asarray(nearest_d_dist, dtype=np.float64).reshape(-1)
            out['%%<%dm@Optimum' % rad] = float(np.sum(opt < rad)) / float(len(top_n)) * 100
    return out


def evaluate_localization(model, ref_set, ref_indices, query_set, query_indices, images_per_pass,
                          k=5, plots=None):
    """-> (metrics dict, nearest_latent_indices [Q,k] into ``ref_indices``)."""
    from sklearn.neighbors import KDTree
    ref_f = extract_features(model, ref_set, ref_indices, images_per_pass)
    qry_f = extract_features(model, query_set, query_indices, images_per_pass)
    k = min(k, len(ref_indices))
    _, nearest = retrieval.topn_l2(ref_f, qry_f, k)          # KDTree(ref).query(query, k=5)
    nearest = nearest.cpu().numpy()
    ref_xy = np.asarray(ref_set.xy)[np.asarray(ref_indices, dtype=int)]
    qry_xy = np.asarray(query_set.xy)[np.asarray(query_indices, dtype=int)]
    g = np.linalg.norm(qry_xy[:, None, :] - ref_xy[nearest], axis=2)
    nearest_d, _ = KDTree(ref_xy).query(qry_xy, k=1)         # the optimum curve (:1184-1185)
    if plots is not None:                                    # (out_dir, mode, out_name): the three PDFs
        save_localization_plots(plots[0], plots[1], plots[2], g, nearest_d)
    return localization_metrics(g, nearest_d), nearest


def eval_loss(step_loss, sampler, image_set, indices, tuples_per_batch, tuple_shape, device):
    """``get_eval_loss``: mean loss over the listed anchors' tuples, sampled WITHOUT hard
    negatives (eval_loss_cpu_thread passes False, :182); batches whose tuple cannot be built
    are skipped.  ``step_loss(distances, images) -> float``."""
    losses = []
    idx = np.asarray(indices, dtype=int)
    for s in range(0, len(idx) - tuples_per_batch + 1, tuples_per_batch):
        distances, every = sampler.get_tuple(idx[s:s + tuples_per_batch], tuple_shape, False)
        if len(every) != tuples_per_batch * sum(tuple_shape):
            continue
        images = torch.from_numpy(image_set.load_images(every)).to(device)
        losses.append(float(step_loss(distances, images)))
    return (float(np.mean(losses)) if losses else None), len(losses)


def save_example_pictures(out_dir, mode, out_name, query_set, query_indices, ref_set, ref_indices,
                          nearest, rng=None, count=10):
    """The visual examples of the localisation check (train/train.py:400-420): for ``count`` random
    queries the query frame, the retrieved frame and the geographically nearest reference frame side
    by side, captioned with their distances, as ``<out_dir>/<mode>_<out_name>/<query file name>``.
    ``nearest`` [Q,k]: the retrieval result of ``evaluate_localization`` (indices into
    ``ref_indices``).  Returns the folder."""
    import os
    from sklearn.neighbors import KDTree
    from ..util import cv, io
    rng = rng or np.random
    q_idx, r_idx = np.asarray(query_indices, dtype=int), np.asarray(ref_indices, dtype=int)
    ref_xy, qry_xy = np.asarray(ref_set.xy)[r_idx], np.asarray(query_set.xy)[q_idx]
    nearest = np.asarray(nearest, dtype=int)
    d_top = np.linalg.norm(qry_xy - ref_xy[nearest[:, 0]], axis=1)
    d_opt, i_opt = KDTree(ref_xy).query(qry_xy, k=1)
    folder = os.path.join(out_dir, mode + '_' + out_name)
    os.makedirs(folder, exist_ok=True)
    for q in rng.choice(len(q_idx), min(count, len(q_idx)), replace=False):
        query = cv.put_text('Query', query_set.load_raw(q_idx[q]))
        got = cv.put_text('Retrieved {}'.format(d_top[q]), ref_set.load_raw(r_idx[nearest[q, 0]]))
        best = cv.put_text('Optimal {}'.format(d_opt[q][0]), ref_set.load_raw(r_idx[i_opt[q][0]]))
        merged = cv.merge_images(cv.merge_images(query, got), best)
        name = os.path.basename(query_set.path(q_idx[q])) if hasattr(query_set, 'path') else '%d.png' % q_idx[q]
        io.save_img(merged, os.path.join(folder, name))
    return folder


def save_localization_plots(out_dir, mode, out_name, top_g_dists, nearest_d_dist, radii=(50, 25, 10)):
    """The three PDFs of a localisation check (train/train.py:368-396):
    ``<out_dir>/<mode>_<out_name>_<rad>.pdf`` — % of queries localised within a tolerance, one curve
    per Top-1 .. Top-k plus the optimum, AUC@Top1 and %<rad m@Top1 written into the plot.  Returns
    the paths; [] when matplotlib is not importable."""
    import os
    try:
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
    except ImportError:
        return []
    import sklearn.metrics
    g = np.asarray(top_g_dists, dtype=np.float64)
    top_n = np.minimum.accumulate(g, axis=1)
    opt = np.asarray(nearest_d_dist, dtype=np.float64).reshape(-1)
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for rad in radii:
        fig = plt.figure()
        xs = np.linspace(0, rad, num=25)
        for n in range(top_n.shape[1]):
            ys = [float(np.sum(top_n[:, n] < x)) / float(len(top_n)) * 100 for x in xs]
            plt.plot(xs, ys)
            if n == 0:
                plt.text(0.5 * float(rad), 8, 'AUC@Top1={:7.2f}'.format(sklearn.metrics.auc(xs, ys)))
                plt.text(0.5 * float(rad), 2, '%<{}m@Top1={:7.2f}'.format(rad, ys[-1]))
        plt.plot(xs, [float(np.sum(opt < x)) / float(len(top_n)) * 100 for x in xs])
        plt.legend(['Top-%d' % (n + 1) for n in range(top_n.shape[1])] + ['Optimum'])
        plt.ylabel('Correctly localized')
        plt.xlabel('Tolerance [m]')
        plt.xlim(0, rad)
        plt.title(os.path.basename(os.path.dirname(out_dir)) + '\n' + os.path.basename(out_dir) + '\n' +
                  mode + ' ' + out_name)
        path = os.path.join(out_dir, mode + '_' + out_name + '_{}.pdf'.format(rad))
        plt.savefig(path)
        plt.close(fig)
        paths.append(path)
    return paths
