"""Evaluation / replication harness around the accelerated classes: the text
interchange format of the reference's replication study and its error summaries.

Mirrors /root/reference/test.py:
  * article_test_load_data_util        test.py:18-67    one replication read from whitespace text files
  * rmse_ (in vipsy_amd.vi)            test.py:70-91
  * multiprocess_article_test_...      test.py:94-127   `try_count` replications (here: sequential on the GPU,
                                                        or one per rank when a process group is up)
  * article_test_util                  test.py:144-201  generate (vipsy_amd.random_data) -> dump text -> fit -> errors
  * multiprocess_article_test_util     test.py:204-236  `try_count` of those
  * print_rmse                         test.py:130-141  mean / std of each error over the replications
File layout (np.savetxt / np.loadtxt, whitespace separated): `<prefix>_<k>.txt` the N x J responses (0 / 1 / nan),
`<prefix>_b_<k>.txt`, `_a_`, `_c_`, `_d_` the true item parameters stored transposed (J x D), test.py:43-58.
"""
import os
from collections import namedtuple

import numpy as np
import torch


def file_prefix(model_name, sample_size, item_size, x_feature_size):
    return "irt_%s_sample_%d_item_%d_dim_%d" % (model_name, sample_size, item_size, x_feature_size)   # test.py:29


def load_responses(path):
    """Whitespace / tab separated 0 / 1 / nan text (np.loadtxt, test.py:43; lsat.dat is tab-separated ints) ->
    uint8 [N][J] with 255 for a missing cell (the storage contract of include/vipsy_amd.h)."""
    y = np.loadtxt(path, dtype=np.float64, ndmin=2)
    miss = np.isnan(y)
    if not np.isin(y[~miss], (0.0, 1.0)).all():
        raise ValueError("%s: responses must be 0, 1 or nan" % path)
    out = np.where(miss, 255, y).astype(np.uint8)
    return np.ascontiguousarray(out)


def save_responses(path, y_u8):
    """Inverse of load_responses (np.savetxt text; 255 -> nan)."""
    y = np.asarray(y_u8).astype(np.float64)
    y[np.asarray(y_u8) == 255] = np.nan
    np.savetxt(path, y, fmt="%g")


def save_case(folder, model_name, y_u8, items, x_feature_size, file_postfix=0):
    """Write one replication in the reference's layout (test.py:165-176 writes the same files)."""
    n, j = np.asarray(y_u8).shape
    pre = os.path.join(folder, file_prefix(model_name, n, j, x_feature_size))
    save_responses("%s_%d.txt" % (pre, file_postfix), y_u8)
    for k, v in items.items():
        v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        np.savetxt("%s_%s_%d.txt" % (pre, k, file_postfix), np.atleast_2d(v).T)
    return pre


def load_case(folder, model_name, sample_size, item_size, x_feature_size, file_postfix=0):
    """-> (y uint8 [N][J], namedtuple R of true parameters shaped like param(name))  (test.py:29-58)."""
    pre = os.path.join(folder or "", file_prefix(model_name, sample_size, item_size, x_feature_size))
    full = "irt_" + model_name
    attrs = ["b"] + (["a"] if full != "irt_1pl" else []) + (["c"] if full in ("irt_3pl", "irt_4pl") else [])
    attrs += ["d"] if full == "irt_4pl" else []
    R = namedtuple("R", attrs)
    vals = {}
    for k in attrs:
        v = np.loadtxt("%s_%s_%d.txt" % (pre, k, file_postfix), ndmin=2)
        vals[k] = torch.from_numpy(v.T.copy()).float()                                            # stored transposed
    return load_responses("%s_%d.txt" % (pre, file_postfix)), R(**vals)


def article_test_load_data_util(model_name, sample_size, item_size, x_feature_size, file_postfix=0, vi_class=None,
                                vi_class_kwargs=None, vi_fit_kwargs=None, folder=None, device=None):
    """One replication from files: fit, then the reference's error metric (test.py:18-67)."""
    from . import vi
    vi_class = vi_class if vi_class is not None else vi.VaeIRT
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    y, r = load_case(folder, model_name, sample_size, item_size, x_feature_size, file_postfix)
    kw = {"data": torch.from_numpy(y).to(device), "model": "irt_" + model_name, "x_feature": x_feature_size}
    kw.update(vi_class_kwargs or {})
    model = vi_class(**kw)
    fit_kw = {"optim": vi.Adam({"lr": 1e-2}), "max_iter": 10000, "progress": False}
    fit_kw.update(vi_fit_kwargs or {})
    model.fit(random_instance=r, **fit_kw)
    out = vi.rmse_(item_size, "irt_" + model_name, r, x_feature_size)
    vi.clear_param_store()
    return out


def article_test_util(sample_size=500, item_size=50, vi_class=None, vi_class_kwargs=None, vi_fit_kwargs=None,
                      random_class=None, random_class_kwargs=None, file_postfix=0, folder=None, device=None, seed=None):
    """One replication of the simulation study (test.py:144-201): draw item parameters and responses with `random_class`
    (vipsy_amd.random_data), dump them in the whitespace-text layout the R comparison script reads (test.py:165-190:
    `<name>_sample_<N>_item_<J>_dim_<D>[_a|_b|_c|_d]_<k>.txt`, item parameters transposed), fit, return the reference's
    error metric.  `seed` replaces the reference's `torch.manual_seed(int(random.random() * 1000))`."""
    import random as _random
    from . import vi, random_data
    vi_class = vi_class if vi_class is not None else vi.VaeIRT
    random_class = random_class if random_class is not None else random_data.RandomIrt2PL
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    model_name = random_class.name
    rkw = {"sample_size": sample_size, "item_size": item_size, "device": device}
    rkw.update(random_class_kwargs or {})
    torch.manual_seed(int(_random.random() * 1000) if seed is None else int(seed))
    r = random_class(**rkw)
    y = r.y
    x_feature = int(r.x.shape[1])
    if folder is not None:
        os.makedirs(folder, exist_ok=True)
    pre = os.path.join(folder or "", "%s_sample_%d_item_%d_dim_%d" % (model_name or "data", sample_size, item_size, x_feature))
    save_responses("%s_%d.txt" % (pre, file_postfix), y.cpu().numpy())
    np.savetxt("%s_b_%d.txt" % (pre, file_postfix), r.b.numpy())
    for k in ("a", "c", "d"):
        if hasattr(r, k):
            np.savetxt("%s_%s_%d.txt" % (pre, k, file_postfix), getattr(r, k).T.numpy())
    kw = {"data": y, "model": model_name, "subsample_size": 100, "x_feature": x_feature}
    kw.update(vi_class_kwargs or {})
    model = vi_class(**kw)
    fit_kw = {"optim": vi.Adam({"lr": 1e-2}), "max_iter": 10000, "progress": False}
    fit_kw.update(vi_fit_kwargs or {})
    model.fit(random_instance=r, **fit_kw)
    out = vi.rmse_(item_size, model_name, r, x_feature)
    vi.clear_param_store()
    return out


def multiprocess_article_test_util(sample_size=500, item_size=50, vi_class=None, vi_class_kwargs=None, vi_fit_kwargs=None,
                                   random_class=None, random_class_kwargs=None, start_idx=0, try_count=10, process_size=None,
                                   folder=None, device=None, seed=None):
    """`try_count` replications of article_test_util (test.py:204-236): one after the other on this GPU, or replication k on
    rank k % world under torch.distributed (every replication owns its problem: no collective inside a fit)."""
    import torch.distributed as dist
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_available() and dist.is_initialized() else (1, 0)
    mine = []
    for i in range(start_idx, start_idx + try_count):
        if (i - start_idx) % world == rank:
            mine.append((i, article_test_util(sample_size, item_size, vi_class, vi_class_kwargs, vi_fit_kwargs, random_class,
                                               random_class_kwargs, file_postfix=i, folder=folder, device=device,
                                               seed=None if seed is None else seed + i)))
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        mine = sorted(sum(gathered, []), key=lambda t: t[0])
    res = [d for _, d in mine]
    if rank == 0:
        print_rmse(res)
    return summarize_rmse(res)


def summarize_rmse(res_lt):
    """{key: (mean, std)} over replications (print_rmse, test.py:130-141, numpy population std)."""
    acc = {}
    for d in res_lt:
        for k, v in d.items():
            acc.setdefault(k, []).append(float(v))
    return {k: (float(np.mean(v)), float(np.std(v))) for k, v in acc.items()}


def print_rmse(res_lt):
    for k, (m, s) in summarize_rmse(res_lt).items():
        print("%s_mean:%s" % (k, m))
        print("%s_std:%s" % (k, s))


def multiprocess_article_test_load_data_util(model_name, sample_size, item_size, x_feature_size, vi_class=None,
                                             try_count=10, vi_class_kwargs=None, vi_fit_kwargs=None, process_size=None,
                                             start_idx=0, folder=None, device=None):
    """`try_count` replications (test.py:94-127).  The reference farms them to a CPU process pool; here a replication
    saturates a GPU, so they run one after the other -- or, under torch.distributed, replication k runs on rank
    k % world and rank 0 gathers the errors.  Every replication is built WITHOUT a process group (the model classes
    share a problem across ranks only when handed `group=`), so nothing inside a fit is collective and ranks may run
    different numbers of replications; the single all_gather_object at the end is the only exchange."""
    import torch.distributed as dist
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_available() and dist.is_initialized() else (1, 0)
    mine = []
    for i in range(start_idx, start_idx + try_count):
        if (i - start_idx) % world == rank:
            mine.append((i, article_test_load_data_util(model_name, sample_size, item_size, x_feature_size,
                                                         file_postfix=i, vi_class=vi_class,
                                                         vi_class_kwargs=vi_class_kwargs, vi_fit_kwargs=vi_fit_kwargs,
                                                         folder=folder, device=device)))
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        mine = sorted(sum(gathered, []), key=lambda t: t[0])
    res = [d for _, d in mine]
    if rank == 0:
        print_rmse(res)
    return summarize_rmse(res)
