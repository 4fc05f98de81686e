"""Validation-time retrieval metric with the reference's function names (src/utils.py:256-272, 380-426):
the per-row similarity ranking runs in one fused HIP kernel instead of a Python loop over argsorts."""
import numpy as np
import torch

from . import ops
from ._lib import check, lib, ptr, stream_ptr


def retrieval_ranks(embs1, embs2):
    """rank[i] = number of rows of embs1 more similar (cosine) to embs2[i] than its partner embs1[i]."""
    if embs1.shape != embs2.shape:
        raise ValueError("retrieval ranks need two (N, D) embedding sets of equal shape")
    e1, _ = ops.l2norm_fwd(embs1.detach().float().contiguous())          # cosine_similarity normalises both, :268-269
    e2, _ = ops.l2norm_fwd(embs2.detach().float().contiguous())
    n, d = e1.shape
    rank = torch.empty(n, dtype=torch.int32, device=e1.device)
    L = lib()
    nb = L.msn_infonce_workspace_bytes(n, n, n, n, d)
    ws = torch.empty(max(nb, 16) // 4 + 1, dtype=torch.float32, device=e1.device)
    check(L.msn_retrieval_rank(ptr(e1), d, ptr(e2), d, n, d, ptr(rank), ptr(ws), nb, stream_ptr()), "msn_retrieval_rank")
    return rank


def get_ROC_data(embs1, embs2):
    """(thresholds, fraction_correct): fraction of rows whose partner is inside the top int(threshold * N)
    of the similarity ranking, for 100 thresholds in [0, 1] (ref src/utils.py:380-411)."""
    ranks = retrieval_ranks(embs1, embs2).cpu().numpy()
    n = len(ranks)
    thresholds = np.linspace(0, 1, 100)
    top = np.array([int(t * n) for t in thresholds])
    fraction_correct = (ranks[None, :] < top[:, None]).sum(axis=1) / n
    return thresholds, fraction_correct


def get_AUC(embs1, embs2):
    """Area under that curve (ref src/utils.py:414-426)."""
    thresholds, fraction_correct = get_ROC_data(embs1, embs2)
    return np.trapezoid(fraction_correct, thresholds) if hasattr(np, "trapezoid") else np.trapz(fraction_correct, thresholds)
