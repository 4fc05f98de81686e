"""Oracle for the row-sharded InfoNCE kernel interface (TEST INFRASTRUCTURE).

Same call signature as multimodal_supernovae_amd.loss.HipPairKernels, in dense torch ops: the
(b x N) row / column slabs of the logit matrix are materialised.  Used (a) to check the HIP
kernels shard by shard on the GPU and (b) as the injected compute backend of the world_size-2
gloo tests, which exercise the all-gather / offset / sum-over-ranks algebra on CPU.
Follows ref src/loss.py:22-37 restricted to the rows / columns a rank owns.
"""
import torch


class OraclePairKernels:
    @staticmethod
    def forward(e1_loc, e2_loc, e1_all, e2_all, q_offset, log_scale, bias):
        s = torch.exp(log_scale)
        n = min(e1_all.shape[0], e2_all.shape[0])
        lse_row = torch.logsumexp((e2_loc @ e1_all.T) * s + bias, dim=1)   # rows of S owned here
        lse_col = torch.logsumexp((e1_loc @ e2_all.T) * s + bias, dim=1)   # columns of S owned here
        nb = min(e1_loc.shape[0], e2_loc.shape[0])
        diag = (e1_loc[:nb] * e2_loc[:nb]).sum(dim=1) * s + bias
        valid = (q_offset + torch.arange(nb, device=e1_loc.device)) < n
        loss = ((lse_row[:nb] + lse_col[:nb] - 2 * diag) * valid).sum() / (2 * n)
        return lse_row, lse_col, loss

    @staticmethod
    def _side(q_loc, k_all, lse_q_all, lse_k_all, q_offset, n, s, bias):
        S = (q_loc @ k_all.T) * s + bias                                    # (b, N): query x key
        qi = q_offset + torch.arange(q_loc.shape[0], device=q_loc.device)
        kj = torch.arange(k_all.shape[0], device=q_loc.device)
        q_in, k_in = (qi < n), (kj < n)
        lq = torch.where(q_in, lse_q_all[qi.clamp(max=lse_q_all.shape[0] - 1)], torch.zeros_like(S[:, 0]))
        lk = torch.where(k_in, lse_k_all[kj.clamp(max=lse_k_all.shape[0] - 1)], torch.zeros_like(S[0]))
        G = torch.exp(S - lq[:, None]) * q_in[:, None] + torch.exp(S - lk[None, :]) * k_in[None, :]
        G = G - 2.0 * ((qi[:, None] == kj[None, :]) & q_in[:, None])
        return G, S

    @classmethod
    def backward(cls, e1_loc, e2_loc, e1_all, e2_all, q_offset, log_scale, bias, lse_row_all, lse_col_all, g):
        s = torch.exp(log_scale)
        n = min(e1_all.shape[0], e2_all.shape[0])
        c = g / (2 * n)
        G0, S0 = cls._side(e2_loc, e1_all, lse_row_all, lse_col_all, q_offset, n, s, bias)
        G1, _ = cls._side(e1_loc, e2_all, lse_col_all, lse_row_all, q_offset, n, s, bias)
        d2 = c * s * (G0 @ e1_all)
        d1 = c * s * (G1 @ e2_all)
        return d1, d2, c * (G0 * (S0 - bias)).sum(), c * G0.sum()
