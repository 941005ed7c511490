"""Test doubles (test infrastructure): CPU stand-ins for the two GPU entry points so
the HOST logic of the product classes can be exercised by `-m "not gpu"` tests.
They are built on the oracle and are never importable from the package."""
import numpy as np
import torch

from oracle import oracle


class OracleIndex:
    """FAISS/FlatIPIndex-shaped object backed by oracle.ip_topk."""

    def __init__(self, d, dtype="f32"):
        self.d, self.dtype = d, dtype
        self.rows = np.zeros((0, d), np.float32)
        self.id_offset = 0

    @property
    def ntotal(self):
        return self.rows.shape[0]

    def add(self, x, normalize=False):
        x = x.detach().cpu().float().numpy() if torch.is_tensor(x) else np.asarray(x, np.float32)
        if normalize:
            x = oracle.normalize_embeddings(x).astype(np.float32)
        self.rows = np.concatenate([self.rows, oracle.quantize(x, self.dtype)], 0)

    def set_id_offset(self, off):
        self.id_offset = int(off)

    def reserve(self, n):
        pass

    def search(self, q, k, exact_dense=False):
        was_tensor = torch.is_tensor(q)
        qn = q.detach().cpu().float().numpy() if was_tensor else np.asarray(q, np.float32)
        D, I = oracle.ip_topk(self.rows, oracle.quantize(qn, self.dtype), int(k), id_offset=self.id_offset)
        return (torch.from_numpy(D), torch.from_numpy(I)) if was_tensor else (D, I)

    def reconstruct_n(self, i0=0, n=None):
        n = self.ntotal - i0 if n is None else n
        return self.rows[i0:i0 + n].copy()


def oracle_maxsim(q, packed, off, mode="maxsim"):
    offs = off.cpu().numpy()
    p = packed.detach().cpu().float().numpy()
    docs = [p[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    return torch.from_numpy(oracle.maxsim_scores(q.detach().cpu().float().numpy(), docs, mode))


def oracle_merge(scores, ids):
    D, I = oracle.merge_topk(scores.cpu().numpy(), ids.cpu().numpy(), scores.shape[2])
    return torch.from_numpy(D), torch.from_numpy(I)


def oracle_maxsim_indexed(q, store, starts, lens, mode="maxsim"):
    st = store.detach().cpu().float().numpy()
    docs = [st[int(a): int(a) + int(n)] for a, n in zip(starts.cpu().tolist(), lens.cpu().tolist())]
    return torch.from_numpy(oracle.maxsim_scores(q.detach().cpu().float().numpy(), docs, mode))


def oracle_maxsim_indexed_batch(q_packed, q_off, store, starts, lens, c_off, mode="maxsim"):
    out = [oracle_maxsim_indexed(q_packed[q_off[j]:q_off[j + 1]], store, starts[c_off[j]:c_off[j + 1]],
                                 lens[c_off[j]:c_off[j + 1]], mode)
           for j in range(len(q_off) - 1) if c_off[j + 1] > c_off[j]]
    return torch.cat(out) if out else torch.zeros(0)
