"""numpy restatement of the contracts of include/dartindex.h's entry points -- TEST infrastructure for the CPU suite: it lets
dart_amd/index_build.py's suffix_array_hip (bucket bookkeeping, the 31-symbol round-0 key with its length field, in-place rank refinement,
termination) run without a GPU, against the plain torch.sort prefix doubler and the reference indexer's digests.  The HIP kernels
themselves are checked on the GPU against the same digests (tests/test_gpu_index.py).  Never imported by the package."""
import numpy as np
import torch

TILE = 4096


def pack_text(fwd: np.ndarray) -> torch.Tensor:
    """di_pack_text: forward + reverse complement, 32 symbols per u64 word, first symbol in the top bits, two words of padding."""
    L = len(fwd)
    n = 2 * L
    words = (n + 31) // 32 + 2
    sym = np.zeros(words * 32, dtype=np.uint64)
    sym[:L] = fwd
    sym[L:n] = 3 - fwd[::-1]
    sym = sym.reshape(words, 32)
    w = np.zeros(words, dtype=np.uint64)
    for j in range(32):
        w = (w << np.uint64(2)) | sym[:, j]
    return torch.from_numpy(w.view(np.int64))


def _symbols(T: torch.Tensor) -> np.ndarray:
    w = T.numpy().view(np.uint64)
    out = np.zeros((len(w), 32), dtype=np.uint8)
    for j in range(32):
        out[:, j] = (w >> np.uint64(62 - 2 * j)) & np.uint64(3)
    return out.reshape(-1)


class EmulOps:
    def sync(self):
        pass

    def bucket_hist(self, T, n, table):
        sym = _symbols(T).astype(np.int64)
        tiles = (n + 1 + TILE - 1) // TILE
        i = np.arange(0, n - 1)                                   # suffixes with two real symbols
        pair = sym[i] * 4 + sym[i + 1]
        t = table.numpy().reshape(16, tiles)
        t[:] = 0
        np.add.at(t, (pair, i // TILE), 1)

    def bucket_keys(self, T, n, pair, base, keys, vals):
        sym = _symbols(T).astype(np.uint64)
        i = np.arange(0, n - 1)
        hit = i[(sym[i] * np.uint64(4) + sym[i + 1]) == pair]
        code = np.zeros(len(hit), dtype=np.uint64)
        for j in range(29):
            code = (code << np.uint64(2)) | sym[hit + 2 + j]     # zero past the end (the padding words)
        f = np.minimum(29, n - hit - 2).astype(np.uint64)
        # the members leave in text order, tile by tile, from base[tile] on
        b = base.numpy().astype(np.int64)
        first_in_tile = np.searchsorted(hit // TILE, np.arange(len(b)))
        at = b[hit // TILE] + (np.arange(len(hit)) - first_in_tile[hit // TILE])
        keys.numpy().view(np.uint64)[at] = (code << np.uint64(5)) | f
        vals.numpy()[at] = hit

    def doubling_keys(self, sa, rank, lo, pos, m, k, N, r2_bits, keys, vals):
        s = sa.numpy()[lo + pos.numpy()[:m].astype(np.int64)]
        r = rank.numpy()
        r1 = (r[s] - lo).astype(np.uint64)
        nxt = s + k
        r2 = np.where(nxt < N, r[np.minimum(nxt, N - 1)] + 1, 0).astype(np.uint64)
        keys.numpy().view(np.uint64)[:m] = (r1 << np.uint64(r2_bits)) | r2
        vals.numpy()[:m] = s

    def sort(self, keys, vals, tk, tv, m, key_bits):
        k = keys.numpy().view(np.uint64)[:m]
        mask = np.uint64((1 << key_bits) - 1) if key_bits < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
        order = np.argsort(k & mask, kind="stable")
        keys.numpy().view(np.uint64)[:m] = k[order]
        vals.numpy()[:m] = vals.numpy()[:m][order]

    def regroup(self, keys, vals, pos, m, lo, rank, sa, new_pos, scratch) -> int:
        k = keys.numpy().view(np.uint64)[:m]
        v = vals.numpy()[:m]
        P = np.arange(m, dtype=np.int64) if pos is None else pos.numpy()[:m].astype(np.int64).copy()
        head = np.ones(m + 1, dtype=bool)
        head[1:m] = k[1:] != k[:-1]
        start = np.maximum.accumulate(np.where(head[:m], P, 0))
        sa.numpy()[lo + P] = v
        rank.numpy()[v] = lo + start
        tied = ~(head[:m] & head[1:])
        t = int(tied.sum())
        new_pos.numpy()[:t] = P[tied]
        return t
