"""CPU: the index arithmetic of three round-3 kernels restated in numpy (the kernels themselves are checked on the GPU):

* csrc/cholesky.hip blocked_steps -- the 64 x 64 diagonal block factored 4 x 16 rows, the rows below a block updated at once with
  S = U_blk^T U_blk (only the tiles ti <= tj of the register image are touched);
* csrc/decode_head.hip -- the runs of lm_head rows the waves own (balanced to one row), the per-workgroup (max, smallest index)
  partials and their reduction in the next step's embed launch = torch.argmax's first maximal index;
* csrc/gptq_qfnb.hip -- the granule protocol's tag / buffer parity (a workgroup can only be one column ahead of the slowest)."""
import numpy as np
import pytest


def blocked_factor(A):
    """cholesky.hip blocked_steps on a 64 x 64 SPD block: returns the upper factor U (A = U^T U)"""
    n = 64
    a = A.astype(np.float64).copy()                      # a[r][c]: register r of lane c
    for kb in range(4):
        r0, r1 = 16 * kb, 16 * kb + 16
        for j in range(r0, r1):                          # factor_block: the 16 rows of the block
            piv = a[j, j]
            assert piv > 0
            a[j, :j] = 0.0
            a[j, j:] = a[j, j:] / np.sqrt(piv)
            for i in range(j + 1, r1):                   # row_update_to: only the rows of this block
                a[i, :] -= a[j, i] * a[j, :]
        if kb < 3:
            Ub = a[r0:r1, :]                             # finished rows, 0 left of the diagonal
            S = Ub.T @ Ub
            for i in range(r1, n):
                for c in range(n):
                    if (i >> 4) <= (c >> 4):             # the lane's column tile: tiles ti <= tj only
                        a[i, c] -= S[i, c]
    return np.triu(a)


def test_blocked_diagonal_factorisation_is_a_cholesky_factorisation():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((200, 64))
    A = X.T @ X / 200 + 0.01 * np.eye(64)
    U = blocked_factor(A)
    np.testing.assert_allclose(U.T @ U, A, rtol=0, atol=1e-12 * np.abs(A).max())
    np.testing.assert_allclose(U, np.linalg.cholesky(A).T, rtol=0, atol=1e-12)


@pytest.mark.parametrize("vocab,nwg", [(50272, 256), (32000, 256), (777, 256), (4099, 64)])
def test_head_rows_partials_and_the_embed_reduction(vocab, nwg):
    rng = np.random.default_rng(vocab)
    logits = rng.standard_normal(vocab).astype(np.float16)
    logits[rng.integers(0, vocab, 5)] = logits.max()                 # ties for the maximum
    nwaves = nwg * 16
    lo = [(w * vocab) // nwaves for w in range(nwaves)]
    hi = [((w + 1) * vocab) // nwaves for w in range(nwaves)]
    assert lo[0] == 0 and hi[-1] == vocab and all(hi[w] == lo[w + 1] for w in range(nwaves - 1))
    assert max(h - l for l, h in zip(lo, hi)) - min(h - l for l, h in zip(lo, hi)) <= 1        # runs differ by at most one row
    part_val = np.full(nwg, -np.inf, np.float32)
    part_idx = np.full(nwg, 0x7fffffff, np.int64)
    for wg in range(nwg):
        best, bidx = -np.inf, 0x7fffffff
        for w in range(16):                                           # waves in ascending row order, strict > keeps the first maximum
            for r in range(lo[wg * 16 + w], hi[wg * 16 + w]):
                if float(logits[r]) > best:
                    best, bidx = float(logits[r]), r
        part_val[wg], part_idx[wg] = best, bidx
    # embed_kernel: (v > bv) or (v == bv and idx < bi), entries without rows skipped
    bv, bi = -np.inf, 0x7fffffff
    for v, ix in zip(part_val, part_idx):
        if ix >= 0 and ix != 0x7fffffff and (v > bv or (v == bv and ix < bi)):
            bv, bi = v, ix
    assert bi == int(np.argmax(logits.astype(np.float32)))           # np.argmax / torch.argmax: first maximal index


def test_granule_parity_protocol():
    """a workgroup writes tag t into buffer t & 1 only after it has read every granule of tag t - 1, which every workgroup wrote only
    after reading tag t - 2: so when anyone overwrites a tag-(t-2) granule, nobody still needs it.  Simulated with random scheduling."""
    rng = np.random.default_rng(1)
    G, ncol = 5, 40
    gran = np.zeros((2, G), np.int64)                                # tags only
    state = [dict(col=1, phase="write", seen=0) for _ in range(G)]   # col = tag being processed
    done = 0
    steps = 0
    while done < G and steps < 100000:
        steps += 1
        w = int(rng.integers(0, G))
        st = state[w]
        if st["col"] > ncol:
            continue
        buf = st["col"] & 1
        if st["phase"] == "write":
            old = gran[buf, w]
            assert old in (0, st["col"] - 2)                         # never overwrites something newer or unread-by-protocol
            gran[buf, w] = st["col"]
            st["phase"], st["seen"] = "poll", 0
        else:
            i = st["seen"]
            tag = gran[buf, i]
            assert tag in (0, st["col"] - 2, st["col"])              # never a tag from the future
            if tag == st["col"]:
                st["seen"] += 1
                if st["seen"] == G:
                    st["col"] += 1
                    st["phase"] = "write"
                    if st["col"] > ncol:
                        done += 1
    assert done == G
