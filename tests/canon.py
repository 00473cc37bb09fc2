"""numpy restatement of the FAST path's canonical fp32 fold of a Q80 projection (nano_amd/csrc/kernels.h q80_canonical(),
gemm_q80_g6.hip): what the device must produce BIT FOR BIT when strict mode is off.

    integer group sums   ival[r, g] = sum_k wq[r, g, k] * xq[g, k]              exact (int32)                 infer/infer.c:668-671
    products             p[r, g]    = ((float)ival * ws[r, g]) * xs[g]          two fp32 roundings, as the reference  infer.c:672
    unit sums            S[r, u]    = ((p[8u] + p[8u+1]) + ...) + p[8u+7]       8 groups = 512 bytes of the row, ascending
    row value            out[r]     = ((S[0] + S[1]) + S[2]) + ...              units ascending

The reference adds all groups of a row in ascending order (strict mode does too); the canonical shape differs from it only in the
association of fp32 additions -- held to SURVEY 7 tier ii's 1e-5 relative in the tests, next to this bit-exact restatement.
Test infrastructure only."""
import numpy as np


def matmul_q80_canon(xq, xs, wq, ws, n, rows, gs=64):
    assert gs == 64 and n % 256 == 0
    ng = n // gs
    W = np.asarray(wq, np.int8).reshape(rows, ng, gs).astype(np.int32)
    X = np.asarray(xq, np.int8).reshape(ng, gs).astype(np.int32)
    ival = np.einsum("rgk,gk->rg", W, X).astype(np.int32)
    p = (ival.astype(np.float32) * np.asarray(ws, np.float32).reshape(rows, ng)).astype(np.float32)
    p = (p * np.asarray(xs, np.float32)[None, :]).astype(np.float32)
    out = None
    for u in range((ng + 7) // 8):
        s = p[:, 8 * u].copy()
        for k in range(1, min(8, ng - 8 * u)):
            s = (s + p[:, 8 * u + k]).astype(np.float32)
        out = s if out is None else (out + s).astype(np.float32)
    return out
