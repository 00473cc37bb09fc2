#!/usr/bin/env python3
"""g7_check.py -- G7 (gemm_q80_g7.hip) against the numpy restatement of the canonical fold (tests/canon.py), shape by shape, with a
DIAGNOSIS of a mismatch (which tokens / rows / how far) instead of a bare assertion -- for the first runs of a new kernel on the device.
Test infrastructure (uses the oracle's quantizer); the pytest form of the same cases is tests/test_gpu_fused_roles.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from canon import matmul_q80_canon          # noqa: E402
from nano_amd import binding as nb          # noqa: E402
from oracle import binding as ob            # noqa: E402

CASES = [(64, 1, 1024, (256,)), (17, 1, 512, (512,)), (64, 1, 2048, (1024,)), (64, 0, 1024, (2048, 1024, 1024)), (64, 0, 1024, (3072, 3072)),
         (19, 1, 768, (512,)), (33, 1, 4096, (2560,)), (64, 1, 9728, (2560,)), (48, 0, 2560, (4096, 1024, 1024)), (64, 0, 2560, (9728, 9728)),
         (47, 0, 1024, (2064, 1040, 1008))]


def main():
    o = ob.load_oracle()
    bad_total = 0
    for nb_, kind, n, rows in CASES:
        rng = np.random.default_rng(nb_ + n)
        x = (rng.integers(-32, 33, size=(nb_, n)).astype(np.float32) / np.float32(16.0)).astype(np.float32)
        nw = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32) if kind == 0 else None
        segs = []
        for r in rows:
            segs.append((rng.integers(-127, 128, size=r * n, dtype=np.int8), rng.uniform(1e-4, 2e-3, size=r * n // 64).astype(np.float32), r))
        old = rng.standard_normal((nb_, sum(rows))).astype(np.float32)
        try:
            out, route = nb.op_fused_gemv(0x80, kind, n, segs, x, nw, gs=64, nb=nb_, resid=old if kind == 1 else None, use_gemm=True, want_route=True)
        except Exception as e:                      # noqa: BLE001
            print(f"case nb={nb_} kind={kind} n={n} rows={rows}: LAUNCH FAILED: {e}")
            bad_total += 1
            continue
        ref = np.empty_like(out)
        for b in range(nb_):
            act = o.rmsnorm(x[b], nw) if nw is not None else x[b]
            xq, xs = o.quantize_q80(act, 64)
            r = np.concatenate([matmul_q80_canon(xq, xs, wq, ws, n, rr, 64) for wq, ws, rr in segs])
            ref[b] = (old[b] + r).astype(np.float32) if kind == 1 else r
        ne = out.view(np.uint32) != ref.view(np.uint32)
        nbad = int(ne.sum())
        bad_total += nbad
        msg = f"case nb={nb_} kind={kind} n={n} rows={rows} route={route}: "
        if nbad == 0:
            print(msg + "bit-exact")
            continue
        toks = np.nonzero(ne.any(axis=1))[0]
        rws = np.nonzero(ne.any(axis=0))[0]
        rel = float(np.nanmax(np.abs(out - ref)) / np.abs(ref).max())
        print(msg + f"{nbad} of {ne.size} differ; max rel {rel:.3e}; nan {int(np.isnan(out).sum())}; tokens {toks[:20].tolist()}{'...' if len(toks) > 20 else ''} "
              f"({len(toks)}); rows {rws[:24].tolist()}{'...' if len(rws) > 24 else ''} ({len(rws)}); rows mod 16 {sorted(set((rws % 16).tolist()))}")
        b, r = int(toks[0]), int(np.nonzero(ne[toks[0]])[0][0])
        print(f"    first: token {b} row {r}: got {out[b, r]!r} want {ref[b, r]!r}")
    print("g7_check:", "ALL BIT-EXACT" if bad_total == 0 else f"{bad_total} mismatches")
    return 0 if bad_total == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
