"""Random shapes through the persistent chain (option persist = 2) against the oracle: the checks of
tests/test_engine_gpu.py::test_persistent_chain on shapes the parametrised test does not list (geometry switches of the backward
kernel at C > 896, partial 32-cell slots, ragged fold tiles, wide inputs through the batched x-projection, 1..16 streams, short and
long T).  Diagnostic: prints one line per shape, exits non-zero on the first failure."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_engine_gpu import run_chunks, check
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = False
for it in range(n):
    C = int(rng.choice([64, 96, 200, 264, 520, 800, 904, 1000, 1024]))
    R = int(rng.choice([32, 64, 128, 256, 512]))
    R = min(R, C)
    I = int(rng.choice([40, 8, 64, 128, 512]))
    S = int(rng.randint(1, 17))                            # (round 6: 9..16 streams as three / four interleaved chains where the shape allows)
    T = int(rng.choice([8, 9, 13, 20, 33]))
    big = C > 96                                         # (the 0.3-scale parameters of the small test shapes saturate wider layers)
    try:
        recs = run_chunks(I, C, R, S, T, nchunks=2, scale=0.01 if big else 0.3, momentum=0.9, lr=1e-5 if big else 1e-3,
                          want_in_diff=bool(rng.randint(0, 2)), od_scale=0.1 if big else 1.0, persist=2, waves=0, tpw=0)
        check(recs, tol_act=3e-5, tol_grad=3e-4 if big else 1e-4, C=C, S=S, T=T)
        print("ok   I=%d C=%d R=%d S=%d T=%d" % (I, C, R, S, T), flush=True)
    except Exception as ex:
        print("FAIL I=%d C=%d R=%d S=%d T=%d: %s" % (I, C, R, S, T, str(ex)[:300]), flush=True)
        bad = True
sys.exit(1 if bad else 0)
