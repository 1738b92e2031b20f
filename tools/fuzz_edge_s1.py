"""Fuzz of the stride-1 single-channel-side kernels of round 6 (k_wgrad_c1e, k_down_s1_in1m) and their neighbours
(k_down_s1_c1, k_wgrad_c1<1>): 5x5 stride-1 layers between 1 / 2 channels and 16 / 32 on maps of 4 k rows x 64 k columns
and on maps just off those sizes, every split of the 4 padding rows / columns, all roles of the Conv2d and the
ConvTranspose2d entry points against float64 with the tests' own gate.

    python tools/fuzz_edge_s1.py [seed] [count]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from tests import test_gpu_kernels as T  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rng = np.random.RandomState(seed)
    bad = 0
    for i in range(count):
        N = int(rng.choice([1, 2, 3, 5]))
        C = int(rng.choice([1, 1, 2]))
        K = int(rng.choice([16, 16, 32, 24, 8]))
        H = int(rng.choice([4, 8, 12, 36, 64, 30, 7]))
        W = int(rng.choice([64, 64, 128, 192, 60, 68, 72]))
        pt, pl = int(rng.randint(0, 5)), int(rng.randint(0, 5))
        conv = ('edge%d_%dx%d_c%d_k%d_n%d_p%d%d' % (i, H, W, C, K, N, pt, pl), N, C, H, W, K, 5, 1, (pt, 4 - pt), (pl, 4 - pl))
        # the mirrored transposed layer: K channels on the same map onto C (crop = the conv's padding)
        convT = (conv[0] + '_T', N, K, H, W, C, 5, 1, 0, (pl, 4 - pl, pt, 4 - pt), 0)
        for fn, case in ((T.test_conv2d_fwd, conv), (T.test_conv2d_bwd, conv), (T.test_convT2d_fwd, convT),
                         (T.test_convT2d_bwd, convT)):
            try:
                if fn in (T.test_conv2d_fwd, T.test_convT2d_fwd):
                    fn(case, T._hip.ACT_LRELU)
                else:
                    fn(case)
            except BaseException as err:                          # noqa: BLE001
                bad += 1
                print('FAIL %s %s: %s' % (fn.__name__, case, (str(err).splitlines() or [type(err).__name__])[0][:300]), flush=True)
                torch.cuda.synchronize()
    print('fuzz_edge_s1 seed %d: %d geometries x 4 tests, %d failures' % (seed, count, bad))
    return bad


if __name__ == '__main__':
    sys.exit(min(main(), 255))
