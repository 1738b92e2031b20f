"""Fuzz of the convolution entry points beyond the seeded sweeps of the test suite:

    python tools/fuzz_conv.py [seed] [count]

draws `count` geometries of each family (tests/test_gpu_kernels.py: _random_conv_cases small / big kernels,
_random_stride5_cases, plus a wilder family here: kernels 1-12, strides 1-5, odd channel counts, any padding) and runs
forward, both data gradients and the weight / bias gradients of the Conv2d AND the ConvTranspose2d entry points against
float64 with the tests' own gate.  Prints every failure with its geometry; exit status = number of failures."""
import os
import sys
import traceback

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from tests import test_gpu_kernels as T  # noqa: E402


def wild_cases(seed, count):
    rng = np.random.RandomState(seed)
    cases = []
    while len(cases) < count:
        R = int(rng.choice([1, 2, 3, 4, 5, 5, 6, 7, 8, 9, 10, 11, 12]))
        st = int(rng.choice([1, 1, 2, 2, 2, 3, 4, 5]))
        H, W = int(rng.randint(R, 50)), int(rng.randint(R, 50))
        C = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 31, 32, 33, 64, 100]))
        K = int(rng.choice([1, 2, 4, 7, 16, 24, 32, 40, 64, 96]))
        N = int(rng.choice([1, 2, 3, 7, 13]))
        pt, pb = int(rng.randint(0, R)), int(rng.randint(0, R))
        pl, pr = int(rng.randint(0, R)), int(rng.randint(0, R))
        P, Q = (H + pt + pb - R) // st + 1, (W + pl + pr - R) // st + 1
        if P < 1 or Q < 1 or N * C * H * W > 3e6 or N * K * P * Q > 3e6:
            continue
        # (the transposed entry points want the exact relation H = (P - 1) st + R - pt - pb)
        if (P - 1) * st + R - pt - pb != H or (Q - 1) * st + R - pl - pr != W:
            continue
        cases.append(('wild%d_s%d_k%d_%dx%d_c%d_k%d_n%d_p%d%d%d%d' % (len(cases), st, R, H, W, C, K, N, pt, pb, pl, pr),
                      N, C, H, W, K, R, st, (pt, pb), (pl, pr)))
    return cases


def many_frames_cases(seed, count):
    """Small maps, MANY frames: the detours that walk the batch in passes (column matrices below 512 MB, tiled / shifted
    copies below 2 GB, frame groups of the stride-1 big-kernel blocks) must get their last, partly filled pass right."""
    rng = np.random.RandomState(seed)
    cases = []
    while len(cases) < count:
        R = int(rng.choice([3, 4, 5, 5, 7, 9]))
        st = int(rng.choice([1, 2, 2, 3]))
        H, W = int(rng.randint(max(R, 4), 28)), int(rng.randint(max(R, 4), 28))
        C = int(rng.choice([1, 2, 3, 16, 32, 33, 64]))
        K = int(rng.choice([1, 2, 16, 32, 48, 64]))
        N = int(rng.choice([64, 97, 200, 256, 300]))
        pt, pb = int(rng.randint(0, R)), int(rng.randint(0, R))
        pl, pr = int(rng.randint(0, R)), int(rng.randint(0, R))
        P, Q = (H + pt + pb - R) // st + 1, (W + pl + pr - R) // st + 1
        if P < 1 or Q < 1 or N * C * H * W > 2e7 or N * K * P * Q > 2e7:
            continue
        if (P - 1) * st + R - pt - pb != H or (Q - 1) * st + R - pl - pr != W:
            continue
        cases.append(('many%d_s%d_k%d_%dx%d_c%d_k%d_n%d_p%d%d%d%d' % (len(cases), st, R, H, W, C, K, N, pt, pb, pl, pr),
                      N, C, H, W, K, R, st, (pt, pb), (pl, pr)))
    return cases


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    families = [
        ('small kernels', T._random_conv_cases(seed, count), T.test_random_geometries_all_roles),
        ('big kernels', T._random_conv_cases(seed + 1, count // 3, big=True), T.test_random_geometries_all_roles),
        ('stride 5', T._random_stride5_cases(seed + 2, count // 2), T.test_random_stride5_geometries_all_roles),
        ('wild', wild_cases(seed + 3, count), T.test_random_geometries_all_roles),
        ('many frames', many_frames_cases(seed + 6, count // 3), T.test_random_geometries_all_roles),
    ]
    bad = 0
    total = 0
    for label, cases, fn in families:
        for case in cases:
            total += 1
            try:
                fn(case)
            except BaseException as err:                          # noqa: BLE001
                bad += 1
                msg = str(err).splitlines()[0][:300] if str(err) else type(err).__name__
                print('FAIL [%s] %s: %s' % (label, case, msg), flush=True)
                if os.environ.get('BN_FUZZ_TRACE') == '1':
                    traceback.print_exc()
                torch.cuda.synchronize()
    # the transposed entry points on the small- and big-kernel families
    for case in T._random_conv_cases(seed + 4, count // 2) + T._random_conv_cases(seed + 5, count // 6, big=True):
        tc = (case[0] + '_T', case[1], case[5], (case[3] + sum(case[8]) - case[6]) // case[7] + 1,
              (case[4] + sum(case[9]) - case[6]) // case[7] + 1, case[2], case[6], case[7], 0,
              (case[9][0], case[9][1], case[8][0], case[8][1]), 0)
        total += 1
        try:
            T.test_random_geometries_all_roles_transposed(tc)
        except BaseException as err:                              # noqa: BLE001
            bad += 1
            print('FAIL [transposed] %s: %s' % (tc, (str(err).splitlines() or [type(err).__name__])[0][:300]), flush=True)
            torch.cuda.synchronize()
    print('fuzz seed %d: %d geometries, %d failures' % (seed, total, bad))
    return bad


if __name__ == '__main__':
    sys.exit(min(main(), 255))
