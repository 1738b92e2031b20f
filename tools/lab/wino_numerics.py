"""Numerical cost of Winograd on the four phase convolutions of a stride-2 5x5 layer.

enc.conv2 (64 -> 128 channels, 32x32 -> 16x16, offsets (1, 1)) on activations of the magnitude the layer sees
(LeakyReLU outputs of noise frames), float32 arithmetic throughout, against the float64 direct convolution:

    out[k, p, q] = sum_{c, r, s} in[c, 2p + r - 1, 2q + s - 1] w[k, c, r, s]

Taps of equal row / column parity form the four sub-kernels 3x3, 3x2, 2x3, 2x2 acting at stride 1 on the four
phases of the input; F(2, 3) / F(2, 2) per axis gives 4 / 3 points per 2 outputs: 16 + 12 + 12 + 9 = 49 products
per 2x2 output block instead of 100.

    python tools/lab/wino_numerics.py        (CPU, numpy)
"""
import numpy as np

# F(2, 3): 2 outputs of a 3-tap filter from 4 inputs, 4 products
BT3 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G3 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT3 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
# F(2, 2): 2 outputs of a 2-tap filter from 3 inputs, 3 products
BT2 = np.array([[1, -1, 0], [0, 1, 0], [0, -1, 1]], dtype=np.float64)
G2 = np.array([[1, 0], [.5, .5], [0, 1]], dtype=np.float64)      # m1 = d1 (g0 + g1) ... see check below
AT2 = np.array([[1, 1, 0], [0, 1, 1]], dtype=np.float64)


def _check_1d():
    rng = np.random.default_rng(0)
    for BT, G, AT, taps in ((BT3, G3, AT3, 3), (BT2, G2, AT2, 2)):
        d = rng.standard_normal(taps + 1)
        g = rng.standard_normal(taps)
        want = np.array([np.dot(d[i:i + taps], g) for i in range(2)])
        got = AT @ ((G @ g) * (BT @ d))
        assert np.allclose(got, want), (taps, got, want)


def fix_f22():
    """F(2, 2) with 3 products: y0 = d0 g0 + d1 g1, y1 = d1 g0 + d2 g1:
    m0 = (d0 - d1) g0, m1 = d1 (g0 + g1), m2 = (d2 - d1) g1 -> y0 = m0 + m1, y1 = m1 + m2."""
    global BT2, G2, AT2
    BT2 = np.array([[1, -1, 0], [0, 1, 0], [0, -1, 1]], dtype=np.float64)
    G2 = np.array([[1, 0], [1, 1], [0, 1]], dtype=np.float64)
    AT2 = np.array([[1, 1, 0], [0, 1, 1]], dtype=np.float64)


def direct64(x, w):
    C, H, W = x.shape
    K = w.shape[0]
    P, Q = H // 2, W // 2
    xp = np.zeros((C, H + 4, W + 4))
    xp[:, 1:1 + H, 1:1 + W] = x
    out = np.zeros((K, P, Q))
    for r in range(5):
        for s in range(5):
            patch = xp[:, r:r + 2 * P:2, s:s + 2 * Q:2]             # in[c, 2p + r - 1, 2q + s - 1]
            out += np.einsum('kc,cpq->kpq', w[:, :, r, s], patch)
    return out


def wino32(x, w):
    """float32 Winograd on the phases; every product and every transform add in float32."""
    f = np.float32
    C, H, W = x.shape
    K = w.shape[0]
    P, Q = H // 2, W // 2
    xp = np.zeros((C, H + 6, W + 6), dtype=f)
    xp[:, 1:1 + H, 1:1 + W] = x.astype(f)
    out = np.zeros((K, P, Q), dtype=f)
    tabs = {3: (BT3.astype(f), G3.astype(f), AT3.astype(f)), 2: (BT2.astype(f), G2.astype(f), AT2.astype(f))}
    for rpar, rt in ((0, 3), (1, 2)):                                # row taps r = rpar, rpar + 2, ...
        for spar, stp in ((0, 3), (1, 2)):
            BTr, Gr, ATr = tabs[rt]
            BTs, Gs, ATs = tabs[stp]
            g = w[:, :, rpar::2, spar::2].astype(f)                  # (K, C, rt, stp)
            U = np.einsum('ia,kcab,jb->kcij', Gr, g, Gs).astype(f)   # filter transform (once per step)
            ph = xp[:, rpar::2, spar::2]                             # phase image: ph[c, p + a, q + b] = in[c, 2(p+a) + rpar - 1, ..]
            for p0 in range(0, P, 2):
                for q0 in range(0, Q, 2):
                    d = ph[:, p0:p0 + rt + 1, q0:q0 + stp + 1]       # (C, rt + 1, stp + 1)
                    V = np.einsum('ia,cab,jb->cij', BTr, d, BTs).astype(f)
                    M = np.einsum('kcij,cij->kij', U, V).astype(f)   # the products, summed over channels in float32
                    Y = np.einsum('ai,kij,bj->kab', ATr, M, ATs).astype(f)
                    out[:, p0:p0 + 2, q0:q0 + 2] += Y
    return out


def main():
    fix_f22()
    _check_1d()
    rng = np.random.default_rng(1)
    C, K, H = 64, 128, 32
    # inputs: LeakyReLU(0.05) of a roughly unit-variance pre-activation, as enc.conv2 sees them
    pre = rng.standard_normal((C, H, H))
    x = np.where(pre > 0, pre, 0.05 * pre)
    w = (rng.random((K, C, 5, 5)) - 0.5) * (2.0 / np.sqrt(C * 25))   # torch's default Conv2d init range
    want = direct64(x, w)
    got_w = wino32(x, w).astype(np.float64)
    # the float32 direct convolution (what the MFMA kernel computes: fp32 FMA chain)
    xf, wf = x.astype(np.float32), w.astype(np.float32)
    got_d = np.zeros_like(want, dtype=np.float32)
    xp = np.zeros((C, H + 4, H + 4), dtype=np.float32)
    xp[:, 1:1 + H, 1:1 + H] = xf
    for r in range(5):
        for s in range(5):
            got_d += np.einsum('kc,cpq->kpq', wf[:, :, r, s], xp[:, r:r + H:2, s:s + H:2]).astype(np.float32)
    scale = np.abs(want).max()
    e_w = np.abs(got_w - want).max() / scale
    e_d = np.abs(got_d.astype(np.float64) - want).max() / scale
    print('enc.conv2 forward, one frame, 64 -> 128 channels, 32x32 -> 16x16')
    print('max |error| / max |out|:  direct float32 %.2e   Winograd-on-phases float32 %.2e   (ratio %.1f)' % (
        e_d, e_w, e_w / e_d))
    print('rms error / rms out:      direct float32 %.2e   Winograd-on-phases float32 %.2e' % (
        np.sqrt(np.mean((got_d - want) ** 2)) / np.sqrt(np.mean(want ** 2)),
        np.sqrt(np.mean((got_w - want) ** 2)) / np.sqrt(np.mean(want ** 2))))
    print('products per 2x2 output block: 49 (16 + 12 + 12 + 9) against 100')


if __name__ == '__main__':
    main()
