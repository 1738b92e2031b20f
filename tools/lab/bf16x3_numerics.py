"""Split-bf16 question, numerics half (runs on the CPU): would the stride-2 middle layers keep fp32
accuracy if every fp32 operand were split ONCE into three bf16 terms (x = x1 + x2 + x3) and the
convolution were evaluated as six bf16 x bf16 products accumulated in fp32
(x1 w1, x1 w2, x2 w1, x2 w2, x1 w3, x3 w1: what six v_mfma_f32_32x32x16_bf16 per K-chunk would do)?

Emulation: a bf16 x bf16 product is exact in fp32 (8 + 8 mantissa bits), so a fp32 convolution of
bf16-VALUED tensors has the arithmetic of the MFMA path up to the order of the fp32 additions; the
six partial convolutions are added in fp32, smallest first.  Judged with the gate of the GPU tests
(tests/test_gpu_kernels.py close()): err <= max(8 * err_cpu32, 3e-6), errors = max |. - f64| / max |f64|.

    python tools/lab/bf16x3_numerics.py            (E1-E3, forward / data gradient / weight gradient)
"""
import torch
import torch.nn.functional as F


def split3(t):
    a = t.to(torch.bfloat16).float()
    r = t - a
    b = r.to(torch.bfloat16).float()
    c = (r - b).to(torch.bfloat16).float()
    return a, b, c


def six(fn, x, w):
    """fn(x_term, w_term) for the six retained products, added smallest first in fp32."""
    x1, x2, x3 = split3(x)
    w1, w2, w3 = split3(w)
    small = (fn(x1, w3) + fn(x3, w1)) + fn(x2, w2)
    mid = fn(x1, w2) + fn(x2, w1)
    return (small + mid) + fn(x1, w1)


def four(fn, x, w):
    """the cheaper variant: two terms per operand, three products (x1 w1, x1 w2, x2 w1)"""
    x1, x2, _ = split3(x)
    w1, w2, _ = split3(w)
    return (fn(x1, w2) + fn(x2, w1)) + fn(x1, w1)


def errs(got, want32, want64):
    scale = want64.abs().max().item()
    return (got.double() - want64).abs().max().item() / scale, \
           (want32.double() - want64).abs().max().item() / scale


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    layers = [('E1', 32, 64, 64), ('E2', 64, 128, 32), ('E3', 128, 256, 16)]
    n = 8
    print('%-4s %-7s | %-28s | %-28s | %-10s | gate = max(8 fp32, 3e-6)' % (
        'lay', 'role', 'six products: err, vs gate', 'three products: err, vs gate', 'fp32 err'))
    for name, cin, cout, hw in layers:
        # activations as they are in the network: LeakyReLU(0.05) outputs of noise-frame features
        x = F.leaky_relu(torch.randn(n, cin, hw, hw) * 0.3 + 0.1, 0.05)
        bound = 1.0 / (cin * 25) ** 0.5
        w = (torch.rand(cout, cin, 5, 5) * 2 - 1) * bound
        dy = torch.randn(n, cout, hw // 2, hw // 2) * 1e-3
        pad = (1, 2, 1, 2)

        def fwd(a, b):
            return F.conv2d(F.pad(a, pad), b, stride=2)

        def dgrad(g, b):      # data gradient of the padded convolution, cropped
            return F.conv_transpose2d(g, b, stride=2)[:, :, 1:1 + hw, 1:1 + hw]

        def wgrad(a, g):
            ap = F.pad(a, pad)
            return torch.nn.grad.conv2d_weight(ap, (cout, cin, 5, 5), g, stride=2)

        for role, fn, args in (('fwd', fwd, (x, w)), ('bwd-d', dgrad, (dy, w)), ('bwd-w', wgrad, (x, dy))):
            want64 = fn(args[0].double(), args[1].double())
            want32 = fn(*args)
            e6, e32 = errs(six(fn, *args), want32, want64)
            e3, _ = errs(four(fn, *args), want32, want64)
            gate = max(8 * e32, 3e-6)
            print('%-4s %-7s | %.2e  %-18s | %.2e  %-18s | %.2e' % (
                name, role, e6, 'PASS' if e6 <= gate else 'FAIL (%.1fx)' % (e6 / gate),
                e3, 'PASS' if e3 <= gate else 'FAIL (%.1fx)' % (e3 / gate), e32))


if __name__ == '__main__':
    main()
