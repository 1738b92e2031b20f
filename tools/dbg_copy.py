import torch, numpy as np
torch.cuda.init()
s2 = torch.cuda.Stream()
for n in (5120, 6144, 7168, 1000, 4097):
    host = torch.from_numpy(np.random.default_rng(0).integers(1, 200, size=(3 * n,), dtype=np.uint8)).pin_memory()
    big = torch.full((1 << 20,), 0xEE, dtype=torch.uint8, device='cuda')
    off = 1 << 19
    dst = big[off:off + n]
    torch.cuda.synchronize()
    with torch.cuda.stream(s2):
        dst.copy_(host[n:2 * n], non_blocking=True)
    torch.cuda.synchronize()
    out = big.cpu().numpy()
    ok_in = np.array_equal(out[off:off + n], host[n:2 * n].numpy())
    before = (out[:off] != 0xEE).sum(); after = (out[off + n:] != 0xEE).sum()
    print('n', n, 'interior ok', ok_in, 'clobbered before', before, 'after', after)
