import sys, numpy as np, torch
sys.path.insert(0, '.')
from ganet_amd import dcn
rng = np.random.default_rng(0)
B, Cin, H, W, Cout, K = 2, 256, 12, 20, 64, 7
x = torch.from_numpy(rng.standard_normal((B, Cin, H, W)).astype(np.float32)).cuda()
w = torch.from_numpy((rng.standard_normal((Cout, Cin, 1, K)) / 40).astype(np.float32)).cuda()
off = torch.from_numpy((rng.standard_normal((B, 2 * K, H, W)) * 2).astype(np.float32)).cuda()
for dt in (torch.float32, torch.bfloat16, torch.float16):
    xx = x.to(dt).contiguous(memory_format=torch.channels_last)
    a = dcn.deform_conv_forward(xx, off, w.to(dt), (1, 1), (0, 3), (1, 1), algo=1).float()
    b = dcn.deform_conv_forward(xx, off, w.to(dt), (1, 1), (0, 3), (1, 1), algo=2).float()
    d = (a - b).abs()
    print(dt, "max diff", d.max().item(), "scale", a.abs().max().item())
    bad = (d > 0.05 * a.abs().max()).nonzero()
    print(" bad count", len(bad), "of", d.numel())
    if len(bad):
        print(" bad b:", bad[:, 0].unique().tolist(), " o:", bad[:, 1].unique().tolist()[:70])
        print(" h:", bad[:, 2].unique().tolist(), " w:", bad[:, 3].unique().tolist())
