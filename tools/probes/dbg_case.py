import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch, torch.nn.functional as F, ctypes as C
from tools import gpu_lab as lab
from dmmfods_amd import _lib
B,H,W,Cin,Cout,R,S,stride,pad = 2,12,20,72,32,1,1,1,0
g = torch.Generator().manual_seed(0)
dt = torch.float32
x = (torch.randn(B, Cin, H, W, generator=g) * 2 + 0.5)
scale = torch.rand(Cin, generator=g) + 0.5
shift = torch.randn(Cin, generator=g) * 0.5
w = torch.randn((Cout, Cin, R, S), generator=g) / (Cin * R * S) ** 0.5
a = F.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
y = F.conv2d(a, w, stride=stride, padding=pad)
for mfma in (0, 1):
    d = _lib.ConvDesc(dtype=0, use_mfma=mfma, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=R, S=S, stride=stride, pad=pad, transposed=0, mode=0, bn_relu=1)
    nsc = lab.L.dmm_conv_scratch_bytes(C.byref(d))
    scratch = torch.zeros(nsc, dtype=torch.uint8, device='cuda')
    xd = lab.nhwc(x, dt).cuda()
    hd = torch.cat([shift, torch.zeros(Cin), torch.ones(Cin)]).cuda()
    yd = torch.full((B, H, W, Cout), float('nan'), device='cuda')
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
    wdev, sdev = w.cuda(), scale.cuda()   # (kept alive: a temporary's block is handed to the next allocation at once)
    _lib.check(lab.L.dmm_conv_forward(C.byref(d), xd.data_ptr(), wdev.data_ptr(), sdev.data_ptr(), hd.data_ptr(), yd.data_ptr(), stats.data_ptr(), scratch.data_ptr(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    yo = lab.nchw(yd).cpu()
    bad = (yo - y).abs() > 1e-3
    print('mfma', mfma, 'bad count', bad.sum().item(), 'per channel', bad.sum(dim=(0,2,3)).tolist()[:8])
    if bad.any():
        yf = y.permute(0,2,3,1).reshape(-1, Cout); of = yo.permute(0,2,3,1).reshape(-1, Cout)
        rows = bad.permute(0,2,3,1).reshape(-1, Cout)
        fullbad = [i for i in range(rows.shape[0]) if rows[i].sum() > 4]
        print('rows with >4 bad channels:', fullbad)
        for i in fullbad[:3]: print(' row', i, 'got', of[i, :8].tolist(), 'want', yf[i, :8].tolist())
        for i in [0, 1, 2, 130, 131]:
            print(' row', i, 'ch0 got', of[i,0].item(), 'want', yf[i,0].item(), 'sum_sq(row)', (yf[i]**2).sum().item(), 'sum(row)', yf[i].sum().item(), 'sum ch0..3', yf[i,:4].sum().item())
        print('col sums of true y, ch0:', yf[:,0].sum().item(), 'sq', (yf[:,0]**2).sum().item())
        print('stats[0], stats[32]:', stats[0].item(), stats[32].item())
