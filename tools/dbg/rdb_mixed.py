import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch, torch.nn.functional as F
from resselt_amd.engine import lib as L, ops, tensors
from resselt_amd.engine.tensors import PF_F16
from test_conv_fp16_gpu import _rand, _h, _conv
device = torch.device('cuda:0')
n, h, w, pf, pg = 1, 37, 70, 8, 4
g = torch.Generator().manual_seed(5)
x = _rand((n, 64, h, w), 5); x0 = _rand((n, 64, h, w), 6)
ws = tensors.Planes.empty(n, pf + 4 * pg, h, w, device, True, PF_F16, lo_planes=pf)
ws.hi.zero_()
src = tensors.nchw_to_planes(x.to(device), True, PF_F16)
ws.hi[:, :pf] = src.hi; ws.lo[:, :pf] = src.lo
r0src = tensors.nchw_to_planes(x0.to(device), True, PF_F16)
r0 = tensors.Planes.empty(n, pf + 4 * pg, h, w, device, True, PF_F16, lo_planes=pf)
r0.hi[:, :pf] = r0src.hi; r0.lo[:, :pf] = r0src.lo
out = tensors.Planes.empty(n, pf, h, w, device, True, PF_F16)
cat = _h(tensors.planes_to_nchw(src, 64).cpu())
for j in range(1, 6):
    cin, cout = 64 + 32 * (j - 1), 32 if j < 5 else 64
    wt = (torch.rand((cout, cin, 3, 3), generator=g) * 2 - 1) / (cin * 9) ** 0.5
    b = (torch.rand((cout,), generator=g) * 2 - 1) * 0.1
    wts = ops.ConvWeights.from_oihw(wt, b, 1, device=device, fmt=PF_F16)
    y = _conv(cat, _h(wt), b)
    if j < 5:
        y = F.leaky_relu(y, 0.2)
        p = ops.conv_params(wts, ws, h, w, cin_planes=cin // 8, out=ws, out_plane_off=pf + (j - 1) * pg, act=L.ACT_LRELU, act_param=0.2)
        ops.run_convs([p], device); torch.cuda.synchronize()
        got = tensors.planes_to_nchw(tensors.Planes(ws.hi[:, pf + (j-1)*pg: pf + j*pg].contiguous(), None), 32).cpu()
        e = (got - _h(y)).abs()
        print('conv', j, L.conv_kernel_name(p), 'err', e.max().item(), 'argmax', [int(v) for v in (e == e.max()).nonzero()[0]], 'scale', y.abs().max().item())
        cat = torch.cat((cat, _h(y)), 1)
    else:
        xs = tensors.planes_to_nchw(src, 64).cpu(); r0v = tensors.planes_to_nchw(r0src, 64).cpu()
        for variant in ('r1', 'r1r2'):
            kw = dict(res1=(ws, 0), alpha=0.2)
            ref = y * 0.2 + xs
            if variant == 'r1r2':
                kw.update(res2=(r0, 0), beta=0.2); ref = ref * 0.2 + r0v
            p = ops.conv_params(wts, ws, h, w, cin_planes=cin // 8, out=out, **kw)
            ops.run_convs([p], device); torch.cuda.synchronize()
            got = tensors.planes_to_nchw(out, 64).cpu()
            e = (got - ref).abs()
            print('conv5', variant, L.conv_kernel_name(p), 'err', e.max().item(), 'argmax', [int(v) for v in (e == e.max()).nonzero()[0]], 'mean', e.mean().item())
            # per-channel-group error
            print('  per 16ch', [round(e[:, c:c+16].max().item(), 5) for c in range(0, 64, 16)], 'rows', [round(e[:, :, r:r+4].max().item(), 5) for r in range(0, h, 4)])
print('aborts', L.ring_aborts())
