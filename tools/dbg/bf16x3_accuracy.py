"""Accuracy of the three-way bf16 split with six products (csrc/sdf_mlp_x3.h) on the reference network shape, emulated with torch on the
CPU: plain fp32 and the split, both against float64.   python tools/dbg/bf16x3_accuracy.py"""
import numpy as np, torch
torch.manual_seed(0)
def bf16_round(x):  # x float32 tensor -> bf16-rounded float32
    return x.to(torch.bfloat16).to(torch.float32)
def split3(x):
    h = bf16_round(x); r1 = x - h; m = bf16_round(r1); r2 = r1 - m; l = bf16_round(r2)
    return h, m, l
def mm6(W, X):   # W [out,in], X [in,n]
    wh, wm, wl = split3(W); xh, xm, xl = split3(X)
    # small terms first or as MFMA chain order? emulate one accumulator chain: acc += each term in order
    acc = wh @ xh
    acc = acc + wh @ xm; acc = acc + wm @ xh
    acc = acc + wh @ xl; acc = acc + wl @ xh; acc = acc + wm @ xm
    return acc
def softplus(x, dt):
    return torch.nn.functional.softplus(x, beta=100)
# geometric init like the reference (SDF nets): weights ~ N(0, sqrt(2)/sqrt(out))
n=20000
d=256
X = torch.rand(39, n)*2-1
Ws=[torch.randn(d,39)*np.sqrt(2)/np.sqrt(d)]+[torch.randn(d,d)*np.sqrt(2)/np.sqrt(d) for _ in range(6)]
bs=[torch.randn(d,1)*0.01 for _ in range(7)]
w7=torch.randn(1,d)*np.sqrt(np.pi)/np.sqrt(d); b7=torch.tensor([[-0.5]])
def run(kind):
    if kind=='f64':
        h=X.double()
        for W,b in zip(Ws,bs): h=softplus(W.double()@h+b.double(),0)
        return (w7.double()@h+b7.double())
    h=X.clone()
    for W,b in zip(Ws,bs):
        z = (W@h) if kind=='f32' else mm6(W,h)
        h=softplus(z+b,0)
    return (w7@h+b7)   # head in fp32 both
ref=run('f64'); a=run('f32'); c=run('x3')
print('f32 vs f64 max abs', (a.double()-ref).abs().max().item(), 'mean', (a.double()-ref).abs().mean().item())
print('x3  vs f64 max abs', (c.double()-ref).abs().max().item(), 'mean', (c.double()-ref).abs().mean().item())
print('x3 vs f32 max abs', (c-a).abs().max().item(), ' |out| mean', ref.abs().mean().item())
print('sign flips f32 vs f64', ((a>0)!=(ref>0)).sum().item(), 'x3 vs f64', ((c>0)!=(ref>0)).sum().item(), 'x3 vs f32', ((c>0)!=(a>0)).sum().item())
