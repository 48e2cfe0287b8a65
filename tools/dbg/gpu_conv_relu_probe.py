"""does MIOpen's fused conv+bias+relu (aten::miopen_convolution_relu) beat conv + add_ + clamp_min for the AlexNet-LPIPS shapes?"""
import torch, time
torch.backends.cudnn.benchmark = True
dev = 'cuda'
shapes = [((4, 3, 1024, 1024), (64, 3, 11, 11), 4, 2), ((4, 64, 127, 127), (192, 64, 5, 5), 1, 2), ((4, 192, 63, 63), (384, 192, 3, 3), 1, 1),
          ((4, 384, 63, 63), (256, 384, 3, 3), 1, 1), ((4, 256, 63, 63), (256, 256, 3, 3), 1, 1)]
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for xs, ws, st, pd in shapes:
    for cl in (False, True):
        x = torch.randn(*xs, device=dev); w = torch.randn(*ws, device=dev) * 0.05; b = torch.randn(ws[0], device=dev)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last); w = w.contiguous(memory_format=torch.channels_last)
        f_plain = lambda: torch.relu(torch.nn.functional.conv2d(x, w, b, stride=st, padding=pd))
        f_nobias = lambda: torch.nn.functional.conv2d(x, w, None, stride=st, padding=pd)
        try:
            f_fused = lambda: torch.ops.aten.miopen_convolution_relu(x, w, b, [st, st], [pd, pd], [1, 1], 1)
            y0, y1 = f_plain(), f_fused()
            err = (y0 - y1).abs().max().item()
            tf = timeit(f_fused)
        except Exception as e:
            err, tf = str(e)[:80], float('nan')
        print(xs, 'channels_last' if cl else 'nchw', 'conv+bias+relu %.0f us   conv only %.0f us   fused %.0f us   max diff %s' % (timeit(f_plain), timeit(f_nobias), tf, err))
