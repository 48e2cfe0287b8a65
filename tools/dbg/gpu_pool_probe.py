"""isolated cost of torch's max_pool2d (3x3, stride 2) forward / backward on the two AlexNet-LPIPS shapes, channels-last"""
import torch, time
dev = 'cuda'
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for shp in ((4, 64, 255, 255), (4, 192, 127, 127)):
    for cl in (True, False):
        x = torch.randn(*shp, device=dev)
        if cl: x = x.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        y = torch.nn.functional.max_pool2d(x, 3, 2)
        g = torch.randn_like(y)
        tf = timeit(lambda: torch.nn.functional.max_pool2d(x, 3, 2))
        tb = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
        tr = timeit(lambda: torch.relu(x))
        print(shp, 'channels_last' if cl else 'nchw', 'pool fwd %.0f us  bwd %.0f us   relu %.0f us   (tensor %.0f MB)' % (tf, tb, tr, x.numel() * 4 / 1e6))
