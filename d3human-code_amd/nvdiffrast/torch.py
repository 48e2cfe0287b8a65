"""Shim under the third-party name so `import nvdiffrast.torch as dr` (train.py:19, render/*.py) resolves to the
MI355X kernels of csrc/raster.hip.  Only the entry points the reference calls are provided."""
from d3h.raster import antialias, interpolate, rasterize as _rasterize, texture  # noqa: F401


class RasterizeGLContext:
    """opaque handle (train.py:1674); the HIP rasterizer needs no GL/CUDA context"""
    def __init__(self, *a, **k):
        pass


RasterizeCudaContext = RasterizeGLContext


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True):
    return _rasterize(pos, tri, resolution)


class DepthPeeler:
    """render/render.py:400-403 uses exactly one layer; the first layer is a plain rasterize"""
    def __init__(self, glctx, pos, tri, resolution):
        self.pos, self.tri, self.res = pos, tri, resolution
        self.layer = 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def rasterize_next_layer(self, want_db=True):
        """want_db False (extension): the caller reads no pixel derivatives; the second result is None"""
        if self.layer > 0:
            raise NotImplementedError('d3h DepthPeeler: only the first layer (the reference asserts num_layers == 1)')
        self.layer += 1
        return _rasterize(self.pos, self.tri, self.res, want_db=want_db)
