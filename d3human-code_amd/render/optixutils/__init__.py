"""render.optixutils stand-in.  There is no OptiX on AMD and the reference's ray-traced shading is dead code under the
hard-wired bsdf='kd' (render/render.py:120); the BVH the reference rebuilds 5x per getMesh_* (geometry/hmsdf.py:464-516) is never
read.  The context / build calls are kept as no-ops so geometry/hmsdf.py-style callers work; the shading entry points raise."""

__all__ = ['OptiXContext', 'optix_build_bvh', 'optix_env_shade', 'bilateral_denoiser']


class OptiXContext:
    def __init__(self):
        pass


def optix_build_bvh(optix_ctx, verts, tris, rebuild):
    return None


def optix_env_shade(*a, **k):
    raise NotImplementedError('optix_env_shade: environment-light ray tracing is unreachable (bsdf is forced to "kd", render/render.py:120)')


def bilateral_denoiser(*a, **k):
    raise NotImplementedError('bilateral_denoiser: only reachable from the dead pbr branch (render/render.py:134-136)')
