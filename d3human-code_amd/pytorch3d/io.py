def load_obj(*a, **k):
    raise NotImplementedError('pytorch3d.io.load_obj is not part of the hot path')
