"""lap_loss.py of the reference (body_laplacian_loss :40-47, body_normal_loss :50-55, find_edges :57-77, find_connected_faces :79-...)
on the fixed-topology kernels of d3h.meshops."""
from d3h import meshops as _M

find_edges = _M.find_edges
find_connected_faces = _M.find_connected_faces


def body_laplacian_loss(mesh):
    """mean_i |(L V)_i|^2 with the uniform Laplacian of mesh.edges (lap_loss.py:40-47)"""
    return _M.laplacian_loss(mesh.v_pos, mesh.edges)


def body_normal_loss(mesh):
    """lap_loss.py:50-55 -> Mesh.normal_consistency()"""
    return mesh.normal_consistency()
