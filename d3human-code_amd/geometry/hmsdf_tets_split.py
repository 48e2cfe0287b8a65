"""hmSDF_Tets with the reference's call signature (geometry/hmsdf_tets_split.py:254-454): the same marching tets with mSDF negated
(under no_grad, :261-264) for type == "body"."""
from .gshell_tets import GShell_Tets


class hmSDF_Tets(GShell_Tets):
    def __call__(self, pos_nx3, sdf_n, msdf_n, tet_fx4, type, output_watertight_template=True, _before_face_sync=None, _spec_hook=None):
        if type not in ('cloth', 'body'):
            raise ValueError(f'hmSDF_Tets: type must be "cloth" or "body", got {type!r}')
        return super().__call__(pos_nx3, sdf_n, msdf_n, tet_fx4, output_watertight_template, _body=(type == 'body'),
                                _before_face_sync=_before_face_sync, _spec_hook=_spec_hook)
