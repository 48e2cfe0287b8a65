"""render/render_mask.py of the reference: render.render_mesh plus a `mesh_id` buffer (render_mask.py:201-203,313,391-396,462-463) --
the face label of the covering triangle, composited WITHOUT antialiasing: [label, 1] where the pixel is covered and the label is
non-zero, [0, 0] elsewhere.  render_seq turns it into the cloth / body masks (geometry/hmsdf.py:787-797)."""
import torch

from . import render as _R


def render_mesh(FLAGS, idx, ctx, mesh, mesh_original, mtx_in, view_pos, lgt, resolution, spp=1, num_layers=1, msaa=False, background=None,
                optix_ctx=None, bsdf=None, denoiser=None, shadow_scale=1.0, use_uv=True, finetune_normal=True, extra_dict=None, xfm_lgt=None,
                shade_data=False, buffers=None, _grad_buffers=None):
    if buffers is not None:
        buffers = tuple(b for b in buffers if b != 'mesh_id') + ('_rast',)
    out = _R.render_mesh(FLAGS, idx, ctx, mesh, mesh_original, mtx_in, view_pos, lgt, resolution, spp=spp, num_layers=num_layers, msaa=msaa,
                         background=background, optix_ctx=optix_ctx, bsdf=bsdf, denoiser=denoiser, shadow_scale=shadow_scale, use_uv=use_uv,
                         finetune_normal=finetune_normal, extra_dict=extra_dict, xfm_lgt=xfm_lgt, shade_data=shade_data, buffers=buffers,
                         _keep_rast=True, _grad_buffers=_grad_buffers)
    rast = out.pop('_rast')
    with torch.no_grad():           # labels carry no gradient (interpolating a per-face constant: the three corners are equal)
        tri_id = rast[..., 3].long()                                       # triangle id + 1, 0 = empty
        labels = torch.cat([mesh.face_labels.new_zeros(1), mesh.face_labels]).float()
        lab = labels[tri_id][..., None]                                    # == dr.interpolate(face_labels, rast, (f, f, f))
        on = ((tri_id > 0)[..., None] & (lab != 0)).float()
        out['mesh_id'] = torch.cat((lab, torch.ones_like(lab)), dim=-1) * on
    return out
