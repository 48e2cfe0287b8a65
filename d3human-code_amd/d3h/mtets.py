"""Host side of the sort-free marching-tets kernels (csrc/marching_tets.hip).

`TetGrid` holds the static per-grid data (built once): int32 tets, the global sorted-unique edge list
(== HmSDFTetsGeometry.generate_edges, geometry/hmsdf.py:382-388) and the per-tet edge ids.
`marching_tets()` mirrors GShell_Tets.__call__ / hmSDF_Tets.__call__ (geometry/gshell_tets.py:253-447).
"""
import os

import torch
from d3h._lib import cur_stream as _cur_stream

from . import _lib as L

# Speculative extraction (csrc/marching_tets.hip: d3h_mtets_emit_spec): from the second extraction on a grid, the emit kernels are queued
# BEFORE the host reads the output sizes, into buffers at the previous sizes x 1.25 + 256; the three sizes travel to the host on a copy stream
# meanwhile.  The host then only narrows views -- the GPU no longer waits ~130 us per step for the emit launches after the read-back
# (tools/dbg/gpu_host_window.py).  An extraction that outgrows a capacity writes nothing and is repeated at the exact sizes.  '0': always exact.
SPECULATE = os.environ.get('D3H_MTETS_SPECULATE', '1') != '0'
SPEC_STATS = {'speculated': 0, 'overflowed': 0}

_BASE_EDGES = [0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3]


class TetGrid:
    _cache = {}

    def __init__(self, tets):
        # the cache key is the tensor's address: the entry keeps the tensor alive, so that address cannot be handed to another tensor (a
        # freed [nt,4] grid and the next one of the same shape otherwise share a key -- and the stale edge tables: found as a once-in-a-few
        # -runs failure of the shuffled-grid golden, whose test allocates grids of one shape back to back)
        self.src = tets
        t = tets.long()
        e = t[:, _BASE_EDGES].reshape(-1, 2)
        e = torch.sort(e, dim=1)[0]
        uniq, inv = torch.unique(e, dim=0, return_inverse=True)       # init-time only (static grid)
        self.tets32 = t.int().contiguous()
        self.edges32 = uniq.int().contiguous()
        self.all_edges = uniq                                         # int64, as hmsdf.py:387
        self.tet_edge32 = inv.reshape(-1, 6).int().contiguous()
        self.nt = int(t.shape[0])
        self.ne = int(uniq.shape[0])
        dev = tets.device
        nbe, nbt = max(1, (self.ne + 255) // 256), max(1, (self.nt + 255) // 256)
        # per-grid scratch shared by every extraction on this grid: extraction is SINGLE-STREAM by contract (the training loop issues
        # it from the main stream only; _MTetsFn.forward asserts it) -- two overlapping extractions on different streams would race here
        self.tet_code = torch.zeros(max(1, self.nt), dtype=torch.uint8, device=dev)
        self._stream = None
        self.blk_e = torch.zeros(nbe, dtype=torch.int32, device=dev)
        self.blk_t = torch.zeros(nbt * 8, dtype=torch.int32, device=dev)
        self.blk_t2 = torch.zeros(nbt * 8, dtype=torch.int32, device=dev)
        self.edge_vid = torch.zeros(max(1, self.ne), dtype=torch.int32, device=dev)
        self.counts = torch.zeros(16, dtype=torch.int32, device=dev)
        self.caps = {}                 # msdf sign (+1 garment / init, -1 body pass) -> (cap_pwt, cap_n1, cap_n2) from the last extraction
        self._host, self._seq, self._spec_seq = None, 0, None

    def _note_sizes(self, sign, pwt, n1, n2):
        grow = lambda v, hi: min(int(hi), int(v * 1.25) + 256)
        self.caps[sign] = (grow(pwt, self.ne), grow(n1, self.nt), grow(n2, self.nt))

    def _publish_sizes(self):
        """queue, right behind the count kernels, the one-thread kernel that writes (pwt, n1, n2) of this extraction into coherent host memory
        (csrc/marching_tets.hip: mt_publish); -> the sequence number to wait for.  No stream synchronisation: kernels queued after it keep the
        GPU busy while the host wakes."""
        import ctypes
        lib = L.lib()
        if self._host is None:
            hp = ctypes.POINTER(ctypes.c_int)()
            L.check(lib.d3h_host_flags_alloc(ctypes.byref(hp)), 'host_flags_alloc')
            self._host = hp
        self._seq = (self._seq % 1000000007) + 1
        L.check(lib.d3h_mtets_publish_sizes(L.ptr(self.counts), self._host, L.i32(self._seq), L.stream()), 'mtets_publish_sizes')
        return self._seq

    def _wait(self, slot, seq, first, n):
        L.check(L.lib().d3h_host_flag_wait(self._host, L.i32(slot), L.i32(seq), L.i32(60000)), 'host_flag_wait (the GPU did not publish the sizes within 60 s)')
        return [int(self._host[first + k]) for k in range(n)]

    @classmethod
    def get(cls, tets):
        key = (tets.data_ptr(), tuple(tets.shape), str(tets.device), tets._version)
        g = cls._cache.get(key)
        if g is None:
            if len(cls._cache) > 8:
                cls._cache.clear()
            g = cls(tets)
            cls._cache[key] = g
        return g


class _MTetsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, sdf, msdf, grid, msdf_sign, msdf_grad, hook=None):
        lib = L.lib()
        dev = pos.device
        pos = pos.contiguous().float()
        sdf_shape = sdf.shape
        sdf = sdf.reshape(-1).contiguous().float()
        msdf = msdf.contiguous().float()
        g = grid
        if pos.is_cuda and not L.emulated():
            cur = _cur_stream()
            if g._stream is not None and g._stream != cur:
                # the scratch buffers are per grid, not per stream: a caller that moves to another stream first waits for the extraction
                # still in flight on the old one (extractions on two streams must never overlap)
                cur.wait_stream(g._stream)
            g._stream = cur
        L.check(lib.d3h_mtets_count(L.ptr(sdf), L.ptr(g.tets32), L.i32(g.nt), L.ptr(g.edges32), L.i32(g.ne), L.ptr(g.tet_code),
                                    L.ptr(g.blk_e), L.ptr(g.blk_t), L.ptr(g.counts), L.stream()), 'mtets_count')
        # the backward's zero-filled outputs (grid-sized: independent of what is extracted) are allocated and filled NOW, before the host blocks in
        # the read-back: the host is idle here, and after the read-back every launch it has to make delays the first render kernel
        # (tools/dbg/gpu_host_window.py); the backward itself sits on the launch-bound tail of the iteration
        ctx.zeros = (L.zeros_like(pos), L.zeros_like(sdf), L.zeros_like(msdf) if msdf_grad else None) \
            if any(ctx.needs_input_grad[:3]) else None
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        caps = g.caps.get(msdf_sign) if SPECULATE else None
        spec = None
        if caps is not None:
            # ---- speculative: every emit kernel queued now, at the capacities; the sizes arrive on the copy stream meanwhile ------------------
            cw, c1, c2 = caps
            cp, cb = cw + 3 * c1 + 4 * c2, 3 * c1 + 4 * c2
            seq = g._publish_sizes()
            spec = dict(verts_wt=torch.empty(cw, 3, **f32), msdf_vert=torch.empty(cw, **f32), vert_edge=torch.empty(cw, 2, **i32),
                        faces_wt=torch.empty(c1 + 2 * c2, 3, **i32), faces_wt64=torch.empty(c1 + 2 * c2, 3, dtype=torch.int64, device=dev),
                        verts_aug=torch.empty(cp, 3, **f32), msdf_aug=torch.empty(cp, **f32), bnd_edge=torch.empty(cb, 2, **i32),
                        faces_aug=torch.empty(2 * c1 + 4 * c2, 3, **i32), faces_aug64=L.zeros((2 * c1 + 4 * c2, 3), torch.int64, dev),
                        used=torch.empty(max(cp, 1), dtype=torch.uint8, device=dev))
            L.check(lib.d3h_mtets_emit_spec(L.ptr(pos), L.ptr(sdf), L.ptr(msdf), L.f32(msdf_sign), L.ptr(g.edges32), L.i32(g.ne), L.ptr(g.tet_edge32),
                                            L.i32(g.nt), L.ptr(g.tet_code), L.ptr(g.blk_e), L.ptr(g.blk_t), L.ptr(g.blk_t2), L.ptr(g.counts),
                                            L.ptr(g.edge_vid), L.i32(cw), L.i32(c1), L.i32(c2), L.ptr(spec['verts_wt']), L.ptr(spec['msdf_vert']),
                                            L.ptr(spec['vert_edge']), L.ptr(spec['faces_wt']), L.ptr(spec['faces_wt64']), L.ptr(spec['verts_aug']),
                                            L.ptr(spec['msdf_aug']), L.ptr(spec['bnd_edge']), L.ptr(spec['faces_aug']), L.ptr(spec['faces_aug64']),
                                            L.ptr(spec['used']), g._host, L.i32(seq), L.stream()), 'mtets_emit_spec')
            if hook is not None:
                # work that needs the extracted vertices only (or tolerates the zero-padded face list), queued NOW at the capacity with the row
                # count read on the device: the nearest-vertex search, LBS, the surface sampler and the first sweep of the eikonal chain
                # (geometry/hmsdf.py:_extract) -- everything the GPU does until the render is then queued before the host even wakes
                hook.launch(spec['verts_aug'], spec['faces_aug64'], g.counts)
            pwt, n1, n2 = g._wait(7, seq, 0, 3)                # host sync #1 (output sizes): returns when the COUNT kernels are done
            g._spec_seq = seq
            SPEC_STATS['speculated'] += 1
            if pwt > cw or n1 > c1 or n2 > c2:                 # outgrown: the kernels wrote nothing; repeat at the exact sizes below
                SPEC_STATS['overflowed'] += 1
                spec = g._spec_seq = None
        else:
            g._spec_seq = None
            pwt, n1, n2 = g.counts[:3].tolist()                # host sync #1 (output sizes)
        if hook is not None:
            hook.ok = spec is not None and hook.launched
        g._note_sizes(msdf_sign, pwt, n1, n2)
        p = pwt + 3 * n1 + 4 * n2
        fwt = n1 + 2 * n2
        faug = 2 * n1 + 4 * n2
        if spec is not None:
            # the exact-size results are the leading rows of the capacity buffers: views, no copy.  (faces_aug64 was zero-filled at its capacity:
            # the rows between the real cut faces and the bound 2 n1 + 4 n2 are degenerate (0, 0, 0), as on the exact path)
            verts_wt, msdf_vert, vert_edge = spec['verts_wt'][:pwt], spec['msdf_vert'][:pwt], spec['vert_edge'][:pwt]
            faces_wt, faces_wt64 = spec['faces_wt'][:fwt], spec['faces_wt64'][:fwt]
            verts_aug, msdf_aug, bnd_edge = spec['verts_aug'][:p], spec['msdf_aug'][:p], spec['bnd_edge'][:max(p - pwt, 0)]
            faces_aug, faces_aug64, used = spec['faces_aug'][:faug], spec['faces_aug64'][:faug], spec['used'][:max(p, 1)]
        else:
            verts_wt = torch.empty(pwt, 3, **f32)
            msdf_vert = torch.empty(pwt, **f32)
            vert_edge = torch.empty(pwt, 2, **i32)
            faces_wt = torch.empty(fwt, 3, **i32)
            faces_wt64 = torch.empty(fwt, 3, dtype=torch.int64, device=dev)
            L.check(lib.d3h_mtets_emit_wt(L.ptr(pos), L.ptr(sdf), L.ptr(msdf), L.f32(msdf_sign), L.ptr(g.edges32), L.i32(g.ne),
                                          L.ptr(g.tet_edge32), L.i32(g.nt), L.ptr(g.tet_code), L.ptr(g.blk_e), L.ptr(g.blk_t),
                                          L.ptr(g.blk_t2), L.ptr(g.counts), L.ptr(g.edge_vid), L.ptr(verts_wt), L.ptr(msdf_vert),
                                          L.ptr(vert_edge), L.ptr(faces_wt), L.ptr(faces_wt64), L.stream()), 'mtets_emit_wt')
            # The number of cut faces (six group counts, still on the device) only sizes the face list: a 1-triangle tet yields at most 2 of
            # them, a 2-triangle tet at most 4, so the list is allocated at that bound and narrowed by marching_tets() AFTER the caller had the
            # chance to queue the work that needs vertices only (nearest SMPL-X vertex + LBS of both meshes): host sync #2 then waits behind
            # ~200 us of queued kernels instead of opening a bubble.
            verts_aug = torch.empty(p, 3, **f32)
            msdf_aug = torch.empty(p, **f32)
            bnd_edge = torch.empty(max(p - pwt, 0), 2, **i32)
            faces_aug = torch.empty(faug, 3, **i32)
            # (zero-filled, not empty: the rows beyond the actual count are then DEGENERATE faces (0, 0, 0) of zero area -- consumers that tolerate
            # those, like the surface sampler, may use the padded list before host sync #2 tells how many rows are real)
            faces_aug64 = L.zeros((faug, 3), torch.int64, dev)
            used = torch.empty(max(p, 1), dtype=torch.uint8, device=dev)
            L.check(lib.d3h_mtets_emit_aug(L.ptr(g.tet_edge32), L.i32(g.nt), L.ptr(g.tet_code), L.ptr(g.blk_t), L.ptr(g.blk_t2),
                                           L.ptr(g.counts), L.ptr(g.edge_vid), L.ptr(verts_wt), L.ptr(msdf_vert), L.i32(pwt), L.i32(p),
                                           L.ptr(verts_aug), L.ptr(msdf_aug), L.ptr(bnd_edge), L.ptr(faces_aug), L.ptr(faces_aug64),
                                           L.ptr(used), L.stream()), 'mtets_emit_aug')
        ctx.save_for_backward(pos, sdf, msdf, verts_wt, msdf_vert, vert_edge, bnd_edge, used)
        ctx.set_materialize_grads(False)       # five index outputs + whichever of the three value outputs nobody differentiates: None, not zero fills
        ctx.meta = (pwt, p, msdf_sign, msdf_grad)
        ctx.sdf_shape = sdf_shape
        for t in (faces_aug64, faces_wt64, faces_aug, faces_wt, bnd_edge):
            ctx.mark_non_differentiable(t)
        return verts_aug, msdf_aug, verts_wt, faces_aug64, faces_wt64, faces_aug, faces_wt, bnd_edge

    @staticmethod
    def backward(ctx, g_verts, g_msdf, g_wt, *_):
        pos, sdf, msdf, verts_wt, msdf_vert, vert_edge, bnd_edge, used = ctx.saved_tensors
        pwt, p, msdf_sign, msdf_grad = ctx.meta
        lib = L.lib()
        z, ctx.zeros = getattr(ctx, 'zeros', None), None             # (one use: a second backward through a retained graph fills its own)
        d_pos, d_sdf, d_msdf = z if z is not None else (L.zeros_like(pos), L.zeros_like(sdf), L.zeros_like(msdf) if msdf_grad else None)
        if pwt > 0:
            scratch = torch.empty(5 * pwt, dtype=torch.float32, device=pos.device)
            c = lambda t: None if t is None else t.contiguous()
            L.check(lib.d3h_mtets_bwd(L.ptr(c(g_verts)), L.ptr(c(g_msdf)), L.ptr(c(g_wt)), L.ptr(used), L.ptr(bnd_edge), L.ptr(vert_edge),
                                      L.ptr(verts_wt), L.ptr(msdf_vert), L.ptr(pos), L.ptr(sdf), L.ptr(msdf), L.f32(msdf_sign),
                                      L.i32(pwt), L.i32(p), L.ptr(scratch), L.ptr(d_pos), L.ptr(d_sdf), L.ptr(d_msdf), L.stream()),
                    'mtets_bwd')
        return d_pos, d_sdf.reshape(ctx.sdf_shape), d_msdf, None, None, None, None


def marching_tets(pos, sdf, msdf, tets, body=False, before_face_sync=None, spec_hook=None):
    """-> dict(verts, faces, verts_wt, faces_wt, msdf, n_wt, faces32, faces_wt32).  body=True is hmSDF_Tets(type='body').
    before_face_sync(verts, verts_wt, faces_padded): called once every kernel of the extraction is queued and before the host reads the
    cut-face count (see _MTetsFn.forward) -- the place to queue work that depends on the vertices only, or that tolerates the face list at
    its allocation bound (`faces_padded` int64: the real faces followed by zero-area (0, 0, 0) rows).
    spec_hook: an object with .launch(verts_cap, faces_cap, counts) / .launched / .ok -- when the extraction runs speculatively, launch() is
    called with the vertex and face buffers AT THEIR CAPACITY and the device-side counters (rows = counts[0] + 3 counts[1] + 4 counts[2])
    before the host knows the sizes; .ok tells afterwards whether what it queued is valid (False: not speculative, or the capacity was outgrown)."""
    grid = TetGrid.get(tets)
    sign = -1.0 if body else 1.0
    # hmsdf_tets_split.py:261-264 negates msdf under no_grad: the body pass sends no gradient to msdf
    verts, msdf_aug, verts_wt, faces, faces_wt, faces32, faces_wt32, bnd_edge = _MTetsFn.apply(pos, sdf, msdf, grid, sign, not body, spec_hook)
    if before_face_sync is not None:
        before_face_sync(verts, verts_wt, faces)
    # host sync #2 (cut-face count): published to host memory behind the emit kernels by the speculative path, else read back through the stream
    c = grid._wait(15, grid._spec_seq, 8, 6) if grid._spec_seq is not None else grid.counts[3:9].tolist()
    faug = c[0] + 2 * c[1] + c[2] + 2 * c[3] + 3 * c[4] + 4 * c[5]
    faces, faces32 = faces[:faug], faces32[:faug]
    return {'verts': verts, 'faces': faces, 'verts_wt': verts_wt, 'faces_wt': faces_wt, 'msdf': msdf_aug,
            'n_wt': verts_wt.shape[0], 'faces32': faces32, 'faces_wt32': faces_wt32, 'bnd_edge': bnd_edge}
