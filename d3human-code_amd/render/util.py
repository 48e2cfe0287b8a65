"""Math / image helpers with the names and behaviour of the reference's render/util.py (:19-530), so that render/*.py of this build AND
the reference's own train.py / light.py / material.py / texture.py / denoiser (which import `render.util` and find this module first on
the path) keep working.  The hot path uses dot / safe_normalize / pixel_grid / rgb_to_srgb / scale_img_nhwc / avg_pool_nhwc; the rest is
start-up and logging surface.  `device=None` arguments default to the GPU when there is one (the reference hard-codes 'cuda')."""
import os

import numpy as np
import torch


def dot(x, y):
    return torch.sum(x * y, -1, keepdim=True)


def reflect(x, n):
    return 2 * dot(x, n) * n - x


def length(x, eps=1e-20):
    return torch.sqrt(torch.clamp(dot(x, x), min=eps))


def safe_normalize(x, eps=1e-20):
    return x / length(x, eps)


def to_hvec(x, w):
    return torch.nn.functional.pad(x, pad=(0, 1), mode='constant', value=w)


_grid_cache = {}


def pixel_grid(width, height, center_x=0.5, center_y=0.5, device=None):
    """util.py:61-66: [H,W,2] of ((x + cx)/W, (y + cy)/H)"""
    device = device or ('cuda' if torch.cuda.is_available() else 'cpu')
    key = (width, height, center_x, center_y, str(device))
    g = _grid_cache.get(key)
    if g is None:
        y, x = torch.meshgrid((torch.arange(0, height, dtype=torch.float32, device=device) + center_y) / height,
                              (torch.arange(0, width, dtype=torch.float32, device=device) + center_x) / width, indexing='ij')
        g = torch.stack((x, y), dim=-1)
        _grid_cache[key] = g
    return g


def _rgb_to_srgb(f):
    return torch.where(f <= 0.0031308, f * 12.92, torch.pow(torch.clamp(f, 0.0031308), 1.0 / 2.4) * 1.055 - 0.055)


def rgb_to_srgb(f):
    return torch.cat((_rgb_to_srgb(f[..., 0:3]), f[..., 3:4]), dim=-1) if f.shape[-1] == 4 else _rgb_to_srgb(f)


def _srgb_to_rgb(f):
    return torch.where(f <= 0.04045, f / 12.92, torch.pow((torch.clamp(f, 0.04045) + 0.055) / 1.055, 2.4))


def srgb_to_rgb(f):
    return torch.cat((_srgb_to_rgb(f[..., 0:3]), f[..., 3:4]), dim=-1) if f.shape[-1] == 4 else _srgb_to_rgb(f)


def scale_img_nhwc(x, size, mag='bilinear', min='area'):
    y = x.permute(0, 3, 1, 2)
    if x.shape[1] > size[0] and x.shape[2] > size[1]:
        y = torch.nn.functional.interpolate(y, size, mode=min)
    elif mag in ('bilinear', 'bicubic'):
        y = torch.nn.functional.interpolate(y, size, mode=mag, align_corners=True)
    else:
        y = torch.nn.functional.interpolate(y, size, mode=mag)
    return y.permute(0, 2, 3, 1).contiguous()


def scale_img_hwc(x, size, mag='bilinear', min='area'):
    return scale_img_nhwc(x[None, ...], size, mag, min)[0]


def avg_pool_nhwc(x, size):
    return torch.nn.functional.avg_pool2d(x.permute(0, 3, 1, 2), size).permute(0, 2, 3, 1).contiguous()


def perspective(fovy=0.7854, aspect=1.0, n=0.1, f=1000.0, device=None):
    y = np.tan(fovy / 2)
    return torch.tensor([[1 / (y * aspect), 0, 0, 0], [0, 1 / -y, 0, 0], [0, 0, -(f + n) / (f - n), -(2 * f * n) / (f - n)], [0, 0, -1, 0]],
                        dtype=torch.float32, device=device)


def translate(x, y, z, device=None):
    return torch.tensor([[1, 0, 0, x], [0, 1, 0, y], [0, 0, 1, z], [0, 0, 0, 1]], dtype=torch.float32, device=device)


def _plane_rotation(i, j, a, device):
    """4x4 rotation by angle a in the (i, j) coordinate plane, i < j, with the sign convention of the reference's rotate_* helpers
    (m[i][j] = +sin a)"""
    m = torch.eye(4, dtype=torch.float32, device=device)
    m[i, i] = m[j, j] = float(np.cos(a))
    m[i, j], m[j, i] = float(np.sin(a)), -float(np.sin(a))
    return m


def rotate_x(a, device=None):
    return _plane_rotation(1, 2, a, device)


def rotate_y(a, device=None):
    return _plane_rotation(0, 2, a, device)


def rotate_z(a, device=None):
    return _plane_rotation(0, 1, a, device)


def scale(s, device=None):
    return torch.diag(torch.tensor([s, s, s, 1.0], dtype=torch.float32, device=device))


# ---- the rest of the reference's util.py surface (render/util.py:34-60,70-93,112-190,216-240,250-272,307-420,440-530): colour
# helpers, dilation, cube maps, camera / sampling helpers, image IO.  None of it is on the per-iteration path; it exists so that
# train.py, light.py, material.py, texture.py and the denoiser import and run against this module unchanged. --------------------------
def ycocg2rgb(ycocg):
    y, co, cg = ycocg.unbind(-1)
    return torch.stack((y + co - cg, y + cg, y - co - cg), dim=-1)


def hsv2rgb(image):
    """h, s, v in [0, 1] -> rgb (the sextant formula: channel n has k = (n + 6 h) mod 6, value v - v s clamp(min(k, 4 - k), 0, 1))"""
    h, s, v = image[..., 0:1], image[..., 1:2], image[..., 2:3]
    k = (torch.tensor([5.0, 3.0, 1.0], dtype=image.dtype, device=image.device) + h * 6) % 6
    return v - v * s * torch.clamp(torch.minimum(k, 4 - k), 0, 1)


def dilate(x, x_avg, mask, N):
    """fill the pixels outside `mask` ([B,H,W,1]) of x ([B,H,W,C]) with the mask-normalised N x N Gaussian average of the pixels
    inside it (x_avg where no masked pixel is in reach); pixels inside the mask are returned unchanged"""
    C = x.shape[3]
    lin = torch.linspace(-1, 1, N, dtype=torch.float32, device=x.device)
    var = (1.0 / 2.5) ** 2
    k = torch.exp(-(lin[:, None] ** 2 + lin[None, :] ** 2) / (2 * var))
    k = (k / k.sum())[None, None]
    m = mask.permute(0, 3, 1, 2)
    cover = torch.nn.functional.conv2d(m, k, padding=N // 2).permute(0, 2, 3, 1)
    blur = torch.nn.functional.conv2d((x * mask).permute(0, 3, 1, 2), k.expand(C, 1, N, N), padding=N // 2, groups=C).permute(0, 2, 3, 1)
    eps = 1e-6
    fill = torch.where(cover > eps, blur / torch.clamp(cover, min=eps), x_avg)
    return fill * (1 - mask) + x * mask


def reinhard(f):
    return f / (1 + f)


def mse_to_psnr(mse):
    return -10.0 * np.log10(mse)


def psnr_to_mse(psnr):
    return 10.0 ** (-0.1 * psnr)


def get_miplevels(texture):
    return np.floor(np.log2(min(texture.shape[0], texture.shape[1])))


def tex_2d(tex_map, coords, filter='nearest'):
    """tex_map [H,W,C], coords [N,2] in [0,1] -> [N,C]"""
    t = torch.nn.functional.grid_sample(tex_map[None].permute(0, 3, 1, 2), coords[None, None, ...] * 2 - 1, mode=filter, align_corners=False)
    return t.permute(0, 2, 3, 1)[0, 0]


_CUBE_FACES = ((('one', 1), ('y', -1), ('x', -1)), (('one', -1), ('y', -1), ('x', 1)), (('x', 1), ('one', 1), ('y', 1)),
               (('x', 1), ('one', -1), ('y', -1)), (('x', 1), ('y', -1), ('one', 1)), (('x', -1), ('y', -1), ('one', -1)))


def cube_to_dir(s, x, y):
    """direction through texel (x, y) in [-1, 1]^2 of cube face s (+x, -x, +y, -y, +z, -z)"""
    src = {'x': x, 'y': y, 'one': torch.ones_like(x)}
    return torch.stack([src[n] * sg for n, sg in _CUBE_FACES[s]], dim=-1)


def latlong_to_cubemap(latlong_map, res):
    import nvdiffrast.torch as dr
    dev = latlong_map.device
    faces = []
    for s in range(6):
        gy, gx = torch.meshgrid(torch.linspace(-1.0 + 1.0 / res[0], 1.0 - 1.0 / res[0], res[0], device=dev),
                                torch.linspace(-1.0 + 1.0 / res[1], 1.0 - 1.0 / res[1], res[1], device=dev), indexing='ij')
        v = safe_normalize(cube_to_dir(s, gx, gy))
        tu = torch.atan2(v[..., 0:1], -v[..., 2:3]) / (2 * np.pi) + 0.5
        tv = torch.acos(torch.clamp(v[..., 1:2], min=-1, max=1)) / np.pi
        faces.append(dr.texture(latlong_map[None, ...], torch.cat((tu, tv), dim=-1)[None, ...].contiguous(), filter_mode='linear')[0])
    return torch.stack(faces)


def cubemap_to_latlong(cubemap, res):
    import nvdiffrast.torch as dr
    dev = cubemap.device
    gy, gx = torch.meshgrid(torch.linspace(0.0 + 1.0 / res[0], 1.0 - 1.0 / res[0], res[0], device=dev),
                            torch.linspace(-1.0 + 1.0 / res[1], 1.0 - 1.0 / res[1], res[1], device=dev), indexing='ij')
    st, ct, sp, cp = torch.sin(gy * np.pi), torch.cos(gy * np.pi), torch.sin(gx * np.pi), torch.cos(gx * np.pi)
    refl = torch.stack((st * sp, ct, -st * cp), dim=-1)
    return dr.texture(cubemap[None, ...], refl[None, ...].contiguous(), filter_mode='linear', boundary_mode='cube')[0]


def segment_sum(data, segment_ids):
    """tf.segment_sum: rows of `data` with equal (sorted) segment id summed"""
    n = torch.unique_consecutive(segment_ids).shape[0]
    out = torch.zeros(n, *data.shape[1:], dtype=torch.float32, device=data.device)
    return out.index_add(0, segment_ids.reshape(-1) if segment_ids.dim() == 1 else segment_ids[(slice(None),) + (0,) * (data.dim() - 1)], data)


def fovx_to_fovy(fovx, aspect):
    return np.arctan(np.tan(fovx / 2) / aspect) * 2.0


def focal_length_to_fovy(focal_length, sensor_height):
    return 2 * np.arctan(0.5 * sensor_height / focal_length)


def perspective_offcenter(fovy, fraction, rx, ry, aspect=1.0, n=0.1, f=1000.0, device=None):
    """projection of the sub-frustum that starts at (rx, ry) of the full one and spans `fraction` of it"""
    y = np.tan(fovy / 2)
    w, h = 2 * aspect * y, 2 * y
    l, b = -aspect * y + w * rx, -y + h * ry
    r, t = l + w * fraction, b + h * fraction
    return torch.tensor([[2 / (r - l), 0, (r + l) / (r - l), 0], [0, -2 / (t - b), (t + b) / (t - b), 0],
                         [0, 0, -(f + n) / (f - n), -(2 * f * n) / (f - n)], [0, 0, -1, 0]], dtype=torch.float32, device=device)


def lookAt(eye, at, up):
    w = torch.nn.functional.normalize(eye - at, dim=0)
    u = torch.nn.functional.normalize(torch.linalg.cross(up, w), dim=0)
    v = torch.linalg.cross(w, u)
    m = torch.eye(4, dtype=eye.dtype, device=eye.device)
    m[:3, :3] = torch.stack((u, v, w))
    m[:3, 3] = -(m[:3, :3] @ eye)
    return m


def _random_frame():
    m = np.random.normal(size=[3, 3])
    m[1] = np.cross(m[0], m[2])
    m[2] = np.cross(m[0], m[1])
    out = np.eye(4)
    out[:3, :3] = m / np.linalg.norm(m, axis=1, keepdims=True)
    return out


@torch.no_grad()
def random_rotation_translation(t, device=None):
    m = _random_frame()
    m[:3, 3] = np.random.uniform(-t, t, size=[3])
    return torch.tensor(m, dtype=torch.float32, device=device)


@torch.no_grad()
def random_rotation(device=None):
    return torch.tensor(_random_frame(), dtype=torch.float32, device=device)


def lines_focal(o, d):
    """least-squares point closest to the lines o_i + t d_i"""
    d = safe_normalize(d)
    P = d[..., :, None] * d[..., None, :] - torch.eye(3, dtype=o.dtype, device=o.device)
    return torch.linalg.pinv(P.sum(0)) @ (P @ o[..., None]).sum(0).squeeze(1)


@torch.no_grad()
def cosine_sample(N, size=None):
    """cosine-weighted direction(s) around the normal N"""
    N = N / torch.linalg.norm(N)
    a = torch.stack((torch.zeros_like(N[0]), N[2], -N[1]))
    b = torch.stack((-N[2], torch.zeros_like(N[0]), N[0]))
    dx = torch.where(dot(a, a) > dot(b, b), a, b)
    dx = dx / torch.linalg.norm(dx)
    dy = torch.linalg.cross(N, dx)
    dy = dy / torch.linalg.norm(dy)
    if size is None:
        phi, s = 2.0 * np.pi * np.random.uniform(), np.random.uniform()
        return dx * (np.cos(phi) * np.sqrt(1.0 - s)) + dy * (np.sin(phi) * np.sqrt(1.0 - s)) + N * np.sqrt(s)
    phi = 2.0 * np.pi * torch.rand(*size, 1, dtype=N.dtype, device=N.device)
    s = torch.rand(*size, 1, dtype=N.dtype, device=N.device)
    return dx * (torch.cos(phi) * torch.sqrt(1.0 - s)) + dy * (torch.sin(phi) * torch.sqrt(1.0 - s)) + N * torch.sqrt(s)


def bilinear_downsample(x, spp):
    """log2(spp) halvings of an NHWC image with the separable (1 3 3 1)/8 kernel, edge-replicated"""
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=torch.float32, device=x.device) / 8.0
    g = x.shape[-1]
    w = (k1[:, None] * k1[None, :]).expand(g, 1, 4, 4)
    x = x.permute(0, 3, 1, 2)
    for _ in range(int(np.log2(spp))):
        x = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode='replicate'), w, stride=2, groups=g)
    return x.permute(0, 2, 3, 1).contiguous()


_display_warned = False


def display_image(image, title=None):
    """the reference opens a GLFW window (util.py:440-480); on a headless MI355X node there is no display: returns False once GLFW /
    PyOpenGL are missing (the training loop only uses the return value to stop early)"""
    global _display_warned
    try:
        import glfw  # noqa: F401
        import OpenGL.GL  # noqa: F401
    except Exception:
        if not _display_warned:
            print('display_image: no GLFW / OpenGL in this environment; images are only saved')
            _display_warned = True
        return False
    raise NotImplementedError('display_image: interactive display is outside this build (use save_image)')


def _png_bytes(a):
    import struct
    import zlib
    h, w, c = a.shape
    raw = b''.join(b'\x00' + a[r].tobytes() for r in range(h))
    chunk = lambda tag, data: struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)
    ctype = {1: 0, 3: 2, 4: 6}[c]
    return b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, ctype, 0, 0, 0)) + chunk(b'IDAT', zlib.compress(raw, 3)) + chunk(b'IEND', b'')


def save_image_raw(fn, x):
    try:
        try:
            import imageio
            imageio.imwrite(fn, x)
        except ImportError:
            a = np.asarray(x)
            if a.ndim == 2:
                a = a[..., None]
            if os.path.splitext(fn)[1].lower() == '.png' and a.dtype == np.uint8 and a.shape[-1] in (1, 3, 4):
                with open(fn, 'wb') as f:
                    f.write(_png_bytes(np.ascontiguousarray(a)))
            else:
                np.save(fn + '.npy', a)           # float / HDR data without an image library: raw array next to the requested name
    except Exception as e:
        print('WARNING: FAILED to save image %s (%s)' % (fn, e))


def save_image(fn, x):
    """x float in [0, 1] (HWC) -> 8-bit image file"""
    save_image_raw(fn, np.clip(np.rint(np.asarray(x) * 255.0), 0, 255).astype(np.uint8))


def load_image_raw(fn):
    if fn.endswith('.npy'):
        return np.load(fn)
    import imageio
    return imageio.imread(fn)


def load_image(fn):
    img = load_image_raw(fn)
    return img if img.dtype == np.float32 else img.astype(np.float32) / 255


def time_to_text(x):
    for limit, div, unit in ((3600, 3600, 'h'), (60, 60, 'm')):
        if x > limit:
            return '%.2f %s' % (x / div, unit)
    return '%.2f s' % x


def checkerboard(res, checker_size):
    """[H,W,3] grey checkerboard (0.66 / 0.33), top-left tile bright"""
    yy, xx = np.meshgrid(np.arange(res[0]) // checker_size, np.arange(res[1]) // checker_size, indexing='ij')
    check = ((yy + xx) % 2 == 0) * 0.33 + 0.33
    return np.stack((check,) * 3, axis=-1)
