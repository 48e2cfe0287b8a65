"""Build-time guard of the bf16-MFMA kernels against the packed-f32 co-residency hazard (reproducers:
tools/probe/mfma_pk_hazard.cpp -- self-contained -- and tools/probe/coresidency_repro.cpp -- against this library).

Measured on MI355X (profiles/r5_hazard_*.txt): a wave executing packed-f32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32,
what -O3 SLP-packs scalar f32 arithmetic into) gets wrong values in lanes 48..63 when it shares a SIMD with a wave of another kernel that
is issuing bf16 MFMAs back to back; never when the kernels are stream-ordered.  The library's rule, verified to give 0 wrong launches where
the bare kernels give 10 %: every kernel that issues bf16 MFMAs (i) is allocated all 256 VGPRs with 512-thread workgroups, so that its two
waves per SIMD own the register file and no foreign wave fits beside them, and (ii) passes a workgroup barrier after its last MFMA, so that
no wave of the workgroup leaves its SIMD (making room for a foreign wave) while a sibling still issues matrix instructions.
This test compiles the two sources for gfx950 (no GPU needed) and checks (i) and (ii) in the ISA of every kernel with a bf16 MFMA."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'd3human-code_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def _kernels(asm):
    """{kernel name: (body text, descriptor text)} of a gfx950 .s file"""
    out = {}
    for m in re.finditer(r'^(_Z\w+):\s*; @\1\n(.*?)^\s*\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel', asm, re.S | re.M):
        out[m.group(1)] = (m.group(2), m.group(3))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
@pytest.mark.timeout(900)
@pytest.mark.parametrize('src', ['sdf_mlp_x3.hip', 'sdf_mlp_bwd.hip', 'texmlp.hip'])
def test_every_bf16_mfma_kernel_claims_the_register_file_and_ends_its_matrix_phase_with_a_barrier(src, tmp_path):
    from d3h import build as B
    out = tmp_path / (src + '.s')
    flags = [f for f in B.FLAGS if f not in ('-shared', '-fPIC')]
    subprocess.check_call([HIPCC] + flags + ['--cuda-device-only', '-S', os.path.join(CSRC, src), '-I', CSRC, '-o', str(out)], stderr=subprocess.DEVNULL)
    ks = _kernels(out.read_text())
    assert ks, 'no kernels parsed'
    checked = []
    for name, (body, desc) in ks.items():
        lines = [ln.strip() for ln in body.splitlines()]
        mf = [i for i, ln in enumerate(lines) if re.match(r'v_mfma_f32_\w*(bf16|f16)\b', ln)]       # (fp16 MFMAs -- the "h2" sweeps -- are held to the same rule)
        if not mf:
            continue
        vg = int(re.search(r'\.amdhsa_next_free_vgpr (\d+)', desc).group(1))
        acc = re.search(r'\.amdhsa_accum_offset (\d+)', desc)
        assert vg == 256, f'{name}: bf16 MFMAs but {vg} VGPRs allocated (accum_offset {acc.group(1) if acc else "?"}): foreign waves fit beside it'
        assert any(ln.startswith('s_barrier') for ln in lines[mf[-1]:]), f'{name}: no workgroup barrier after the last bf16 MFMA'
        assert re.search(r'\.amdhsa_private_segment_fixed_size', desc)
        checked.append(name)
    assert checked, f'{src}: expected bf16-MFMA kernels'
    # the host side launches these kernels with 512 threads (two waves per SIMD x 256 VGPRs = the whole file): NTHREADS / launch_bounds(512)
    text = open(os.path.join(CSRC, src)).read()
    assert 'D3H_X3_CLAIM_SIMD()' in text or 'v_mov_b32 v255, 0' in text
    print(src, 'bf16-MFMA kernels checked:', len(checked))
