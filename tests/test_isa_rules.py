"""ISA-level rules of the two bf16-MFMA kernels (K20 split linear, K1 split window attention), checked on the
cross-compiled gfx950 assembly (no GPU needed).

Measured on MI355X (tools/experiments/pk_mfma_probe.hip, profiles/r03_pk_mfma_probe.txt): while a wave that mixes bf16
MFMAs with LDS traffic (LDS-DMA loads in K20, ds reads in K1) is resident, v_pk_fma_f32 instructions with an SGPR source
executed by OTHER waves of the same SIMD -- another kernel's included -- return wrong low halves in lanes 48..63.  Both
kernels therefore (a) claim the whole register file of their SIMDs, so that no other kernel's wave is ever resident
beside them, and (b) keep their own packed f32 arithmetic on VGPR operands.  Both are properties of the generated code,
not of the source, so they are tested on the assembly."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _asm(tmp_path_factory, source):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / (source + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    "--cuda-device-only", "-S", "-o", str(out),
                    os.path.join(ROOT, "neurips2023_soc_amd", "csrc", source)], check=True)
    return out.read_text()


@pytest.fixture(scope="module")
def k20_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "linear_split.hip")


@pytest.fixture(scope="module")
def k1_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "win_attn3d.hip")


def _kernel_bodies(asm, needle):
    # "<symbol>:" ... "s_endpgm"
    for m in re.finditer(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)s_endpgm" % needle, asm, re.S | re.M):
        yield m.group(1), m.group(2)


def test_k20_keeps_packed_f32_math_off_sgpr_operands(k20_asm):
    bodies = list(_kernel_bodies(k20_asm, "linear_split_kernel"))
    assert len(bodies) == 5, [b[0] for b in bodies]
    for name, body in bodies:
        assert "global_load_lds_dwordx4" in body and "v_mfma_f32_32x32x16_bf16" in body, name
        bad = [ln.strip() for ln in body.splitlines()
               if re.search(r"\bv_pk_(fma|mul|add)_f32\b", ln) and re.search(r"[ ,]s\[\d+:\d+\]", ln)]
        assert not bad, (name, bad[:4])


def test_k20_claims_the_whole_register_file(k20_asm):
    counts = re.findall(r"\.name:\s+(_Z\w*linear_split_kernel\w*)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", k20_asm)
    assert len(counts) == 5, counts
    for name, vgprs in counts:      # 2 waves per SIMD x 256 = the 512-entry file: nothing else fits on the CU
        assert int(vgprs) == 256, (name, vgprs)
    assert not re.search(r"linear_split_kernel\w*\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+[1-9]", k20_asm)


def test_k1_split_follows_the_same_rules(k1_asm):
    bodies = list(_kernel_bodies(k1_asm, "win_attn3d_split_kernel"))
    assert len(bodies) == 2, [b[0] for b in bodies]
    for name, body in bodies:
        assert "v_mfma_f32_16x16x32_bf16" in body and "ds_read_b64_tr_b16" in body, name
        bad = [ln.strip() for ln in body.splitlines()
               if re.search(r"\bv_pk_(fma|mul|add)_f32\b", ln) and re.search(r"[ ,]s\[\d+:\d+\]", ln)]
        assert not bad, (name, bad[:4])
    counts = re.findall(r"\.name:\s+(_Z\w*win_attn3d_split_kernel\w*)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", k1_asm)
    assert len(counts) == 2 and all(int(v) == 256 for _, v in counts), counts


@pytest.fixture(scope="module")
def k13b_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "ws_linear_split.hip")


def test_k13b_follows_the_same_rules(k13b_asm):
    bodies = list(_kernel_bodies(k13b_asm, "ws_linear_split_kernel"))
    assert len(bodies) >= 20, len(bodies)
    for name, body in bodies:
        assert "v_mfma_f32_16x16x32_bf16" in body and "s_barrier" in body, name
        bad = [ln.strip() for ln in body.splitlines()
               if re.search(r"\bv_pk_(fma|mul|add)_f32\b", ln) and re.search(r"[ ,]s\[\d+:\d+\]", ln)]
        assert not bad, (name, bad[:4])
    counts = re.findall(r"\.name:\s+(_Z\w*ws_linear_split_kernel\w*)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", k13b_asm)
    assert len(counts) == len(bodies) and all(int(v) == 256 for _, v in counts), counts[:3]
    spills = re.findall(r"\.name:\s+_Z\w*ws_linear_split_kernel\w*\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", k13b_asm)
    assert spills and all(int(v) == 0 for v in spills), spills


@pytest.fixture(scope="module")
def k22_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "ffn_split.hip")


def test_k22_follows_the_same_rules(k22_asm):
    bodies = list(_kernel_bodies(k22_asm, "ffn_split_kernel"))
    assert len(bodies) == 1
    name, body = bodies[0]
    assert "v_mfma_f32_16x16x32_bf16" in body and "global_load_lds_dwordx4" in body and "s_barrier" in body
    bad = [ln.strip() for ln in body.splitlines()
           if re.search(r"\bv_pk_(fma|mul|add)_f32\b", ln) and re.search(r"[ ,]s\[\d+:\d+\]", ln)]
    assert not bad, bad[:4]
    counts = re.findall(r"\.name:\s+(_Z\w*ffn_split_kernel\w*)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", k22_asm)
    assert counts and all(int(v) == 256 for _, v in counts), counts
