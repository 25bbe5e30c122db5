"""ISA-level rules of EVERY kernel that issues bf16 MFMAs (K20 split linear, K1 split window attention, K13b, K23, and
whatever comes next), checked on the cross-compiled gfx950 assembly (no GPU needed).  The source files are DISCOVERED -- any
csrc/*.hip that names a bf16 MFMA builtin -- so a new kernel cannot skip the lint by not being listed here.

Measured on MI355X (tools/experiments/pk_mfma_probe.hip, profiles/r03_pk_mfma_probe.txt): while a wave that mixes bf16
MFMAs with LDS traffic (LDS-DMA loads in K20 / K23, ds reads in K1 / K13b) is resident, v_pk_fma_f32 instructions with
an SGPR source executed by OTHER waves of the same SIMD -- another kernel's included -- return wrong low halves in lanes
48..63.  These kernels therefore (a) claim the whole register file of their SIMDs, so that no other kernel's wave is ever
resident beside them, (b) keep their own packed f32 arithmetic on VGPR operands, (c) retire their waves behind a barrier.
All three are properties of the generated code, not of the source, so they are tested on the assembly."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neurips2023_soc_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
BF16_MFMA_SRC = re.compile(r"__builtin_amdgcn_mfma_f32_\d+x\d+x\d+_?bf16|mfma6\(")
BF16_MFMA_ISA = re.compile(r"\bv_mfma_f32_\d+x\d+x\d+_bf16\b")


def _text_with_local_includes(path, seen=None):
    """The text of a source file followed by that of the csrc/ headers it includes (kernel templates live in headers when
    two translation units share them: xs_linear_split.h)."""
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return ""
    seen.add(path)
    with open(path) as f:
        text = f.read()
    for inc in re.findall(r'#include\s+"([^"]+)"', text):
        text += _text_with_local_includes(os.path.join(CSRC, inc), seen)
    return text


def bf16_mfma_sources():
    """csrc/*.hip that issue bf16 MFMAs: directly, through split_math.h's mfma6(), or through a kernel header they include."""
    out = []
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        own = open(path).read()
        body = own + "".join(_text_with_local_includes(os.path.join(CSRC, inc)) for inc in re.findall(r'#include\s+"([^"]+)"', own)
                             if inc != "split_math.h")
        if BF16_MFMA_SRC.search(body):
            out.append(os.path.basename(path))
    return out


SOURCES = bf16_mfma_sources()


def test_every_known_bf16_mfma_kernel_file_is_discovered():
    assert {"linear_split.hip", "win_attn3d.hip", "ws_linear_split.hip", "mlp_split.hip", "xs_linear_split.hip",
            "xs_linear_split_wide.hip"} <= set(SOURCES), SOURCES


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    """gfx950 assembly of every discovered source, compiled side by side (the template-heavy files take over a minute each)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    from concurrent.futures import ThreadPoolExecutor
    outdir = tmp_path_factory.mktemp("isa")

    def compile_one(src):
        out = outdir / (src + ".s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                        "--cuda-device-only", "-S", "-o", str(out), os.path.join(CSRC, src)], check=True)
        return src, out

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES), len(os.sched_getaffinity(0)))) as pool:
        return dict(pool.map(compile_one, SOURCES))


@pytest.fixture(scope="module", params=SOURCES)
def unit(request, assembly):
    src = request.param
    out = assembly[src]
    asm = out.read_text()
    # a kernel body runs to its .Lfunc_end label, not to the first s_endpgm: K24's surplus workgroups (grid padded to whole
    # groups of eight row groups) return in front of everything else
    bodies = {m.group(1): m.group(2) for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M)}
    meta = {}
    for m in re.finditer(r"- \.agpr_count:\s+(\d+)\n(.*?)\.wavefront_size:", asm, re.S):
        block = m.group(0)
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        meta[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
                      for k in ("agpr_count", "vgpr_count", "max_flat_workgroup_size", "private_segment_fixed_size")}
    kernels = {n: b for n, b in bodies.items() if BF16_MFMA_ISA.search(b)}
    assert kernels, f"{src}: no kernel with a bf16 MFMA found in the assembly"
    assert set(kernels) <= set(meta), sorted(set(kernels) - set(meta))[:3]
    return src, kernels, meta


def test_packed_f32_math_stays_off_sgpr_operands(unit):
    src, kernels, _ = unit
    for name, body in kernels.items():
        bad = [ln.strip() for ln in body.splitlines()
               if re.search(r"\bv_pk_(fma|mul|add)_f32\b", ln) and re.search(r"[ ,]s\[\d+:\d+\]", ln)]
        assert not bad, (src, name, bad[:4])


def test_the_whole_register_file_is_claimed(unit):
    """2 waves per SIMD x 256 registers or 1 wave x 512 = the 512-entry file: nothing else fits on the CU (`.vgpr_count` of the
    code object metadata is the unified allocation, accumulation registers included)."""
    src, kernels, meta = unit
    for name in kernels:
        m = meta[name]
        waves_per_simd = max(1, m["max_flat_workgroup_size"] // 256)
        assert m["vgpr_count"] * waves_per_simd == 512, (src, name, m)


def test_the_waves_retire_behind_a_barrier(unit):
    src, kernels, _ = unit
    for name, body in kernels.items():
        # every exit that can follow an MFMA: from each s_endpgm walk back to the nearest s_barrier -- no MFMA in between (the
        # compiler places out-of-line blocks behind the last s_endpgm in the TEXT, so "the text behind the last barrier" is not
        # the path to an exit; an s_endpgm in front of the first MFMA -- K24's surplus workgroups -- has issued none)
        assert "s_barrier" in body, (src, name)
        first_mfma = BF16_MFMA_ISA.search(body).start()
        exits = [m.start() for m in re.finditer(r"\bs_endpgm\b", body) if m.start() > first_mfma]
        assert exits, (src, name)
        for e in exits:
            b = body.rfind("s_barrier", 0, e)
            assert b > 0 and not BF16_MFMA_ISA.search(body[b:e]), (src, name, "MFMAs between the last barrier and an exit")


def test_no_scratch_inside_the_mfma_stream(unit):
    """A spilled register is re-loaded through the vector-memory counter the LDS-DMA ring is paced by (K23 / K24: a scratch load
    in the block loop would wait for every piece in flight).  A slot used in the row prologue only (K24 at K = 384 with a
    LayerNorm in front: 12 bytes, stored and re-loaded before the first MFMA; since the column-span loop of round 5 keeps the
    split rows alive across the epilogue, up to 16 dwords of loop invariants at K = 256 / 384) is tolerated; nothing from the
    first MFMA of a kernel on -- the ring, the epilogue of a range, the way back into the span loop -- may touch scratch."""
    src, kernels, meta = unit
    for name, body in kernels.items():
        ms = [m.start() for m in BF16_MFMA_ISA.finditer(body)]
        if "win_attn3d_stream_kernel" in name:
            # K1's streaming form (round 6) has no LDS-DMA ring; its MFMAs come in three places (the shared tile's chunks, the
            # unrolled tile, the two-pass tile) with the tile loop's entry and exit between them, where a few values that live
            # ACROSS the loop (the shared tile's partial row sum, lane ids) may pass through scratch.  Inside a run of MFMAs --
            # the unrolled tile is ~300 of them within ~4 500 lines -- nothing may.
            lines = body.splitlines()
            mf = [i for i, ln in enumerate(lines) if BF16_MFMA_ISA.search(ln)]
            for i, ln in enumerate(lines):
                if "scratch_" in ln:
                    near = sum(1 for j in mf if abs(j - i) <= 200)
                    assert near <= 12, (src, name, i, ln.strip(), near)      # (a slot of the unrolled tile is ~9 lines: 200 lines of it hold ~45 MFMAs)
            assert meta[name]["private_segment_fixed_size"] <= 160, (src, name, meta[name])
            continue
        assert "scratch_" not in body[ms[0]:ms[-1]], (src, name)
        # behind the last MFMA (the epilogue of a range, inside the span loop): nothing either, except K24's GELU epilogue at
        # K = 384 with 16 / 18 column tiles (no layer of the shipped configs: stage-2 fc1 runs in K23), whose erf polynomial
        # pushes three or four row fragments through scratch per range
        gelu384 = re.search(r"xs_linear_kernelILi384ELi2ELb[01]ELi1[68]E", name) is not None
        assert gelu384 or "scratch_" not in body[ms[-1]:], (src, name)
        # prologue-only slots: a few loop invariants; at K = 1024 with a LayerNorm in front the row's 256 f32 values alone fill
        # the architectural half of the register file (Swin-B stage 3: ~90 registers pass through scratch once per pass)
        limit = 384 if "ILi1024E" in name else 128
        assert meta[name]["private_segment_fixed_size"] <= limit, (src, name, meta[name])


def test_k23_k24_ring_discipline(unit):
    """mlp_split.hip and xs_linear_split.hip issue their LDS-DMA from inline assembly (so that the compiler keeps counted lgkmcnt waits for the fragment
    reads) and paces the ring with its own counted vmcnt in front of every barrier: m0 is written inside those statements
    only and every hand-off barrier follows its own counted s_waitcnt vmcnt."""
    src, kernels, _ = unit
    if src not in ("mlp_split.hip", "xs_linear_split.hip", "xs_linear_split_wide.hip"):
        pytest.skip("K23 / K24 only")
    for name, body in kernels.items():
        n_dma = n_handoff = 0
        in_asm = False
        prev = ""
        for raw in body.splitlines():
            ln = raw.strip()
            if "#ASMSTART" in ln:
                in_asm = True
                continue
            if "#ASMEND" in ln:
                in_asm = False
                continue
            if not ln or ln.startswith(";"):
                continue
            if re.search(r"\bm0\b", ln):
                assert in_asm, (name, ln)
            if ln.startswith("global_load_lds"):
                assert in_asm, (name, ln)
                n_dma += 1
            if ln.startswith("s_barrier") and in_asm:      # a ring hand-off (the two __syncthreads() sit outside the ring)
                assert prev.startswith("s_waitcnt vmcnt("), (name, prev, ln)
                n_handoff += 1
            prev = ln
        assert n_dma > 0 and n_handoff > 0, name


def test_k20_hand_off_barriers_follow_a_drained_vector_memory_counter(unit):
    """K20 (linear_split.hip) still issues its LDS-DMA through the compiler builtin and relies on hipcc draining the vector-memory
    counter in front of every hand-off barrier (the gfx9 memory model does not force that; today's compiler emits
    it).  Pinned here: walking back from every s_barrier of a kernel that contains an LDS-DMA, an `s_waitcnt ... vmcnt(0)` comes
    before any LDS-DMA instruction (register loads issued behind the wait do not touch LDS)."""
    src, kernels, _ = unit
    if src != "linear_split.hip":
        pytest.skip("K20 only")
    checked = 0
    for name, body in kernels.items():
        if "global_load_lds" not in body:
            continue
        lines = [ln.strip() for ln in body.splitlines() if ln.strip() and not ln.strip().startswith(";")]
        for i, ln in enumerate(lines):
            if not ln.startswith("s_barrier"):
                continue
            for back in reversed(lines[:i]):
                if back.startswith("s_waitcnt") and "vmcnt(0)" in back:
                    break
                assert "global_load_lds" not in back and not re.search(r"buffer_load\w* .* lds", back), (name, back)
            else:
                raise AssertionError((name, "no vmcnt(0) in front of a barrier"))
            checked += 1
    assert checked >= 5
