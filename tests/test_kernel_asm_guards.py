"""Build-time guards on the oscillator-bank kernel's generated code (no GPU needed:
hipcc cross-compiles).  The kernel writes its LDS tile with ds_write_addtid_b32, whose
address comes from M0: nothing else in the kernel may write M0, there must be no
scratch (spilled registers), and descriptors/profiles must be scalar loads."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "openpbso_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_iir_kernel_generated_code(tmp_path):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize",
                    "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                    os.path.join(CSRC, "kernels_iir.hip"), "-o", str(tmp_path / "k.s")], check=True, capture_output=True)
    asm = open(tmp_path / "k.s").read()
    kernels = re.split(r"\n(?=_ZN4pbso10iir_scalar15iir_bank_kernel\S*:)", asm)[1:]
    assert len(kernels) >= 20
    headline = [k for k in kernels if k.startswith("_ZN4pbso10iir_scalar15iir_bank_kernelILi2ELi0ELi1ELi256E")]
    assert len(headline) == 1
    for k in kernels:
        body = k.split("s_endpgm")[0]
        name = body.split(":")[0]
        m0_writes = [l for l in body.splitlines() if re.search(r"\bm0\b", l) and not l.strip().startswith(";")]
        assert all("s_mov_b32 m0" in l for l in m0_writes), (name, m0_writes[:3])
        assert "ds_write_addtid_b32" in body, name
    h = headline[0].split("s_endpgm")[0]
    assert "scratch_" not in h                                   # no spills in the headline shape
    assert "s_load_dwordx8" in h                                 # BufDesc via one scalar load
    hot = [l for l in h.splitlines() if re.match(r"\s+v_(fma|fmac|mul|add)_f32", l)]
    assert hot and all("_e64" not in l for l in hot)            # VOP2 only in the arithmetic
    meta = asm[asm.find(".amdgpu_metadata"):]
    m = re.search(r"\.name:\s+_ZN4pbso10iir_scalar15iir_bank_kernelILi2ELi0ELi1ELi256E.*?\.vgpr_count:\s+(\d+)", meta, re.S)
    if m:
        assert int(m.group(1)) <= 128                            # 4 waves per SIMD need <= 128 VGPRs


@pytest.fixture(scope="module")
def block_asm(tmp_path_factory):
    """kernels_block.hip -> gfx950 assembly, the R = 4 builds only (-DPBSO_ONLY_R4: the headline shape's; ~30 s)"""
    out = tmp_path_factory.mktemp("kb") / "kb.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-DPBSO_ONLY_R4",
                    "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                    os.path.join(CSRC, "kernels_block.hip"), "-o", str(out)], check=True, capture_output=True)
    asm = open(out).read()
    kernels = {}
    for k in re.split(r"\n(?=_ZN4pbso9iir_block16iir_block_kernel\S*:)", asm)[1:]:
        t = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d+)ELb(\d)ELb(\d)E", k.split(":")[0])
        # key: (R, QNM, PROJ, DUMP, FORCED, CHUNKED)
        kernels[tuple(int(t.group(i)) for i in (1, 2, 3, 4, 6, 7))] = k.split("s_endpgm")[0]
    return asm, kernels


def _meta(asm, mangled_part):
    meta = asm[asm.find(".amdgpu_metadata"):]
    m = re.search(r"\.name:\s+_ZN4pbso9iir_block16iir_block_kernel" + mangled_part + r".*?\.private_segment_fixed_size:\s+(\d+).*?"
                  r"\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", meta, re.S)
    return tuple(int(x) for x in m.groups())


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_block_kernel_generated_code(block_asm):
    """K1b, the kernel behind the headline: the f32 build <R=4, qnorm, f32 projection> and the split-bf16 build keep
    0 bytes of scratch and <= 256 VGPRs (two waves per SIMD), the matrix work per buffer is the count the roofline prices
    (8 slices x 32 = 256 v_mfma_f32_16x16x4_f32; 8 x 12 = 96 v_mfma_f32_16x16x32_bf16), the LDS-DMA of a direct hit sets M0 right
    before every global_load_lds_dword and nothing else touches M0."""
    asm, kernels = block_asm
    # 8 builds that walk the buffers in order + their 6 time-chunked twins (K5; the builds that keep block states for a listener
    # mix have none)
    assert len(kernels) == 14 and sum(1 for k in kernels if k[5]) == 6
    f32 = kernels[(4, 2, 0, 0, 0, 0)]
    f32_noqn = kernels[(4, 0, 0, 0, 0, 0)]
    bf16 = kernels[(4, 2, 1, 0, 0, 0)]
    # the chunked twin of the headline build: the same matrix work per buffer body, no scratch, the same register budget
    f32_tc = kernels[(4, 2, 0, 0, 0, 1)]
    assert len(re.findall(r"\bv_mfma_f32_16x16x4_f32\b", f32_tc)) == 256 and "scratch_" not in f32_tc
    for part in ("ILi4ELi2ELi0ELb0ELi512ELb0ELb0E", "ILi4ELi0ELi0ELb0ELi512ELb0ELb0E", "ILi4ELi2ELi1ELb0ELi512ELb0ELb0E",
                 "ILi4ELi0ELi1ELb0ELi512ELb0ELb0E", "ILi4ELi2ELi0ELb0ELi512ELb0ELb1E", "ILi4ELi2ELi1ELb0ELi512ELb0ELb1E"):
        scratch, vgprs, spilled = _meta(asm, part)
        assert scratch == 0 and spilled == 0 and vgprs <= 256, (part, scratch, vgprs, spilled)
    for body in (f32, f32_noqn, bf16):
        assert "scratch_" not in body
    # matrix instructions of ONE buffer body (the slice pipeline is fully unrolled; the per-sample path has none)
    assert len(re.findall(r"\bv_mfma_f32_16x16x4_f32\b", f32)) == 256 and "v_mfma_f32_16x16x32_bf16" not in f32
    assert len(re.findall(r"\bv_mfma_f32_16x16x32_bf16\b", bf16)) == 96 and "v_mfma_f32_16x16x4_f32" not in bf16
    # the forced block path (dense force profiles) adds its own: per group R x 32 + 4 (FIR), + R x 32 for the taps
    forced = kernels[(4, 2, 0, 0, 1, 0)]
    assert len(re.findall(r"\bv_mfma_f32_16x16x4_f32\b", forced)) == 256 + 2 * (4 * 32 + 4) + 4 * 32
    for body in kernels.values():
        lines = body.splitlines()
        m0 = [i for i, l in enumerate(lines) if re.search(r"\bm0\b", l) and not l.strip().startswith(";")]
        dma = [i for i, l in enumerate(lines) if "global_load_lds_dword" in l]
        assert dma and len(m0) == len(dma)
        assert all("s_mov_b32 m0" in lines[i] for i in m0)
        assert all(i - 1 in m0 for i in dma)                       # M0 (the LDS address) is written right before its load
        # no bf16 conversion instruction anywhere near the recurrence (profiles/r02_valu_issue.txt); the only other packed ops are the
        # few v_pk_add_f32 that add accumulator pairs
        assert not re.search(r"\bv_pk_mul_f32\b", body) and "v_cvt_pk_bf16" not in body
        assert len(re.findall(r"\bv_pk_add_f32\b", body)) <= 16
    # Round 6, the f32 projection's pipeline (builds without DUMP): the vector burst is ONE asm statement -- sixteen coarse steps of two
    # v_pk_fma_f32 on the (Q, D) register pair, 8 slices x 32 per buffer body, the block-start states parked by 8 x 8 ds_write2_b64
    # between the steps -- and the matrix burst refills its B operands by ds_read_b128 (8 per slice, 7 slices + the first fill; the
    # forced builds fill three more times per group).  The few ds_read_b32 left are the landing area of a direct hit and the taps.
    # With the steps as separate asm statements the compiler put an s_nop behind each: 269 per body (round 5), now a few dozen.
    for key, body in kernels.items():
        f32_pipeline = key[2] == 0 and key[3] == 0
        assert len(re.findall(r"\bv_pk_fma_f32\b", body)) == (256 if f32_pipeline else 0), key
        if key[2] == 0:
            assert len(re.findall(r"\bds_write2_b64\b", body)) == 64, key
            assert len(re.findall(r"\bds_read_b128\b", body)) >= 64 and len(re.findall(r"\bds_read_b32\b", body)) <= 24, key
        if f32_pipeline:
            assert len(re.findall(r"\bs_nop\b", body)) <= 130, key
    # descriptors come through scalar loads (one s_load_dwordx8 per BufDesc, or x4 + x2 for the words the build uses), never
    # through per-lane vector loads
    assert re.search(r"s_load_dwordx[48]", f32) and re.search(r"s_load_dwordx[48]", bf16)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_scan_kernel_generated_code(tmp_path):
    """K5's scan (kernels_scan.hip), both builds: no scratch (the rows of a group of buffers and their weights stay in
    registers: every index into them is a compile-time constant), and the rows are fetched by plain vector loads that the
    compiler can count -- no load sits behind a branch on the descriptor, so a group's steps wait for THAT group only."""
    out = tmp_path / "kscan.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", os.path.join(CSRC, "kernels_scan.hip"), "-o", str(out)], check=True, capture_output=True)
    asm = open(out).read()
    meta = asm[asm.find(".amdgpu_metadata"):]
    for direct in (0, 1):
        blk = [b for b in meta.split("- .agpr_count") if "iir_scan_kernelILb%dE" % direct in b][0]
        scratch, vgpr, spill = (int(re.search(r"\.%s:\s+(\d+)" % f, blk).group(1)) for f in
                                ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count"))
        assert scratch == 0 and spill == 0 and vgpr <= 256, (direct, scratch, vgpr, spill)
    assert "scratch_" not in asm.split(".amdgpu_metadata")[0]


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_pipe_kernel_generated_code(tmp_path):
    """K1p (kernels_pipe.hip), both builds: no scratch, <= 256 VGPRs, static LDS under 64 KB; the consumers' projection + FIR
    once (32 + 4 MFMAs: the loop over a consumer's groups is not unrolled), the producer's increment products F . T twice
    (group 0, group 1) in both builds (with qnorm rows the consumers re-step the samples for the sums only); and the
    per-sample loops take their profile values from LDS (ds_read_b128), not from scalar loads."""
    out = tmp_path / "kp.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", os.path.join(CSRC, "kernels_pipe.hip"), "-o", str(out)], check=True, capture_output=True)
    asm = open(out).read()
    meta = asm[asm.find(".amdgpu_metadata"):]
    bodies = {int(re.search(r"iir_pipe_kernelILi(\d)E", k).group(1)): k.split("s_endpgm")[0]
              for k in re.split(r"\n(?=_ZN4pbso8iir_pipe15iir_pipe_kernel\S*:)", asm)[1:]}
    assert set(bodies) == {0, 2}
    for qnm, body in bodies.items():
        blk = [b for b in meta.split("- .agpr_count") if "iir_pipe_kernelILi%dE" % qnm in b][0]
        lds, scratch, vgpr, spill = (int(re.search(r"\.%s:\s+(\d+)" % f, blk).group(1)) for f in
                                     ("group_segment_fixed_size", "private_segment_fixed_size", "vgpr_count", "vgpr_spill_count"))
        assert scratch == 0 and spill == 0 and vgpr <= 256 and lds <= 64 * 1024, (qnm, lds, scratch, vgpr, spill)
        n_mfma = len(re.findall(r"\n\s+v_mfma_f32_16x16x4_f32", body))
        assert n_mfma == 32 + 4 + 2 * 32, (qnm, n_mfma)
        assert "s_load_dwordx16" not in body and "ds_read_b128" in body


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_round5_kernels_generated_code(tmp_path):
    """the kernels of round 5: the five-role pipeline teams (kernels_pipe.hip, iir_pipe5_kernel: no scratch, at most 168 VGPRs -- a
    workgroup of ten waves needs three per SIMD --, two teams' LDS within a CU, the roles' matrix work: projection 32 + the FIR's 4,
    increments 32), dense_increment_kernel (64 MFMAs per profile row and 64 columns, no scratch, four waves per SIMD) and the scan cut
    along the time axis (no scratch; its chunk matrix in fp64)"""
    out = tmp_path / "kp.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", os.path.join(CSRC, "kernels_pipe.hip"), "-o", str(out)], check=True, capture_output=True)
    asm = open(out).read()
    meta = asm[asm.find(".amdgpu_metadata"):]
    bodies = {int(re.search(r"iir_pipe5_kernelILi(\d)E", k).group(1)): k.split("s_endpgm")[0]
              for k in re.split(r"\n(?=_ZN4pbso8iir_pipe16iir_pipe5_kernel\S*:)", asm)[1:]}
    assert set(bodies) == {0, 2}
    for qnm, body in bodies.items():
        blk = [b for b in meta.split("- .agpr_count") if "iir_pipe5_kernelILi%dE" % qnm in b][0]
        lds, scratch, vgpr, spill = (int(re.search(r"\.%s:\s+(\d+)" % f, blk).group(1)) for f in
                                     ("group_segment_fixed_size", "private_segment_fixed_size", "vgpr_count", "vgpr_spill_count"))
        assert scratch == 0 and spill == 0 and vgpr <= 168 and lds <= 160 * 1024, (qnm, lds, scratch, vgpr, spill)
        assert len(re.findall(r"\n\s+v_mfma_f32_16x16x4_f32", body)) == 32 + 4 + 32, qnm
        assert "s_barrier" in body and "s_setprio 3" in body
    out = tmp_path / "ks.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", os.path.join(CSRC, "kernels_scan.hip"), "-o", str(out)], check=True, capture_output=True)
    asm = open(out).read()
    meta = asm[asm.find(".amdgpu_metadata"):]
    inc = [k for k in re.split(r"\n(?=_ZN4pbso8iir_scan22dense_increment_kernel\S*:)", asm)[1:]][0].split("s_endpgm")[0]
    assert len(re.findall(r"\n\s+v_mfma_f32_16x16x4_f32", inc)) == 64 and "scratch_" not in inc
    blk = [b for b in meta.split("- .agpr_count") if "dense_increment_kernel" in b][0]
    assert int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)) <= 128
    segs = re.split(r"\n(?=_ZN4pbso8iir_scan19iir_scan_seg_kernel\S*:)", asm)[1:]
    assert len(segs) == 4
    for k in segs:
        body = k.split("s_endpgm")[0]
        assert "scratch_" not in body and "v_fma_f64" in body and "s_barrier" in body
