"""Compile-time guard on the gfx950 kernels (hipcc cross-compiles here, no GPU needed): no kernel of the library may use scratch
memory.  Round 3 found k_chunk_units with 936 bytes of scratch per lane after an innocent-looking bounds check (the compiler
turned a select chain on the unit's kind into a table on the stack and kept a copy of the kernel's argument block there): 26 GB
of HBM writes per run and twice the kernel's time, with every parity test green.  rocprofv3 showed it; this test would have."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "epidemicsimulator_amd", "csrc")


def test_no_kernel_uses_scratch_and_the_hot_kernels_keep_their_occupancy(tmp_path):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off",
           "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", str(tmp_path / "k.o"), os.path.join(CSRC, "esim_api.hip")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            kernels[cur] = {}
            continue
        m = re.search(r"remark:\s+([\w \[\]/]+): (\d+)", line)
        if m and cur:
            kernels[cur][m.group(1).strip()] = int(m.group(2))
    chunk = {k: v for k, v in kernels.items() if "k_chunk" in k or "k_decide" in k or "k_future" in k}
    assert len(chunk) >= 10, sorted(kernels)
    for name, r in kernels.items():
        assert r.get("ScratchSize [bytes/lane]", 0) == 0, "%s uses %d bytes of scratch per lane" % (name, r["ScratchSize [bytes/lane]"])
        assert r.get("VGPRs Spill", 0) == 0, name
    # the chunk pass launches 4 wavefronts per SIMD (1024 workgroups of 256): the draw kernels must be able to hold them
    for name, r in kernels.items():
        if "k_chunk_draw" in name or "k_chunk_units" in name or "k_chunk_marks" in name:
            assert r["Occupancy [waves/SIMD]"] >= 4, (name, r)
            assert r["VGPRs"] <= 128, (name, r)
        # ... and k_chunk_draw a fifth (its grid is four times the marks grid: the dispatcher keeps five wavefronts per SIMD resident;
        # a change that took it from 94 to 106 registers cost 10 % of the kernel with every parity test green)
        if "k_chunk_draw" in name:
            assert r["Occupancy [waves/SIMD]"] >= 5 and r["VGPRs"] <= 96, (name, r)
