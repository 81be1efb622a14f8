"""The C ABI exercised from plain C (tests/test_cabi.c): create -> upload -> run -> step -> download_state -> destroy on the
synthetic York, compared with the CPU oracle inside the C program -- no Python between the caller and the library, as the
reference's own FFI binding would use it (run/src/load_data.rs:120-124, run/src/main.rs:306)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_cabi_program():
    out_dir = os.path.join(ROOT, "build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "test_cabi")
    lib_dir, orc_dir = os.path.join(ROOT, "epidemicsimulator_amd"), os.path.join(ROOT, "oracle")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + orc_dir,
                           os.path.join(ROOT, "tests", "test_cabi.c"), "-o", exe,
                           os.path.join(lib_dir, "libesim.so"), os.path.join(orc_dir, "libesim_oracle.so"),
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + orc_dir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_c_program_links_against_the_abi():
    """CPU: the header compiles as C11 and every entry point the program uses resolves at link time."""
    assert os.path.exists(build_cabi_program())


@pytest.mark.gpu
def test_c_caller_matches_the_oracle():
    exe = build_cabi_program()
    p = subprocess.run([exe, "1200"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "cabi ok: 1200 steps x 197603 citizens" in p.stdout and "vaccination running" in p.stdout, p.stdout
