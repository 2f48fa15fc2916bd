"""include/ripped.hpp (the C++ host-side mirror of the reference API) through tests/cpp/test_ripped.cpp:
the reference's own test list restated in C++.  Host-only part on CPU, everything on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_ripped.bin")


def _build():
    lib = os.path.join(ROOT, "lp_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "test_ripped.cpp"), "-o", EXE, "-L", lib, "-llpipm",
                    f"-Wl,-rpath,{lib}"], check=True)


def test_cpp_host_logic(built):
    _build()
    r = subprocess.run([EXE, "host"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_reference_tests_on_gpu(built):
    _build()
    r = subprocess.run([EXE, "gpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
