"""Builds tests/cpp/test_host_api.cpp (a plain g++ program against gridpp_amd/host/gridpp.hpp + libgridpp_hip.so)
and runs it on the GPU box: the C++ drop-in boundary works end to end."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_program(tmp_path):
    libdir = os.path.join(ROOT, "gridpp_amd", "lib")
    exe = str(tmp_path / "test_host_api")
    cmd = ["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "gridpp_amd", "host"), os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp"),
           "-L", libdir, "-lgridpp_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout


def test_plain_c_caller_with_deferred_calls(tmp_path):
    """tests/cpp/test_async_c_abi.c: gcc, include/gridpp_hip.h, fields in HBM (hipMalloc), GPP_ASYNC + gpp_wait with one call ahead -- the bits of
    the blocking calls, three passes over six observation sets (one of them with other usable observations: the wait runs that call again)."""
    libdir = os.path.join(ROOT, "gridpp_amd", "lib")
    exe = str(tmp_path / "test_async_c_abi")
    cmd = ["gcc", "-std=c11", "-O1", "-I", os.path.join(ROOT, "include"), "-I/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "test_async_c_abi.c"),
           "-L", libdir, "-lgridpp_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout
