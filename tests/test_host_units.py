"""C++ unit tests of the host layer (thread pool, Map window policy, observation lists),
compiled with g++ and run as a subprocess.  No GPU, no oracle."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_units_cpp(tmp_path):
    exe = str(tmp_path / "host_units")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_units.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all host unit tests passed" in r.stdout
