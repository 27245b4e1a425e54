"""The multiply-shift dividers of the work-item plan (item -> pixel, pixel -> row) against integer division, on the host:
every divisor up to 70 000, random and extreme ones beyond, with edge and random numerators."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def test_dividers_agree_with_integer_division(tmp_path):
    exe = str(tmp_path / "divider_check")
    subprocess.run([HIPCC, "-O2", "-std=c++17", "-x", "hip", "--cuda-host-only", f"-I{ROOT}/include", f"-I{ROOT}/raytracing_amd/csrc",
                    os.path.join(ROOT, "tests", "native", "divider_check.cpp"), "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("0 of "), r.stdout
