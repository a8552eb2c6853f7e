"""yt_libm.h — the reference platform's libm (glibc 2.35) restated for the device — compiled
for the HOST and compared with the live glibc bit for bit (tests/cpp/libm_check.cpp):
16.7 million arguments per one-argument function here (every exponent, stride 257; the
committed record of the full 2^32 sweep is profiles/r02_libm_check.txt), 33 M seeded pairs
+ special values for powf / atan2f."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("flags", [[], ["-DYT_LIBM_NO_TABLES"]], ids=["shipped", "no-tables"])
def test_restated_libm_equals_glibc(tmp_path, flags):
    """(no-tables: the round-5 lead that picks atanf's interval constants without a table load)"""
    exe = str(tmp_path / "libm_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-pthread", *flags, "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "libm_check.cpp"), "-lm"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "libm_check: OK" in r.stdout, r.stdout + r.stderr
    for fn in ["sinf", "cosf", "sincosf.sin", "sincosf.cos", "expf", "exp2f", "logf", "atanf", "acosf", "powf",
               "atan2f"]:
        assert f"ok       {fn}" in r.stdout, r.stdout
