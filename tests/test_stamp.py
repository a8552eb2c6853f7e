"""The drop-in's residency stamps (yocto-gl_amd/host/yt_stamp.h): the device mirrors of scene / bvh / lights are
re-sent when the content stamp of the caller's arrays changes.  Since round 4 the large arrays are stamped by a
64-bit hash of EVERY byte (a pool of host threads), so that an in-place edit of one vertex is seen as the reference
sees it (it reads its arguments fresh on every call, yocto_trace.cpp:1595-1619); tests/cpp/stamp_check.cpp checks
the hash on its own, tests/cpp/dropin_test.cpp §5b checks the render on the GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_full_content_stamp_sees_every_single_bit_edit(tmp_path):
    exe = str(tmp_path / "stamp_check")
    subprocess.run(["g++", "-O3", "-std=c++17", "-pthread", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "stamp_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "stamp_check: OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    print(r.stdout)
