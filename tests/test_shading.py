"""yt_shading.h — the device's fused lobe evaluator (eval_lobe = {eval_bsdfcos, sample_bsdfcos_pdf} from
one evaluation of the shared terms, sample_lobe, the delta forms, the medium) — compiled for the HOST and
compared bit for bit with the reference's own lobe functions (libs/yocto/yocto_shading.h, included from
the reference tree) behind the reference's material dispatch: tests/cpp/shading_check.cpp.  Every material
type, rough and delta, random and degenerate direction pairs.  Needs the reference headers, so it runs
where /root/reference exists (here; not on the GPU box, where the `-m gpu` renders are the check)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("YOCTO_REF", "/root/reference")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "libs", "yocto", "yocto_shading.h")),
                    reason="reference headers not present")
def test_fused_lobe_evaluator_equals_the_reference_lobes(tmp_path):
    exe = str(tmp_path / "shading_check")
    subprocess.run(["g++", "-O3", "-DNDEBUG", "-std=c++17", "-ffp-contract=off", f"-I{REF}/libs", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "shading_check.cpp"), "-lm"], check=True)
    r = subprocess.run([exe, "300000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "shading_check: OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert " 0 mismatches" in r.stdout
