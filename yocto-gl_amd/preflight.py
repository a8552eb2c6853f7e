"""A sacrificial first GPU process.

On this pool the very first GPU process of a freshly provisioned MI355X box has, three times in
about ten, died with `Memory access fault by GPU node-2 ... Reason: Unknown` during its first
large host-to-device copies / kernels — never a later process on the same box (hundreds of
runs, same binaries, same inputs).  Everything that is about to use the GPU for something
that matters (the GPU test session, bench.py, smoke()) therefore first runs this module in a
process of its own and ignores what happens to it: it creates a context, uploads the
1M-triangle plane, builds its tree on the device and renders a few samples — the same first
steps the real run takes.

    python yocto-gl_amd/preflight.py [device]        (exit code irrelevant)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def _body(device):
    sys.path.insert(0, HERE)
    import ythip as yt
    import scenes as ysc
    flat = ysc.plane_scene()
    try:
        ctx = yt.Context(device)
    except yt.YthipError:
        sys.exit(77)  # no device: nothing to warm up (and nothing to retry)
    ctx.upload_scene(flat)
    ctx.make_trace_bvh(flat)
    ctx.make_trace_lights(flat)
    p = yt.trace_params(sampler="path", resolution=1280, samples=8, batch=4)
    ctx.make_trace_state(flat, p)
    ctx.trace_samples(p)
    ctx.trace_samples(p)
    ctx.download_state()
    ctx.close()


def run(device=0, timeout=300):
    """Run the preflight in a child process; returns its exit code (None on timeout).  Never raises."""
    if os.environ.get("YTHIP_NO_PREFLIGHT"):
        return 0
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    import time
    rc = None
    for attempt in range(3):  # (the process after a faulted one failed too, once: give the device a moment)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(device)], capture_output=True, text=True,
                               timeout=timeout, env=env)
            rc = r.returncode
            if rc in (0, 77):
                return rc
            print(f"[preflight] attempt {attempt + 1}: the sacrificial GPU process ended with rc {rc}: "
                  f"{(r.stderr or '').strip().splitlines()[-1:] or ''}", file=sys.stderr, flush=True)
        except Exception as e:  # timeout, missing interpreter, ...
            print(f"[preflight] attempt {attempt + 1}: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        time.sleep(5)
    return rc


if __name__ == "__main__":
    _body(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
