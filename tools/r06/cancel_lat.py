import sys, time, threading, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "yocto-gl_amd")
import parity as P
from parity import yt
flat = P.SCENES["cornellbox"]()
for sched, fin in ((0, 250), (1, 250), (1, 0)):
    ctx = P.gpu_context(flat); ctx.set_traversal("wide"); ctx.set_scheduler(sched); ctx.set_stream_finish(fin)
    p = yt.trace_params(sampler="path", resolution=1280, samples=1 << 20, batch=4096)
    ctx.make_trace_state(flat, p)
    q = yt.trace_params(sampler="path", resolution=1280, samples=1 << 20, batch=8)
    ctx.trace_samples(q)
    stop = np.zeros(1, np.int32)
    t_set = [0.0]
    def fire():
        t_set[0] = time.time(); stop[0] = 1
    threading.Timer(0.2, fire).start()
    try:
        ctx.trace_samples(p, stop=stop)
    except yt.YthipError as e:
        pass
    print(f"scheduler {sched} finish {fin}: returned {1e3 * (time.time() - t_set[0]):.1f} ms after the flag; info {ctx.stream_info()['generations']} generations, launched {ctx.stream_info()['launched']}, finish_rays {ctx.stream_info()['finish_rays']}", flush=True)
    ctx.close()
