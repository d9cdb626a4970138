import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B, sharding, synth_scenes as S
W, H, K = 1920, 1080, 40
cam = dict(S.BENCH_CAMERA); cam["dir"] = S.camera_dir(cam)
r = B.Renderer(W, H, cam)
r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
r.set_primitives(B.make_prims(S.scene_d()))
r.set_max_path_segments(8)
r.backend.set_timing(0)
for n in (1, 8):
    if n > 1:
        y0, rows, band, stride, _ = sharding.interleaved_rows(0, n, H)
        r.set_interleaved_tile(0, y0, W, rows, band, stride)
    r.restart_path_tracing(1, 3); [r.path_tracing_pass() for _ in range(3)]; r.finish()
    r.set_seed(5489); r.restart_path_tracing(1, K)
    t0 = time.perf_counter()
    for _ in range(K): r.path_tracing_pass()
    t1 = time.perf_counter()
    r.finish()
    t2 = time.perf_counter()
    print("N=%d: host enqueue %.3f ms/pass, total %.3f ms/pass" % (n, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
