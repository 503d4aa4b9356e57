"""Per-phase cycle counts of the step-parallel dvr kernels (csrc/dvr_par_kernels.h), EXPERIMENT build only:

    VIDAR_EXTRA_HIPCC_FLAGS="-DVIDAR_PAR_TIMING" VIDAR_EXTRA_HIPCC_ONLY=dvr_family.hip python vidar_amd/build.py
    python tools/dvr_par_timing.py            # on the GPU box; rebuild without the flag afterwards

The flag makes lane 0 of every workgroup add `s_memtime` deltas at its barriers into a device array that
`vidar_dbg_par_cycles` (only exported by that build) reads back: setup | allocation + chain 1 | rank | chain 2 | consume,
summed over workgroups.  Numbers of round 5: profiles/r05_kbench_dvr_traversal.log."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from vidar_amd.synthetic import ray_set
from vidar_amd.third_lib import dvr as D
from vidar_amd._lib import ptr, stream_of
from vidar_amd._lib import lib
t = lambda a: torch.from_numpy(a).cuda()
L = lib()
buf = (ctypes.c_ulonglong * 8)()
sigma, origin, points, tindex = map(t, ray_set(seed=0, N=1, T=1, rays_per_frame=30000))
N, M = tindex.shape
pred = torch.empty((N, M), device="cuda"); gt = torch.empty((N, M), device="cuda")
L.vidar_dvr_set_traversal(1)
def fwd(flags):
    L.vidar_dvr_render_forward_f32(ptr(sigma), ptr(origin), ptr(points), ptr(tindex), ptr(pred), ptr(gt), N, M, 1, 1, 16, 200, 200, 1, stream_of(sigma))
for name, flags in (("render_forward", 0),):
    fwd(flags); L.vidar_dbg_par_cycles(buf, 1)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fwd(flags)
    e1.record(); torch.cuda.synchronize()
    L.vidar_dbg_par_cycles(buf, 1)
    nwg = (M + 15) // 16
    per = [buf[i] / 10 / nwg for i in range(5)]
    print(name, "ms %.4f" % (e0.elapsed_time(e1) / 10), "cycles per WG: setup %.0f chain1 %.0f rank %.0f chain2 %.0f consume %.0f  total %.0f" % (*per, sum(per)))
