"""Where the holes of the minimizer lists come from: chunks used per workgroup of desc_owner_split_kernel, one window."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tsxcount_amd as T
world, reads, k, l, seed = 8, int(os.environ.get("SIM_READS", "271750")), 31, 30, 20261004
nb, nk, _ = T.synth_sizes(seed, 0, reads, k)
buf = torch.empty(nb + 256, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
T.synth_fastq_device(seed, 0, reads, k, buf.data_ptr(), nb)
m = T.TSXHashMapHIP(l, 0, k)
L, vp = m._lib, ctypes.c_void_p
cap = ctypes.c_size_t(0)
L.tsx_hip_mini_capacity(m.handle, nb + 256, world, ctypes.byref(cap))
cap = cap.value
i64 = dict(dtype=torch.int64, device="cuda:0")
dsc = torch.empty((2 * cap * world,), **i64)
cnt = torch.zeros((world + 4,), **i64)
emit = torch.zeros((2,), **i64)
assert L.tsx_hip_mini_window_device(m.handle, vp(buf.data_ptr()), nb, 0, nb, world, vp(dsc.data_ptr()), cap, vp(cnt.data_ptr()), vp(emit.data_ptr()), None) == 0
m.sync()
c = [int(x) for x in cnt.tolist()]
G = 1024
for o in range(world):
    lst = dsc[2 * o * cap:2 * (o * cap + c[o])].view(-1, 2)
    valid = ((lst[:, 1] >> 32) & 0xFFFF) != 0
    J = c[o] // (G * 64)
    per = valid.view(J, G, 64).sum(2)            # descriptions per chunk
    used = (per > 0).sum(0)                      # chunks used per workgroup
    tot = per.sum(0)
    print("owner %d: list %d, valid %d (%.1f %% holes); J = %d; descriptions per workgroup min %d mean %.0f max %d; chunks used min %d max %d"
          % (o, c[o], int(valid.sum()), 100 - 100.0 * int(valid.sum()) / c[o], J, int(tot.min()), float(tot.float().mean()), int(tot.max()), int(used.min()), int(used.max())))
