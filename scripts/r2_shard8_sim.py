"""One GPU playing shard 0 of N (argv[1], default 8): its own descriptions walked N times (stand-ins for the N GPUs' lists), then
level 2 + build -- the compute side of a description-exchange step at N = 8, without the exchange."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tsxcount_amd as T
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reads, k, l, seed = 1087000, 31, 30, 20261004
nb, nk, _ = T.synth_sizes(seed, 0, reads, k)
buf = torch.empty(nb + 256, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
T.synth_fastq_device(seed, 0, reads, k, buf.data_ptr(), nb)
m = T.TSXHashMapHIP(l, 0, k, shard_bits=world.bit_length() - 1, shard_index=0)
L, vp = m._lib, ctypes.c_void_p
cap = ctypes.c_size_t(0)
L.tsx_hip_shard_desc_capacity(m.handle, nb + 256, 1, ctypes.byref(cap))
i64 = dict(dtype=torch.int64, device="cuda:0")
dsc = torch.empty((4 * cap.value,), **i64); cnt = torch.zeros((1,), **i64); emit = torch.zeros((2,), **i64)
st = torch.cuda.Stream()
def step():
    m.clear()
    emit.zero_()
    rc = L.tsx_hip_shard_desc_window_device(m.handle, vp(buf.data_ptr()), nb, 0, nb, 1, vp(dsc.data_ptr()), cap.value,
                                            vp(cnt.data_ptr()), vp(emit.data_ptr()), None)
    assert rc == 0, rc
    m.sync()          # the count is written on the map's own stream
    n = int(cnt.item())
    for s in range(world):
        rc = (L.tsx_hip_shard_filter_device if os.environ.get('SIM_FILTER') == '1' else L.tsx_hip_shard_walk_device)(m.handle, vp(dsc.data_ptr()), n, 1, s, world, int(n * 64 * 1.1), vp(emit[1:].data_ptr()), None)
        assert rc == 0, rc
    rc = L.tsx_hip_shard_build_l1_device(m.handle, None)
    assert rc == 0, rc
    m.sync()
for _ in range(2): step()
t0 = time.perf_counter()
for _ in range(5): step()
dt = (time.perf_counter() - t0) / 5
print("world=%d filter=%s flush_q=%s: %.2f ms per step (desc + N walks + level 2 + build); kept %d of %d described" % (world, os.environ.get("SIM_FILTER", "0"), os.environ.get("TSX_HIP_WALK_FLUSHQ", "auto"), dt * 1e3, int(emit[1].item()), int(emit[0].item())))
