"""Diagnostic: rounds / cycles / lane utilisation of build_segments_stream_kernel (TSX_HIP_DEBUG=16)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TSX_HIP_DEBUG"] = "16"
import torch
import tsxcount_amd as T
k, l, reads, seed = 31, 30, 1087000, 20261004
nb, nk, _ = T.synth_sizes(seed, 0, reads, k)
buf = torch.empty(nb + 256, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
T.synth_fastq_device(seed, 0, reads, k, buf.data_ptr(), nb)
m = T.TSXHashMapHIP(l, 0, k)
m.clear(); m.countFastqDevice(buf.data_ptr(), nb); m.sync()
out = (ctypes.c_uint64 * 8)()
m._lib.tsx_hip_debug_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
m._lib.tsx_hip_debug_counters(m.handle, out)
rounds, cyc, lanes, tail, tailcyc = [int(x) for x in out[:5]]
waves = 65536 * 16
print("wave-passes", waves, "rounds/wave %.1f" % (rounds / waves), "cycles/round %.0f" % (cyc / max(rounds, 1)),
      "probing lanes/round %.1f" % (lanes / max(rounds, 1)), "probes/key %.2f" % (lanes / 804329712),
      "tail rounds/wave %.1f" % (tail / waves), "cycles/tail round %.0f" % (tailcyc / max(tail, 1)),
      "insert cycles/wave %.0f" % (cyc / waves))
