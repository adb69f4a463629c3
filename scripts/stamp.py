import sys, ctypes
sys.path.insert(0,'.')
import torch, numpy as np
import tsxcount_amd as T
nb, nk, _ = T.synth_sizes(1, 0, 1087000, 31)
text = torch.empty(nb+256, dtype=torch.uint8, device='cuda:0'); torch.cuda.synchronize()
T.synth_fastq_device(1, 0, 1087000, 31, text.data_ptr(), nb)
m = T.TSXHashMapHIP(30, 0, 31); m.set_path('partitioned')
for it in range(2):
    m.clear(); m.countFastqDevice(text.data_ptr(), nb); m.sync()
buf = (ctypes.c_ulonglong*16)()
m._lib.tsx_hip_debug_stats.argtypes=[ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
m._lib.tsx_hip_debug_stats(m.handle, buf)
names=['load+classify','linescan+bar','roundA(per-pos+dedup)','bar','phaseB(emit)','bar']
tot=sum(buf[8+i] for i in range(6))
for i in range(6): print(names[i], buf[8+i], '%.1f%%'%(100.0*buf[8+i]/max(tot,1)))
print('total cycles', tot)
