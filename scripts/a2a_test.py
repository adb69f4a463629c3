import os, sys
import torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29645')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
for nbytes in (1 << 20, 1 << 27, 1200000000, 6400000000):
    n64 = nbytes // 8
    src = torch.randint(0, 2**62, (n64,), dtype=torch.int64, device='cuda:0')
    for name, view in (('int64', lambda t: t), ('int32', lambda t: t.view(torch.int32)), ('uint8', lambda t: t.view(torch.uint8))):
        s = view(src); d = torch.zeros_like(s)
        dist.all_to_all_single(d, s, output_split_sizes=[s.numel()], input_split_sizes=[s.numel()])
        torch.cuda.synchronize()
        bad = int((d != s).sum().item())
        print('a2a_single split', nbytes, name, 'mismatch elements', bad, 'of', s.numel())
        d.zero_()
        dist.all_to_all_single(d, s)
        torch.cuda.synchronize()
        print('a2a_single even ', nbytes, name, 'mismatch', int((d != s).sum().item()))
        del d
    # all_gather_into_tensor
    d = torch.zeros_like(src); dist.all_gather_into_tensor(d, src); torch.cuda.synchronize(); print('all_gather', nbytes, int((d != src).sum().item()))
    del src, d
dist.destroy_process_group()
