import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
mode = sys.argv[1]
if mode == 'torch_first':
    import torch
    print('torch first: cuda avail', torch.cuda.is_available())
    from video_segment_amd import _lib
    print('lib devices', _lib.lib().vsg_device_count())
    x = torch.zeros(4, device='cuda'); print(x.sum().item())
else:
    from video_segment_amd import _lib
    print('lib devices', _lib.lib().vsg_device_count())
    import torch
    print('lib first: cuda avail', torch.cuda.is_available())
