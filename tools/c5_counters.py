"""Counters per sample of the fused three-window recipe (C5) in steady launches: how many replay steps a sample costs
(measuring passes included), python3 tools/c5_counters.py"""
import sys, json, numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import cudabrot_amd as cb
from tools.seq_profile import CONFIGS
w, h, windows, (x0, x1, y0, y1) = CONFIGS["C5"]
dev = torch.device("cuda", 0)
dims = cb.FractalDimensions.make(w, h, x0, x1, y0, y1)
T = cb.CB_DEFAULT_THREADS
planes = len(windows)
hist = torch.zeros(planes * w * h, dtype=torch.int64, device=dev)
states = torch.empty(cb.rng_state_bytes(T), dtype=torch.uint8, device=dev)
counters = torch.zeros(17, dtype=torch.int64, device=dev)
carry = torch.zeros(cb.carry_bytes(T), dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
cb.initialize_rng(1337, 0, T, states.data_ptr(), stream)
spt = 3200
wsb = cb.scatter_workspace_bytes(dims, T, spt, n_channels=planes)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
for k in range(4):
    if k == 2:
        torch.cuda.synchronize(); counters.zero_()
    cb.draw_buddhabrot_channels(dims, hist.data_ptr(), windows, states.data_ptr(), T, spt, counters.data_ptr(), cb.CB_KERNEL_DEFAULT, stream, ws.data_ptr(), wsb, carry.data_ptr())
    cb.flush_scatter_channels(dims, hist.data_ptr(), planes, T, ws.data_ptr(), wsb, stream)
torch.cuda.synchronize()
c = dict(zip(cb.Counters().as_dict().keys(), (int(v) for v in counters.cpu().numpy().view(np.uint64))))
n = c["samples"]
print({k: round(v / n, 4) for k, v in c.items() if k in ("iterate_steps", "skipped_steps", "replay_steps", "increments", "recorded", "too_fast", "never_escaped")})
