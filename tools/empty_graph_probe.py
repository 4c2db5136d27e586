import torch
s = torch.cuda.Stream()
x = torch.zeros(8, device="cuda")
pool = torch.cuda.graph_pool_handle()
with torch.cuda.stream(s):
    g1 = torch.cuda.CUDAGraph(); g1.capture_begin(pool=pool, capture_error_mode="thread_local"); x.add_(1); g1.capture_end()
    g2 = torch.cuda.CUDAGraph(); g2.capture_begin(pool=pool, capture_error_mode="thread_local"); g2.capture_end()
    g3 = torch.cuda.CUDAGraph(); g3.capture_begin(pool=pool, capture_error_mode="thread_local"); y = x * 2; g3.capture_end()
torch.cuda.synchronize()
for g in (g1, g2, g3):
    try:
        g.replay(); print("replay ok")
    except Exception as e:
        print("replay failed:", type(e).__name__, str(e)[:200])
torch.cuda.synchronize()
print(x, y)
