"""does a capture survive a stream that waits on ITSELF, or two 'different' torch streams that are the same pool stream?  (torch.cuda.Stream()
hands out 32 pool streams per priority round-robin: the 33rd object aliases the 1st.)  python tools/stream_alias_probe.py <case>"""
import sys
import torch
case = sys.argv[1]
x = torch.zeros(1 << 20, device="cuda")
cap = torch.cuda.Stream()
others = [torch.cuda.Stream() for _ in range(31)]
alias = torch.cuda.Stream()  # 33rd: the same pool stream as `cap`
print("cap", cap.stream_id, "alias", alias.stream_id, "equal", cap == alias, flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=cap):
    y = x * 2
    side = {"self_wait": cap, "alias_side": alias, "distinct": others[0]}[case]
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        z = y + 1
    torch.cuda.current_stream().wait_stream(side)
    out = y + z
g.replay()
torch.cuda.synchronize()
print(case, "ok", float(out[0]), flush=True)
