"""ROCm 7.2 / PyTorch 2.10: hipStreamEndCapture segfaults when, inside ONE stream capture, a stream that is itself a fork of the
origin forks (or merely waits on an event of) another non-origin stream and is waited on by it in turn.  Plain torch ops suffice.
Forks from the ORIGIN stream work.  Found while trying to run the value head's first layer beside the policy head's second layer
inside session.capture_pair (round 4); kept as the evidence.     python tools/capture_nested_fork_repro.py x"""
import faulthandler, sys, torch
faulthandler.enable()
variant = sys.argv[1]
dev = torch.device("cuda:0")
s0, s1, s2 = (torch.cuda.Stream(device=dev) for _ in range(3))
x = torch.zeros(1 << 20, device=dev); y = torch.zeros(1 << 20, device=dev); z = torch.zeros(1 << 20, device=dev)
torch.cuda.synchronize()
keep = []
def wait(waiter, on):
    ev = torch.cuda.Event()
    ev.record(on)
    waiter.wait_event(ev)
    keep.append(ev)          # alive until the capture has ended
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s0, capture_error_mode="thread_local"):
    x.add_(1)
    wait(s1, s0)
    with torch.cuda.stream(s1):
        y.add_(1)
    wait(s2, s1)
    with torch.cuda.stream(s2):
        z.add_(1)
    with torch.cuda.stream(s1):
        y.add_(1)
    wait(s1, s2)
    with torch.cuda.stream(s1):
        y.add_(z)
    wait(s0, s1)
print("captured", variant, flush=True)
g.replay(); torch.cuda.synchronize()
print("ok", variant, float(y[0]), flush=True)
