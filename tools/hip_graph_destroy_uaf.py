"""Repro WITHOUT this package (torch only) of the host-heap damage behind the eval-latent fits:
    gcc -O1 -g -shared -fPIC -o /tmp/heap_guard.so tools/heap_guard.c -ldl
    HEAP_GUARD_FENCE_SIZE=920 LD_PRELOAD="/tmp/heap_guard.so${LD_PRELOAD:+:$LD_PRELOAD}" python tools/hip_graph_destroy_uaf.py [fork|line] [inflight|idle] [cycles] [wide] [depth]
A HIP graph whose capture FORKED a second stream is instantiated with internal streams of its own.  hipGraphLaunch keeps a reference to
the executable graph until the launch's last command completes; when the user's handle is destroyed while a launch is in flight, that
reference is the LAST one and it is dropped by the completion callback, on ROCr's async-events thread:
    amd::roc::HsaAmdSignalHandler -> VirtualGPU::updateCommandsState -> amd::Event::setStatus -> Event::processCallbacks
      -> hip::GraphExec::~GraphExec -> hip::Stream::terminate -> amd::HostQueue::terminate -> free(roc::VirtualGPU, 920 bytes)
and HsaAmdSignalHandler then goes on using the VirtualGPU it was called for -- the one just freed when the completing command ran on one
of the graph's own streams (a decrement at offset 152: glibc's "free(): invalid pointer" / "corrupted size vs. prev_size" much later,
in whoever owns that memory by then).  libamdhip64.so 7.0.51831 (torch 2.10.0+rocm7.0 wheel).
Measured on an MI355X box under the fenced guard (round 6, GPU call 11; gpurun_out/r6/call11_uaf_*.log, profiles/r06_heap_uaf.txt):
    fork inflight, 3 side streams x 8 kernels:  SIGSEGV at HsaAmdSignalHandler+0xca, offset 0xb8 of a freed page; 27 of 27 VirtualGPU frees came from the handler
    fork inflight, 4 side streams x 30 kernels: the same
    fork idle (synchronize, wait 50 ms, then destroy), 3 x 8: clean, 121 frees, none from the handler (all from hipGraphExecDestroy on the caller's thread)
    fork inflight, 2 x 4: 80 of 81 frees from the handler, no fault in 40 cycles (the completing command must sit on a freed stream)
    one side stream / line (no fork): the executable graph owns no stream it frees at destruction: clean."""
import ctypes, sys, time
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else "fork"
when = sys.argv[2] if len(sys.argv) > 2 else "inflight"
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 40
wide = int(sys.argv[4]) if len(sys.argv) > 4 else 3    # forked side streams
depth = int(sys.argv[5]) if len(sys.argv) > 5 else 8   # kernels per branch
libc = ctypes.CDLL(None)
sweep = getattr(libc, "heap_guard_sweep", None)
if sweep is not None:
    sweep.restype, sweep.argtypes = ctypes.c_long, [ctypes.c_char_p]
x = torch.zeros(1 << 26, device="cuda")
sides = [torch.cuda.Stream() for _ in range(wide)]
def body():
    y = x * 2
    if mode == "fork":  # `wide` more branches on other streams, forked and joined twice before the capture ends
        for rnd in range(2):
            zs = []
            for s_ in sides:
                s_.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s_):
                    z = y + 1
                    for _ in range(depth):
                        z = z * 1.0001
                zs.append(z)
            for s_ in sides:
                torch.cuda.current_stream().wait_stream(s_)
            y = y + sum(zs)
        return y
    return y + (x + 1)
print("torch", torch.__version__, "hip", torch.version.hip, "guard", sweep is not None, mode, when, "wide", wide, "depth", depth, flush=True)
for i in range(cycles):
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = body()
    for _ in range(3):
        graph.replay()
    if when == "idle":
        torch.cuda.synchronize()
        time.sleep(0.05)
    del graph, out  # hipGraphExecDestroy (+ hipGraphDestroy): in flight -> the launch's reference is the last one
    torch.cuda.synchronize()
    time.sleep(0.01)
    if sweep is not None and sweep(b"after destroying graph %d" % i):
        print("HEAP DAMAGE first seen after destroying graph", i, flush=True)
        sys.exit(77)
print("clean:", cycles, "capture / replay / destroy cycles", flush=True)
