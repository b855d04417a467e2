"""Captured-HIP-graph helper for fixed-shape loops (not in the reference; SURVEY has no counterpart).

A render / optimisation step at small batch sizes (example 2: one 512x512 view; BASELINE config 2: 16 views) is bound by
host time: ~15 kernel launches, a dozen allocations and two autograd.Function round trips take longer to ISSUE than the
GPU needs to run them.  Every launch of libnr_hip.so goes to the caller's stream and none of them synchronises or calls a
capture-hostile API, so a whole step -- forward, loss, backward, optimizer -- can be captured once with
torch.cuda.CUDAGraph (= hipGraph on ROCm) and replayed with one host call.

    step = nr.graph.capture(lambda: train_step(), device)     # train_step reads / writes fixed tensors
    for _ in range(300):
        step()

Rules of graph capture apply: the callable must use the same tensors every time (update inputs with `.copy_()`), must not
synchronise or move data to the host, and an optimizer must keep its step counter on the device
(`torch.optim.Adam(..., capturable=True)`).  The library's only per-process state -- the dynamic-LDS limit already granted
to a kernel -- is settled by the warm-up calls, which run outside the capture."""
import torch


def capture(fn, device=None, warmup=3):
    """Run `fn` `warmup` times on a side stream (allocator and kernel attributes settle), capture one more call into a
    graph and return a zero-argument callable that replays it.  Tensors created inside `fn` live in the graph's private
    pool: read results from the tensors `fn` assigns to (e.g. `.grad` fields or pre-allocated outputs).

    (Round 3 had a crash here -- a whole-step capture after the operator's own graph-replay mode had run in the same process
    took the process down inside torch's capture.  It came from that round's host code, not from the kernels (round 3's
    Python with this round's library still crashes, this round's Python with round 3's library does not:
    scripts/graph_crash_probe.py), and is gone with the rewritten operator; tests/test_hip_parity.py::
    test_whole_step_capture_after_operator_replay runs the sequence in a subprocess.)"""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        for _ in range(warmup):
            fn()
    torch.cuda.current_stream(device).wait_stream(side)
    torch.cuda.synchronize(device)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    return _Replay(graph)


class _Replay(object):
    """Zero-argument callable that replays a captured graph (and keeps it, with its memory pool, alive)."""

    def __init__(self, graph):
        self.graph = graph

    def __call__(self):
        self.graph.replay()


def backward_on_caller_thread(flag=True):
    """The cheaper remedy for a host-bound loop.  torch's autograd engine hands the backward of CUDA tensors to a device
    thread; waking it and passing the GIL back and forth costs ~100 us per step (scripts/host_profile.py: 283 -> 153 us of
    host time per Rasterize forward + backward; BASELINE config 2 through the plain API 0.205 -> 0.158 ms, example 2
    0.55 -> 0.35 ms per step -- `profiles/r03l_configs.jsonl`).  This is `torch.autograd.set_multithreading_enabled(not
    flag)`: call it once, or use it as a context manager around the loop.  A setting of the calling THREAD's autograd state
    (every backward started from this thread then runs its CUDA nodes here), so the library never switches it on its own."""
    return torch.autograd.set_multithreading_enabled(not flag)
